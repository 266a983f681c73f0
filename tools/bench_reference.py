#!/usr/bin/env python3
"""B-ref of BASELINE.md section 2: the REAL reference timed on host cores -- robosuite 1.5.2 + the `mujoco` wheel, N worker processes with one env each.

    python tools/bench_reference.py [--procs N] [--steps 2000] [--warmup 100] [--env Lift] [--robots Panda] [--backend mujoco|shim]

Protocol (BASELINE.md section 2, B-ref): `suite.make(env, robots=..., has_renderer=False, has_offscreen_renderer=False, use_camera_obs=False, use_object_obs=True,
reward_shaping=True, control_freq=20, horizon=500, ignore_done=True, seed=i)` (environments/base.py:23-42; with ignore_done the timed region holds no resets), the
robot's default controller (controllers/config/robots/default_panda.json: OSC_POSE), `time.perf_counter` around `--steps` calls of `env.step(a)`
(environments/base.py:467-521) after `--warmup`, `a ~ U(-1, 1)^dim` from `default_rng(10**6 + i)`, NUMBA_DISABLE_JIT=1 as in the reference's CI
(.github/workflows/run-tests.yaml:60).  Reported: sum of steps / wall time of the slowest worker (all workers start behind a barrier), one JSON line.

`--backend mujoco` (default) needs the wheel the reference depends on (setup.py:18, mujoco>=3.3.0,<3.10).  It is NOT installed in the build container and cannot be
(no network): the tool then says so and exits 3 -- it never substitutes anything silently.  `--backend shim` runs the same harness over robosuite_amd.shim with the
project's fp64 oracle as arithmetic (test infrastructure: it exists so that the harness is exercised end to end before a wheel is; its number is B-cpu, labelled
"port", never B-ref).  The same session that has the wheel should also run tools/gen_golden_with_mujoco.py: goldens and the true CPU baseline together.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("RSIM_REFERENCE", "/root/reference")


def _worker(i, a, barrier, out):
    os.environ["NUMBA_DISABLE_JIT"] = "1"
    sys.path.insert(0, ROOT)
    try:
        if a.backend == "shim":
            # the mujoco-shaped shim with the fp64 oracle as arithmetic (what tools/gen_golden.py records the fixtures through) + stubs for the optional packages
            from oracle.shim_backend import OracleBackend
            from robosuite_amd import shim
            shim.install(OracleBackend)
        sys.path.insert(0, REF)
        import numpy as np
        import robosuite as suite
        env = suite.make(a.env, robots=a.robots.split(","), has_renderer=False, has_offscreen_renderer=False, use_camera_obs=False, use_object_obs=True,
                         reward_shaping=True, control_freq=20, horizon=500, ignore_done=True, seed=i)
        env.reset()
        rng = np.random.default_rng(10 ** 6 + i)
        lo, hi = env.action_spec
        for _ in range(a.warmup):
            env.step(rng.uniform(-1.0, 1.0, lo.shape[0]))
        barrier.wait()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            env.step(rng.uniform(-1.0, 1.0, lo.shape[0]))
        out.put((i, time.perf_counter() - t0, None))
    except Exception as e:   # noqa: BLE001 -- reported by the parent
        try:
            barrier.abort()
        except Exception:   # noqa: BLE001
            pass
        out.put((i, None, f"{type(e).__name__}: {e}"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--env", default="Lift")
    ap.add_argument("--robots", default="Panda")
    ap.add_argument("--backend", choices=("mujoco", "shim"), default="mujoco")
    a = ap.parse_args()
    if a.backend == "mujoco":
        try:
            import mujoco  # noqa: F401
        except Exception as e:   # noqa: BLE001
            print(json.dumps({"baseline": "B-ref", "error": f"the `mujoco` wheel is not importable ({type(e).__name__}: {e}); B-ref is blocked until it is "
                              "(BASELINE.md section 2).  `--backend shim` exercises this harness on the project's fp64 oracle (B-cpu, not B-ref)."}))
            raise SystemExit(3)
    if not os.path.isdir(os.path.join(REF, "robosuite")):
        print(json.dumps({"baseline": "B-ref", "error": f"no robosuite checkout at {REF} (RSIM_REFERENCE)"}))
        raise SystemExit(3)
    ctx = mp.get_context("spawn")
    barrier, out = ctx.Barrier(a.procs), ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(i, a, barrier, out)) for i in range(a.procs)]
    [p.start() for p in ps]
    res = [out.get() for _ in ps]
    [p.join() for p in ps]
    errs = [r[2] for r in res if r[2]]
    if errs:
        print(json.dumps({"baseline": "B-ref" if a.backend == "mujoco" else "B-cpu", "error": errs[0], "failed_workers": len(errs)}))
        raise SystemExit(1)
    wall = max(r[1] for r in res)
    print(json.dumps({"baseline": "B-ref" if a.backend == "mujoco" else "B-cpu", "kind": "reference" if a.backend == "mujoco" else "port",
                      "value": a.procs * a.steps / wall, "unit": "env-steps/s", "cores": a.procs, "steps_per_worker": a.steps, "warmup": a.warmup,
                      "workload": f"{a.env}/{a.robots}, default controller, control_freq 20 (25 substeps per env.step)", "wall_s": wall,
                      "per_worker_steps_per_s": [a.steps / r[1] for r in sorted(res)],
                      "sample": f"{a.procs} processes x {a.steps} env.step() calls of the unmodified reference Python on "
                                + ("MuJoCo" if a.backend == "mujoco" else "robosuite_amd.shim + the fp64 C oracle (NOT MuJoCo)")}))


if __name__ == "__main__":
    main()
