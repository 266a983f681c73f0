#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the fused control-step hot path on MI355X (BASELINE.json metric).

One "step" = one robosuite `env.step(action)` for every env of the batch = ONE launch of the fused kernel
(25 physics substeps at dt=0.002 + 25 OSC_POSE/GRIP controller evaluations + set_goal; reference
environments/base.py:467-521).  Workload at N=1 = BASELINE configs[1]: Lift / Panda / OSC_POSE, 4096 envs on one GPU,
per-env seeded episodes (cube size, arm noise, cube pose) and per-env action streams.
N>1: every rank owns 4096 envs of the global index range (weak scaling, no data-path collective; one stats all-reduce
after the timed region).  Inputs (state, model tables, the whole action tape) are resident in HBM before the timed region.

Episode phase.  A launch gets slower along an episode (random actions bring the hand to the table: more narrow-phase pairs and Newton
iterations; +30 % from step 0 to step 250), so timing the first steps of 4096 synchronised episodes measures the cheap part only.  By
default the envs are therefore put at episode steps spread uniformly over the horizon before anything is timed: env i starts with its
step counter at o_i = (197 i) mod 500 and `horizon` untimed launches are run, so every env passes its horizon once (on-device reset from
the bank) and then sits o_i genuine steps into its second episode.  Every timed launch then sees the steady-state mix of an RL rollout,
including the ~B/500 on-device episode resets per launch.  `--phase fresh` times synchronised episodes from their first step instead.

Stream groups.  One launch lasts as long as its slowest env (contact-rich envs take 3-4 x the median; measured 3.9 ms against a mean slot
load of 2.5 ms), and the envs are independent of each other.  By default the batch therefore steps as `--groups` (16) contiguous env blocks, each
on its own HIP stream (include/rsim.h rsim_set_stream_groups): `env.step()` still enqueues one control step of all 4096 envs, but a block's
step t + 1 starts when ITS slowest env has finished step t instead of waiting for the slowest env of the whole batch.  Same work, same results
(tools/groups_sweep.py: the reached state is bit-identical for every group count); the timed region is still exactly K steps of every env
between two full synchronisations.  `config.lockstep` reports the same K steps timed with `--groups 1` right after the main region.

Prints ONE JSON line on rank 0.  See DESIGN.md section 6 for the roofline / cpu_baseline definitions.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from robosuite_amd import backend, lift, mjcf, shard  # noqa: E402

ENVS_PER_GPU = 4096
N_SUB = 25
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s
HORIZON = 500
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4  # wave-instructions/s: 256 CUs x 4 SIMDs, one VALU wave-instruction per 4 cycles at 2.4 GHz


def algorithmic_bytes_per_env_step(flat, action_dim):
    """Compulsory HBM bytes one env-step moves through the fused kernel (fp32 words x 4), DESIGN.md section 6:
    read  action + qpos + qvel + qacc_warmstart + ctrl + time + controller state + per-env model deltas (cube: size3 rbound1 mass1 inertia3 subtree1 invw2 dofinvw6)
    write qpos + qvel + qacc_warmstart + ctrl + time + controller state + observation record + reward/done."""
    from robosuite_amd.backend import CSTATE
    OBS_DIM = len(lift.lift_task(flat, json.load(open(os.path.join(ROOT, "robosuite_amd", "assets", "lift_panda.cfg.json"))))["obs"])
    nq, nv, nu = flat.nq, flat.nv, flat.nu
    rd = action_dim + nq + nv + nv + nu + 1 + CSTATE + 17
    wr = nq + nv + nv + nu + 1 + CSTATE + OBS_DIM + 2
    return 4 * (rd + wr)


def cpu_baseline(flat, cfg, budget_s=12.0):
    """The CPU oracle (oracle/rsim_oracle.c: same pipeline, fp64, serial C) timed on this host's cores on a bounded sample of the
    same workload: `cores` threads (ctypes releases the GIL), each stepping its own seeded Lift env with its own action stream."""
    import threading

    from oracle.oracle import OracleController, OracleData, OracleModel

    cores = os.cpu_count() or 1
    counts = [0] * cores
    stop = time.perf_counter() + budget_s

    def work(k):
        sizes, qpos = lift.episode_setup(0, [k])
        f = flat.copy()
        for field, rows in lift.cube_model_rows(flat, sizes).items():
            f.arrays[field] = rows[0].reshape(f.arrays[field].shape)
        om = OracleModel(mjcf.to_blob(f)); od = OracleData(om); oc = OracleController(cfg)
        od.qpos[:] = qpos[0]; od.qvel[:] = 0; od.qacc_warmstart[:] = 0; od.forward(); oc.reset(od)
        acts = lift.env_actions([k], 4000)[:, 0].astype(np.float64)
        n = 0
        while time.perf_counter() < stop and n < len(acts):
            oc.env_step(od, acts[n], N_SUB)
            n += 1
        counts[k] = n

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    return {"value": sum(counts) / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{sum(counts)} env.steps of Lift/Panda/OSC_POSE ({cores} envs x ~{sum(counts)//cores} steps, {cores} threads, fp64 C oracle incl. C controllers) in {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # 200 timed steps after 20 warm-up steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--phase", choices=("staggered", "fresh"), default="staggered", help="episode phase of the envs when the timed region starts (module docstring)")
    ap.add_argument("--groups", type=int, default=16, help="env blocks stepped on their own HIP streams (module docstring); 1 = one launch per step")
    ap.add_argument("--no-lockstep", action="store_true", help="skip the second timed region (same K steps with one launch per step)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- one process per GPU under torch.distributed.run (RCCL rendezvous on 127.0.0.1),
        # same arguments; rank 0 of the child job prints the JSON line, this process only forwards the exit code
        import socket
        import subprocess

        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))

    rank, local_rank, world = shard.init_process_group()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        print(f"bench.py rank {rank}/{world}: no GPU visible (the HIP backend has no CPU fallback)", file=sys.stderr, flush=True)
        raise SystemExit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    adir = os.path.join(ROOT, "robosuite_amd", "assets")
    flat = mjcf.load_model(os.path.join(adir, "lift_panda.rsim"))
    cfg = json.load(open(os.path.join(adir, "lift_panda.cfg.json")))
    B = args.envs_per_gpu
    ids = shard.env_block(B * world, rank, world)
    K, W = args.steps, args.warmup
    P = HORIZON if args.phase == "staggered" else 0   # untimed pre-roll launches
    G = max(1, args.groups)
    K2 = 0 if (args.no_lockstep or G == 1) else K   # second region: the same number of steps, one launch per step
    env = lift.LiftBatch(flat, cfg, ids, device=local_rank, seed0=0, horizon=HORIZON, bank_episodes=2 + (P + W + K + K2) // HORIZON)  # config 2: episodes auto-reset at horizon 500
    tape = torch.tensor(lift.env_actions(ids, P + K + W + K2), device=dev)  # whole action tape resident in HBM
    env.batch.set_stream_groups(G)
    streams = [torch.cuda.ExternalStream(env.batch.group_stream(g), device=dev) for g in range(G)]   # the streams the fused kernel is launched on

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    if P:
        env.batch.set("ep_step", ((197 * ids) % HORIZON).astype(np.int32))   # keyed by the GLOBAL env id: independent of the GPU count
        for t in range(P):
            env.step(tape[t])
        tape = tape[P:]
    for t in range(W):
        env.step(tape[t])
    env.batch.sync(); torch.cuda.synchronize(); barrier()
    def timed(first, n, strs):
        ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in strs] for _ in range(n)]
        t0 = time.perf_counter()
        for t in range(n):
            for g, s_ in enumerate(strs):
                ev[t][g][0].record(s_)
            env.step(tape[first + t])
            for g, s_ in enumerate(strs):
                ev[t][g][1].record(s_)
        env.batch.sync(); torch.cuda.synchronize(); barrier()
        dt_ = shard.max_over_ranks(time.perf_counter() - t0, dev)
        return dt_, [[a.elapsed_time(b) for a, b in row] for row in ev]

    dt, evms = timed(W, K, streams)
    kern_ms = float(np.mean(evms))   # mean duration of one launch of the fused kernel (one env block of B / G envs) incl. its dispatch-order / reset passes
    if os.environ.get("RSIM_BENCH_TRACE"):
        print("per-step ms (group 0):", " ".join(f"{r[0]:.2f}" for r in evms), file=sys.stderr)
    lockstep = None
    if K2:
        env.batch.set_stream_groups(1)
        dt2, ev2 = timed(W + K, K2, [torch.cuda.ExternalStream(env.batch.stream(), device=dev)])
        lockstep = {"value": B * world * K2 / dt2, "ms_per_step": 1e3 * dt2 / K2, "kernel_ms": float(np.mean(ev2)), "steps": K2,
                    "note": "the next K steps of the same envs with one launch of all envs per step (--groups 1)"}

    st = shard.RolloutStats(dev)
    q = env.batch.tensor("qpos")
    # envs that hit the bad-state guard (RSIM_DIVERGED, MuJoCo's mj_checkPos semantics) or hold a non-finite coordinate
    st.add(env_steps=B * K, diverged=int((~torch.isfinite(q).all(dim=1)).sum().item()) + int((env.batch.tensor("diverged") > 0).sum().item()),
           reward_sum=float(env.reward().sum().item()), successes=int(env.success().sum().item()))
    overflow_envs = shard.max_over_ranks(float((env.batch.tensor("overflow") > 0).sum().item()), dev)   # envs that ever dropped a contact / constraint row
    if hasattr(env, "rollout_totals"):
        st.add(**env.rollout_totals())
    tot = st.allreduce()

    if rank == 0:
        OBS_DIM_REPORT = env.model.nobs
        abytes = algorithmic_bytes_per_env_step(flat, env.model.action_dim) * B / G   # one launch = one env block
        ach = abytes / (kern_ms * 1e-3) / 1e9
        # PMC-derived figures are only valid for the library build they were measured on: the files carry the sha of that build
        lib_sha = hashlib.sha256(open(backend.LIB_PATH, "rb").read()).hexdigest()[:16]

        def pmc(name, key):
            try:
                d = json.load(open(os.path.join(ROOT, "profiles", name)))
                return d.get(key) if d.get("lib_sha16") == lib_sha else None
            except Exception:
                return None

        traffic = pmc("hbm_traffic.json", "bytes_per_launch")            # tools/pmc_traffic.py (FETCH_SIZE / WRITE_SIZE passes), one launch of all B envs
        traffic_step = traffic
        if traffic:
            traffic = traffic / G                                         # per launch of one env block, like `achieved`
        valu = pmc("valu_count.json", "valu_per_env_substep")            # tools/pmc_valu.py (SQ_INSTS_VALU pass on this workload)
        issue = None
        if valu:
            rate = valu * B * N_SUB * K / dt   # wave-instructions per second of this GPU over the timed region (launches of different env blocks overlap)
            issue = {"bound": "valu-issue", "valu_instr_per_env_substep": valu, "achieved": rate / 1e9,
                     "peak": VALU_ISSUE_PEAK / 1e9, "unit": "G wave-instr/s", "frac": rate / VALU_ISSUE_PEAK,
                     "source": "profiles/valu_count.json (rocprofv3 --pmc SQ_INSTS_VALU on this build, same workload)"}
        out = {
            "metric": "env-steps/sec (whole node), Lift/Panda/OSC_POSE @4096 envs/GPU", "value": tot["env_steps"] / dt, "unit": "env-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * dt / K, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Lift/Panda/OSC_POSE, 25 substeps x dt 0.002 + OSC_POSE/GRIP per substep, fused in one launch (BASELINE configs[1])",
                       "envs_per_gpu": B, "global_envs": B * world, "n_sub": N_SUB, "per_env_seeded_reset": True, "horizon": HORIZON, "on_device_auto_reset": True,
                       "stream_groups": G, "lockstep": lockstep,
                       "episode_phase": ("uniform over the horizon: step counters offset by (197 i) mod 500, then 500 untimed pre-roll launches" if P else "fresh: all envs at step W of their first episode"),
                       "overflow_envs": int(overflow_envs), "lib_sha16": lib_sha, "obs_dim": OBS_DIM_REPORT, "sharding": f"env-block x{world}",
                       "diverged_envs": int(tot["diverged"]), "reward_sum": tot["reward_sum"], "successes": int(tot["successes"])},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_step", "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": abytes, "concurrent_launches": G, "traffic_per_control_step": traffic_step,
                         "aggregate_achieved": abytes * G * K / dt / 1e9,   # GB/s of all env blocks together (their launches overlap)
                         "note": "latency/VALU/LDS-bound by design (state LDS-resident for 25 substeps); see DESIGN.md section 6",
                         # the fraction that describes this kernel: VALU issue slots used (PMC instruction count of THIS build x measured rate); null
                         # when profiles/valu_count.json was measured on another build
                         "issue": issue},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(flat, cfg)
        elif world == 1:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
