"""bench.py's control flow and output contract, run on the CPU against a stand-in batch (no kernel, no oracle): the driver parses ONE JSON line from rank 0, so a
NameError on a path only a GPU box reaches would cost the round its measurement.  Everything device-side is replaced (torch.cuda, the env the factory would build, the
child processes of the secondary regions); what runs is main() itself: argument handling, pre-roll / warm-up / timed regions, the order of the regions (the secondary
configurations directly behind the headline region, ahead of the open-loop and double-buffered ones), statistics, PMC evidence lookup and the JSON assembly."""
import json
import os
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "robosuite_amd", "librsim_hip.so")), reason="needs the built library (its sha goes into the line)")


class _Event:
    def __init__(self, enable_timing=False):
        pass

    def record(self, stream=None):
        pass

    def elapsed_time(self, other):
        return 1.5


class _Batch:
    maxcon, maxefc = 16, 64

    def __init__(self, n, log):
        self.n, self.log = n, log

    def stream(self):
        return 0

    def group_stream(self, g):
        return g

    def sync(self):
        pass

    def set(self, name, value):
        self.log.append(("set", name))

    def set_stream_groups(self, g):
        self.log.append(("groups", g))

    def randomize_dynamics(self, seed, step):
        self.log.append(("dr", step))

    def tier_stats(self):
        return (3 * self.log.count(("step", self.n)), self.log.count(("step", self.n)))

    def tensor(self, name):
        if name == "qpos":
            return torch.zeros(self.n, 9)
        if name == "cap_need":
            return torch.ones(self.n * 2, dtype=torch.int32)
        return torch.zeros(self.n, dtype=torch.int32)


class _Env:
    def __init__(self, n, log):
        self.batch = _Batch(n, log)
        self.model = types.SimpleNamespace(action_dim=7, nobs=60, cstate_size=32, int=lambda name: 128)
        self.n, self.log, self.steps = n, log, 0

    def step(self, a):
        assert a.shape == (self.n, 7)
        self.steps += 1
        self.log.append(("step", self.n))

    def bank_stats(self):
        return {"steps": self.steps, "polls": self.steps // 2, "rows": 3 * self.steps, "tick_s": 1e-6 * self.steps, "upkeep_s": 2e-6 * self.steps}

    def bank_quiesce(self):
        self.log.append(("quiesce", self.n))

    def _bank_stop(self):
        pass

    def _bank_patch_offsets(self):
        return [0, 1, 2]

    def reward(self):
        return torch.zeros(self.n)

    def success(self):
        return torch.zeros(self.n)


def _run_main(monkeypatch, capsys, argv, child=None):
    import bench

    log = []
    fake_cuda = types.SimpleNamespace(is_available=lambda: True, set_device=lambda d: None, synchronize=lambda: None, Event=_Event,
                                      ExternalStream=lambda s, device=None: s, empty_cache=lambda: None)

    class _Torch:                                   # the real torch, except that "cuda" devices are the CPU and torch.cuda is the stand-in above
        cuda = fake_cuda

        def __getattr__(self, k):
            return getattr(torch, k)

        def device(self, *a):
            return torch.device("cpu")

    monkeypatch.setattr(bench, "torch", _Torch())
    monkeypatch.setattr(bench, "build_env", lambda config, flat, cfg, ids, device, episodes: _Env(len(ids), log))
    monkeypatch.setattr(bench.shard, "max_over_ranks", lambda x, device="cpu": float(x))

    def default_child(cmd, **kw):
        assert "--secondary-only" in cmd
        oc = cmd[cmd.index("--secondary-only") + 1]
        log.append(("child", oc, cmd[cmd.index("--other-steps") + 1], cmd[cmd.index("--other-preroll") + 1]))
        return types.SimpleNamespace(returncode=0, stderr="", stdout="noise\n" + json.dumps({"workload": oc, "value": 1.0, "ms_per_step": 2.0}) + "\n")

    monkeypatch.setattr(subprocess, "run", child or default_child)
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    bench.main()
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines                  # ONE JSON line
    return json.loads(lines[0]), log


def test_default_command_flow_and_contract(monkeypatch, capsys):
    d, log = _run_main(monkeypatch, capsys, ["--steps", "4", "--warmup", "2", "--preroll", "3", "--envs-per-gpu", "8", "--groups", "2", "--no-cpu-baseline"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f32"
    assert d["unit"] == "env-steps/s" and d["value"] > 0 and d["vs_baseline"] is None and "Lift" in d["metric"] and "workload" in d["config"]
    assert d["config"]["envs_per_gpu"] == 8 and d["config"]["open_loop"]["stream_groups"] == 2 and d["config"]["double_buffered"]["halves"] == [4, 4]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["kernel_ms"] == 1.5 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert isinstance(r["traffic"], (float, str)) and r["traffic"] != "absent"       # a figure, or "stale:<sha>" said out loud
    # the three secondary configurations, each in its child process, on the quick protocol ...
    oc = d["config"]["other_configs"]
    assert set(oc) == {"stack", "peg", "pickplace"} and all(v["value"] == 1.0 for v in oc.values())
    kids = [e for e in log if e[0] == "child"]
    assert kids == [("child", "stack", "50", "500"), ("child", "peg", "50", "500"), ("child", "pickplace", "20", "500")]       # pre-roll = the horizon: steady state
    # the slow-window diagnostics of the round-5 review: per-step distribution and what the capacity tier did in the timed region, at top level
    assert d["step_ms"] == {"min": 1.5, "p50": 1.5, "p90": 1.5, "max": 1.5} and d["tier_env_steps"] == 3 * 4 and d["tier_changes_in_mid_step"] == 4
    assert d["collective"] == "none (one rank)" and d["config"]["steady_state"] is False and d["config"]["episode_window"] == [5, 9]
    # ... directly behind the headline region: 3 pre-roll + 2 warm-up + 4 timed steps of the full batch, the children, THEN stream groups and the two half batches
    first_child, groups_on = log.index(kids[0]), log.index(("groups", 2))
    assert [e for e in log[:first_child] if e[0] == "step"] == [("step", 8)] * 9 and log[first_child - 1] == ("quiesce", 8)
    assert first_child < groups_on and not any(e == ("step", 4) for e in log[:groups_on])
    assert [e for e in log[groups_on:] if e == ("step", 8)] == [("step", 8)] * 4 and sum(e == ("step", 4) for e in log) == 2 * (3 + 2 + 4)


def test_other_config_line_and_a_failing_child(monkeypatch, capsys):
    d, log = _run_main(monkeypatch, capsys, ["--config", "stack", "--steps", "2", "--warmup", "1", "--preroll", "0", "--envs-per-gpu", "4", "--no-open-loop", "--no-cpu-baseline"])
    assert "Stack" in d["metric"] and d["config"]["other_configs"] is None and d["config"]["open_loop"] is None and not any(e[0] == "child" for e in log)

    # a child that dies (a fault of the GPU queue aborts the process that owns it) is reported in the record; the headline line stands
    def dying(cmd, **kw):
        return types.SimpleNamespace(returncode=-6, stderr="Memory access fault by GPU node-2\n", stdout="")

    d, log = _run_main(monkeypatch, capsys, ["--steps", "2", "--warmup", "1", "--preroll", "0", "--envs-per-gpu", "4", "--no-open-loop", "--no-cpu-baseline"], child=dying)
    oc = d["config"]["other_configs"]
    assert set(oc) == {"stack", "peg", "pickplace"} and all("Memory access fault" in v["error"] for v in oc.values()) and d["value"] > 0


def test_reference_baseline_harness_is_blocked_loudly_without_mujoco_and_runs_on_the_shim():
    """tools/bench_reference.py (BASELINE.md section 2, B-ref): without the `mujoco` wheel it must say so and exit 3 -- never substitute anything silently; with
    `--backend shim` the same harness (N processes, barrier, suite.make with the B-ref kwargs, env.step loop) runs end to end on the fp64 oracle and labels its number
    "port" / B-cpu, so that the day a wheel exists only the backend changes."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "bench_reference.py")
    if importlib.util.find_spec("mujoco") is None:
        r = subprocess.run([sys.executable, tool, "--procs", "1", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        assert r.returncode == 3 and d["baseline"] == "B-ref" and "not importable" in d["error"]
    if not os.path.isdir("/root/reference/robosuite"):
        pytest.skip("no reference checkout: the shim run needs robosuite's own Python")
    r = subprocess.run([sys.executable, tool, "--procs", "2", "--steps", "3", "--warmup", "1", "--backend", "shim"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["baseline"] == "B-cpu" and d["kind"] == "port" and d["cores"] == 2 and d["value"] > 0 and "NOT MuJoCo" in d["sample"] and len(d["per_worker_steps_per_s"]) == 2
