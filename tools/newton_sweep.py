"""fp32 stopping rules of the Newton solver (rsim_step.hip solve_newton: RSIM_NEWTON_NS / NA / NG): iterations, launch time and accuracy against
the fp64 oracle for a list of settings.  Throughput / iteration counts: the bench workload from episode step `nskip` on (4096 Lift envs, lockstep).
Accuracy: ONE set of reached states (recorded from the first setting's rollout: the Newton-heaviest envs of a launch plus envs spread over the
batch) is loaded into a batch of every setting and evaluated by forward(), so the settings are compared on identical inputs.
Usage (GPU box): python tools/newton_sweep.py [nskip=200] "ns,na,ng" "ns,na,ng" ..."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd import lift, mjcf
from tests.test_full_size_parity import compare_reached_states, spread
nskip = int(sys.argv[1]) if len(sys.argv) > 1 else 200
settings = [tuple(float(x) for x in s.split(",")) for s in sys.argv[2:]] or [(0, 0, 0)]
B = 4096
adir = os.path.join(ROOT, "robosuite_amd", "assets")
flat = mjcf.load_model(os.path.join(adir, "lift_panda.rsim")); cfg = json.load(open(os.path.join(adir, "lift_panda.cfg.json")))
tape = torch.tensor(lift.env_actions(np.arange(B), nskip + 41), device="cuda")
saved = None
for ns, na, ng in settings:
    os.environ["RSIM_NEWTON_NS"], os.environ["RSIM_NEWTON_NA"], os.environ["RSIM_NEWTON_NG"] = str(ns), str(na), str(ng)
    env = lift.LiftBatch(flat, cfg, np.arange(B), seed0=0)
    for t in range(nskip): env.step(tape[t])
    env.batch.sync(); t0 = time.perf_counter()
    for t in range(nskip, nskip + 20): env.step(tape[t])
    env.batch.sync(); ms = 1e3 * (time.perf_counter() - t0) / 20
    env.batch.profile(True); env.batch.profile_env(-1)
    env.step(tape[nskip + 20]); env.batch.sync()
    w = env.batch.wavelog(); p = env.batch.profile(False)
    cnt = w[:, 4:8].astype(np.int64)
    if saved is None:
        pick = np.unique(np.concatenate([np.argsort(-cnt[:, 2])[:48], spread(B, 24)]))
        saved = {k: env.batch.get(k) for k in ("qpos", "qvel", "qacc_warmstart", "ctrl")} | {"pick": pick}
    b = env.batch
    for k in ("qpos", "qvel", "qacc_warmstart", "ctrl"): b.set(k, saved[k])
    res = compare_reached_states(flat, b, saved["pick"])
    ok = [r for r in res if r["same"]]
    f = np.array([r["force"] / max(1.0, r["fscale"]) for r in ok]); a = np.array([r["qacc"] / max(1.0, r["ascale"]) for r in ok])
    print(f"NS {ns:g} NA {na:g} NG {ng:g}: {ms:.3f} ms/step lockstep; newton {p['n_newton'] / max(1, p['n_sub']):.3f} ls {p['n_ls'] / max(1, p['n_sub']):.3f} per env-substep; "
          f"per launch newton mean {cnt[:, 2].mean():.1f} p99 {np.percentile(cnt[:, 2], 99):.0f} max {cnt[:, 2].max()}; same {len(saved['pick'])} states vs oracle ({len(ok)} agree in structure): "
          f"rel dforce max {f.max():.1e} median {np.median(f):.1e}; rel dqacc max {a.max():.1e} median {np.median(a):.1e}; forward() newton iterations mean {b.get('niter')[saved['pick']].mean():.2f}", flush=True)
    del env
