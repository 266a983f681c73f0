"""How well can the dispatch order of a lockstep launch be predicted?  Records per-env wavefront durations and event counts of N consecutive control steps
(wave log), then replays list scheduling offline (envs dispatched in a given order onto S resident slots, each env taking its MEASURED duration) for several
predictors of the duration: the launch's own durations (oracle = best possible order), durations of step t - 1 / t - 2 (what k_order uses: t - 2), decaying
maxima and means, event counts, index order.
Usage (GPU box): python tools/sched_study.py [task=stack] [nskip=300] [nsteps=12] [B=4096]"""
import json, os, sys, heapq
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd import lift, mjcf, peg_in_hole, stack
TASK = sys.argv[1] if len(sys.argv) > 1 else "stack"
nskip = int(sys.argv[2]) if len(sys.argv) > 2 else 300
N = int(sys.argv[3]) if len(sys.argv) > 3 else 12
B = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
STEM, CLS, SLOTS = {"lift": ("lift_panda", lift.LiftBatch, 2048), "stack": ("stack_panda", stack.StackBatch, 2048), "peg": ("peg_baxter_joint_velocity", peg_in_hole.PegBatch, 1280)}[TASK]   # resident envs: 8 / 8 / 5 per CU (round 6)
adir = os.path.join(ROOT, "robosuite_amd", "assets")
flat = mjcf.load_model(os.path.join(adir, STEM + ".rsim")); cfg = json.load(open(os.path.join(adir, STEM + ".cfg.json")))
env = CLS(flat, cfg, np.arange(B), seed0=0)
tape = torch.tensor(lift.env_actions(np.arange(B), nskip + N, action_dim=env.model.action_dim), device="cuda")
if os.environ.get("SPREAD", "1") == "1":
    env.batch.set("ep_step", ((197 * np.arange(B)) % 500).astype(np.int32))   # the bench's episode-phase mix
for t in range(nskip): env.step(tape[t])
env.batch.sync(); env.batch.profile(True); env.batch.profile_env(-2)
D, C, SP = [], [], []
for t in range(N):
    env.step(tape[nskip + t]); env.batch.sync()
    w = env.batch.wavelog()
    D.append((w[:, 3].astype(np.int64) - w[:, 2].astype(np.int64)) / 100.0); C.append(w[:, 4:8].astype(np.float64))
    SP.append((w[:, 3].astype(np.int64).max() - w[:, 2].astype(np.int64).min()) / 100.0)
D, C = np.array(D), np.array(C)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, 'gpurun_out', f'sched_study_{TASK}.npz'), D=D.astype(np.float32), C=C.astype(np.float32), span=np.array(SP))


def makespan(dur, order, slots=SLOTS):
    h = [0.0] * slots
    heapq.heapify(h)
    end = 0.0
    for e in order:
        t0 = heapq.heappop(h); t1 = t0 + dur[e]; end = max(end, t1); heapq.heappush(h, t1)
    return end


print(f"{TASK} B={B} slots={SLOTS}: steps {nskip}..{nskip + N - 1}; measured span us per step:", [int(x) for x in SP])
print("durations us: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f; sum/slots %.0f" % (D.mean(), np.percentile(D, 50), np.percentile(D, 90), np.percentile(D, 99), D.max(), D.sum(1).mean() / SLOTS))
for lag in (1, 2, 3):
    print(f"corr(dur[t], dur[t-{lag}]) = %.3f" % np.mean([np.corrcoef(D[t], D[t - lag])[0, 1] for t in range(lag, N)]))
preds = {}
for t in range(4, N):
    P = {"oracle (own duration)": D[t], "dur[t-1]": D[t - 1], "dur[t-2] (k_order)": D[t - 2], "max(dur[t-2], dur[t-3])": np.maximum(D[t - 2], D[t - 3]),
         "mean(dur[t-2..t-4])": D[t - 4:t - 1].mean(0), "max(dur[t-2..t-4])": D[t - 4:t - 1].max(0),
         "decay max 0.5": np.maximum.reduce([D[t - 2], 0.5 * D[t - 3] + 0.5 * D[t - 2].mean(), 0.25 * D[t - 4]]),
         "n_cand[t-2]": C[t - 2][:, 3], "n_newton[t-2]": C[t - 2][:, 2], "index order": -np.arange(B, dtype=np.float64),
         "n_cand[t-2] * 1e3 + dur[t-2]": C[t - 2][:, 3] * 1e3 + D[t - 2]}
    for k, v in P.items():
        preds.setdefault(k, []).append(makespan(D[t], np.argsort(-v, kind="stable")))
print("list-scheduling makespan us (mean over steps) by dispatch-order predictor; lower bound sum/slots = %.0f, longest env %.0f" % (D[4:].sum(1).mean() / SLOTS, D[4:].max(1).mean()))
for k, v in preds.items():
    print(f"  {k:34s} {np.mean(v):8.0f}")

# round 6: how much of a step's duration is predictable from what the previous step knew?  Least squares of dur[t] on (1, dur[t-1], dur[t-2], the four event counts of t-1),
# fitted on the first half of the recorded steps, replayed on the second half
h = max(5, N // 2)
X = lambda t: np.column_stack([np.ones(B), D[t - 1], D[t - 2], C[t - 1]])
A = np.concatenate([X(t) for t in range(2, h)]); y = np.concatenate([D[t] for t in range(2, h)])
w = np.linalg.lstsq(A, y, rcond=None)[0]
res = {"oracle (own duration)": [], "dur[t-1]": [], "least squares on t-1": [], "dur[t-1] + 0.5 (dur[t-1] - dur[t-2])+": []}
for t in range(h, N):
    res["oracle (own duration)"].append(makespan(D[t], np.argsort(-D[t], kind="stable")))
    res["dur[t-1]"].append(makespan(D[t], np.argsort(-D[t - 1], kind="stable")))
    res["least squares on t-1"].append(makespan(D[t], np.argsort(-(X(t) @ w), kind="stable")))
    res["dur[t-1] + 0.5 (dur[t-1] - dur[t-2])+"].append(makespan(D[t], np.argsort(-(D[t - 1] + 0.5 * np.maximum(D[t - 1] - D[t - 2], 0)), kind="stable")))
print("second half of the steps (least squares fitted on the first): weights", np.round(w, 3), " corr(pred, dur) %.3f vs corr(dur[t-1], dur) %.3f" % (
    np.mean([np.corrcoef(X(t) @ w, D[t])[0, 1] for t in range(h, N)]), np.mean([np.corrcoef(D[t - 1], D[t])[0, 1] for t in range(h, N)])))
for k, v in res.items():
    print(f"  {k:42s} {np.mean(v):8.0f}")
print("measured span of those steps %.0f; sum / slots %.0f; longest env %.0f" % (np.mean(SP[h:]), D[h:].sum(1).mean() / SLOTS, D[h:].max(1).mean()))
