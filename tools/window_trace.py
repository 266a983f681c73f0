#!/usr/bin/env python3
"""What a lockstep control step of a window waits for: per-step milliseconds (HIP events on the batch's stream, as bench.py takes them) next to the number of
envs on the wide capacity tier in that step and the number handed over in mid-step (redone), for the driver's invocation shape on several episode windows.

    python tools/window_trace.py [--config lift] [--steps 20] [--warmup 5] [--prerolls 500,700,900] [--out profiles/rNN_window_trace.json]

Same env construction, episode staggering and action tape as bench.py's headline region.  The tier snapshot between steps synchronises the stream (the step
durations are device-side event intervals, unaffected; the host gap between steps is not part of them).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from robosuite_amd import factory, lift, shard  # noqa: E402


def window(config, P, W, K, snapshot=True, wavelog=False):
    label, stem, B, dr, which = bench.CONFIGS[config]
    flat, cfg = factory.load_shipped(stem)
    ids = shard.env_block(B, 0, 1)
    env = bench.build_env(config, flat, cfg, ids, 0, 3 + (P + W + K) // bench.HORIZON)
    dev = torch.device("cuda", 0)
    tape = torch.tensor(lift.env_actions(ids, P + W + K, action_dim=env.model.action_dim), device=dev)
    n = [0]

    def step(t):
        if dr:
            env.batch.randomize_dynamics(seed=11, step=n[0]); n[0] += 1
        env.step(tape[t])

    if P:
        env.batch.set("ep_step", ((197 * ids) % bench.HORIZON).astype(np.int32))
    for t in range(P + W):
        step(t)
    env.batch.sync(); torch.cuda.synchronize()
    s = torch.cuda.ExternalStream(env.batch.stream(), device=dev)
    tier0 = env.batch.tier_snapshot() if snapshot else None
    rows = []
    slow = []
    if wavelog:   # per-env start / end ticks (100 MHz) and event counts of every launch: which env a step waits for
        env.batch.profile(True); env.batch.profile_env(-2)
    for t in range(K):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        step(P + W + t)
        b.record(s)
        if snapshot:
            tier1 = env.batch.tier_snapshot()
            rows.append([a, b, int((tier0 == 1).sum()), int(((tier0 == 0) & (tier1 == 1)).sum())])
            tier0 = tier1
        else:
            rows.append([a, b, None, None])
        if wavelog:
            w = env.batch.wavelog()
            t0_, t1_ = w[:, 2].astype(np.int64), w[:, 3].astype(np.int64)
            dur = (t1_ - t0_) / 100.0
            t1_ = np.where(dur > 200, t1_, 0)   # (the reset-observation pass that follows the step rewrites the records of the few envs it touches: ~45 us entries)
            e = int(np.argmax(t1_))   # the env that finished last
            slow.append({"last_env": e, "last_env_start_us": float((t0_[e] - t0_.min()) / 100.0), "last_env_dur_us": float(dur[e]), "span_us": float((t1_.max() - t0_.min()) / 100.0),
                         "dur_p50_us": float(np.percentile(dur, 50)), "dur_p99_us": float(np.percentile(dur, 99)), "dur_max_us": float(dur.max()), "longest_env": int(np.argmax(dur)), "longest_env_counts": w[int(np.argmax(dur)), 4:8].astype(np.int64).tolist(),
                         "last_env_counts_mpr_support_newton_cand": w[e, 4:8].astype(np.int64).tolist()})
    env.batch.sync(); torch.cuda.synchronize()
    ms = [r[0].elapsed_time(r[1]) for r in rows]
    cn = env.batch.tensor("cap_need").view(B, 2)
    out = {"config": config, "preroll": P, "warmup": W, "steps": K, "ms": [round(x, 3) for x in ms], "wide_list": [r[2] for r in rows], "redone": [r[3] for r in rows],
           "ms_min": min(ms), "ms_p50": float(np.median(ms)), "ms_max": max(ms), "value_from_events": B * K / (sum(ms) * 1e-3),
           "max_contacts_needed": int(cn[:, 0].max().item()), "max_rows_needed": int(cn[:, 1].max().item()), "capacity": [env.batch.maxcon, env.batch.maxefc]}
    if wavelog:
        out["per_step_envs"] = slow
        # three more steps, each with the phase accumulators restricted to the env that ran longest in the step before: where its time goes
        prof = []
        for t in range(3):
            e = slow[-1]["longest_env"] if t == 0 else prof[-1]["next"]
            env.batch.profile(True); env.batch.profile_env(e)
            step(P + W + K - 3 + t)   # (tape rows reused: the content of the actions does not matter here)
            env.batch.sync()
            w = env.batch.wavelog(); p = env.batch.profile(True)
            dur = (w[:, 3].astype(np.int64) - w[:, 2].astype(np.int64)) / 100.0
            ns = max(1, p["n_sub"])
            cyc = {k: int(v / ns) for k, v in p.items() if not k.startswith("n_") and not k.startswith("x") and k not in ("boxbox", "mpr", "plane") and v}
            prof.append({"env": int(e), "dur_us": float(dur[e]), "ticks_per_substep": cyc, "narrow_split": {k: int(p[k] / ns) for k in ("boxbox", "mpr", "plane")},
                         "per_substep": {k[2:]: round(p[k] / ns, 2) for k in p if k.startswith("n_") and k != "n_sub"}, "next": int(np.argmax(np.where(dur > 200, dur, 0)))})
        out["slowest_env_profile"] = prof
        env.batch.profile(False)
    env.bank_quiesce(); env._bank_stop()
    del env, tape
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="lift")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--prerolls", default="500,700,900")
    ap.add_argument("--no-snapshot", action="store_true", help="no stream synchronisation between steps (the steps queue back to back as in bench.py)")
    ap.add_argument("--wavelog", action="store_true", help="arm the in-kernel wave log: per step the env that finished last, its start, duration and event counts")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    res = []
    for P in [int(x) for x in a.prerolls.split(",")]:
        r = window(a.config, P, a.warmup, a.steps, not a.no_snapshot, a.wavelog)
        res.append(r)
        print(f"preroll {P}: {r['value_from_events'] / 1e3:.0f} K env-steps/s  ms min {r['ms_min']:.2f} p50 {r['ms_p50']:.2f} max {r['ms_max']:.2f}  need {r['max_contacts_needed']} / {r['max_rows_needed']}", flush=True)
        print("   ms       " + " ".join(f"{x:5.2f}" for x in r["ms"]))
        if not a.no_snapshot:
            print("   on tier  " + " ".join(f"{x:5d}" for x in r["wide_list"]))
            print("   redone   " + " ".join(f"{x:5d}" for x in r["redone"]), flush=True)
        for t, e in enumerate(r.get("per_step_envs", [])):
            print(f"   step {t:2d}: span {e['span_us']:6.0f} us  last env {e['last_env']:5d} started {e['last_env_start_us']:6.0f} ran {e['last_env_dur_us']:6.0f}  (p50 {e['dur_p50_us']:.0f} p99 {e['dur_p99_us']:.0f} max {e['dur_max_us']:.0f} env {e['longest_env']} mpr/support/newton/cand {e['longest_env_counts']})")
        for q in r.get("slowest_env_profile", []):
            print(f"   env {q['env']} ({q['dur_us']:.0f} us) ticks/substep {q['ticks_per_substep']}  narrow {q['narrow_split']}  per substep {q['per_substep']}")
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
