#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the fused control-step hot path on MI355X (BASELINE.json metric).

One "step" = one robosuite `env.step(action)` for every env of the batch (25 physics substeps at dt = 0.002 + 25 controller evaluations +
set_goal + observation / reward; reference environments/base.py:467-521), i.e. one launch of the fused kernel per env.
Workload at N = 1 = BASELINE configs[1]: Lift / Panda / OSC_POSE, 4096 envs on one GPU, per-env seeded episodes (cube size, arm noise, cube
pose) and per-env action streams.  `--config stack | peg | pickplace` runs BASELINE configs[2..4] under the same
protocol and JSON contract (Stack 4096 envs, TwoArmPegInHole / Baxter / JOINT_VELOCITY 2048, PickPlace / IIWA 8192 with the dynamics re-drawn
before every control step).  N > 1: every rank owns `envs-per-gpu` envs of the global index range (weak scaling, no data-path collective; one
stats all-reduce after the timed region).  State, model tables and the whole action tape are resident in HBM before the timed region.

`value` is the LOCKSTEP figure: every control step is handed the actions of ALL envs and the next one starts when all of them have finished --
what a closed-loop policy that consumes the whole observation batch gets.  (Round 2 reported the stream-groups figure as `value`; that one is
`config.open_loop` now: env blocks on their own streams run ahead of each other, which only an action tape recorded beforehand allows.)

Episode phase.  A launch gets slower along an episode (random actions bring the hand to the table: more narrow-phase pairs and Newton
iterations; +30 % from step 0 to step 250), so timing the first steps of synchronised episodes measures the cheap part only.  By default the
envs are therefore put at episode steps spread uniformly over the horizon before anything is timed: env i starts with its step counter at
o_i = (197 i) mod 500 and `--preroll` (default: the horizon) untimed launches are run, so every env passes its horizon once (on-device reset
from the ring) and then sits o_i genuine steps into its second episode.  Every timed launch then sees the steady-state mix of an RL rollout,
including the ~B/500 on-device episode resets per launch and the asynchronous upkeep of the reset ring (`config.reset_ring`).

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

from robosuite_amd import backend, factory, lift, mjcf, shard  # noqa: E402

N_SUB = 25
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s
HORIZON = 500
# VALU wave-instructions per second the chip can issue.  MEASURED (tools/ubench/valu_peak.hip, profiles/r03_a_valu_peak.txt): independent v_fma_f32
# streams at 8 wavefronts per SIMD reach 1177.8 G/s = one wave64 instruction per 2.09 cycles and SIMD; the guide's figure (MI355X_MICROARCH.md, wave
# scheduling: SIMD-32, a wave64 VALU instruction issues over 2 cycles) is 256 CUs x 4 SIMDs x 2.4 GHz / 2 = 1228.8 G/s.  Rounds 1-2 used 614 G/s
# (4 cycles per instruction, the SIMD-16 figure of earlier parts), which overstated every issue fraction 2x.
VALU_ISSUE_PEAK = 1177.8e9
VALU_ISSUE_PEAK_THEORY = 256 * 4 * 2.4e9 / 2

# BASELINE configs[1..4]: name -> (task class name for the report, asset stem, envs per GPU, per-step dynamics randomisation)
CONFIGS = {
    "lift": ("Lift/Panda/OSC_POSE", "lift_panda", 4096, False, "configs[1]"),
    "stack": ("Stack/Panda/OSC_POSE", "stack_panda", 4096, False, "configs[2]"),
    "peg": ("TwoArmPegInHole/Baxter/JOINT_VELOCITY", "peg_baxter_joint_velocity", 2048, False, "configs[3]"),
    "pickplace": ("PickPlace/IIWA+Robotiq140/OSC_POSE + per-step dynamics randomisation", "pickplace_iiwa", 8192, True, "configs[4]"),
}


def build_env(config, flat, cfg, ids, device, episodes):
    from robosuite_amd import peg_in_hole, pick_place, stack

    if config == "lift":
        return lift.LiftBatch(flat, cfg, ids, device=device, seed0=0, horizon=HORIZON, bank_episodes=episodes)  # episodes auto-reset at horizon 500
    if config == "stack":
        return stack.StackBatch(flat, cfg, ids, device=device, seed0=0, horizon=HORIZON, bank_episodes=episodes)
    if config == "peg":
        return peg_in_hole.PegBatch(flat, cfg, ids, device=device, seed0=0, horizon=HORIZON, bank_episodes=episodes)
    env = pick_place.PickPlaceBatch(flat, cfg, ids, device=device, seed0=0, horizon=HORIZON, bank_episodes=episodes, per_env_params=True)
    env.batch.dr_save_defaults()
    return env


def algorithmic_bytes_per_env_step(env, flat, dr):
    """Compulsory HBM bytes one env-step moves through the fused kernel (fp32 words x 4), DESIGN.md section 6:
    read  action + qpos + qvel + qacc_warmstart + ctrl + time + controller state + the per-env model values that change per episode
    write qpos + qvel + qacc_warmstart + ctrl + time + controller state + observation record + reward/done.
    With per-step dynamics randomisation every env's float table is re-drawn (read defaults, write table) and its constant block rebuilt
    (write) and read once by the control step."""
    nq, nv, nu = flat.nq, flat.nv, flat.nu
    cs, nobs, adim = env.model.cstate_size, env.model.nobs, env.model.action_dim
    per_episode = len(env._bank_patch_offsets())
    rd = adim + nq + nv + nv + nu + 1 + cs + per_episode
    wr = nq + nv + nv + nu + 1 + cs + nobs + 2
    words = rd + wr
    if dr:
        words += 2 * env.model.int("float_table_size") + 2 * (env.model.int("constant_block_bytes") // 4)
    return 4 * words


def cpu_baseline(config, flat, cfg, budget_s=12.0):
    """The CPU oracle (oracle/rsim_oracle.c: same pipeline, fp64, serial C) timed on this host's cores on a bounded sample of the same workload:
    `cores` threads (ctypes releases the GIL), each stepping its own env of the configuration's model with its own action stream."""
    import threading

    from oracle.oracle import OracleController, OracleData, OracleModel, env_step_parts
    from robosuite_amd import peg_in_hole, pick_place, stack

    cores = os.cpu_count() or 1
    counts = [0] * cores
    stop = time.perf_counter() + budget_s
    def work(k):
        f = flat
        if config == "lift":
            sizes, qpos = lift.episode_setup(0, [k])
            f = flat.copy()
            for field, rows in lift.cube_model_rows(flat, sizes).items():
                f.arrays[field] = rows[0].reshape(f.arrays[field].shape)
            q0 = qpos[0]
        elif config == "stack":
            q0 = stack.episode_setup(0, [k])[0]
        elif config == "peg":
            q0 = peg_in_hole.episode_setup(0, [k])[0]
        else:
            q0 = pick_place.episode_setup(cfg, flat.nq, 0, [k])[0]
        om = OracleModel(mjcf.to_blob(f)); od = OracleData(om)
        od.qpos[:] = q0; od.qvel[:] = 0; od.qacc_warmstart[:] = 0; od.forward()
        if "parts" in cfg and not str(cfg.get("type", "")).startswith("OSC"):
            parts = [(OracleController(p), len(p["input_min"])) for p in cfg["parts"]]
            for c, _ in parts:
                c.reset(od)
            na = sum(n for _, n in parts)
            step = lambda a: env_step_parts(od, parts, a, N_SUB)   # noqa: E731
        else:
            oc = OracleController(cfg); oc.reset(od)
            na = len(cfg["input_min"]) + (1 if cfg.get("grip_act") else 0)
            step = lambda a: oc.env_step(od, a, N_SUB)   # noqa: E731
        acts = lift.env_actions([k], 4000, action_dim=na)[:, 0].astype(np.float64)
        n = 0
        while time.perf_counter() < stop and n < len(acts):
            step(acts[n])
            n += 1
        counts[k] = n

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    return {"value": sum(counts) / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{sum(counts)} env.steps of {CONFIGS[config][0].split(' +')[0]} ({cores} envs x ~{sum(counts)//cores} steps, {cores} threads, fp64 C oracle incl. C controllers"
                      f"{'' if config != 'pickplace' else ', without the per-step dynamics randomisation'}) in {dt:.1f} s"}


def pmc_evidence(name, key, lib_sha, config=None):
    """A PMC-derived figure from profiles/<name>, valid only for the kernel it was measured on: the file carries the sha of the library build and (since round 5,
    session 13) of the configuration's own code object.  The library holds one code object per kernel configuration; a later build that changed another
    configuration runs the bit-identical kernel for this one, and its evidence stands.  Evidence of another kernel is reported as the string "stale:<its library
    sha>", a missing file as "absent" -- never as a silent null (round-4 review)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return "absent"
    if d.get("lib_sha16") == lib_sha and d.get("tuning_sha16", backend.tuning_sha16()) == backend.tuning_sha16():
        return d.get(key)
    if config is not None and d.get("code_sha16"):
        # another library build: the figure stands only if everything it depends on is bit-identical -- the configuration's code object, the code object of the
        # capacity tier its envs move to, and the host-side dispatch / solver settings (rsim_tuning_defaults: round-5 advisor finding)
        try:
            from tools.kernel_resources import config_code_sha16, wide_code_sha16
            if (config_code_sha16(backend.LIB_PATH, config) == d["code_sha16"] and d.get("wide_code_sha16") == wide_code_sha16(backend.LIB_PATH, config)
                    and d.get("tuning_sha16") == backend.tuning_sha16()):
                return d.get(key)
        except Exception:
            pass
    return f"stale:{d.get('lib_sha16')}"


# (timed steps, untimed pre-roll launches) of the secondary regions of the default command: the quick-bench protocol of tools/ab_many.sh where a control step takes a few
# milliseconds (2 - 3 s per region), a shorter one for PickPlace (91 ms per step).  Round 5, session 15: ten steps after fifty said 603 K for Stack where the longer
# protocols say 647 K (quick) / 620 - 650 K (full): the first control steps after a cold start carry the redo passes in which envs find their capacity tier.
# Round 6: the pre-roll of every secondary region is the horizon, as for the headline -- every env has then passed an on-device reset and sits at its staggered
# episode step, the tiers are found and the record says "steady_state": true (a shorter --other-preroll says false).  The round-5 PickPlace child (10 steps after 50)
# read 6 % above the full protocol.
OTHER_REGION = {"stack": (50, HORIZON), "peg": (50, HORIZON), "pickplace": (20, HORIZON)}


def other_region(args, config):
    k, p = OTHER_REGION.get(config, (10, 50))
    return (args.other_steps if args.other_steps >= 0 else k), (args.other_preroll if args.other_preroll >= 0 else p)


def step_stats(ms):
    """Distribution of the per-step durations of a timed region (HIP events around every control step on the stream it runs on): a slow window explains itself."""
    a = np.asarray(ms, dtype=np.float64)
    return {"min": float(a.min()), "p50": float(np.percentile(a, 50)), "p90": float(np.percentile(a, 90)), "max": float(a.max())}


def double_buffered_region(config, flat, cfg, ids, local_rank, dev, world, dr, P, K, adim):
    """Closed-loop compatible: the batch as two halves (two rsim batches on their own streams) stepped alternately; the host waits for a half's step t (the point
    where a policy would read that half's observations) before it issues that half's step t + 1, the other half steps meanwhile.  Returns (seconds, bank_stale)."""
    halves = [ids[:len(ids) // 2], ids[len(ids) // 2:]]
    envs2 = [build_env(config, flat, cfg, h, local_rank, 3 + (P + K) // HORIZON) for h in halves]
    tapes2 = [torch.tensor(lift.env_actions(h, P + K, action_dim=adim), device=dev) for h in halves]
    if P:
        for e2, h in zip(envs2, halves):
            e2.batch.set("ep_step", ((197 * h) % HORIZON).astype(np.int32))
    drs = [0, 0]

    def step2(k, t):
        if dr:
            envs2[k].batch.randomize_dynamics(seed=11, step=drs[k]); drs[k] += 1
        envs2[k].step(tapes2[k][t])

    for t in range(P):
        for k in (0, 1):
            step2(k, t)
    for e2 in envs2:
        e2.batch.sync()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for t in range(K):
        for k in (0, 1):
            envs2[k].batch.sync()          # half k's observations of step t - 1 are complete: its policy can act
            step2(k, P + t)
    for e2 in envs2:
        e2.batch.sync()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt3 = shard.max_over_ranks(time.perf_counter() - t0, dev)
    stale3 = sum(float(e2.batch.tensor("bank_stale").sum().item()) for e2 in envs2)
    for e2 in envs2:
        e2.bank_quiesce(); e2._bank_stop()
    return dt3, int(stale3), [len(h) for h in halves]


def secondary_region(config, rank, local_rank, world, dev, K, P, double_buffer=True):
    """K lockstep control steps of another BASELINE configuration at its stated batch size, after P untimed launches from staggered episode steps (same
    protocol as the headline region): ms per step and its distribution, env-steps/s, dropped / diverged envs, tier statistics, the VALU issue fraction and
    HBM traffic per launch when PMC evidence of this build exists, and the two-half (double-buffered) figure."""
    label, stem, B, dr, which = CONFIGS[config]
    flat, cfg = factory.load_shipped(stem)
    ids = shard.env_block(B * world, rank, world)
    env = build_env(config, flat, cfg, ids, local_rank, 3 + (P + K) // HORIZON)
    tape = torch.tensor(lift.env_actions(ids, P + K, action_dim=env.model.action_dim), device=dev)
    n = [0]

    def step(t):
        if dr:
            env.batch.randomize_dynamics(seed=11, step=n[0]); n[0] += 1
        env.step(tape[t])

    env.batch.set("ep_step", ((197 * ids) % HORIZON).astype(np.int32))
    for t in range(P):
        step(t)
    env.batch.sync(); torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    tier0 = env.batch.tier_stats()
    s_ = torch.cuda.ExternalStream(env.batch.stream(), device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    t0 = time.perf_counter()
    for t in range(K):
        ev[t][0].record(s_)
        step(P + t)
        ev[t][1].record(s_)
    env.batch.sync(); torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = shard.max_over_ranks(time.perf_counter() - t0, dev)
    tier1 = env.batch.tier_stats()
    q = env.batch.tensor("qpos")
    div = shard.max_over_ranks(float(int((~torch.isfinite(q).all(dim=1)).sum().item()) + int((env.batch.tensor("diverged") > 0).sum().item())), dev)
    ovf = shard.max_over_ranks(float((env.batch.tensor("overflow") > 0).sum().item()), dev)
    cn = env.batch.tensor("cap_need").view(B, 2)
    lib_sha = hashlib.sha256(open(backend.LIB_PATH, "rb").read()).hexdigest()[:16]
    valu = pmc_evidence(f"valu_count_{config}.json", "valu_per_env_substep", lib_sha, config)
    issue = valu if isinstance(valu, str) or valu is None else valu * B * world * N_SUB * K / dt / (world * VALU_ISSUE_PEAK)
    traffic = pmc_evidence(f"hbm_traffic_{config}.json", "bytes_per_launch", lib_sha, config)
    env.bank_quiesce(); env._bank_stop()
    out = {"workload": f"{label} (BASELINE {which})", "envs_per_gpu": B, "steps": K, "preroll": P, "steady_state": bool(P >= HORIZON), "value": B * world * K / dt, "unit": "env-steps/s",
           "ms_per_step": 1e3 * dt / K, "step_ms": step_stats([a.elapsed_time(b_) for a, b_ in ev]), "overflow_envs": int(ovf), "diverged_envs": int(div),
           "tier_env_steps": tier1[0] - tier0[0], "tier_changes_in_mid_step": tier1[1] - tier0[1],
           "max_contacts_needed": int(cn[:, 0].max().item()), "max_rows_needed": int(cn[:, 1].max().item()), "capacity": [env.batch.maxcon, env.batch.maxefc],
           "issue_frac": issue, "traffic": traffic, "algorithmic_bytes_per_launch": algorithmic_bytes_per_env_step(env, flat, dr) * B,
           "dynamics_randomisation": "re-drawn before every control step" if dr else None}
    adim = env.model.action_dim
    del env, tape
    torch.cuda.empty_cache()
    if double_buffer:
        dt3, stale3, hl = double_buffered_region(config, flat, cfg, ids, local_rank, dev, world, dr, P, K, adim)
        out["double_buffered"] = {"value": B * world * K / dt3, "ms_per_step": 1e3 * dt3 / K, "steps": K, "halves": hl, "bank_stale": stale3}
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # 200 timed steps after 20 warm-up steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="lift", help="BASELINE configuration (module docstring); the driver's default is lift = configs[1]")
    ap.add_argument("--envs-per-gpu", type=int, default=0, help="default: the configuration's BASELINE batch size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--phase", choices=("staggered", "fresh"), default="staggered", help="episode phase of the envs when the timed region starts (module docstring)")
    ap.add_argument("--preroll", type=int, default=-1, help="untimed launches before the warm-up in the staggered phase (default: the horizon)")
    ap.add_argument("--groups", type=int, default=16, help="env blocks on their own HIP streams for the secondary open-loop figure; 1 = skip it")
    ap.add_argument("--no-open-loop", action="store_true", help="skip the second and third timed regions (the same K steps with stream groups; two half-batches alternating)")
    ap.add_argument("--no-double-buffer", action="store_true", help="skip the third timed region (two half-batches stepped alternately, closed-loop compatible)")
    ap.add_argument("--no-other-configs", action="store_true", help="lift only: skip the short secondary regions of BASELINE configs[2..4] (config.other_configs)")
    ap.add_argument("--other-steps", type=int, default=-1, help="timed lockstep control steps of each secondary configuration (default: OTHER_REGION, 50 where a step takes a few ms, 10 for PickPlace)")
    ap.add_argument("--other-preroll", type=int, default=-1, help="untimed launches before each secondary region (episode steps staggered as in the headline region; default: OTHER_REGION, 300 / 50)")
    ap.add_argument("--strict-collective", action="store_true", help="N > 1: exit with code 4 if the C-ABI RCCL communicator cannot be formed (default: reduce the rollout statistics through torch.distributed and say so at top level, \"collective\": \"fallback: ...\", and on stderr)")
    ap.add_argument("--secondary-only", choices=sorted(CONFIGS), default=None, help="internal: run one secondary region and print its record (the default run starts one child per configuration)")
    args = ap.parse_args()

    if args.secondary_only:
        if not torch.cuda.is_available():
            raise SystemExit(3)
        torch.cuda.set_device(0)
        print(json.dumps(secondary_region(args.secondary_only, 0, 0, 1, torch.device("cuda", 0), *other_region(args, args.secondary_only))), flush=True)
        return

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

    label, stem, b_default, dr, which = CONFIGS[args.config]
    flat, cfg = factory.load_shipped(stem)
    B = args.envs_per_gpu or b_default
    ids = shard.env_block(B * world, rank, world)
    K, W = args.steps, args.warmup
    P = (HORIZON if args.preroll < 0 else args.preroll) if args.phase == "staggered" else 0   # untimed pre-roll launches
    G = max(1, args.groups)
    K2 = 0 if (args.no_open_loop or G == 1) else K   # second region: the same number of steps with stream groups (open loop)
    env = build_env(args.config, flat, cfg, ids, local_rank, 3 + (P + W + K + K2) // HORIZON)
    adim = env.model.action_dim
    tape = torch.tensor(lift.env_actions(ids, P + K + W + K2, action_dim=adim), device=dev)  # whole action tape resident in HBM
    dr_step = [0]

    def step(t):
        if dr:
            env.batch.randomize_dynamics(seed=11, step=dr_step[0]); dr_step[0] += 1
        env.step(tape[t])

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    if P:
        env.batch.set("ep_step", ((197 * ids) % HORIZON).astype(np.int32))   # keyed by the GLOBAL env id: independent of the GPU count
        for t in range(P):
            step(t)
        tape = tape[P:]
    for t in range(W):
        step(t)
    env.batch.sync(); torch.cuda.synchronize(); barrier()
    ring0 = env.bank_stats()
    tier0 = env.batch.tier_stats()

    def timed(first, n, strs):
        ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in strs] for _ in range(n)]
        t0 = time.perf_counter()
        for t in range(n):
            for g, s_ in enumerate(strs):
                ev[t][g][0].record(s_)
            step(first + t)
            for g, s_ in enumerate(strs):
                ev[t][g][1].record(s_)
        env.batch.sync(); torch.cuda.synchronize(); barrier()
        dt_ = shard.max_over_ranks(time.perf_counter() - t0, dev)
        return dt_, [[a.elapsed_time(b) for a, b in row] for row in ev]

    # ---- the headline region: one control step of all envs at a time, events on the stream the fused kernel is launched on
    dt, evms = timed(W, K, [torch.cuda.ExternalStream(env.batch.stream(), device=dev)])
    kern_ms = float(np.mean(evms))   # one control step of all B envs: dispatch order + k_step + reset passes
    ring1 = env.bank_stats()
    tier1 = env.batch.tier_stats()
    if os.environ.get("RSIM_BENCH_TRACE"):
        print("per-step ms:", " ".join(f"{r[0]:.2f}" for r in evms), file=sys.stderr)
    # ---- secondary regions: BASELINE configs[2..4] at their stated batch sizes, a short lockstep region each, so that the clock of whoever runs the
    # default command also covers them.  After the headline region (which they cannot disturb) and BEFORE the open-loop / double-buffered regions of the headline
    # configuration: behind those (sixteen stream groups, two more batches) the children of session 16 read 6 - 9 % lower than the same regions stand-alone or
    # behind a parent that had run the headline region only (session 17: identical); 12 s of sustained load alone costs 1 % (session 18, tools/sustained_load.py).
    # Every rank takes part (weak scaling like the headline).
    other = None
    if args.config == "lift" and not args.no_other_configs:
        env.bank_quiesce()
        other = {}
        for oc in ("stack", "peg", "pickplace"):
            if world == 1:
                # one child process per configuration: a fault of the GPU queue there (which aborts the process that owns it) is reported in the record
                # instead of taking the headline line with it
                import subprocess
                try:
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--secondary-only", oc, "--other-steps", str(other_region(args, oc)[0]), "--other-preroll", str(other_region(args, oc)[1])],
                                       capture_output=True, text=True, timeout=900)
                    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                    other[oc] = json.loads(lines[-1]) if r.returncode == 0 and lines else {"error": f"exit code {r.returncode}: {r.stderr.strip().splitlines()[-1][:200] if r.stderr.strip() else ''}"}
                except Exception as e:
                    other[oc] = {"error": f"{type(e).__name__}: {e}"}
            else:
                # under torch.distributed.run every rank takes part (weak scaling like the headline), in process
                try:
                    other[oc] = secondary_region(oc, rank, local_rank, world, dev, *other_region(args, oc))
                except Exception as e:   # a failing secondary region is reported, it does not take the headline line with it
                    other[oc] = {"error": f"{type(e).__name__}: {e}"}

    open_loop = None
    if K2:
        env.batch.set_stream_groups(G)
        dt2, ev2 = timed(W + K, K2, [torch.cuda.ExternalStream(env.batch.group_stream(g), device=dev) for g in range(G)])
        env.batch.set_stream_groups(1)
        open_loop = {"value": B * world * K2 / dt2, "ms_per_step": 1e3 * dt2 / K2, "kernel_ms_per_env_block": float(np.mean(ev2)), "steps": K2, "stream_groups": G,
                     "note": "the next K steps of the same envs as G env blocks on their own HIP streams (rsim_set_stream_groups): a block's step t + 1 does not wait "
                             "for the other blocks' step t -- reachable only with actions known in advance (an action tape), not by a closed-loop policy"}

    # ---- double-buffered closed loop: the batch as two halves stepped alternately, each half's step t + 1 issued only after the HOST has seen its
    # step t complete (stream synchronisation = the point where a policy would read that half's observations) -- while the other half is stepping
    double_buffered = None
    if K2 and not args.no_double_buffer:
        dt3, stale3, hl = double_buffered_region(args.config, flat, cfg, ids, local_rank, dev, world, dr, P + W, K, adim)
        double_buffered = {"value": B * world * K / dt3, "ms_per_step": 1e3 * dt3 / K, "steps": K, "halves": hl, "bank_stale": stale3,
                           "note": "closed-loop compatible: two half-batches (two rsim batches on their own streams) stepped alternately; the host waits for a half's "
                                   "step t (its observations) before it issues that half's step t + 1, the other half steps meanwhile -- what a policy evaluated per "
                                   "half can reach; `value` above is the stricter one-policy-call-per-step protocol"}

    # the end-of-rollout statistics go through the C-ABI's own collective (rsim_comm_* / rsim_allreduce_stats over RCCL) when there is more than one rank.  If
    # that communicator cannot be formed the job's torch process group carries the five numbers and the line says so LOUDLY -- top-level "collective":
    # "fallback: <why>" and a warning on stderr from every rank -- so that a multi-GPU line that never touched the C-ABI collective cannot look like one that did
    # (round-5 review); --strict-collective turns it into exit code 4.  shard.hip_comm raises on every rank or on none.
    comm, comm_note = None, None
    if world > 1:
        try:
            comm = shard.hip_comm(rank, world, local_rank)
        except shard.CommUnavailable as e:
            comm_note = str(e)
            print(f"bench.py rank {rank}: WARNING: the C-ABI collective is unavailable ({e}); the rollout statistics go through torch.distributed (reported as \"collective\": \"fallback\")", file=sys.stderr, flush=True)
            if args.strict_collective:
                torch.distributed.destroy_process_group()
                raise SystemExit(4)
    st = shard.RolloutStats(dev, comm=comm)
    q = env.batch.tensor("qpos")
    # envs that hit the bad-state guard (RSIM_DIVERGED, MuJoCo's mj_checkPos semantics) or hold a non-finite coordinate
    st.add(env_steps=B * K, diverged=int((~torch.isfinite(q).all(dim=1)).sum().item()) + int((env.batch.tensor("diverged") > 0).sum().item()),
           reward_sum=float(env.reward().sum().item()), successes=int(env.success().sum().item()))
    overflow_envs = shard.max_over_ranks(float((env.batch.tensor("overflow") > 0).sum().item()), dev)   # envs that ever dropped a contact / constraint row
    bank_stale = shard.max_over_ranks(float(env.batch.tensor("bank_stale").sum().item()), dev)
    cn = env.batch.tensor("cap_need").view(B, 2)   # largest contact / row demand of any substep since the batch was created (pre-roll included)
    need_con, need_efc = shard.max_over_ranks(float(cn[:, 0].max().item()), dev), shard.max_over_ranks(float(cn[:, 1].max().item()), dev)
    need_hist = {f"contacts>{t}": int((cn[:, 0] > t).sum().item()) for t in (16, 24, 32, 48)} | {f"rows>{t}": int((cn[:, 1] > t).sum().item()) for t in (64, 80, 96, 128, 160)}   # rank-local
    tot = st.allreduce()

    if rank == 0:
        abytes = algorithmic_bytes_per_env_step(env, flat, dr) * B   # one control step = one launch of all B envs
        ach = abytes / (kern_ms * 1e-3) / 1e9
        # PMC-derived figures are only valid for the library build they were measured on: the files carry the sha of that build
        lib_sha = hashlib.sha256(open(backend.LIB_PATH, "rb").read()).hexdigest()[:16]
        sfx = "" if args.config == "lift" else "_" + args.config

        traffic = pmc_evidence(f"hbm_traffic{sfx}.json", "bytes_per_launch", lib_sha, args.config)      # tools/pmc_traffic.py (FETCH_SIZE / WRITE_SIZE passes), one launch of all B envs
        valu = pmc_evidence(f"valu_count{sfx}.json", "valu_per_env_substep", lib_sha, args.config)      # tools/pmc_valu.py (SQ_INSTS_VALU pass on this workload)
        issue = valu if isinstance(valu, str) else None       # "stale:<sha>" / "absent": said out loud, never a silent null
        if valu and not isinstance(valu, str):
            rate = valu * B * N_SUB * K / dt   # wave-instructions per second of this GPU over the timed region
            issue = {"bound": "valu-issue", "valu_instr_per_env_substep": valu, "achieved": rate / 1e9,
                     "peak": VALU_ISSUE_PEAK / 1e9, "unit": "G wave-instr/s", "frac": rate / VALU_ISSUE_PEAK,
                     "peak_source": "measured: tools/ubench/valu_peak.hip, 8 wavefronts per SIMD of independent v_fma_f32 (profiles/r03_a_valu_peak.txt); "
                                    f"guide figure {VALU_ISSUE_PEAK_THEORY / 1e9:.1f} (one wave64 VALU instruction per 2 cycles and SIMD)",
                     "source": f"profiles/valu_count{sfx}.json (rocprofv3 --pmc SQ_INSTS_VALU on this build, same workload)"}
        dsteps = max(1, ring1["steps"] - ring0["steps"])
        out = {
            "metric": f"env-steps/sec (whole node), {label.split(' +')[0]} @{B} envs/GPU", "value": tot["env_steps"] / dt, "unit": "env-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * dt / K,
            # per-step durations of the timed region on rank 0 (HIP events on the batch's stream) and what the capacity tier did in it: a slow window explains itself
            "step_ms": step_stats([r[0] for r in evms]), "tier_env_steps": tier1[0] - tier0[0], "tier_changes_in_mid_step": tier1[1] - tier0[1],
            "collective": ("none (one rank)" if world == 1 else ("rsim_allreduce_stats (C-ABI, RCCL)" if comm is not None else f"fallback: torch.distributed; {comm_note}")),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{label}, 25 substeps x dt 0.002 + controllers per substep, fused in one launch per env (BASELINE {which})",
                       "envs_per_gpu": B, "global_envs": B * world, "n_sub": N_SUB, "per_env_seeded_reset": True, "horizon": HORIZON, "on_device_auto_reset": True,
                       "protocol": "lockstep: one control step of all envs per call, the next starts when all have finished (closed-loop compatible)",
                       "open_loop": open_loop, "double_buffered": double_buffered, "dynamics_randomisation": "re-drawn before every control step" if dr else None,
                       "steady_state": bool(P >= HORIZON), "episode_window": [P + W, P + W + K],
                       "episode_phase": (f"uniform over the horizon: step counters offset by (197 i) mod 500, then {P} untimed pre-roll launches" if P else "fresh: all envs at step W of their first episode"),
                       "reset_ring": {"bank_stale": int(bank_stale), "polls_in_region": ring1["polls"] - ring0["polls"], "rows_refilled_in_region": ring1["rows"] - ring0["rows"],
                                      "stepping_thread_ms_per_1000_steps": 1e6 * (ring1["tick_s"] - ring0["tick_s"]) / dsteps,
                                      "upkeep_thread_ms_per_1000_steps": 1e6 * (ring1["upkeep_s"] - ring0["upkeep_s"]) / dsteps,
                                      "note": "asynchronous: episode counters polled on a side stream, rows drawn by a host thread from persistent per-env generators, "
                                              "scattered through pinned staging (reset_bank.py); the stepping thread never reads the device"},
                       "overflow_envs": int(overflow_envs),
                       "capacity": {"contacts": env.batch.maxcon, "rows": env.batch.maxefc, "max_contacts_needed": int(need_con), "max_rows_needed": int(need_efc), "envs_by_demand": need_hist,
                                    "note": "compiled contact / constraint-row capacity per env against the largest demand of any substep of any env over pre-roll, warm-up and "
                                            "timed region (RSIM_CAP_NEED); overflow_envs counts the envs that ever dropped one"},
                       "other_configs": other,
                       "lib_sha16": lib_sha, "obs_dim": env.model.nobs, "action_dim": adim, "sharding": f"env-block x{world}",
                       "stats_allreduce": st.path + (f"; {comm_note}" if comm_note else ""),
                       "diverged_envs": int(tot["diverged"]), "reward_sum": tot["reward_sum"], "successes": int(tot["successes"])},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_step", "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": abytes,
                         "note": "latency/VALU/LDS-bound by design (state LDS-resident for 25 substeps); see DESIGN.md section 6",
                         # the fraction that describes this kernel: VALU issue slots used (PMC instruction count of THIS build x measured rate); null
                         # when profiles/valu_count.json was measured on another build
                         "issue": issue},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.config, flat, cfg)
        elif world == 1:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
