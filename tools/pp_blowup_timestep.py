"""The fp64 oracle's own PickPlace blow-ups (no randomisation: 2 of 1024 episodes non-finite within 35 control steps): are they the
restated contact model or the time discretisation?  Step 1 finds the episodes whose velocities run away at the reference's dt = 0.002 (25 substeps per
control step); step 2 re-runs exactly those episodes -- same reset, same action stream -- at dt / 2 and dt / 4 (50 / 100 substeps per control step, so
the controller and the actions see the same 20 Hz), and reports the largest velocity and the first substep with a velocity above 100 rad/s.
A model bug (a contact that injects energy) survives a smaller step; integration stiffness does not.
Usage: python tools/pp_blowup_timestep.py [n_envs=1024] [steps=35] [procs=8]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(i, steps, div, draw=False):
    from robosuite_amd import pick_place
    from tests.util import load_golden, make_oracle
    g, cfg, flat = load_golden("seed0_full", "pickplace_iiwa")
    flat = flat.copy()
    flat.arrays["timestep"] = np.array([0.002 / div])
    om, od, oc = make_oracle(flat, cfg)
    q0 = pick_place.episode_setup(cfg, flat.nq, 0, [i], block=0)[0]
    od.qpos[:] = q0; od.qvel[:] = 0; od.qacc_warmstart[:] = 0; od.ctrl[:] = 0; od.forward(); oc.reset(od)
    rng = np.random.default_rng(10**7 + i)
    vmax, dof, first = 0.0, -1, None
    simp = om.field("geom_solimp"); base = simp.copy()
    for t in range(steps):
        u = rng.random((flat.ngeom, 5))             # the draw tools/dr_solimp_oracle.py makes (ignored with the randomisation off): same action stream
        if draw:                                    # the reference's per-control-step solimp draw (ratio 0.1, mjmod.py:1705-1728), nothing else randomised
            simp[:] = np.clip(base * (1.0 + 0.1 * (2.0 * u.reshape(base.shape) - 1.0)), 0.0, np.inf)
        oc.env_step(od, rng.uniform(-1, 1, 7), 25 * div)
        v = np.abs(np.asarray(od.qvel))
        if not np.isfinite(v).all():
            return i, div, t, float("inf"), -1, first
        if v.max() > 100.0 and first is None:
            first = (t, int(v.argmax()), float(v.max()), int(od.ncon))
        if v.max() > vmax:
            vmax, dof = float(v.max()), int(v.argmax())
    return i, div, -1, vmax, dof, first


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 35
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    draw = len(sys.argv) > 4 and sys.argv[4] == "solimp"      # 4th argument "solimp": with the reference's solimp draw before every control step
    from multiprocessing import Pool
    t0 = time.time()
    with Pool(procs) as p:
        res = p.starmap(run, [(i, steps, 1, draw) for i in range(n)], chunksize=4)
    vm = np.array([r[3] for r in res])
    bad = [r[0] for r in res if r[3] > 1e3]
    print(f"dt 0.002: {n} episodes x {steps} control steps, {'solimp draw before every control step' if draw else 'no randomisation'}: max |qvel| median {np.median(vm):.1f} p99 {np.percentile(vm, 99):.0f}; above 1e3: {len(bad)} {bad}, non-finite: {int(np.isinf(vm).sum())}   [{time.time() - t0:.0f} s]", flush=True)
    for r in res:
        if r[3] > 1e3:
            print(f"   episode {r[0]}: non-finite at control step {r[2]}" if np.isinf(r[3]) else f"   episode {r[0]}: max {r[3]:.0f} on dof {r[4]}", "; first velocity above 100: (control step, dof, value, contacts)", r[5], flush=True)
    for div in (2, 4):
        with Pool(procs) as p:
            res2 = p.starmap(run, [(i, steps, div, draw) for i in bad], chunksize=1)
        v2 = np.array([r[3] for r in res2])
        print(f"dt 0.002 / {div}: of those {len(bad)} episodes: above 1e3: {int((v2 > 1e3).sum())}, non-finite: {int(np.isinf(v2).sum())}, median max |qvel| {np.median(v2):.1f}, p90 {np.percentile(v2, 90):.1f}", flush=True)
        for r in res2[:8]:
            print(f"dt 0.002 / {div}: episode {r[0]}: " + (f"non-finite at control step {r[2]}" if np.isinf(r[3]) else f"max |qvel| {r[3]:.1f} on dof {r[4]}") + f"; first above 100: {r[5]}   [{time.time() - t0:.0f} s]", flush=True)
