"""Parity at the batch sizes BASELINE.json states, at the states the workloads actually reach.

Each test builds one BASELINE configuration at its full size (Lift 4096, Stack 4096, TwoArmPegInHole / Baxter / JOINT_VELOCITY 2048,
PickPlace / IIWA 8192 with dynamics randomisation before every control step), runs >= 50 control steps of full-range random actions
through the C-ABI, then takes envs spread over the batch and compares ONE forward evaluation of the reached state (contact list,
constraint-row count, constraint forces, accelerations) against the fp64 oracle carrying that env's own model parameters.  The state
itself is the kernel's (fp32); both sides evaluate the same numbers, so the comparison is free of trajectory divergence.

Discrete decisions (a pair sitting within rounding of its margin, an MPR portal choice on an interpenetrating pose) can fall differently
in fp32 and fp64; the tests therefore require the contact / row structure to agree for nearly all sampled envs (bounds written below) and
hold the continuous quantities to tolerances on the envs where it does.
"""
import json
import os

import numpy as np
import pytest

from robosuite_amd import lift, mjcf
from tests.util import load_golden

pytestmark = pytest.mark.gpu
# PickPlace @ 8192 with per-step dynamics randomisation, oracle fed the kernel's contact geometry (measured values: profiles/r03_*_full_size_parity_pickplace.txt)
# What single precision delivers on this model: the Hessian M + J' D J spans 1e-5 (an object's or finger link's rotational inertia) to 1e6 (a squeezed
# contact) and its Cholesky factor resolves the soft directions to a few per cent, so the kernel stops at an acceleration whose OBJECTIVE is optimal
# to ~1e-4 in the worst env of the batch (median 1e-11, 90th percentile 1e-8) while, in the worst env of a sample, the light bodies' accelerations and the forces that balance them differ by up to the
# size of the env's largest (tools/pp_dump.py + tools/pp_solver_metric.py, profiles/r03_f_pickplace_solver_metric.txt: a 2 kN squeeze on the bread,
# cost gap 2.1e-5, a gradient of 1.5 N m left on its 4 g body).  Asserted therefore: the objective gap on every env, the arm (armature >= 0.1) on every
# env, and the MEDIAN env for the forces, the six gripper joints and the four objects (the maxima are printed).
# The gap is heavy-tailed: over 198 envs of the batch p50 5e-11, p90 2e-8 .. 4e-8, p99 1e-5 .. 8e-5, max 7e-5 .. 2e-4, the same for the builds before
# and after the round-3 solver changes (tools/pp_gap_stats.py, profiles/r03_m_pickplace_gap_stats.txt); the maximum of a 32-env sample is whatever the
# one worst-conditioned env in it gives (1.6e-7 and 3.6e-4 seen on consecutive builds), so the bound on it is loose and the 90th percentile carries the claim.
# Round 4: two refinement passes with an fp64 gradient behind the fp32 iteration (solve_newton: residuals, forces, J^T f and the objective in fp64 from the float
# data, the acceleration as a pair of floats, the fp32 factor reused, a pass kept only if it lowers the fp64 objective and leaves the active set alone) bring the
# typical env two orders closer -- over 406 envs of the batch (tools/pp_gap_stats.py 400, profiles/r04_u_pickplace_refinement.txt): objects p50 5e-5 -> 1e-6, p90 1.2e-2 ->
# 9e-4, forces p50 2e-5 -> 7e-6, p90 7e-4 -> 2e-4.  The tail (p99 ~ 1e-1, one env per few hundred with the maximum near 1) is where the fp32 iteration itself stops on the
# wrong side of a state change or the factor resolves nothing of the soft direction; it is unchanged.  So the 90th percentile is now asserted, an order below what
# the medians were held to in round 3, and the medians two orders below.
# Round 5: the tail is gone.  It was traced (RSIM_POLISH diagnostics, per-state fp32-input floors, dumps of the worst envs analysed on the CPU: tools/analyze_parity_dump.py,
# tools/emulate_polish.py, profiles/r05_g_*) to ONE mechanism: the fp32 iteration ends on a kink of the piecewise objective -- in every one of the worst envs a condim-4
# contact of an object sat in the cone (or satisfied) state at the kernel's answer and in the quadratic state at the minimiser -- where a plain Newton step gains nothing
# even with an exact Hessian, and only an exact line search ACROSS the kink gets on (the fp64 oracle, started at the kernel's answer, needed 3 - 10 more iterations).
# Not conditioning: rounding the fp64 solve's inputs to float32 moves it by 1e-7 of the group's largest on those very states (`g_floor`).  The polish behind the wide
# Newton solver is now that iteration (fp64 evaluation + exact fp64 line search, fp32 factor; csrc/rsim_step.hip solve_newton), and the assertions below are per-env
# MAXIMA over the sample (197 envs measured: objects 1e-3, gripper 1e-3, forces 4e-3, objective gap 4e-8; before: 3e-1, 9e-2, 1e+0, 9e-4).
PP_COST_GAP = 2e-6
PP_COST_GAP_P90 = 1e-8
PP_ARM_TOL = 1e-3
PP_FORCE_MEDIAN = 2e-4
PP_FORCE_MAX = 1e-2
PP_GROUP_MEDIAN = {"gripper": 5e-4, "objects": 2e-4}
PP_GROUP_MAX = {"gripper": 5e-3, "objects": 5e-3}
torch = pytest.importorskip("torch")

# float model arrays an env may carry its own values for (rsim_model_param_set / domain randomisation / per-episode patches)
PARAM_FIELDS = ("body_pos", "body_quat", "body_ipos", "body_iquat", "body_mass", "body_inertia", "body_invweight0", "body_subtreemass", "jnt_pos", "jnt_axis",
                "jnt_range", "jnt_margin", "jnt_solref", "jnt_solimp", "dof_armature", "dof_damping", "dof_frictionloss", "dof_solref", "dof_solimp",
                "dof_invweight0", "geom_size", "geom_pos", "geom_quat", "geom_friction", "geom_solref", "geom_solimp", "geom_solmix", "geom_margin", "geom_gap",
                "geom_rbound", "site_pos", "site_quat", "actuator_gear", "actuator_gainprm", "actuator_biasprm", "actuator_ctrlrange", "actuator_forcerange")


def oracle_for_env(flat, hb, e):
    """fp64 oracle model + data carrying the LIVE parameters of env `e` of the HIP batch (read back through rsim_model_param_get)."""
    from oracle.oracle import OracleData, OracleModel

    f = flat.copy()
    for k in PARAM_FIELDS:
        if k in f.arrays:
            f.arrays[k] = hb.param_get(k, e, 1)[0].reshape(f.arrays[k].shape).astype(np.float64)
    opt = hb.param_get("opt", e, 1)[0]
    f.arrays["timestep"] = np.array([opt[0]]); f.arrays["gravity"] = opt[1:4].copy(); f.arrays["density"] = np.array([opt[4]])
    f.arrays["viscosity"] = np.array([opt[5]]); f.arrays["impratio"] = np.array([opt[6]]); f.arrays["wind"] = opt[7:10].copy()
    blob = mjcf.to_blob(f)
    om = OracleModel(blob)
    om.blob_bytes = blob          # kept for RSIM_PARITY_DUMP (offline analysis of single envs on the CPU)
    return om, OracleData(om)


def compare_reached_states(flat, hb, pick, mpr_geoms=(), with_contacts=0, ignore_pair=None, dof_groups=None):
    """forward() on the whole batch (writes the compat arrays, advances nothing), then per picked env the oracle on the same state.
    `with_contacts` > 0 adds that many envs that currently HAVE contacts (spread over the batch) to the sample.
    Returns per-env dicts with the discrete agreement flag and the error measures."""
    q, v, ws, ctrl = hb.get("qpos"), hb.get("qvel"), hb.get("qacc_warmstart"), hb.get("ctrl")
    hb.forward()
    ncon, nefc, qacc, efc, con = hb.get("ncon"), hb.get("nefc"), hb.get("qacc"), hb.get("efc_force"), hb.get("contact")
    niter_all = hb.get("polish") if "polish" in hb.shapes else hb.get("niter")
    if with_contacts:
        have = np.nonzero(ncon > 0)[0]
        if len(have):
            pick = np.unique(np.concatenate([pick, have[np.linspace(0, len(have) - 1, min(with_contacts, len(have))).astype(int)]]))
    out = []
    for e in pick:
        om, od = oracle_for_env(flat, hb, int(e))
        od.qpos[:] = q[e]; od.qvel[:] = v[e]; od.qacc_warmstart[:] = ws[e]; od.ctrl[:] = ctrl[e]
        od.forward()
        oc = od.contacts()
        r = dict(env=int(e), ncon=(int(ncon[e]), od.ncon), nefc=(int(nefc[e]), od.nefc))
        hc = hb.contacts(int(e))
        same = ncon[e] == od.ncon and nefc[e] == od.nefc and all((a["geom1"], a["geom2"], a["dim"]) == (b["geom1"], b["geom2"], b["dim"]) for a, b in zip(hc, oc))
        r["same"] = bool(same)
        if same:
            # contact geometry: depth to 1e-5 m and normal to 0.1 degree on every contact (`ignore_pair(g1, g2)`: pairs left out of this verdict)
            keep = [(a, b) for a, b in zip(hc, oc) if not (ignore_pair and ignore_pair(b["geom1"], b["geom2"]))]
            r["angle"] = max([float(np.degrees(np.arccos(np.clip(np.dot(a["frame"][0], b["frame"][0]), -1.0, 1.0)))) for a, b in keep], default=0.0)
            r["geom_ok"] = r["angle"] <= 0.1 and max([abs(a["dist"] - b["dist"]) for a, b in keep], default=0.0) <= 1e-5
            r["dist"] = max([abs(a["dist"] - b["dist"]) for a, b in zip(hc, oc)], default=0.0)
            r["dist_box"] = max([abs(a["dist"] - b["dist"]) for a, b in zip(hc, oc) if a["geom1"] not in mpr_geoms and a["geom2"] not in mpr_geoms], default=0.0)
            r["pos"] = max([np.abs(a["pos"] - b["pos"]).max() for a, b in zip(hc, oc)], default=0.0)
            of = np.asarray(od.efc_force) if od.nefc else np.zeros(0)
            r["fscale"] = float(np.abs(of).max()) if od.nefc else 0.0
            r["force"] = float(np.abs(efc[e][:od.nefc] - of).max()) if od.nefc else 0.0
            r["ascale"] = float(np.abs(od.qacc).max())
            r["qacc"] = float(np.abs(qacc[e] - od.qacc).max())
            r["qacc_arm"] = float(np.abs(qacc[e][:7] - od.qacc[:7]).max())
            # the solver / dynamics half on identical inputs: the oracle evaluated again with the KERNEL's contact geometry (its own pairs, dimensions
            # and materials); what is left is constraint assembly, the Newton solve and the accelerations of every dof
            if od.forward_with_contact_geometry(hc):
                of = np.asarray(od.efc_force) if od.nefc else np.zeros(0)
                r["g_fscale"], r["g_ascale"] = (float(np.abs(of).max()) if od.nefc else 0.0), float(np.abs(od.qacc).max())
                r["g_force"] = float(np.abs(efc[e][:od.nefc] - of).max()) if od.nefc else 0.0
                r["g_qacc"] = float(np.abs(qacc[e] - od.qacc).max())
                # ... and in the solver's own metric: the objective the Newton solver minimises, evaluated in fp64 at the kernel's acceleration and at
                # the oracle's (the unique minimiser).  Where the Hessian is nearly flat -- a 1e-5 kg m^2 object or finger link under contacts of
                # D ~ 1e6 -- accelerations far apart have costs equal to single precision; this number says how far from optimal the kernel stopped
                (c_opt, g_opt), c_hip = od.cost(np.array(od.qacc), True), od.cost(qacc[e].astype(np.float64))
                r["g_cost_gap"] = float((c_hip - c_opt) / max(1.0, abs(c_opt)))
                # the yardstick itself: the oracle's Newton iteration stops on MuJoCo's criteria too, and in single envs it stops short (gradient of 0.4 - 0.6 left on an
                # object dof while the kernel's answer has the LOWER objective: profiles/r05_m_*).  Such an env says nothing about the kernel
                r["oracle_grad"] = float(np.abs(g_opt).max())
                if dof_groups:
                    r["g_groups"] = {k: (float(np.abs(qacc[e][ix] - od.qacc[ix]).max()), float(np.abs(od.qacc[ix]).max())) for k, ix in dof_groups.items()}
                    # conditioning of THIS state: how far the fp64 solve itself moves when its inputs are rounded to float32 (oracle/rsim_oracle.c round_inputs_f32)
                    a_exact = np.array(od.qacc)
                    od.set_round_rows(True); od.qacc_warmstart[:] = ws[e]
                    if od.forward_with_contact_geometry(hc):
                        r["g_floor"] = {k: float(np.abs(np.asarray(od.qacc)[ix] - a_exact[ix]).max()) for k, ix in dof_groups.items()}
                    od.set_round_rows(False)
                r["niter"] = int(niter_all[e])
                if os.environ.get("RSIM_PARITY_DUMP"):
                    r["dump"] = dict(blob=np.frombuffer(om.blob_bytes, dtype=np.uint8).copy(), qpos=q[e].copy(), qvel=v[e].copy(), ws=ws[e].copy(), ctrl=ctrl[e].copy(), qacc=qacc[e].copy(),
                                     efc=efc[e][:od.nefc].copy(), geo=np.array([[c["dist"], *c["pos"], *np.asarray(c["frame"]).ravel(), c["geom1"], c["geom2"], c["dim"]] for c in hc]))
        out.append(r)
    return out


def summarize(name, res):
    ok = [r for r in res if r["same"]]
    print(f"\n[{name}] envs {len(res)}  structure agrees {len(ok)}  ncon range {min(r['ncon'][0] for r in res)}..{max(r['ncon'][0] for r in res)}"
          f"  nefc range {min(r['nefc'][0] for r in res)}..{max(r['nefc'][0] for r in res)}")
    for k in ("dist", "dist_box", "pos"):
        print(f"   max |d{k}| = {max(r[k] for r in ok):.3e}")
    print("   max |dforce| / max force per env:", " ".join(f"{r['force'] / max(1.0, r['fscale']):.1e}" for r in ok))
    print("   max |dqacc| / max(1, |qacc|) per env:", " ".join(f"{r['qacc'] / max(1.0, r['ascale']):.1e}" for r in ok))
    tight = [r for r in ok if r["geom_ok"]]
    print(f"   contact geometry (depth 1e-5 m, normal 0.1 deg) agrees in {len(tight)} of {len(ok)}; on those: max rel dforce {max([r['force'] / max(1.0, r['fscale']) for r in tight], default=0):.1e}"
          f"  max rel dqacc {max([r['qacc'] / max(1.0, r['ascale']) for r in tight], default=0):.1e}  arm dqacc {max([r['qacc_arm'] / max(1.0, r['ascale']) for r in tight], default=0):.1e}")
    loose = [r for r in ok if not r["geom_ok"]]
    print("   geometry differs (env, depth, normal deg, rel force):", [(r["env"], f"{r['dist']:.1e}", f"{r['angle']:.2f}", f"{r['force'] / max(1.0, r['fscale']):.1e}") for r in loose])
    print(f"   all structure-agreeing envs: max arm rel dqacc {max(r['qacc_arm'] / max(1.0, r['ascale']) for r in ok):.1e}")
    gg = [r for r in ok if "g_qacc" in r]
    if gg:
        print(f"   oracle fed the kernel's contact geometry ({len(gg)} envs): max rel dforce {max(r['g_force'] / max(1.0, r['g_fscale']) for r in gg):.1e}"
              f"  max rel dqacc (all dofs) {max(r['g_qacc'] / max(1.0, r['g_ascale']) for r in gg):.1e}"
              f"  solver objective above its minimum, relative: max {max(r['g_cost_gap'] for r in gg):.1e} median {float(np.median([r['g_cost_gap'] for r in gg])):.1e}")
        for k in (gg[0].get("g_groups") or {}):
            print(f"      {k}: max |dqacc| {max(r['g_groups'][k][0] for r in gg):.2e}  relative to the group's largest {max(r['g_groups'][k][0] / max(1.0, r['g_groups'][k][1]) for r in gg):.1e}")
            print(f"      {k} per env, relative, sorted:", " ".join(f"{x:.0e}" for x in sorted(r['g_groups'][k][0] / max(1.0, r['g_groups'][k][1]) for r in gg)))
        print("      objective gap per env, sorted:", " ".join(f"{x:.0e}" for x in sorted(r['g_cost_gap'] for r in gg)))
        if gg[0].get("g_groups") and "objects" in gg[0]["g_groups"]:
            worst = sorted(gg, key=lambda r: -max(r["g_groups"][k][0] / max(1.0, r["g_groups"][k][1]) for k in r["g_groups"]))[:16]
            print("      worst envs (env; per group: error / fp32-input floor of the fp64 solve, relative to the group's largest; objective gap; fscale; RSIM_POLISH = exit|-log10 grad|passes|fp64|000):")
            for r in worst:
                fl = r.get("g_floor", {})
                print("        ", r["env"], " ".join(f"{k} {r['g_groups'][k][0] / max(1.0, r['g_groups'][k][1]):.0e}/{fl.get(k, float('nan')) / max(1.0, r['g_groups'][k][1]):.0e}" for k in r["g_groups"]),
                      f"gap {r['g_cost_gap']:.0e} fmax {r['g_fscale']:.0f} niter {r.get('niter', -1):08d} ncon {r['ncon'][0]} nefc {r['nefc'][0]}")
        print("      rel dforce per env, sorted:", " ".join(f"{x:.0e}" for x in sorted(r['g_force'] / max(1.0, r['g_fscale']) for r in gg)))
    for r in res:
        if not r["same"]:
            print("   structure differs:", r)
    return ok


def continue_on_the_oracle(name, flat, cfg, env, pick, actions, dq, dv, min_over=1, assert_all=False, collect=None):
    """ONE more control step of the whole batch through the fused path (capacity tiers included), and for the picked envs the same control step on the fp64
    oracle from the kernel's own state: positions, velocities, warm start, actuator commands and the controller record (goal, initial joints, gripper
    action) are copied over, so the comparison covers what the debug entry rsim_forward cannot -- a step in which the env needs MORE contacts / rows
    than the native configuration holds (the oracle holds 64 contacts).  Asserts the bounds on the envs whose step did exceed the native capacity
    (at least `min_over` of them) and prints all."""
    from tests.util import make_oracle

    b = env.batch
    pre = {k: b.get(k) for k in ("qpos", "qvel", "qacc_warmstart", "ctrl", "cstate", "time")}
    b.set("cap_need", 0)
    env.step(actions)
    q1, v1, need = b.get("qpos"), b.get("qvel"), b.get("cap_need")
    over = 0
    for e in pick:
        e = int(e)
        _, _, oc = make_oracle(flat, cfg)
        om, od = oracle_for_env(flat, b, e)      # the env's LIVE model (per-episode values such as Lift's cube size, randomised dynamics), not the shipped defaults
        od.qpos[:] = pre["qpos"][e]; od.qvel[:] = pre["qvel"][e]; od.qacc_warmstart[:] = pre["qacc_warmstart"][e]; od.ctrl[:] = pre["ctrl"][e]
        od.forward(); oc.reset(od)
        st = oc.state
        st[:20] = pre["cstate"][e][:20]; st[20:24] = pre["cstate"][e][20:24]; st[24:28] = pre["cstate"][e][20:24]
        oc.env_step(od, actions[e].cpu().numpy().astype(np.float64), env.n_sub)
        eq, ev = float(np.abs(q1[e] - od.qpos).max()), float(np.abs(v1[e] - od.qvel).max())
        beyond = need[e, 0] > b.maxcon or need[e, 1] > b.maxefc
        print(f"   [{name}] env {e}: demand {need[e, 0]} contacts / {need[e, 1]} rows (native {b.maxcon} / {b.maxefc}){' -> stepped by the wide configuration' if beyond else ''}: one control step vs the oracle |dq| {eq:.1e} |dv| {ev:.1e}")
        if beyond:
            over += 1
        if collect is not None:
            collect.append(dict(env=e, dq=np.abs(q1[e] - od.qpos), dv=np.abs(v1[e] - od.qvel), beyond=beyond, finite=bool(np.isfinite(od.qpos).all())))     # the caller's verdict
        elif beyond or assert_all:
            assert eq < dq and ev < dv, (e, eq, ev)
    assert over >= min_over, f"{name}: none of the picked envs exceeded the native capacity in this step"
    assert int(b.get("overflow").sum()) == 0
    return over


# one whole control step of a PickPlace env through the fused path against the oracle.  Measured (round 6, 22 envs of control step 31, a dynamics draw before the step): arm joints
# |dq| p50 4.7e-7 max 5.4e-4; gripper joints (the Robotiq's interpenetrating four-bar links: MPR facet choices, see the 8192-env test) p50 4.2e-4 max 1.1e-2; object
# coordinates p50 4.4e-4 max 3.4e-3 -- 25 substeps of bodies of 1e-5 .. 4e-3 kg m^2 whose single-solve accelerations agree to the 5e-3 of that test
PP_STEP_DQ_MEDIAN, PP_STEP_DQ_ARM, PP_STEP_DQ_MAX, PP_STEP_DQ_ARM_MEDIAN = 3e-3, 3e-3, 5e-2, 1e-5
# one whole control step of a Lift env (25 substeps), fused kernel against the fp64 oracle from the same state and the env's live model.  Measured (round 6, 24 envs of
# control step 251): |dq| 3e-7 typical, 2.2e-5 on the most contact-rich env (12 contacts / 50 rows); |dv| 5e-6 typical, 2.1e-4 worst
LIFT_STEP_DQ, LIFT_STEP_DV = 2e-4, 5e-3


def spread(B, n=32):
    return np.unique(np.concatenate([[0, 1, 63, 64, B // 2 - 1, B // 2, B - 2, B - 1], np.linspace(0, B - 1, n).astype(int)]))


def test_lift_4096_late_episode_states_of_the_bench_workload():
    """BASELINE configs[1]: the bench workload itself (per-env seeded episodes and action streams) at control steps 200-250, where launches are
    slowest (hand at the table, finger / cube / table contacts), 4096 envs."""
    adir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "robosuite_amd", "assets")
    flat = mjcf.load_model(os.path.join(adir, "lift_panda.rsim"))
    cfg = json.load(open(os.path.join(adir, "lift_panda.cfg.json")))
    B = 4096
    ids = np.arange(B)
    env = lift.LiftBatch(flat, cfg, ids, seed0=0, horizon=500, bank_episodes=2)
    tape = torch.tensor(lift.env_actions(ids, 251), device="cuda")
    checked = agree = 0
    worst = dict(dist=0.0, pos=0.0, force=0.0, qacc=0.0, cost_gap=0.0)
    for t in range(250):
        if t == 249:
            env.batch.set("cap_need", 0)     # demand of the last step alone: picks the envs of the whole-step comparison below
        env.step(tape[t])
        if t in (199, 224, 249):
            assert int(env.batch.get("overflow").sum()) == 0          # control steps never drop a contact (capacity tiers) ...
            res = compare_reached_states(flat, env.batch, spread(B, 24))
            env.batch.set("overflow", 0)                                # ... rsim_forward in there is a debug entry: native capacity, drops counted
            ok = summarize(f"Lift step {t + 1}", res)
            checked += len(res); agree += len(ok)
            for r in ok:
                worst["dist"] = max(worst["dist"], r["dist"]); worst["pos"] = max(worst["pos"], r["pos"])
                worst["force"] = max(worst["force"], r["force"] / max(1.0, r["fscale"])); worst["qacc"] = max(worst["qacc"], r["qacc"] / max(1.0, r["ascale"]))
                worst["cost_gap"] = max(worst["cost_gap"], r.get("g_cost_gap", 0.0))
    assert int((env.batch.get("diverged") > 0).sum()) == 0 and int(env.batch.get("overflow").sum()) == 0
    assert checked >= 72 and agree >= checked - 2, (checked, agree)          # contact / row structure: at most 2 knife-edge envs in ~90
    # contact geometry: depth to 5e-6 m everywhere; the POINT of an MPR contact is a barycentric blend of the portal's witness points, which on a
    # thin portal (finger hull flat on the table) slides along the contact face at rounding level: position to 1e-4 m (measured 2e-5; depth 1e-7)
    assert worst["dist"] < 5e-6 and worst["pos"] < 1e-4, worst
    assert worst["force"] < 2e-3 and worst["qacc"] < 2e-3, worst              # constraint forces and accelerations, relative to the env's largest
    assert worst["cost_gap"] < 1e-6, worst                                    # the solver's objective at the kernel's acceleration: within 1e-6 of its minimum
    # Round-5 review: everything above reads the debug entry's forces at the native capacity; the fused control-step kernel's own arithmetic is seen through its
    # trajectories only.  So: ONE more control step of the whole batch through the fused kernel (both bodies of it) and, for the envs that asked for the most
    # contacts in the step before (the hand held against the table / the mount: the envs a launch waits for) plus envs spread over the batch, the same control step
    # on the fp64 oracle from the kernel's own state -- 25 substeps of collision, solve and controller each, nothing fed from one side to the other in between.
    b = env.batch
    b.set("overflow", 0)
    need = b.get("cap_need")
    pick = np.unique(np.concatenate([np.argsort(-need[:, 0], kind="stable")[:12], spread(B, 6)]))
    continue_on_the_oracle("Lift", flat, cfg, env, pick, tape[250], dq=LIFT_STEP_DQ, dv=LIFT_STEP_DV, min_over=0, assert_all=True)


def test_stack_4096_reached_states():
    """BASELINE configs[2] (per GPU): Stack / Panda / OSC_POSE, 4096 envs, 50 control steps of full-range random actions."""
    from robosuite_amd import stack
    g, cfg, flat = load_golden("seed0_full", "stack_panda")
    B = 4096
    ids = np.arange(B)
    env = stack.StackBatch(flat, cfg, ids, seed0=0, horizon=500, bank_episodes=2)
    tape = torch.tensor(lift.env_actions(ids, 400), device="cuda")
    for t in range(50):
        env.step(tape[t])
    res = compare_reached_states(flat, env.batch, spread(B, 32))
    ok = summarize("Stack step 50", res)
    assert int((env.batch.get("diverged") > 0).sum()) == 0
    assert len(res) >= 32 and len(ok) >= len(res) - 1
    assert max(r["dist"] for r in ok) < 5e-6 and max(r["pos"] for r in ok) < 5e-6
    assert max(r["force"] / max(1.0, r["fscale"]) for r in ok) < 2e-3 and max(r["qacc"] / max(1.0, r["ascale"]) for r in ok) < 2e-3
    # capacity tiers: later in the episodes (the hand pressing a cube onto the other or the table) some envs need more than the native 64 rows; none is ever
    # truncated (models/assets/base.xml:5: nconmax = 5000), and a control step of such an env agrees with the oracle's
    env.batch.set("overflow", 0)           # rsim_forward above is a debug entry: native capacity
    over, t, seen = 0, 50, 0
    while over == 0 and t < 399:
        env.batch.set("cap_need", 0)
        env.step(tape[t]); t += 1
        need = env.batch.get("cap_need")     # demand of this one control step
        seen = max(seen, int(need[:, 1].max()))
        if (need[:, 1] > 58).any():           # envs at or beyond the native capacity right now: their next control step on the oracle as well
            over = continue_on_the_oracle("Stack", flat, cfg, env, np.nonzero(need[:, 1] > 58)[0][:16], tape[t], dq=4e-4, dv=2e-2, min_over=0); t += 1     # dq: 2.2e-4 seen on one env of one build (round 5), 1e-4 typical: a whole control step of a stacked pair near the capacity
    print(f"   Stack: largest single-step demand seen {seen} rows up to control step {t}")
    assert over >= 1 and int(env.batch.get("overflow").sum()) == 0


def test_baxter_joint_velocity_2048_reached_states_with_contacts():
    """BASELINE configs[3]: TwoArmPegInHole / Baxter / JOINT_VELOCITY, 2048 envs, 50 control steps; accelerations and constraint forces are
    compared WITH contacts present (self-collisions of the arms, peg against hole)."""
    from robosuite_amd import peg_in_hole
    g, cfg, flat = load_golden("ctl_joint_velocity", "peg_baxter")
    B = 2048
    ids = np.arange(B)
    env = peg_in_hole.PegBatch(flat, cfg, ids, seed0=0, horizon=500, bank_episodes=2)
    adim = env.model.action_dim
    tape = torch.tensor(lift.env_actions(ids, 50, action_dim=adim), device="cuda")
    for t in range(50):
        env.step(tape[t])
    cyl = {i for i in range(flat.ngeom) if flat.geom_type[i] not in (0, 6)}     # everything but planes and boxes goes through MPR
    res = compare_reached_states(flat, env.batch, spread(B, 16), mpr_geoms=cyl, with_contacts=24)
    ok = summarize("Baxter step 50", res)
    withcon = [r for r in ok if r["ncon"][0] > 0]
    assert int((env.batch.get("diverged") > 0).sum()) == 0
    assert len(res) >= 32 and len(ok) >= len(res) - 2
    assert len(withcon) >= 16, "the sample must contain contact states"
    # The contacts of this workload are the elbow cylinders resting against the torso hull at micrometre depths.  MPR's last portal is then a
    # sliver; where the closest point falls on its edge instead of its interior the fp32 normal is good to ~1e-7 / depth only.  Bound: at
    # least 85 % of the contact envs agree in geometry (measured 94 %), and on those forces / accelerations are held to the numbers below
    # (measured 6e-2 / 1.3e-2 of the env's largest with the tree sums as lane loops, 1.8e-1 / 6.5e-3 with them on the matrix cores -- the reached
    # states differ in the last bits and so does the sample: friction rows of D ~ 1e4 under kilonewton normal forces from saturated velocity PIDs).
    good = [r for r in withcon if r["geom_ok"]]
    assert len(good) >= 0.85 * len(withcon), (len(good), len(withcon))
    # raw constraint forces (each side on its OWN contact geometry) are printed by summarize(), not asserted: a 1e-5 m difference in the depth of a
    # sliver contact under a kilonewton load moves a friction row by a quarter of the env's largest force without either side being wrong; the
    # comparison on identical geometry below carries the force claim (2e-3 on every contact env), the raw accelerations stay bounded here
    assert max(r["qacc"] / max(1.0, r["ascale"]) for r in good) < 4e-2
    nocon = [r for r in ok if r["ncon"][0] == 0]
    assert max(r["qacc"] / max(1.0, r["ascale"]) for r in nocon) < 2e-4
    # With the kernel's contact geometry in the oracle the kilonewton forces of this workload are compared on identical rows: every env with
    # contacts, not only those whose MPR portals agree, at the tolerances of the contact-free states (round 2 held raw forces to 25 %)
    fed = [r for r in withcon if "g_qacc" in r]
    assert len(fed) == len(withcon)
    assert max(r["g_force"] / max(1.0, r["g_fscale"]) for r in fed) < 2e-3 and max(r["g_qacc"] / max(1.0, r["g_ascale"]) for r in fed) < 2e-3


def test_pickplace_8192_with_dynamics_randomisation_reached_states():
    """BASELINE configs[4]: PickPlace (4 objects) / IIWA + Robotiq140 / OSC_POSE, 8192 envs, dynamics randomisation re-drawn before every
    control step (randomize_every_n_steps = 1), 50 control steps of full-range random actions."""
    from robosuite_amd import pick_place
    g, cfg, flat = load_golden("seed0_full", "pickplace_iiwa")
    B = 8192
    ids = np.arange(B)
    env = pick_place.PickPlaceBatch(flat, cfg, ids, seed0=0, horizon=500, bank_episodes=2, per_env_params=True)
    b = env.batch
    b.dr_save_defaults()
    tape = torch.tensor(lift.env_actions(ids, 50), device="cuda")
    for t in range(50):
        b.randomize_dynamics(seed=11, step=t)
        env.step(tape[t])
    # The Robotiq140's finger / knuckle collision meshes interpenetrate by ~1 cm in every pose (adjacent links of its four-bar linkages, not
    # parent and child, so the pair is not filtered).  MPR between two fine polytopes in deep penetration ends on one of several coplanar /
    # neighbouring portal triangles of the same facet, picked by near-ties in the vertex scans; fp32 and fp64 break those ties differently and
    # the closest point then sits on a different triangle edge (normals degrees apart, depths 1e-4 m apart, measured in half of the envs).
    # Those pairs are left out of the geometry verdict; everything else (objects on the bin floor, objects against walls and each other,
    # arm against bins) must agree.
    grip = {i for i in range(flat.ngeom) if (flat.names["geom"][i] or "").startswith("gripper0_")}
    arm, fing = np.asarray(cfg["dof_idx"]), np.asarray(cfg["grip_dof_idx"])
    groups = {"arm": arm, "gripper": fing, "objects": np.setdiff1d(np.arange(flat.nv), np.concatenate([arm, fing]))}
    # capacity tiers: the envs whose substeps asked for more than the native 32 contacts / 128 rows were stepped by the 64 x 256 configuration; nothing was dropped
    need = b.get("cap_need")
    print(f"   PickPlace demand: max {need[:, 0].max()} contacts / {need[:, 1].max()} rows; envs beyond the native capacity so far: {int(((need[:, 0] > 32) | (need[:, 1] > 128)).sum())}")
    # (before rsim_forward below: a debug entry, native capacity, drops counted.)  With the solimp draw a few envs of 8192 blow up within these 50 steps (DESIGN.md
    # section 8: the fp64 oracle does the same on this model), and a body thrown through the others can ask for more than even 64 contacts / 256 rows (71 / 242
    # seen): only such an env may have dropped anything, and at most a handful of them
    ov = np.nonzero(b.get("overflow") > 0)[0]
    assert len(ov) <= 3 and all(need[e, 0] > 64 or need[e, 1] > 256 for e in ov), (ov, need[ov])
    res = compare_reached_states(flat, b, spread(B, int(os.environ.get("RSIM_PARITY_SAMPLE", "32"))), ignore_pair=lambda g1, g2: g1 in grip and g2 in grip, dof_groups=groups)
    ok = summarize("PickPlace step 50", res)
    if os.environ.get("RSIM_PARITY_DUMP"):
        worst = sorted([r for r in res if "dump" in r], key=lambda r: -max(r["g_groups"][k][0] / max(1.0, r["g_groups"][k][1]) for k in r["g_groups"]))[:12]
        np.savez_compressed(os.environ["RSIM_PARITY_DUMP"], **{f"e{r['env']}_{k}": a for r in worst for k, a in r["dump"].items()}, envs=np.array([r["env"] for r in worst]),
                            polish=np.array([r["niter"] for r in worst]))
    nit, iters = b.get("polish"), b.get("niter")        # of the forward evaluation above: RSIM_POLISH = the polish's diagnostics (include/rsim.h)
    ex = nit // 10000000
    print(f"   Newton iterations median {int(np.median(iters))} max {int(iters.max())}; "
          f"polish exits (0 not run, 1 gradient below tolerance, 2 improvement below tolerance, 3 pass budget, 5 no descent direction, 6 a step raised the objective): {np.bincount(ex, minlength=8).tolist()}; "
          f"passes histogram {np.bincount((nit // 10000) % 10, minlength=8).tolist()}")
    assert np.isfinite(b.get("qpos")).all() and np.isfinite(b.get("obs")).all()
    assert len(res) >= 32 and len(ok) >= len(res) - 4
    good = [r for r in ok if r["geom_ok"]]
    assert len(good) >= 0.8 * len(ok), (len(good), len(ok))
    # arm and object accelerations: the gripper's 5e-5 kg m^2 links turn a 1e-3 N m residual into 20 rad/s^2, so the bound is on the arm dofs
    assert max(r["qacc_arm"] / max(1.0, r["ascale"]) for r in ok) < 2e-2
    # Everything downstream of the narrow phase on ALL 37 dofs (round 2 asserted the 7 arm dofs only): with the kernel's contact geometry in the oracle
    # -- the gripper's own finger / knuckle pairs included -- constraint forces and the accelerations of the arm, the six gripper joints and the
    # four free objects are compared on identical rows.  Bounds are relative to the env's largest force / each group's largest acceleration.
    fed = [r for r in ok if "g_qacc" in r]
    assert len(fed) == len(ok)
    # The kernel says which solves its polish did not finish (RSIM_POLISH exit 3 / 6: the pass budget ran out, or a searched step raised the objective -- MuJoCo's own solver
    # reports non-convergence through mjData warnings): they must be few, and every OTHER sampled env is held to the per-env bounds.  Exit 5 (the fp32 factor of the last
    # states gave no descent direction of the fp64 objective: the passes end at the kept point, which no pass has raised) is counted and bounded, and those envs stay IN
    # the sample that is held to the per-env maxima: 2 % of the solves end that way (profiles/r05_z_parity_pickplace.txt: 163 of 8192, none of them among the worst envs).
    ex = nit // 10000000
    unfinished, nodescent = np.isin(ex, (3, 6)), ex == 5
    print(f"   solves the polish reports unfinished (budget / raised objective): {int(unfinished.sum())} of {B} ({100.0 * unfinished.mean():.2f} %); in the sample: {[r['env'] for r in fed if unfinished[r['env']]]}")
    print(f"   solves ended for want of a descent direction: {int(nodescent.sum())} of {B} ({100.0 * nodescent.mean():.2f} %); in the sample (env, worst group error relative): "
          f"{[(r['env'], '%.1e' % max(r['g_groups'][k][0] / max(1.0, r['g_groups'][k][1]) for k in r['g_groups'])) for r in fed if nodescent[r['env']]]}")
    assert unfinished.mean() < 0.01 and nodescent.mean() < 0.05
    fed = [r for r in fed if not unfinished[r["env"]]]
    short = [r["env"] for r in fed if r["oracle_grad"] > 1e-3 and r["g_cost_gap"] < 1e-7]
    print(f"   sampled envs in which the ORACLE stopped short (its own gradient > 1e-3 at its answer, the kernel's objective not higher): {short}")
    assert len(short) <= 0.03 * len(fed) + 1
    fed = [r for r in fed if r["env"] not in short]
    gaps = np.array([r["g_cost_gap"] for r in fed])
    assert gaps.max() < PP_COST_GAP and float(np.percentile(gaps, 90)) < PP_COST_GAP_P90 and float(np.median(gaps)) < 1e-7, np.percentile(gaps, [50, 90, 100])
    med = lambda xs: float(np.median(list(xs)))   # noqa: E731
    assert med(r["g_force"] / max(1.0, r["g_fscale"]) for r in fed) < PP_FORCE_MEDIAN
    assert max(r["g_groups"]["arm"][0] / max(1.0, r["g_groups"]["arm"][1]) for r in fed) < PP_ARM_TOL
    assert max(r["g_force"] / max(1.0, r["g_fscale"]) for r in fed) < PP_FORCE_MAX
    for k, tol in PP_GROUP_MEDIAN.items():
        assert med(r["g_groups"][k][0] / max(1.0, r["g_groups"][k][1]) for r in fed) < tol, k
        assert max(r["g_groups"][k][0] / max(1.0, r["g_groups"][k][1]) for r in fed) < PP_GROUP_MAX[k], k          # every sampled env, not a percentile
    # the same rollout without the solimp draw (the one dynamics parameter whose per-step re-draw makes the restated model itself run away, fp64 oracle
    # included: DESIGN.md section 8): no env may hit the bad-state guard
    env2 = pick_place.PickPlaceBatch(flat, cfg, ids[:2048], seed0=0, horizon=500, bank_episodes=2, per_env_params=True)
    env2.batch.dr_save_defaults()
    for t in range(35):
        env2.batch.randomize_dynamics(seed=11, step=t, solimp_ratio=0.0)
        env2.step(tape[t][:2048])
    assert int(env2.batch.get("diverged").sum()) == 0


def test_stack_one_whole_control_step_of_the_slowest_envs_against_the_oracle():
    """The envs a Stack launch waits for: the 48 Newton-heaviest envs of control step 300 of the 4096-env bench workload plus 48 envs spread over the batch
    (tests/golden/stack_hard_envs_step300.npz: their states, warm starts, commands, controller records and actions before that step, recorded by
    tools/newton_hard_envs.py).  ONE whole control step -- 25 substeps with set_goal, both controllers, up to 10 contacts and 8 Newton iterations per substep --
    through the fused kernel and through the oracle from identical inputs.  Typical env to rounding; the tail is bounded by what one contact's MPR facet choice
    in fp32 does to a light cube (profiles/r04_x10_line_search.txt: with the kernel's contact geometry the two solvers agree to 3e-9)."""
    import json

    from robosuite_amd import mjcf

    adir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "robosuite_amd", "assets")
    flat = mjcf.load_model(os.path.join(adir, "stack_panda.rsim"))
    cfg = json.load(open(os.path.join(adir, "stack_panda.cfg.json")))
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stack_hard_envs_step300.npz"))
    n, n_sub = len(z["envs"]), int(z["n_sub"])
    from tests.util import make_hip, make_oracle

    hm, hb = make_hip(flat, cfg, B=n)
    for f in ("qpos", "qvel", "qacc_warmstart", "ctrl", "cstate"):
        hb.set(f, z[f])
    hb.control_step(torch.tensor(z["actions"], dtype=torch.float32, device="cuda"), n_sub)
    q1 = hb.get("qpos")
    assert int(hb.get("overflow").sum()) == 0 and int(hb.get("diverged").sum()) == 0
    dq, it = [], []
    for i in range(n):
        om, od, oc = make_oracle(flat, cfg)
        od.qpos[:] = z["qpos"][i]; od.qvel[:] = z["qvel"][i]; od.qacc_warmstart[:] = z["qacc_warmstart"][i]; od.ctrl[:] = z["ctrl"][i]
        od.forward(); oc.reset(od)
        st, cs = oc.state, z["cstate"][i]
        st[:20] = cs[:20]; st[20:24] = cs[20:24]; st[24:28] = cs[20:24]
        a = z["actions"][i].astype(np.float64)
        k = 0
        for s in range(n_sub):
            od.step1()
            if s == 0:
                oc.set_goal(od, a)
            oc.run(od)
            od.step2()
            k += od.solver_iter
        dq.append(float(np.abs(q1[i] - od.qpos).max())); it.append(k / n_sub)
    dq, it = np.array(dq), np.array(it)
    print(f"   [Stack, slowest envs] |dq| after one control step: p50 {np.median(dq):.1e} p90 {np.percentile(dq, 90):.1e} max {dq.max():.1e}; oracle Newton iterations per substep "
          f"mean {it.mean():.2f} max {it.max():.2f} (kernel, recorded: {z['newton'].mean() / n_sub:.2f} / {z['newton'].max() / n_sub:.2f})")
    assert np.median(dq) < 5e-6 and np.percentile(dq, 90) < 2e-4 and dq.max() < 5e-2, (np.median(dq), np.percentile(dq, 90), dq.max())


def test_pickplace_one_whole_control_step_of_the_fused_path_against_the_oracle():
    """BASELINE configs[4], the fused control step itself (native pass, tier pass, redo pass) rather than the debug entry: after 30 control steps of the bench's workload
    with a dynamics draw before every step, ONE more control step (a fresh draw first) of the whole batch, and for the envs that asked for the most contacts plus envs
    spread over the batch the same 25 substeps on the fp64 oracle from the kernel's own state with the env's LIVE model (this step's draw).  Nothing is fed from one side
    to the other in between, so a light object under a kilonewton squeeze amplifies what one MPR facet choice differs by: the typical env is held to rounding, the arm
    in every env, the objects' tail to what a contact-rich step of a 1e-5 kg m^2 body does with a 1e-5 difference in a normal (round 3 - 5 analyses)."""
    from robosuite_amd import pick_place
    g, cfg, flat = load_golden("seed0_full", "pickplace_iiwa")
    B = 2048
    ids = np.arange(B)
    env = pick_place.PickPlaceBatch(flat, cfg, ids, seed0=0, horizon=500, bank_episodes=2, per_env_params=True)
    b = env.batch
    b.dr_save_defaults()
    tape = torch.tensor(lift.env_actions(ids, 31), device="cuda")
    for t in range(30):
        if t == 29:
            b.set("cap_need", 0)
        b.randomize_dynamics(seed=11, step=t)
        env.step(tape[t])
    need = b.get("cap_need")
    pick = np.unique(np.concatenate([np.argsort(-need[:, 0], kind="stable")[:8], spread(B, 8)]))
    b.randomize_dynamics(seed=11, step=30)
    rows = []
    continue_on_the_oracle("PickPlace", flat, cfg, env, pick, tape[30], dq=0.0, dv=0.0, min_over=0, collect=rows)
    arm, armq, gripq = np.asarray(cfg["dof_idx"]), np.asarray(cfg["qpos_idx"]), np.asarray(cfg["grip_qpos_idx"])
    objq = np.setdiff1d(np.arange(flat.nq), np.concatenate([armq, gripq]))
    dq_all = np.array([r["dq"].max() for r in rows]); dq_arm = np.array([r["dq"][armq].max() for r in rows]); dv_arm = np.array([r["dv"][arm].max() for r in rows])
    dq_grip = np.array([r["dq"][gripq].max() for r in rows]); dq_obj = np.array([r["dq"][objq].max() for r in rows])
    print(f"   [PickPlace, fused step] {len(rows)} envs: |dq| all coordinates p50 {np.median(dq_all):.1e} p90 {np.percentile(dq_all, 90):.1e} max {dq_all.max():.1e}; arm joints |dq| max {dq_arm.max():.1e} |dv| max {dv_arm.max():.1e}")
    print(f"      gripper joints |dq| p50 {np.median(dq_grip):.1e} max {dq_grip.max():.1e}; object coordinates |dq| p50 {np.median(dq_obj):.1e} max {dq_obj.max():.1e}; arm p50 {np.median(dq_arm):.1e}")
    assert all(r["finite"] for r in rows) and int((b.get("diverged")[pick] > 0).sum()) == 0
    assert np.median(dq_all) < PP_STEP_DQ_MEDIAN and dq_arm.max() < PP_STEP_DQ_ARM and dq_all.max() < PP_STEP_DQ_MAX, (np.median(dq_all), dq_arm.max(), dq_all.max())
    assert np.median(dq_arm) < PP_STEP_DQ_ARM_MEDIAN, np.median(dq_arm)


def _oracle_dr_episode(job):
    """One PickPlace env on the fp64 oracle under the bench's own per-step dynamics randomisation: the env's draws of control step t are those of the device
    (robosuite_amd.dr.randomize_host = the host mirror of k_randomize, keyed by seed / step / env), written into the oracle's model before every control step;
    same initial state, same action tape.  Returns (env, first control step after which a coordinate was non-finite or beyond the kernel's bad-state bound, or -1;
    largest |qvel| seen at a control-step boundary)."""
    env_id, T, seed0, dr_seed = job
    import json
    from robosuite_amd import dr, pick_place
    from robosuite_amd.backend import DEFAULT_DYNAMICS_ARGS
    from tests.util import make_oracle
    adir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "robosuite_amd", "assets")
    flat = mjcf.load_model(os.path.join(adir, "pickplace_iiwa.rsim")); cfg = json.load(open(os.path.join(adir, "pickplace_iiwa.cfg.json")))
    om, od, oc = make_oracle(flat, cfg)
    cg = sorted(set(int(g) for g in flat.pair_geom1) | set(int(g) for g in flat.pair_geom2))
    base = {k: np.asarray(v, dtype=np.float64).copy() for k, v in flat.arrays.items()}
    od.qpos[:] = pick_place.episode_setup(cfg, flat.nq, seed0, [env_id], block=0)[0]
    od.qvel[:] = 0; od.qacc_warmstart[:] = 0; od.ctrl[:] = 0; od.forward(); oc.reset(od)
    acts = lift.env_actions(np.array([env_id]), T, action_dim=7)[:, 0]
    vmax = 0.0
    for t in range(T):
        o = dr.randomize_host(flat, cg, DEFAULT_DYNAMICS_ARGS, dr_seed, t, env_id, base)
        for k, v in o.items():
            om.field(k)[:] = np.asarray(v, dtype=np.float64).ravel()
        oc.env_step(od, acts[t], 25)
        q, v = np.asarray(od.qpos), np.asarray(od.qvel)
        if not (np.isfinite(q).all() and np.isfinite(v).all()) or np.abs(q).max() > 1e10 or np.abs(v).max() > 1e10:
            return env_id, t, float("inf")
        vmax = max(vmax, float(np.abs(v).max()))
    return env_id, -1, vmax


def test_pickplace_dr_bad_state_rate_against_the_oracle_on_the_same_draws():
    """Round-4 review 1(c).  The bench's PickPlace line reports a handful of envs of 8192 that hit the bad-state guard (MuJoCo's mj_checkPos / mj_checkVel semantics,
    RSIM_DIVERGED) under the reference's default dynamics randomisation re-drawn before every control step (wrappers/domain_randomization_wrapper.py:47-81,
    utils/mjmod.py:1705-1729).  Is that the restated model's physics or the kernel's arithmetic?  The SAME 8192 episodes for 60 control steps on the kernel; then,
    on the fp64 oracle with the same initial states, the same action tapes and the same draws (host mirror of the device's counter-based generator): an UNBIASED
    sample of 264 envs spread over the batch, and every env the kernel flagged.
    What the first run of this test showed (profiles/r05_c_dr_bad_state.txt): the kernel flagged 12 of 8192 envs (0.15 %); the oracle lost 2 of the 261 unflagged
    sample envs (0.8 %) and 2 of the kernel's 12 -- blow-ups are chaotic events (a Robotiq link or an object spinning up over a few control steps after an impedance
    re-draw), WHICH env goes is not reproducible across arithmetics, HOW MANY go is.  Asserted therefore as rates: the kernel's rate is small, and not above
    three times the oracle's rate on the unbiased sample plus one env's worth of slack."""
    import multiprocessing as mp
    from robosuite_amd import pick_place
    g, cfg, flat = load_golden("seed0_full", "pickplace_iiwa")
    B, T = 8192, 60
    ids = np.arange(B)
    env = pick_place.PickPlaceBatch(flat, cfg, ids, seed0=0, horizon=500, bank_episodes=2, per_env_params=True)
    b = env.batch
    b.dr_save_defaults()
    tape = torch.tensor(lift.env_actions(ids, T), device="cuda")
    first = np.full(B, -1)
    for t in range(T):
        b.randomize_dynamics(seed=11, step=t)
        env.step(tape[t])
        if t % 5 == 4 or t == T - 1:
            d = b.get("diverged") > 0
            first[(first < 0) & d] = t
    flagged = np.nonzero(first >= 0)[0]
    vk = np.abs(b.get("qvel")).max(axis=1)
    print(f"\n   kernel: {len(flagged)} of {B} envs hit the bad-state guard within {T} control steps (first seen at steps {sorted(first[flagged].tolist())}); "
          f"envs above 1e3 rad/s at the end: {int((vk > 1e3).sum())}")
    sample = spread(B, 256)
    subset = np.unique(np.concatenate([flagged[:64], sample]))
    with mp.get_context("spawn").Pool(min(32, os.cpu_count() or 1)) as pool:
        res = pool.map(_oracle_dr_episode, [(int(e), T, 0, 11) for e in subset], chunksize=2)
    gone = {e for e, t, v in res if t >= 0 or v > 1e3}          # non-finite / beyond 1e10, or running away (|qvel| > 1e3 rad/s)
    fl, smp = set(int(e) for e in flagged), set(int(e) for e in sample)
    rate_k, rate_o = len(fl) / B, len(gone & smp) / len(smp)
    print(f"   oracle, same draws: {len(gone & smp)} of the {len(smp)} sample envs lost (rate {rate_o:.2%}; kernel over the batch {rate_k:.2%}, on the sample {len(fl & smp)}); "
          f"of the kernel's {len(fl)} flagged envs the oracle loses {len(fl & gone)}; oracle-only in the sample: {sorted((gone & smp) - fl)[:12]}")
    assert rate_k <= 0.005
    assert rate_k <= 3.0 * rate_o + 1.0 / len(smp)
