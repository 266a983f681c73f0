"""Dynamics randomisation: the host mirror of the device kernel (robosuite_amd/dr.py) against the reference's DynamicsModder.randomize run over the shim
(utils/mjmod.py:1596-1640, 1705-1729; wrappers/domain_randomization_wrapper.py:47-81), and the device kernel against the mirror."""
import os

import numpy as np
import pytest

from robosuite_amd import backend, dr, factory
from tests.util import load_golden

HAVE_REF = os.path.isdir("/root/reference/robosuite")
ARGS = {k: v for k, v in backend.DEFAULT_DYNAMICS_ARGS.items()}


def _cgeoms(model):
    flat = model.flat
    idx = [model._L.rsim_model_cgeom(model.ptr, g) for g in range(flat.ngeom)]
    out = [None] * (max(idx) + 1)
    for g, c in enumerate(idx):
        if c >= 0:
            out[c] = g
    return out


def test_counter_based_draws_are_uniform_and_independent_of_the_batch():
    u = dr.dr_uniform(11, 3, 5, np.arange(200000))
    assert u.min() >= -1 and u.max() < 1 and abs(float(u.mean())) < 5e-3 and abs(float(u.var()) - 1 / 3) < 5e-3
    assert np.array_equal(dr.dr_uniform(11, 3, 5, np.arange(50)), u[:50]) and not np.array_equal(dr.dr_uniform(11, 4, 5, np.arange(50)), u[:50])
    assert abs(float(np.corrcoef(u[:-1], u[1:])[0, 1])) < 1e-2 and abs(float(np.corrcoef(dr.dr_uniform(11, 3, 6, np.arange(200000)), u)[0, 1])) < 1e-2


@pytest.mark.skipif(not HAVE_REF, reason="reference checkout not present (GPU box)")
def test_host_mirror_has_the_distribution_of_the_reference_dynamics_modder():
    """N draws of the reference's DynamicsModder over the shim (its own np.random.rand draws, ratio / size rules, clips, quaternion normalisation, free-joint and
    zero-stiffness skips) against N draws of the mirror, per parameter: mean and range of value / default (ratio) or value - default (size), and what the clip
    at zero does to parameters whose default is below the perturbation size (frictionloss 0 + U(-0.05, 0.05) -> half of the draws clip to 0)."""
    suite = factory._import_reference()
    from robosuite.utils.mjmod import DynamicsModder

    env = suite.make("Lift", robots="Panda", has_renderer=False, has_offscreen_renderer=False, use_camera_obs=False, seed=0)
    sim = env.sim
    flat = sim.model._model._flat
    model = backend.HipModel(flat)
    cg = _cgeoms(model)
    base = {k: np.array(v, dtype=np.float64).copy() for k, v in flat.arrays.items()}
    np.random.seed(0)
    # the wrapper's magnitudes (the modder's own constructor defaults differ); robosuite.wrappers cannot be imported here (h5py is absent), so the literal is
    # read out of the module's source
    import ast
    src = open("/root/reference/robosuite/wrappers/domain_randomization_wrapper.py").read()
    node = next(n for n in ast.parse(src).body if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "DEFAULT_DYNAMICS_ARGS")
    REF_ARGS = ast.literal_eval(node.value)
    assert {k.replace("_perturbation", ""): v for k, v in REF_ARGS.items() if k.endswith(("_ratio", "_size"))} == {k: v for k, v in ARGS.items() if not k.endswith("_mask")}
    modder = DynamicsModder(sim=sim, random_state=np.random.RandomState(0), **REF_ARGS)
    N = 300
    fields = ("body_pos", "body_quat", "body_inertia", "body_mass", "geom_friction", "geom_solref", "geom_solimp", "dof_frictionloss", "dof_damping", "dof_armature")
    ref = {k: [] for k in fields}
    opt_ref = []
    for _ in range(N):
        modder.randomize()
        for k in fields:
            ref[k].append(np.array(getattr(sim.model, k), dtype=np.float64).copy())
        opt_ref.append((float(sim.model.opt.density), float(sim.model.opt.viscosity)))
    modder.restore_defaults()
    mir = {k: [] for k in fields}
    for n in range(N):
        o = dr.randomize_host(flat, cg, ARGS, seed=7, step=n, env=0, base=base)
        for k in fields:
            mir[k].append(np.asarray(o[k], dtype=np.float64).reshape(base[k].shape))
    cgs = np.array(cg)
    hinge = np.array([i for i in range(flat.nv) if flat.jnt_type[flat.dof_jntid[i]] != 0])
    free = np.array([i for i in range(flat.nv) if flat.jnt_type[flat.dof_jntid[i]] == 0])
    sel = {"body_pos": slice(1, None), "body_quat": slice(1, None), "body_inertia": slice(1, None), "body_mass": slice(1, None), "geom_friction": cgs, "geom_solref": cgs,
           "geom_solimp": cgs, "dof_frictionloss": hinge, "dof_damping": hinge, "dof_armature": hinge}
    kind = {"body_pos": ("size", 0.0015), "body_inertia": ("ratio", 0.02), "body_mass": ("ratio", 0.02), "geom_friction": ("ratio", 0.1), "geom_solref": ("ratio", 0.1),
            "geom_solimp": ("ratio", 0.1), "dof_frictionloss": ("size", 0.05), "dof_damping": ("size", 0.01), "dof_armature": ("size", 0.01)}
    for k, (typ, mag) in kind.items():
        b0 = base[k].reshape(ref[k][0].shape)[sel[k]]
        stats = []
        for arr in (ref[k], mir[k]):
            a = np.stack([x[sel[k]] for x in arr])          # [N, elements...]
            if typ == "ratio":
                ok = np.abs(b0) > 0
                r = (a[:, ok] / b0[ok]).ravel()
            else:
                r = (a - b0[None]).ravel()
            stats.append((float(r.mean()), float(r.min()), float(r.max()), float((a == 0).mean())))
        (m0, lo0, hi0, z0), (m1, lo1, hi1, z1) = stats
        centre = 1.0 if typ == "ratio" else 0.0
        # identical rule => identical support; means agree to sampling error; the share of exact zeros (clip at 0) agrees
        assert abs(lo0 - lo1) < 0.05 * mag + 1e-12 and abs(hi0 - hi1) < 0.05 * mag + 1e-12, (k, stats)
        assert abs(m0 - m1) < 0.1 * mag + 1e-12 and abs(z0 - z1) < 0.05, (k, stats)
        if k == "geom_solref":   # the clip to (0, 1): time constants stay inside, a damping ratio of 1.0 is cut at 1 in half of the draws (the clip asymmetry)
            aa = [np.stack([x[sel[k]] for x in arr])[..., 1] for arr in (ref[k], mir[k])]
            assert max(a.max() for a in aa) <= 1.0 + 1e-6 and abs(aa[0].mean() - aa[1].mean()) < 5e-3 and abs((aa[0] == 1.0).mean() - (aa[1] == 1.0).mean()) < 0.05 and (aa[1] == 1.0).mean() > 0.2
        if k in ("dof_frictionloss", "dof_damping", "dof_armature"):
            assert lo1 >= -max(float(b0.max()), 0) - 1e-9 and z1 > 0.0 if float(b0.min()) < mag else True
    # quaternions: perturbed by +-0.003 per component, then normalised
    qr, qm = np.stack([x[1:] for x in ref["body_quat"]]), np.stack([x[1:] for x in mir["body_quat"]])
    assert np.abs(np.linalg.norm(qm, axis=-1) - 1).max() < 1e-6 and np.abs(np.linalg.norm(qr, axis=-1) - 1).max() < 1e-9
    d_r, d_m = np.abs(qr - base["body_quat"][None, 1:]).max(), np.abs(qm - base["body_quat"][None, 1:]).max()
    assert 0.002 < d_m < 0.0045 and 0.002 < d_r < 0.0045
    # free joints keep their values (mjmod.py:1927); density / viscosity are ratios around the default
    for arr in (ref, mir):
        for k in ("dof_frictionloss", "dof_damping", "dof_armature"):
            assert all(np.array_equal(x[free], base[k][free]) for x in arr[k]), k
    dens = np.array([o[0] for o in opt_ref])
    md = np.array([float(dr.randomize_host(flat, cg, ARGS, 7, n, 0, base)["density"][0]) for n in range(N)])
    assert abs(dens.mean() / float(base["density"][0]) - 1) < 0.02 and abs(md.mean() / float(base["density"][0]) - 1) < 0.02
    # name subsets: only the selected bodies / joints move
    sub = dict(ARGS, **backend.dr_masks(model, body_names=["cube_main"], joint_names=["robot0_joint3"], geom_names=["cube_g0"]))
    o = dr.randomize_host(flat, cg, sub, 7, 0, 0, base)
    cb, j3 = flat.name2id("body", "cube_main"), flat.name2id("joint", "robot0_joint3")
    moved = np.nonzero(np.abs(o["body_mass"].astype(np.float64) - base["body_mass"]) > 1e-6 * np.maximum(1.0, np.abs(base["body_mass"])))[0]
    assert moved.tolist() == [cb]
    assert np.nonzero(np.abs(o["dof_damping"].astype(np.float64) - base["dof_damping"]) > 1e-6)[0].tolist() == [int(flat.jnt_dofadr[j3])]
    with pytest.raises(ValueError):
        backend.dr_masks(model, geom_names=["cube_g0_vis"])          # a visual geom: nothing the dynamics could see


@pytest.mark.gpu
def test_device_randomisation_equals_its_host_mirror():
    """rsim_randomize_dynamics on the device against dr.randomize_host element by element (same counter-based draws; float32 rounding of fused
    multiply-adds aside), with name subsets, for several envs and steps."""
    from tests.util import make_hip
    g, cfg, flat = load_golden("seed1_full")
    hm, hb = make_hip(flat, cfg, B=5, per_env=True)
    hb.dr_save_defaults()
    cg = _cgeoms(hm)
    fields = ("body_pos", "body_quat", "body_inertia", "body_mass", "geom_friction", "geom_solref", "geom_solimp", "dof_frictionloss", "dof_damping", "dof_armature")
    for step, extra in ((0, {}), (4, backend.dr_masks(hm, body_names=["cube_main", "robot0_link3"], joint_names=["robot0_joint2"], geom_names=["cube_g0", "table_collision"]))):
        args = dict(ARGS, **extra)
        hb.randomize_dynamics(seed=21, step=step, **extra)
        for e in (0, 3, 4):
            o = dr.randomize_host(flat, cg, args, 21, step, e)
            for k in fields:
                got = hb.param_get(k, e, 1)[0].reshape(np.asarray(flat.arrays[k]).shape)
                want = o[k].astype(np.float64).reshape(got.shape)
                if extra and step == 4:      # subsets: untouched elements keep the PREVIOUS draw (step 0), as in the reference where they are simply not visited
                    prev = dr.randomize_host(flat, cg, ARGS, 21, 0, e)[k].astype(np.float64).reshape(got.shape)
                    if k.startswith("body_"):
                        sel = np.array([(extra["body_mask"] >> b) & 1 for b in range(flat.nbody)], dtype=bool)
                    elif k.startswith("geom_"):
                        sel = np.zeros(flat.ngeom, dtype=bool); sel[[g for c, g in enumerate(cg) if (extra["geom_mask"] >> c) & 1]] = True
                    else:
                        sel = np.array([(extra["joint_mask"] >> int(flat.dof_jntid[i])) & 1 for i in range(flat.nv)], dtype=bool)
                    want = np.where(sel.reshape((-1,) + (1,) * (got.ndim - 1)), want, prev)
                assert np.abs(got - want).max() <= 2e-6 * max(1.0, np.abs(want).max()), (step, e, k, np.abs(got - want).max())
            opt = hb.param_get("opt", e, 1)[0]
            assert abs(opt[4] - float(o["density"][0])) < 1e-5 * max(1.0, abs(opt[4])) and abs(opt[5] - float(o["viscosity"][0])) < 1e-9 + 1e-5 * abs(opt[5])
