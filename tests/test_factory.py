"""`robosuite_amd.make()`: the suite.make()-shaped entry to the batched path (robosuite_amd/factory.py; reference environments/base.py:23-42)."""
import json
import os

import numpy as np
import pytest

from robosuite_amd import factory, mjcf
from tests.util import load_golden

HAVE_REF = os.path.isdir("/root/reference/robosuite")
PAIRS = (("lift_panda", "seed1_full", "lift_panda"), ("stack_panda", "seed0_full", "stack_panda"),
         ("peg_baxter_joint_velocity", "ctl_joint_velocity", "peg_baxter"), ("pickplace_iiwa", "seed0_full", "pickplace_iiwa"))


@pytest.mark.parametrize("stem,tag,model", PAIRS)
def test_shipped_assets_are_the_fixture_models(stem, tag, model):
    """The shipped configuration of every BASELINE model carries the cfg the reference-recorded fixture carries, and the same compiled arrays
    (appearance aside; PickPlace's visual-object bodies are placed here, they have neither mass nor collision geoms)."""
    flat, cfg = factory.load_shipped(stem)
    g, gcfg, gflat = load_golden(tag, model)
    assert cfg == gcfg
    skip = {"site_rgba", "geom_rgba"} | ({"body_pos"} if model == "pickplace_iiwa" else set())
    for k, a in gflat.arrays.items():
        if k not in skip:
            assert np.array_equal(a, flat.arrays[k]), k
    if model == "pickplace_iiwa":
        vis = [i for i, n in enumerate(flat.names["body"]) if n.startswith("Visual")]
        rest = [i for i in range(flat.nbody) if i not in vis]
        assert np.array_equal(flat.body_pos[rest], gflat.body_pos[rest]) and np.all(flat.body_mass[vis] == 0)


@pytest.mark.skipif(not HAVE_REF, reason="reference checkout not present (GPU box)")
def test_make_reads_model_and_controller_configuration_off_the_reference_objects():
    """from_reference(): the unmodified env class constructed over the shim (KinematicsBackend), its own factory building the controllers; what
    extract() reads equals the shipped assets -- for a default configuration and for one that goes through controller_configs + constructor kwargs."""
    flat, cfg = factory.from_reference("Stack", "Panda", seed=0)
    sflat, scfg = factory.load_shipped("stack_panda")
    assert cfg == scfg and all(np.array_equal(np.ravel(flat.arrays[k]), np.ravel(sflat.arrays[k])) for k in sflat.arrays)
    from robosuite.controllers import load_part_controller_config
    from robosuite.controllers.composite.composite_controller_factory import refactor_composite_controller_config

    cc = refactor_composite_controller_config(load_part_controller_config(default_controller="JOINT_VELOCITY"), "Baxter", ["right", "left"])
    assert factory.controller_type_of(cc, "Baxter") == "JOINT_VELOCITY"
    flat, cfg = factory.from_reference("TwoArmPegInHole", "Baxter", cc, seed=0, env_configuration="single-robot", gripper_types=None)
    sflat, scfg = factory.load_shipped("peg_baxter_joint_velocity")
    assert cfg == scfg and cfg["type"] == "JOINT_VELOCITY" and len(cfg["parts"]) == 2
    assert all(np.array_equal(np.ravel(flat.arrays[k]), np.ravel(sflat.arrays[k])) for k in sflat.arrays)
    # a configuration that is NOT shipped: another arm part type through the reference's own config loader
    cc = refactor_composite_controller_config(load_part_controller_config(default_controller="JOINT_POSITION"), "Panda", ["right"])
    flat, cfg = factory.from_reference("Lift", "Panda", cc, seed=2)
    g, gcfg, gflat = load_golden("ctl_joint_position")
    assert {k: cfg[k] for k in gcfg} == gcfg and np.array_equal(flat.geom_size, gflat.geom_size)


@pytest.mark.skipif(not HAVE_REF, reason="reference checkout not present (GPU box)")
def test_make_extracts_pickplace_single_with_the_set_order_of_this_process():
    """PickPlaceSingle (single_object_mode 1): the reference draws the episode's object from a Python set, so extract() records the order in which
    THIS process iterates that set; the object keys of the observation table are named neutrally (they are the drawn object's, whichever it is);
    everything else equals the fixture recorded under PYTHONHASHSEED=0."""
    flat, cfg = factory.from_reference("PickPlaceSingle", "IIWA", seed=3)
    g, gcfg, gflat = load_golden("seed3", "pickplace_single_iiwa")
    t, gt = cfg["task"], gcfg["task"]
    assert t["single_object_mode"] == 1 and sorted(t["mode1_order"]) == [0, 1, 2, 3]
    assert {k: v for k, v in t.items() if k not in ("mode1_order", "object_id")} == {k: v for k, v in gt.items() if k not in ("mode1_order", "object_id")}
    assert cfg["obs_keys"] == gcfg["obs_keys"] and cfg["obs_keys"][-5:] == ["obj_to_robot0_eef_pos", "obj_to_robot0_eef_quat", "obj_pos", "obj_quat", "obj_id"]
    assert cfg["obs_dims"] == gcfg["obs_dims"] and sum(cfg["obs_dims"]) == 73
    assert all(np.array_equal(np.ravel(flat.arrays[k]), np.ravel(gflat.arrays[k])) for k in gflat.arrays if k not in ("body_pos", "body_quat", "site_rgba"))   # visual twins (and their markers) follow the draw


@pytest.mark.skipif(not HAVE_REF, reason="reference checkout not present (GPU box)")
def test_kinematics_backend_serves_constructors_but_does_not_step():
    flat, _ = factory.load_shipped("lift_panda")
    kb = factory.KinematicsBackend(flat)
    kb.forward()
    M = kb.full_M()
    assert np.allclose(M, M.T) and np.linalg.eigvalsh(M).min() > 0
    site = flat.names["site"].index("gripper0_right_grip_site")
    jp, jr = kb.jac("site", site)
    q = kb.d["qpos"].copy()
    eps, col = 1e-6, 3
    p0 = kb.d["site_xpos"].reshape(-1, 3)[site].copy()
    kb.d["qpos"][col] += eps; kb.forward()
    assert np.abs((kb.d["site_xpos"].reshape(-1, 3)[site] - p0) / eps - jp[:, col]).max() < 1e-5
    kb.d["qpos"][:] = q
    with pytest.raises(RuntimeError):
        kb.step()


def test_make_without_a_matching_asset_or_reference_says_so():
    with pytest.raises(ValueError, match="no shipped assets"):
        factory.make("Lift", "Sawyer", n_envs=2, source="assets")
    assert factory.controller_type_of(None, "Panda") == "OSC_POSE"
    assert factory.controller_type_of({"type": "BASIC", "body_parts": {"right": {"type": "JOINT_VELOCITY"}}}, "Baxter") == "JOINT_VELOCITY"


@pytest.mark.gpu
@pytest.mark.parametrize("name,robot,stem,tag,model,kw", (("Stack", "Panda", "stack_panda", "seed0_full", "stack_panda", {}),
                                                          ("PickPlace", "IIWA", "pickplace_iiwa", "seed0_full", "pickplace_iiwa", {})))
def test_make_from_shipped_assets_equals_the_fixture_path_bitwise(name, robot, stem, tag, model, kw):
    """robosuite_amd.make("Stack", "Panda", n_envs=8) on the GPU box (no reference there: shipped assets) against VecEnv built from the
    reference-recorded fixture: same reset observations, same states / observations / rewards after ten control steps, bit for bit."""
    import torch

    import robosuite_amd
    from robosuite_amd.vec_env import VecEnv

    g, cfg, flat = load_golden(tag, model)
    B = 8
    a = robosuite_amd.make(name, robot, n_envs=B, seed=3, horizon=6, bank_episodes=3, **kw)
    b = VecEnv(name, B, flat, cfg, seed=3, horizon=6, bank_episodes=3)
    oa, ob = a.reset(), b.reset()
    assert torch.equal(oa, ob) and a.action_dim == b.action_dim
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    for t in range(10):
        act = torch.rand(B, a.action_dim, device="cuda", generator=gen) * 2 - 1
        ra, rb = a.step(act), b.step(act)
        assert torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1]) and torch.equal(ra[2], rb[2]), t
    assert np.array_equal(a.env.batch.get("qpos"), b.env.batch.get("qpos")) and a.env.batch.get("ep_index").tolist() == [1] * B


@pytest.mark.gpu
def test_make_baxter_joint_velocity_from_assets_with_the_reference_kwargs():
    import torch

    import robosuite_amd

    cc = {"type": "BASIC", "body_parts": {"right": {"type": "JOINT_VELOCITY"}, "left": {"type": "JOINT_VELOCITY"}}}
    env = robosuite_amd.make("TwoArmPegInHole", "Baxter", n_envs=4, controller_configs=cc, env_configuration="single-robot", gripper_types=None)
    obs = env.reset()
    assert env.action_dim == 14 and tuple(obs.shape) == (4, 109)
    o, r, d, info = env.step(torch.zeros(4, 14, device="cuda"))
    assert torch.isfinite(o).all() and torch.isfinite(r).all()


@pytest.mark.skipif(not HAVE_REF, reason="reference checkout not present (GPU box)")
def test_two_makes_with_one_seed_agree_and_another_seed_differs():
    """The reference's determinism test (tests/test_environments/test_env_determinism.py:27-114: two make(seed=s) + reset() -> identical XML and initial
    state) at this boundary: two extractions of Lift / Panda with one seed give the same compiled model and configuration bit for bit and the same
    host-side episode, another seed another cube."""
    from robosuite_amd import lift
    a_flat, a_cfg = factory.from_reference("Lift", "Panda", seed=7)
    b_flat, b_cfg = factory.from_reference("Lift", "Panda", seed=7)
    c_flat, c_cfg = factory.from_reference("Lift", "Panda", seed=8)
    assert a_cfg == b_cfg and all(np.array_equal(a_flat.arrays[k], b_flat.arrays[k]) for k in a_flat.arrays)
    cube = a_flat.names["geom"].index("cube_g0")
    assert not np.array_equal(a_flat.geom_size[cube], c_flat.geom_size[cube])
    (s1, q1), (s2, q2), (s3, q3) = lift.episode_setup(7, [0, 1]), lift.episode_setup(7, [0, 1]), lift.episode_setup(8, [0, 1])
    assert np.array_equal(q1, q2) and np.array_equal(s1, s2) and not np.array_equal(q1, q3)
    assert np.array_equal(q1[1], q3[0])                       # env i of seed s is default_rng(s + i): the streams are keyed by seed + global env id
