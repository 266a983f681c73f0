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
    def covers(a, b):      # every entry of the fixture's cfg is in the shipped one, nested dicts included
        return all(k in a and (covers(a[k], v) if isinstance(v, dict) else a[k] == v) for k, v in b.items())
    assert covers(cfg, gcfg)      # + cfg["env"] / ["reset"] / ["grasp"] / placement noise since round 4 (the fixtures predate them)
    assert cfg["env"]["n_sub"] == 25 and cfg["env"]["reward_shaping"] is True and cfg["reset"]["noise"] == {"type": "gaussian", "magnitude": 0.02}
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
    assert cfg == scfg and cfg["type"] == "JOINT_VELOCITY" and len(cfg["parts"]) == 2 and cfg["reset"]["peg"] == {"radius": [0.015, 0.03], "length": 0.13}
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
    assert t["placement"].pop("noise") == {"type": "gaussian", "magnitude": 0.02}      # round 4: the robot's initialization_noise travels with the placement
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
    a = robosuite_amd.make(name, robot, n_envs=B, seed=3, horizon=6, bank_episodes=3, reward_shaping=True, **kw)   # the fixtures were recorded with the dense reward; make() defaults to the reference's sparse one
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
    (s1, q1), (s2, q2), (s3, q3) = lift.episode_setup(7, [0, 1], spec=a_cfg["reset"]), lift.episode_setup(7, [0, 1]), lift.episode_setup(8, [0, 1], spec=c_cfg["reset"])
    assert np.array_equal(q1, q2) and np.array_equal(s1, s2) and not np.array_equal(q1, q3)
    assert np.array_equal(q1[1], q3[0])                       # env i of seed s is default_rng(s + i): the streams are keyed by seed + global env id


# ---- make() as a drop-in: what happens to the reference constructor's kwargs (round 4) -------------------------------------------------------
@pytest.mark.parametrize("stem", ("make_lift_panda_sparse", "make_lift_sawyer"))
def test_reset_draws_follow_the_constructor_kwargs_of_the_recorded_env(stem):
    """Fixtures recorded from the reference env under kwargs that change the reset (tools/gen_golden.py record_make: initialization_noise=None on the
    Panda; uniform noise of magnitude 0.05 on a Sawyer, whose init_qpos and gripper differ): cfg["reset"] as extract() read it off the live env drives
    the host-side draws -- qpos after make() (block 0 of default_rng(seed)) and after reset() (block 1) equal the reference's to rounding."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", stem + ".npz"))
    cfg = json.load(open(os.path.join(os.path.dirname(__file__), "golden", stem + ".cfg.json")))
    from robosuite_amd import lift

    spec = cfg["reset"]
    rng = np.random.default_rng(int(g["seed"]))
    q0 = lift.initial_qpos(lift.reset_draws(rng, spec), spec)
    d1 = lift.reset_draws(rng, spec)
    q1 = lift.initial_qpos(d1, spec)
    assert np.abs(q0 - g["make_qpos"]).max() < 1e-12 and np.abs(q1 - g["reset_qpos"]).max() < 1e-12
    if stem == "make_lift_panda_sparse":
        assert spec["noise"]["magnitude"] == 0.0 and np.array_equal(d1["arm"], np.array(spec["arm_init_qpos"]))     # no noise, but the draw was made
        assert cfg["env"] == dict(cfg["env"], reward_shaping=False, reward_scale=3.0, control_freq=10.0, n_sub=50)
    else:
        assert spec["noise"] == {"type": "uniform", "magnitude": 0.05} and np.abs(d1["arm"] - np.array(spec["arm_init_qpos"])).max() <= 0.05
        assert cfg["grip_sign"] == [1.0, -1.0] and cfg["grasp"]["left_pad"] != ["gripper0_right_finger1_pad_collision"]    # RethinkGripper, not the Panda's


def test_task_programs_take_reward_flavour_and_scale_from_the_cfg():
    from robosuite_amd import lift, stack

    g = os.path.join(os.path.dirname(__file__), "golden")
    cfg = json.load(open(os.path.join(g, "make_lift_panda_sparse.cfg.json")))
    flat = mjcf.load_model(os.path.join(g, "make_lift_panda_sparse.rsim"))
    t = lift.lift_task(flat, cfg)
    assert t["reward_shaping"] is False and t["reward_scale"] == 3.0 and len(t["obs"]) == 60
    assert lift.lift_task(flat, dict(cfg, env=dict(cfg["env"], reward_scale=None)))["reward_scale"] == 2.25      # None: the raw 2.25 (lift.py:270-271)
    sflat, scfg = factory.load_shipped("stack_panda")
    assert stack.stack_task(sflat, scfg)["reward_shaping"] is True                                                  # shipped = the benchmark's dense reward
    sparse = factory.apply_host_kwargs(sflat, scfg, dict(reward_shaping=False, reward_scale=None, initialization_noise=None, control_freq=50))
    ts = stack.stack_task(sflat, sparse)
    assert ts["reward_shaping"] is False and ts["reward_scale"] == 2.0 and sparse["env"]["n_sub"] == 10 and sparse["reset"]["noise"]["magnitude"] == 0.0
    with pytest.raises(ValueError, match="whole number"):
        factory.apply_host_kwargs(sflat, scfg, dict(control_freq=30))
    # an observation record without the object keys (use_object_obs=False) gets a program without them, and a record the program cannot produce is refused
    noobj = dict(cfg, obs_keys=[k for k in cfg["obs_keys"] if k.startswith("robot0_")], obs_dims=[d for k, d in zip(cfg["obs_keys"], cfg["obs_dims"]) if k.startswith("robot0_")])
    assert len(lift.lift_task(flat, noobj)["obs"]) == 50
    with pytest.raises(NotImplementedError):
        lift.lift_task(flat, dict(cfg, obs_dims=cfg["obs_dims"] + [3], obs_keys=cfg["obs_keys"] + ["extra"]))


def test_make_refuses_what_it_cannot_honour():
    for kw in (dict(has_renderer=True), dict(use_camera_obs=True), dict(has_offscreen_renderer=True), dict(hard_reset=False)):
        with pytest.raises(NotImplementedError):
            factory.make("Lift", "Panda", n_envs=2, **kw)


@pytest.mark.skipif(not HAVE_REF, reason="reference checkout not present (GPU box)")
def test_extract_reads_constructor_kwargs_off_the_live_env():
    """from_reference with the kwargs the round-3 review found dropped: they arrive in cfg; a shipped cfg patched by make() equals what the reference
    constructor produces; a user's placement_initializer reaches the host-side sampler (its own unseeded generator flagged); a kwarg the reference
    does not know raises there (nothing is dropped on the way)."""
    assert factory._import_reference() is not None
    from robosuite.utils.placement_samplers import UniformRandomSampler

    off = factory.RENDER_OFF
    flat, cfg = factory.from_reference("Lift", "Panda", seed=1, defaults={}, **off, reward_shaping=False, reward_scale=3.0, initialization_noise=None, control_freq=10)
    sflat, scfg = factory.load_shipped("lift_panda")
    patched = factory.apply_host_kwargs(sflat, scfg, dict(reward_shaping=False, reward_scale=3.0, initialization_noise=None, control_freq=10))
    for k in ("reward_shaping", "reward_scale", "control_freq", "n_sub"):
        assert cfg["env"][k] == patched["env"][k], k
    assert cfg["reset"] == patched["reset"] and cfg["grasp"] == patched["grasp"]
    # the reference's own defaults: sparse reward
    _, dcfg = factory.from_reference("Lift", "Panda", seed=1, defaults={}, **off)
    assert dcfg["env"]["reward_shaping"] is False and dcfg["env"]["horizon"] == 1000 and dcfg["reset"]["noise"]["magnitude"] == 0.02
    sampler = UniformRandomSampler(name="ObjectSampler", x_range=[-0.1, 0.1], y_range=[0.0, 0.2], rotation=[0.1, 0.4], rotation_axis="z", z_offset=0.02,
                                   ensure_object_boundary_in_range=True, reference_pos=np.array((0.05, 0, 0.8)))
    _, pcfg = factory.from_reference("Lift", "Panda", seed=1, defaults={}, **off, placement_initializer=sampler)
    sm = pcfg["reset"]["sampler"]
    assert sm["x_range"] == [-0.1, 0.1] and sm["rotation"] == [0.1, 0.4] and sm["z_offset"] == 0.02 and sm["ensure_object_boundary_in_range"] and sm["own_rng"]
    from robosuite_amd import lift
    d = lift.reset_draws(np.random.default_rng(3), pcfg["reset"], aux=np.random.default_rng(4))
    r = float(np.linalg.norm(d["size"][:2]))
    assert -0.1 + r + 0.05 <= d["pos"][0] <= 0.1 - r + 0.05 and r <= d["pos"][1] <= 0.2 - r and abs(d["pos"][2] - (0.82 + d["size"][2])) < 1e-12
    ang = 2 * np.arctan2(d["quat"][3], d["quat"][0])
    assert 0.1 <= ang <= 0.4
    with pytest.raises(NotImplementedError, match="UniformRandomSampler"):
        class MySampler(UniformRandomSampler):      # a sampler with a draw order of its own: refused, not replaced by the default
            def _sample_x(self, r):
                return 0.0
        factory.from_reference("Lift", "Panda", seed=1, defaults={}, **off, placement_initializer=MySampler(name="s", x_range=[0, 0], y_range=[-0.1, 0.1], reference_pos=np.array((0, 0, 0.8))))
    with pytest.raises(TypeError):
        factory.from_reference("Lift", "Panda", seed=1, defaults={}, **off, no_such_kwarg=1)


def _replay_make_fixture(stem, B=2):
    import torch

    from robosuite_amd import lift

    gdir = os.path.join(os.path.dirname(__file__), "golden")
    g = np.load(os.path.join(gdir, stem + ".npz"))
    cfg = json.load(open(os.path.join(gdir, stem + ".cfg.json")))
    flat = mjcf.load_model(os.path.join(gdir, stem + ".rsim"))
    env = lift.LiftBatch(flat, cfg, np.arange(B), per_env_cube=False)      # the fixture's own cube (the model of its reset) in every env
    nq, b = flat.nq, env.batch
    s0 = g["states"][0]
    b.set("qpos", s0[1:1 + nq][None].repeat(B, 0)); b.set("qvel", s0[1 + nq:][None].repeat(B, 0)); b.set("qacc_warmstart", 0); b.set("ctrl", 0); b.set("time", 0)
    b.forward(); b.ctrl_reset()
    out = []
    for t in range(len(g["actions"])):
        env.step(torch.tensor(np.repeat(g["actions"][t][None], B, 0), dtype=torch.float32, device="cuda"))       # n_sub = cfg["env"]["n_sub"]
        out.append((b.get("qpos")[0].copy(), b.get("obs")[0].copy(), float(b.get("reward")[0]), int(b.get("success")[0]), b.get("ctrl")[0].copy()))
    return g, cfg, flat, env, out


@pytest.mark.gpu
def test_sparse_reward_scale_and_control_frequency_of_the_constructor_reach_the_device():
    """Fixture recorded from the reference env built with reward_shaping=False, reward_scale=3.0, initialization_noise=None, control_freq=10 under a
    closed-loop grasp-and-lift policy (tools/gen_golden.py record_make): the batched env built from that cfg steps 50 substeps per control step, pays
    0 until the cube is 4 cm above the table and exactly 3.0 from then on (lift.py:256-271), and reaches success within two steps of the reference."""
    g, cfg, flat, env, out = _replay_make_fixture("make_lift_panda_sparse")
    assert env.n_sub == 50 and env.model.nobs == 60
    nq = flat.nq
    rew, succ = np.array([o[2] for o in out]), np.array([o[3] for o in out])
    assert set(np.round(rew, 5).tolist()) == {0.0, 3.0} and np.array_equal(rew > 0, succ > 0)
    t_ref, t_dev = int(np.argmax(g["success"] > 0)), int(np.argmax(succ > 0))
    assert g["success"].sum() > 10 and abs(t_ref - t_dev) <= 2 and np.array_equal(g["rewards"] > 0, g["success"] > 0)
    for t in range(min(t_ref, t_dev) - 12):        # approach and descent (before the fingers close on the cube): the two trajectories coincide
        assert np.abs(out[t][0][:9] - g["states"][t + 1][1:10]).max() < 2e-3, t
    assert abs(float(env.batch.get("time")[0]) - 0.1 * len(out)) < 1e-3       # 70 control steps of 0.1 s


@pytest.mark.gpu
def test_lift_with_a_sawyer_from_the_make_boundary():
    """Lift / Sawyer + RethinkGripper (kernel configuration 2: 36 bodies), dense reward, uniform joint noise: observation record, reward and actuator
    commands of the batched env against what the reference env.step() returned."""
    g, cfg, flat, env, out = _replay_make_fixture("make_lift_sawyer")
    dims = np.cumsum([0] + cfg["obs_dims"])
    for t, (q, obs, rew, succ, ctrl) in enumerate(out):
        assert np.abs(q - g["states"][t + 1][1:1 + flat.nq]).max() < 2e-3, t
        assert np.abs(ctrl - g["ctrl"][t]).max() < 2e-2 * max(1.0, np.abs(g["ctrl"][t]).max()), t
        for k, key in enumerate(cfg["obs_keys"]):
            ref, got = g["obs"][t][dims[k]:dims[k + 1]], obs[dims[k]:dims[k + 1]]
            if key.endswith("quat") or key.endswith("quat_site"):
                got = got * np.sign(np.dot(got, ref))
            tol = 5e-2 * max(1.0, np.abs(ref).max()) if key.endswith("joint_acc") else (2e-2 if key.endswith("vel") else 2e-3)
            assert np.abs(got - ref).max() < tol, (t, key)
        assert abs(rew - g["rewards"][t]) < 1e-3 and succ == 0, t
