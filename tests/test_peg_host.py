"""Host-side TwoArmPegInHole logic (robosuite_amd/peg_in_hole.py) against the state the reference produced after make() + reset() (seed 0)."""
import numpy as np

from robosuite_amd import peg_in_hole
from tests.util import load_golden


def test_reset_draws_reproduce_the_reference_reset_state():
    g, cfg, flat = load_golden("ctl_joint_torque", "peg_baxter")
    q = peg_in_hole.episode_setup(0, [0], block=1)[0]      # make() consumes block 0, the user's reset() block 1
    assert np.abs(q - g["states"][0][1:15]).max() < 1e-12
    rng = np.random.default_rng(0)
    peg_in_hole.reset_draws(rng)
    d = peg_in_hole.reset_draws(rng)
    peg = flat.names["geom"].index("peg_g0")
    assert abs(d["peg_radius"] - flat.geom_size[peg][0]) < 1e-12   # the fixture's model was compiled from that draw


def test_task_program_matches_the_recorded_observation_layout():
    g, cfg, flat = load_golden("ctl_joint_torque", "peg_baxter")
    t = peg_in_hole.peg_task(flat, cfg)
    assert len(t["obs"]) == sum(cfg["obs_dims"]) == g["obs"].shape[1]


def test_per_episode_peg_rows_equal_a_full_recompile():
    """Closed-form model rows for another peg radius == the arrays of the model the reference built (and this compiler compiled) for that radius."""
    import os
    from robosuite_amd import mjcf
    from tests.util import GOLD
    g, cfg, flat0 = load_golden("ctl_joint_torque", "peg_baxter")
    flat1 = mjcf.load_model(os.path.join(GOLD, "peg_baxter_model_seed1.rsim"))
    r1 = float(flat1.geom_size[flat1.names["geom"].index("peg_g0")][0])
    assert abs(r1 - flat0.geom_size[flat0.names["geom"].index("peg_g0")][0]) > 1e-4
    rows = peg_in_hole.peg_model_rows(flat0, [r1, r1])
    for k, v in rows.items():
        ref = np.asarray(flat1.arrays[k], dtype=np.float64).ravel()
        assert np.abs(v[0] - ref).max() < 1e-9 * max(1.0, np.abs(ref).max()), k
        assert np.array_equal(v[0], v[1])
    # and the draw that produced that model is block 1 of seed 1's generator
    rng = np.random.default_rng(1)
    peg_in_hole.reset_draws(rng)
    assert abs(peg_in_hole.reset_draws(rng)["peg_radius"] - r1) < 1e-12
