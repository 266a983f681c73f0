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
