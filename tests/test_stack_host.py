"""Host-side Stack reset logic (robosuite_amd/stack.py) against reset states recorded from the reference's own reset code
(tools/gen_golden.py record_stack_resets: Stack._reset_internal -> Robot.reset + UniformRandomSampler.sample with rejection)."""
import os

import numpy as np

from robosuite_amd import stack
from tests.util import GOLD, load_golden


def test_reset_draws_reproduce_the_reference_reset_states():
    g = np.load(os.path.join(GOLD, "stack_panda_resets.npz"))
    for seed in g["seeds"]:
        rng = np.random.default_rng(int(seed))
        q0 = stack.initial_qpos(stack.reset_draws(rng))      # block 0: the load inside make()
        q1 = stack.initial_qpos(stack.reset_draws(rng))      # block 1: the user's reset() (hard_reset)
        assert np.abs(q0 - g[f"make_{seed}"]).max() < 1e-12
        assert np.abs(q1 - g[f"reset_{seed}"]).max() < 1e-12
    # the trajectory fixture starts from block 1 of seed 0
    gg, _, _ = load_golden("seed0_full", "stack_panda")
    assert np.abs(stack.episode_setup(0, [0], block=1)[0] - gg["states"][0][1:24]).max() < 1e-12


def test_placement_keeps_the_cubes_apart():
    for seed in range(50):
        d = stack.reset_draws(np.random.default_rng(seed))
        (pa, _), (pb, _) = d["objects"]
        assert np.linalg.norm(pa[:2] - pb[:2]) > np.hypot(0.02, 0.02) + np.hypot(0.025, 0.025)
        assert abs(pa[2] - 0.83) < 1e-12 and abs(pb[2] - 0.835) < 1e-12


def test_stack_task_program_matches_the_recorded_observation_layout():
    g, cfg, flat = load_golden("seed0_full", "stack_panda")
    t = stack.stack_task(flat, cfg)
    assert len(t["obs"]) == sum(cfg["obs_dims"]) == g["obs"].shape[1]
