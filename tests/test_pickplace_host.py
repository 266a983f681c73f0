"""Host-side PickPlace reset logic (robosuite_amd/pick_place.py) against reset states recorded from the reference's own reset code
(tools/gen_golden.py record_pickplace_resets: Robot.reset + SequentialCompositeSampler over the bin samplers)."""
import os

import numpy as np

from robosuite_amd import pick_place
from tests.util import GOLD, load_golden


def test_reset_draws_reproduce_the_reference_reset_states():
    g, cfg, flat = load_golden("seed0_full", "pickplace_iiwa")
    r = np.load(os.path.join(GOLD, "pickplace_iiwa_resets.npz"))
    pl = cfg["task"]["placement"]
    for seed in r["seeds"]:
        rng = np.random.default_rng(int(seed))
        q0 = pick_place.initial_qpos(pick_place.reset_draws(rng, pl), pl, flat.nq)
        q1 = pick_place.initial_qpos(pick_place.reset_draws(rng, pl), pl, flat.nq)
        assert np.abs(q0 - r[f"make_{seed}"]).max() < 1e-12, seed
        assert np.abs(q1 - r[f"reset_{seed}"]).max() < 1e-12, seed
    assert np.abs(pick_place.episode_setup(cfg, flat.nq, 0, [0], block=1)[0] - g["states"][0][1:1 + flat.nq]).max() < 1e-12


def test_task_program_matches_the_recorded_observation_layout():
    g, cfg, flat = load_golden("seed0_full", "pickplace_iiwa")
    t = pick_place.pick_place_task(flat, cfg)
    assert len(t["obs"]) == sum(cfg["obs_dims"]) == g["obs"].shape[1] == 114
    k = cfg["obs_keys"].index("Milk_pos")
    assert t["pos_slot"][0] == sum(cfg["obs_dims"][:k])


def test_single_object_mode_reset_and_observation_layout():
    """PickPlaceCan (single_object_mode 2, pick_place.py:840-847): the sampler still places all four objects (same RNG consumption), then the other
    three are moved out of the scene (base.py:591-602); the observation record holds the can's sensors only."""
    g, cfg, flat = load_golden("seed2_full", "pickplace_can_iiwa")
    assert cfg["task"]["single_object_mode"] == 2 and cfg["task"]["object_id"] == 3
    q_make = pick_place.episode_setup(cfg, flat.nq, 2, [0], block=0)[0]
    q_reset = pick_place.episode_setup(cfg, flat.nq, 2, [0], block=1)[0]
    assert np.abs(q_make - g["make_qpos"]).max() < 1e-12
    assert np.abs(q_reset - g["states"][0][1:1 + flat.nq]).max() < 1e-12
    for o in cfg["task"]["placement"]["objects"][:3]:
        a = o["qposadr"]
        assert list(q_reset[a:a + 7]) == [10, 10, 10, 1, 0, 0, 0]
    t = pick_place.pick_place_task(flat, cfg)
    assert t["single_object_mode"] == 2 and len(t["obs"]) == sum(cfg["obs_dims"]) == g["obs"].shape[1] == 72
    assert t["pos_slot"][3] == sum(cfg["obs_dims"][:cfg["obs_keys"].index("Can_pos")])
