"""Host-side Lift logic (robosuite_amd/lift.py) against values recorded from the reference's own reset code."""
import numpy as np
import pytest

from robosuite_amd import lift, mjcf
from tests.util import load_golden


@pytest.mark.parametrize("tag,seed", [("seed0_gentle", 0), ("seed1_full", 1)])
def test_reset_draws_match_reference(tag, seed):
    g, _, flat = load_golden(tag)
    rng = np.random.default_rng(seed)
    d0 = lift.reset_draws(rng)   # make()  (base.py:154-180)
    d1 = lift.reset_draws(rng)   # reset() (base.py:290-295, hard reset)
    assert lift.initial_qpos(d0) == pytest.approx(g["make_qpos"], abs=1e-12)
    assert lift.initial_qpos(d1) == pytest.approx(g["reset_qpos"], abs=1e-12)
    assert d1["size"] == pytest.approx(g["cube_size"], abs=1e-15)
    sizes, qpos = lift.episode_setup(seed, [0], block=1)
    assert qpos[0] == pytest.approx(g["reset_qpos"], abs=1e-12)


def test_cube_rows_equal_full_recompile():
    """closed-form per-env cube fields == what the MJCF compiler produced for a different cube size (the two fixtures)."""
    _, _, f0 = load_golden("seed0_gentle")
    g1, _, f1 = load_golden("seed1_full")
    rows = lift.cube_model_rows(f0, g1["cube_size"][None])
    for field, v in rows.items():
        assert v[0] == pytest.approx(np.asarray(f1.arrays[field], dtype=np.float64).ravel(), rel=1e-9, abs=1e-12), field


def test_action_streams_are_keyed_by_global_env_id():
    a = lift.env_actions([5, 6, 7], 4)
    b = lift.env_actions([7], 4)
    assert a.shape == (4, 3, 7) and np.array_equal(a[:, 2], b[:, 0])
    assert np.abs(a).max() <= 1.0
