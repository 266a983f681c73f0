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


def test_fast_draw_path_is_bit_equal_to_the_general_one():
    """reset_draws_fast (what the reset ring's upkeep thread runs: Generator.uniform written out as low + (high - low) * random()) against reset_draws, for the
    default spec, uniform joint noise with a bounded rotation and the object kept inside the range, and a fixed rotation about another axis."""
    import copy
    from robosuite_amd import factory
    _, cfg = factory.load_shipped("lift_panda")
    specs = [cfg["reset"]]
    s2 = copy.deepcopy(cfg["reset"]); s2.pop("_np", None); s2["noise"] = dict(type="uniform", magnitude=0.05)
    s2["sampler"].update(rotation=[0.1, 0.4], ensure_object_boundary_in_range=True, x_range=[-0.1, 0.1], y_range=[-0.1, 0.2]); specs.append(s2)
    s3 = copy.deepcopy(cfg["reset"]); s3.pop("_np", None); s3["sampler"].update(rotation=0.3, rotation_axis="y"); specs.append(s3)
    for spec in specs:
        assert lift.fast_path_ok(spec)
        for seed in range(60):
            r1, r2 = np.random.default_rng(seed), np.random.default_rng(seed)
            for _ in range(3):
                a, b = lift.reset_draws(r1, spec), lift.reset_draws_fast(r2, spec)
                assert all(np.array_equal(a[k], b[k]) for k in a)
    own = copy.deepcopy(cfg["reset"]); own["sampler"]["own_rng"] = True
    assert not lift.fast_path_ok(own)


def test_prepared_spec_cache_is_bounded_and_follows_in_place_edits():
    """Round-5 advisor finding: lift.prepared() cached by id(spec) in a module global for ever and returned stale arrays after an in-place edit of the spec.
    Now: the arrays follow the spec's values, and the cache keeps a bounded number of specs."""
    from robosuite_amd import lift

    s = lift.default_reset_spec()
    rng = np.random.default_rng(3)
    a0 = lift.arm_noise(rng, s)
    s["arm_init_qpos"][0] += 0.25                      # edited in place after the first draw
    a1 = lift.arm_noise(np.random.default_rng(3), s)
    assert abs((a1[0] - a0[0]) - 0.25) < 1e-12 and np.allclose(a1[1:], a0[1:])
    s["cube"]["size_max"] = [0.05, 0.05, 0.05]
    assert lift.prepared(s)["size_max"].tolist() == [0.05, 0.05, 0.05]
    for _ in range(4 * lift._PREPARED_MAX):
        lift.prepared(lift.default_reset_spec())
    assert len(lift._PREPARED) <= lift._PREPARED_MAX
    assert lift.prepared(s)["arm"][0] == s["arm_init_qpos"][0]          # evicted meanwhile: rebuilt, same values
