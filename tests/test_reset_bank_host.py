"""Host logic of the reset ring (robosuite_amd/reset_bank.py) against a numpy stand-in for the device side of the C-ABI (rsim_set_reset_bank,
rsim_refill_reset_bank[_async], rsim_bank_poll*, the on-device reset of rsim_control_step): episode k of env i is always the k-th block of env i's
generator, no reset is ever replayed, the asynchronous upkeep keeps ahead of the envs, and the per-env generators are persistent (O(1) per refill)."""
import gc
import threading
import time

import numpy as np
import pytest

from robosuite_amd.reset_bank import EpisodeStreams, ResetBankMixin


class FakeBatch:
    """The ring as the kernel sees it: slot (episode % E) of env e, a tag per slot, a stale counter (include/rsim.h RSIM_BANK_STALE)."""

    def __init__(self, B):
        self.B = B
        self.ep_index = np.zeros(B, dtype=np.int32)
        self.stale = np.zeros(B, dtype=np.int64)
        self.lock = threading.Lock()
        self.async_calls = 0

    def set_reset_bank(self, q, offs, p):
        self.q, self.p, self.E = np.array(q), np.array(p), q.shape[1]
        self.tag = np.tile(np.arange(self.E), (self.B, 1))

    def refill_reset_bank(self, idx, episodes, q, p):
        with self.lock:
            for i, e, qr, pr in zip(idx, episodes, q, p):
                s = int(e) % self.E
                self.q[i, s], self.p[i, s], self.tag[i, s] = qr, pr, int(e)

    def refill_reset_bank_async(self, idx, episodes, q, p):
        self.async_calls += 1
        self.refill_reset_bank(idx, episodes, q, p)

    def get(self, name):
        assert name == "ep_index"
        return self.ep_index.copy()

    def bank_poll_begin(self):
        self._snap = self.ep_index.copy()

    def bank_poll(self, ep, wait=True):
        ep[:] = self._snap
        return True

    def bank_flush(self):
        pass

    def device_reset(self, envs):
        """What the fused kernel does at an env's horizon: next episode from its slot; a slot that does not hold that episode is counted, still used."""
        out = {}
        with self.lock:
            for e in envs:
                self.ep_index[e] += 1
                k = int(self.ep_index[e]); s = k % self.E
                if self.tag[e, s] != k:
                    self.stale[e] += 1
                out[e] = (k, self.q[e, s].copy(), self.p[e, s].copy())
        return out


class FakeTask(ResetBankMixin):
    def __init__(self, B, horizon, seed0=7, first_env=100):
        self.B, self.horizon, self.seed0 = B, horizon, seed0
        self.env_ids = np.arange(first_env, first_env + B)
        self.batch = FakeBatch(B)
        self.draws = 0

    def _draw_fn(self, rng):
        self.draws += 1
        return dict(q=rng.standard_normal(4), size=rng.uniform(0.02, 0.03))

    def _bank_patch_offsets(self):
        return [11]

    def _bank_rows(self, idx, episode):
        d = self.episode_draws(idx, episode)
        return np.array([x["q"] for x in d]).reshape(-1, 4), np.array([[x["size"]] for x in d]).reshape(-1, 1)


def expected(seed0, env_id, episode):
    rng = np.random.default_rng(seed0 + env_id)
    for _ in range(episode + 1):
        q, size = rng.standard_normal(4), rng.uniform(0.02, 0.03)
    return q, size


def test_episode_streams_are_persistent_per_env_generators():
    calls = [0]

    def draw(rng):
        calls[0] += 1
        return rng.standard_normal(3)

    s = EpisodeStreams(5, [10, 11], draw)
    a = [s.draw(0, k) for k in range(4)]
    assert calls[0] == 4                                        # one block per episode, not k + 1 replays
    assert np.array_equal(s.draw(0, 3), a[3]) and calls[0] == 4   # the last block is kept
    rng = np.random.default_rng(15)
    for k in range(4):
        assert np.array_equal(a[k], rng.standard_normal(3))
    assert np.array_equal(s.draw(0, 1), a[1]) and calls[0] == 6   # going back re-seeds and replays (2 blocks)
    r16 = np.random.default_rng(16)
    assert np.array_equal(s.draw(1, 2), [r16.standard_normal(3) for _ in range(3)][2])
    assert [np.array_equal(x, y) for x, y in zip(s.draws([0, 1], 2), [s.draw(0, 2), s.draw(1, 2)])] == [True, True]


@pytest.mark.parametrize("E", (2, 4))
def test_synchronous_upkeep_never_replays_an_episode(E):
    """Horizon below 32: the refill runs from step() (blocking).  Envs finish at different times; every reset loads exactly the next block of its env."""
    B, H = 6, 5
    t = FakeTask(B, H)
    t.install_reset_bank(E)
    steps = np.arange(B) % H                                   # staggered episode phases
    seen = {e: [0] for e in range(B)}
    for step in range(12 * H):
        steps += 1
        done = np.nonzero(steps >= H)[0]
        for e, (k, q, p) in t.batch.device_reset(done).items():
            eq, es = expected(t.seed0, int(t.env_ids[e]), k)
            assert np.allclose(q, eq.astype(np.float32)) and np.isclose(p[0], np.float32(es)), (e, k)
            seen[e].append(k)
        steps[done] = 0
        t._bank_tick()
    assert t.batch.stale.sum() == 0 and t.batch.async_calls == 0
    assert all(v == list(range(len(v))) and len(v) >= 11 for v in seen.values())
    # persistent generators: E initial blocks + one block per refilled row and env, nothing replayed
    assert t.draws <= B * (E + max(len(v) for v in seen.values()) + E)


def test_asynchronous_upkeep_keeps_ahead_and_stops_with_its_env():
    """Horizon >= 32: a daemon thread polls and refills; the stepping thread only counts steps.  No stale slot over 12 episodes of every env."""
    B, H, E = 8, 32, 3
    t = FakeTask(B, H)
    t.install_reset_bank(E)
    steps = (5 * np.arange(B)) % H
    n_resets = 0
    for step in range(12 * H):
        steps += 1
        done = np.nonzero(steps >= H)[0]
        for e, (k, q, p) in t.batch.device_reset(done).items():
            eq, es = expected(t.seed0, int(t.env_ids[e]), k)
            assert np.allclose(q, eq.astype(np.float32)) and np.isclose(p[0], np.float32(es)), (e, k)
            n_resets += 1
        steps[done] = 0
        t._bank_tick()
        if step % (H // 4) == 0:
            t.bank_quiesce()                                    # the device side of this fake has no stream to order against
    t.bank_quiesce()
    st = t.bank_stats()
    assert t.batch.stale.sum() == 0 and n_resets >= 11 * B
    assert st["polls"] >= 10 and st["rows"] >= n_resets - B * E and t.batch.async_calls > 0 and st["steps"] == 12 * H
    assert st["tick_ms_per_1000_steps"] < st["upkeep_ms_per_1000_steps"] + 50.0
    th = t._bank_thread
    assert th.is_alive()
    del t
    gc.collect()
    th.join(timeout=5.0) if th.is_alive() else None
    # the worker holds only a weak reference while it sleeps: without its env it ends at the next wake-up or is a daemon at exit
    assert (not th.is_alive()) or th.daemon


def test_reinstalling_the_ring_restarts_the_episode_numbering():
    """VecEnv.reset(): the ring is re-installed for episodes 0 .. E - 1 (round 2 left the moved-on ring in place and replayed whatever it held)."""
    B, H, E = 4, 6, 2
    t = FakeTask(B, H)
    t.install_reset_bank(E)
    for _ in range(3):
        t.batch.device_reset(range(B)); t.refill_bank()
    assert (t.batch.ep_index == 3).all()
    t.batch.ep_index[:] = 0                                     # what VecEnv.reset() writes
    t.install_reset_bank(E)
    out = t.batch.device_reset(range(B))
    for e, (k, q, p) in out.items():
        eq, es = expected(t.seed0, int(t.env_ids[e]), 1)
        assert k == 1 and np.allclose(q, eq.astype(np.float32))
    assert t.batch.stale.sum() == 0
