"""Host side of the on-device episode reset: a ring of pre-drawn resets per env that never runs dry and never repeats.

The reference draws every reset fresh (MujocoEnv.reset with hard_reset, environments/base.py:277-347; placement samplers,
utils/placement_samplers.py:221-309).  The fused kernel re-initialises an env at its horizon from slot `episode % E` of a per-env
ring (include/rsim.h rsim_set_reset_bank); this module draws the episodes in the reference's RNG order and keeps the ring ahead of every env.

Two things changed in round 3 (both were review findings):
  * `EpisodeStreams` keeps ONE persistent generator per env (seeded by (seed, global env id), block k of its stream = episode k, the
    reference's draw order): drawing the next episode of an env costs one block of draws, whatever the episode number.  The first version
    re-seeded and replayed k + 1 blocks per refill, O(B N^2) over a run.
  * the upkeep no longer stalls `step()`: a daemon thread polls the episode counters with an asynchronous copy on a side stream
    (rsim_bank_poll_begin / rsim_bank_poll), draws the rows for the consumed slots and scatters them through pinned staging
    (rsim_refill_reset_bank_async); the stepping thread only counts steps.  `bank_stats()` reports what the upkeep cost; RSIM_BANK_STALE counts
    resets that found a slot not yet refilled (0 while the thread keeps up -- it has E - 1 whole episodes of slack).
    `refill_bank()` remains as the synchronous form (tests, `sync_bank=True`).
"""
from __future__ import annotations

import threading
import time
import weakref

import numpy as np


class EpisodeStreams:
    """draw(k, episode) -> the `episode`-th hard-reset block of env k's generator (k = LOCAL index).  Sequential requests advance the generator
    by one block; the last block of each env is kept, so asking for it again is free; anything else re-seeds and replays (as the first version
    always did)."""

    def __init__(self, seed0: int, env_ids, draw_fn, aux: bool = False):
        self.seed0, self.env_ids, self.draw_fn = int(seed0), np.asarray(env_ids, dtype=np.int64), draw_fn
        n = len(self.env_ids)
        # aux: a second persistent generator per env, handed to draw_fn as `aux=` -- for parts of a reset the reference draws from a generator of
        # their own (a user-supplied placement sampler built with rng=None owns an unseeded default_rng(): only its distribution can be matched)
        self.aux = bool(aux)
        self._aux = [None] * n
        self._rng = [None] * n
        self._next = np.zeros(n, dtype=np.int64)     # episode number the generator draws next
        self._last = [None] * n                      # draw of episode _next - 1

    def draw(self, k: int, episode: int):
        episode = int(episode)
        if self._rng[k] is None or episode < self._next[k] - 1:
            self._rng[k] = np.random.default_rng(self.seed0 + int(self.env_ids[k]))
            if self.aux:
                self._aux[k] = np.random.default_rng([self.seed0 + int(self.env_ids[k]), 0x5A3F])
            self._next[k] = 0
        if episode == self._next[k] - 1:
            return self._last[k]
        d = None
        while self._next[k] <= episode:
            d = self.draw_fn(self._rng[k], aux=self._aux[k]) if self.aux else self.draw_fn(self._rng[k])
            self._next[k] += 1
        self._last[k] = d
        return d

    def draws(self, idx, episode: int):
        return [self.draw(int(k), episode) for k in idx]


def _bank_worker(ref, wake, quit_flag, B):
    """Upkeep thread: holds only a weak reference to the env while it sleeps, so the env (and its device memory) can be collected."""
    ep = np.empty(B, dtype=np.int32)
    while True:
        wake.wait()
        if quit_flag[0]:
            return
        env = ref()
        if env is None:
            return
        env._bank_upkeep_once(ep)
        if env._bank_error is not None:
            wake.clear()
            return
        del env


class ResetBankMixin:
    """Expects: self.batch (HipBatch), self.B, self.horizon; the task class implements
    _bank_patch_offsets() -> list of float-table offsets patched per episode, and
    _bank_rows(idx, episode) -> (qpos [n, nq], patch values [n, P]) for the LOCAL env indices `idx`."""

    bank_episodes = 0
    sync_bank = False          # True: refill from step() with a blocking read of the episode counters (the round-2 behaviour)
    bank_poll_steps = 0        # control steps between two polls of the episode counters (0: horizon // 4)

    def episode_draws(self, idx, episode: int):
        """The task's reset_draws() block `episode` for the LOCAL env indices idx, from the persistent per-env generators (self._draw_fn: rng -> draw)."""
        if getattr(self, "_streams", None) is None:
            self._streams = EpisodeStreams(self.seed0, self.env_ids, self._draw_fn, aux=bool(getattr(self, "_draw_aux", False)))
        return self._streams.draws(idx, episode)

    def install_reset_bank(self, n_episodes: int):
        self._bank_stop()
        E = max(2, int(n_episodes))
        offs = list(self._bank_patch_offsets())
        idx = np.arange(self.B)
        rows = [self._bank_rows(idx, ep) for ep in range(E)]
        q = np.stack([r[0] for r in rows], axis=1).astype(np.float32)
        p = np.stack([np.asarray(r[1], dtype=np.float32).reshape(self.B, len(offs)) for r in rows], axis=1)
        self.batch.set_reset_bank(q, offs, p)
        self.bank_episodes = E
        self._bank_filled = np.full(self.B, E - 1, dtype=np.int64)     # highest episode number stored for each env
        self._bank_steps = 0
        self._bank_stat = dict(polls=0, rows=0, upkeep_s=0.0, tick_s=0.0, steps=0)

    # ---- the refill itself (either thread) -----------------------------------------------------
    def _bank_refill_from(self, ep_index, asynchronous: bool) -> int:
        """Store every episode a slot is free for: with the env in episode k, slots hold k+1 .. filled; k's own slot is consumed, so up to k+E fits."""
        upto = np.asarray(ep_index, dtype=np.int64) + self.bank_episodes
        n = 0
        while True:
            todo = np.nonzero(self._bank_filled < upto)[0]
            if len(todo) == 0:
                return n
            nxt = self._bank_filled[todo] + 1
            for e in np.unique(nxt):
                idx = todo[nxt == e]
                q, p = self._bank_rows(idx, int(e))
                (self.batch.refill_reset_bank_async if asynchronous else self.batch.refill_reset_bank)(idx, np.full(len(idx), e), q, p)
                self._bank_filled[idx] = e
                n += len(idx)

    def refill_bank(self) -> int:
        """Synchronous upkeep: blocking read of the episode counters, refill, return when the rows are in the ring."""
        if not self.bank_episodes:
            return 0
        with self._bank_lock_():
            n = self._bank_refill_from(self.batch.get("ep_index"), asynchronous=False)
            self.batch.bank_flush()
        return n

    # ---- asynchronous upkeep -----------------------------------------------------------------
    def _bank_lock_(self):
        if not hasattr(self, "_bank_lock"):
            self._bank_lock = threading.Lock()
        return self._bank_lock

    def _bank_upkeep_once(self, ep):
        """One poll + refill on the upkeep thread (holds the bank lock; the wake flag is cleared inside it, see bank_quiesce)."""
        t0 = time.perf_counter()
        try:
            with self._bank_lock_():
                self._bank_wake.clear()
                b = self.batch
                b.bank_poll_begin()
                b.bank_poll(ep, wait=True)          # hipEventSynchronize on the side stream: the GIL is released, the control steps keep running
                n = self._bank_refill_from(ep, asynchronous=True)
            self._bank_stat["polls"] += 1
            self._bank_stat["rows"] += n
        except Exception as exc:   # noqa: BLE001  (surfaced on the stepping thread by the next _bank_tick)
            self._bank_error = exc
        finally:
            self._bank_stat["upkeep_s"] += time.perf_counter() - t0

    def _bank_start(self):
        self._bank_quit, self._bank_error = [False], None
        self._bank_wake = threading.Event()
        self._bank_thread = threading.Thread(target=_bank_worker, args=(weakref.ref(self), self._bank_wake, self._bank_quit, self.B), name="rsim-reset-bank", daemon=True)
        self._bank_thread.start()

    def _bank_stop(self):
        t = getattr(self, "_bank_thread", None)
        if t is not None and t.is_alive():
            self._bank_quit[0] = True
            self._bank_wake.set()
            t.join()
        self._bank_thread = None

    def _bank_tick(self):
        """Called from every step(): counts steps, wakes the upkeep thread every `bank_poll_steps` control steps."""
        if not self.bank_episodes:
            return
        t0 = time.perf_counter()
        self._bank_steps += 1
        self._bank_stat["steps"] += 1
        horizon = int(getattr(self, "horizon", 0) or 1)
        # short episodes (tests: horizons of 2 .. 7 steps with two slots) leave an upkeep thread no slack: they refill from this thread, blocking
        sync = self.sync_bank or horizon < 32
        every = self.bank_poll_steps or max(1, horizon // (2 if sync else 4))
        if self._bank_steps >= every:
            self._bank_steps = 0
            if sync:
                self.refill_bank()
            else:
                if getattr(self, "_bank_thread", None) is None:
                    self._bank_start()
                if self._bank_error is not None:
                    raise self._bank_error
                self._bank_wake.set()
        self._bank_stat["tick_s"] += time.perf_counter() - t0

    def bank_stats(self) -> dict:
        """Cost of the ring upkeep so far: host seconds the stepping thread spent in it (`tick_s`), seconds of the upkeep thread (`upkeep_s`, runs
        beside the control steps), polls, rows refilled, and both per 1000 control steps."""
        s = dict(getattr(self, "_bank_stat", dict(polls=0, rows=0, upkeep_s=0.0, tick_s=0.0, steps=0)))
        k = 1000.0 / max(1, s["steps"])
        s["tick_ms_per_1000_steps"], s["upkeep_ms_per_1000_steps"] = 1e3 * s["tick_s"] * k, 1e3 * s["upkeep_s"] * k
        return s

    def bank_quiesce(self):
        """Wait until the upkeep thread is idle and every refill it issued is in the ring (tests; before reading RSIM_BANK_STALE-sensitive state)."""
        if getattr(self, "_bank_thread", None) is not None:
            while self._bank_wake.is_set():
                time.sleep(0.0005)
            with self._bank_lock_():
                self.batch.bank_flush()
        if getattr(self, "_bank_error", None) is not None:
            raise self._bank_error

    def __del__(self):
        try:
            self._bank_stop()
        except Exception:   # noqa: BLE001
            pass
