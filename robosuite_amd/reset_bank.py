"""Host side of the on-device episode reset: a ring of pre-drawn resets per env that never runs dry and never repeats.

The reference draws every reset fresh (MujocoEnv.reset with hard_reset, environments/base.py:277-347; placement samplers,
utils/placement_samplers.py:221-309).  The fused kernel re-initialises an env at its horizon from slot `episode % E` of a per-env
ring (include/rsim.h rsim_set_reset_bank); this mixin draws the episodes in the reference's RNG order (`episode_setup` of the task
modules: generator seeded by (seed, global env id), block = episode number) and keeps the ring ahead of every env by refilling the
consumed slots from `step()` -- a few rows per control step in steady state, read back with one small device-to-host copy of the
episode counters every `horizon // 2` steps.  RSIM_BANK_STALE counts resets that found a slot not yet refilled (0 when this keeps up).
"""
from __future__ import annotations

import numpy as np


class ResetBankMixin:
    """Expects: self.batch (HipBatch), self.B, self.horizon; the task class implements
    _bank_patch_offsets() -> list of float-table offsets patched per episode, and
    _bank_rows(idx, episode) -> (qpos [n, nq], patch values [n, P]) for the LOCAL env indices `idx`."""

    bank_episodes = 0

    def install_reset_bank(self, n_episodes: int):
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

    def refill_bank(self):
        """Store every episode a slot is free for: with the env in episode k, slots hold k+1 .. filled; k's own slot is consumed, so up to k+E fits."""
        if not self.bank_episodes:
            return 0
        ep = self.batch.get("ep_index").astype(np.int64)
        upto = ep + self.bank_episodes
        n = 0
        while True:
            todo = np.nonzero(self._bank_filled < upto)[0]
            if len(todo) == 0:
                return n
            nxt = self._bank_filled[todo] + 1
            for e in np.unique(nxt):
                idx = todo[nxt == e]
                q, p = self._bank_rows(idx, int(e))
                self.batch.refill_reset_bank(idx, np.full(len(idx), e), q, p)
                self._bank_filled[idx] = e
                n += len(idx)

    def _bank_tick(self):
        if self.bank_episodes:
            self._bank_steps += 1
            if self._bank_steps >= max(1, int(getattr(self, "horizon", 0) or 1) // 2):
                self._bank_steps = 0
                self.refill_bank()
