"""Vectorised counterpart of `suite.make(env_name, robots=...)` + `GymWrapper` for the tasks the fused kernel carries.

    env = VecEnv("Stack", n_envs=4096, flat=..., cfg=...)       # Lift | Stack | TwoArmPegInHole | PickPlace
    obs = env.reset()                                           # device tensor [n_envs, obs_dim], the reference's per-key record concatenated
    obs, reward, done, info = env.step(actions)                 # actions: CUDA float32 [n_envs, action_dim] in [-1, 1]
    flat = env.flat_obs(obs)                                    # GymWrapper layout (wrappers/gym_wrapper.py:45-163): object-state + robot proprio keys
    cube = env.key(obs, "cubeA_pos")                            # one observable by the reference's key name

Semantics follow the reference loop (environments/base.py:277-347, 467-521): `done` is reported on the control step that reaches `horizon`,
the env restarts on the device from its next pre-drawn reset (hard reset draws in the reference's RNG order) and the next `step` continues
the new episode.  Everything returned aliases device memory owned by the backend.
"""
from __future__ import annotations

import numpy as np

from . import lift, peg_in_hole, pick_place, stack

TASKS = {"Lift": lift.LiftBatch, "Stack": stack.StackBatch, "TwoArmPegInHole": peg_in_hole.PegBatch, "PickPlace": pick_place.PickPlaceBatch}
# single-object mode 2 of PickPlace (pick_place.py:810-847): the model / task constants of the env's own fixture carry single_object_mode and object_id
_SINGLE_OBJECT = {"PickPlaceMilk": 0, "PickPlaceBread": 1, "PickPlaceCereal": 2, "PickPlaceCan": 3}   # object_to_id, pick_place.py:218
TASKS.update({n: pick_place.PickPlaceBatch for n in _SINGLE_OBJECT})
TASKS["PickPlaceSingle"] = pick_place.PickPlaceBatch     # single_object_mode = 1: the object is drawn at every reset (pick_place.py:800-807)


class VecEnv:
    def __init__(self, env_name: str, n_envs: int, flat, cfg, device: int = 0, seed: int = 0, horizon: int = 500, env_ids=None, bank_episodes: int = 4, stream_groups: int = 1):
        if env_name not in TASKS:
            raise ValueError(f"{env_name!r} has no on-device task epilogue (have {sorted(TASKS)})")
        ids = np.arange(n_envs) if env_ids is None else np.asarray(env_ids)
        if env_name in _SINGLE_OBJECT:
            # PickPlaceMilk / Bread / Cereal / Can are PickPlace with single_object_mode = 2 and the named object (pick_place.py:810-847): a cfg that
            # does not say so would silently run the four-object task under a single-object name
            t = cfg.get("task", {})
            if int(t.get("single_object_mode", 0)) != 2 or int(t.get("object_id", -1)) != _SINGLE_OBJECT[env_name]:
                raise ValueError(f"{env_name}: cfg['task'] must carry single_object_mode = 2 and object_id = {_SINGLE_OBJECT[env_name]} "
                                 f"(got {t.get('single_object_mode')}, {t.get('object_id')}); build the cfg from a {env_name} env")
        if env_name == "PickPlaceSingle" and int(cfg.get("task", {}).get("single_object_mode", 0)) != 1:
            raise ValueError("PickPlaceSingle: cfg['task'] must carry single_object_mode = 1; build the cfg from a PickPlaceSingle env")
        self.env_name = env_name
        self.env = TASKS[env_name](flat, cfg, ids, device=device, seed0=seed, horizon=horizon, bank_episodes=bank_episodes)
        self.n_envs, self.horizon, self.bank_episodes = len(ids), horizon, bank_episodes
        if stream_groups > 1:   # env blocks stepped on their own HIP streams (rsim_set_stream_groups): same results, no whole-batch tail per step
            self.env.batch.set_stream_groups(min(int(stream_groups), self.n_envs))
        self.action_dim, self.obs_dim = self.env.model.action_dim, self.env.model.nobs
        keys, dims = cfg["obs_keys"], cfg["obs_dims"]
        off = np.cumsum([0] + list(dims))
        self.obs_slices = {k: slice(int(off[i]), int(off[i + 1])) for i, k in enumerate(keys)}
        self._object_keys = [k for k in keys if not k.startswith("robot0_")]
        self._proprio_keys = [k for k in keys if k.startswith("robot0_")]

    @property
    def action_spec(self):
        return -np.ones(self.action_dim), np.ones(self.action_dim)

    def reset(self, seed=None):
        """Every env back to episode 0 of its stream (with `seed`: of the streams keyed by that seed, env i = default_rng(seed + i)).  The reset ring is re-installed for episodes 0 .. E-1: it has moved on by then, and resets
        read from slots that still held later episodes would silently break `episode k of env i = episode_setup(seed, i, k)`."""
        e = self.env
        e._bank_stop()      # the upkeep thread draws from the same per-env generators (EpisodeStreams): it must be gone before they are re-keyed / replayed
        if seed is not None and int(seed) != e.seed0:
            e.seed0, e._streams = int(seed), None
        e.reset(block=0)
        b = e.batch
        b.set("ep_step", 0); b.set("ep_index", 0); b.set("done", 0)
        if self.bank_episodes:
            e.install_reset_bank(self.bank_episodes)
        b.observe()
        return e.obs()

    def step(self, actions):
        self.env.step(actions)
        # gym auto-reset convention: for an env whose episode just ended, `obs` is already the reset observation of its next episode and the
        # last record of the finished one is in info["terminal_obs"] (rows of envs that did not finish are stale)
        return self.env.obs(), self.env.reward(), self.env.batch.tensor("done"), {"success": self.env.success(), "terminal_obs": self.env.batch.tensor("terminal_obs")}

    def key(self, obs, name: str):
        return obs[:, self.obs_slices[name]]

    def flat_obs(self, obs, keys=None):
        """GymWrapper default: ["object-state", "robot0_proprio-state"] = object keys then robot keys; or an explicit list of observable keys."""
        import torch

        keys = self._object_keys + self._proprio_keys if keys is None else keys
        return torch.cat([obs[:, self.obs_slices[k]] for k in keys], dim=1)


class AlternatingVecEnv:
    """The batch as two halves stepped alternately -- the "alternating sampler" of RL frameworks, closed-loop compatible:

        env = AlternatingVecEnv("Lift", 4096, flat, cfg)
        obs = env.reset()                                # [obs of half 0, obs of half 1]
        for k in (0, 1): env.step_half(k, policy(obs[k]))
        while training:
            for k in (0, 1):
                o, r, d, info = env.wait_half(k)         # half k's step has completed: its observations are final
                env.step_half(k, policy(o))              # its next step is enqueued while the OTHER half is still stepping

    A lockstep launch of all envs ends when its slowest env does, with the chip half empty for the last third of it; here that
    drain is filled by the other half's launch, and a half's step t + 1 still depends only on its own step t.  Each half is a VecEnv of its own
    (its own rsim batch and HIP stream); env i behaves exactly as env i of one big batch (per-env seeding by GLOBAL env id).
    Measured by `bench.py` as `config.double_buffered` (Lift 4096: 1.33 M env-steps/s against 1.13 M lockstep on the same box)."""

    def __init__(self, env_name: str, n_envs: int, flat, cfg, env_ids=None, **kw):
        ids = np.arange(n_envs) if env_ids is None else np.asarray(env_ids)
        h = len(ids) // 2
        if h < 1:
            raise ValueError("AlternatingVecEnv needs at least two envs")
        self.halves = [VecEnv(env_name, h, flat, cfg, env_ids=ids[:h], **kw), VecEnv(env_name, len(ids) - h, flat, cfg, env_ids=ids[h:], **kw)]
        self.n_envs, self.action_dim, self.obs_dim = len(ids), self.halves[0].action_dim, self.halves[0].obs_dim

    def reset(self, seed=None):
        return [e.reset(seed=seed) for e in self.halves]

    def step_half(self, k: int, actions):
        """Enqueue one control step of half k (returns at once)."""
        self.halves[k].env.step(actions)

    def wait_half(self, k: int):
        """Block until half k's last enqueued step has completed; (obs, reward, done, info) of that step, device tensors."""
        e = self.halves[k]
        e.env.batch.sync()
        return e.env.obs(), e.env.reward(), e.env.batch.tensor("done"), {"success": e.env.success(), "terminal_obs": e.env.batch.tensor("terminal_obs")}


class _Box:
    """Stand-in for gymnasium.spaces.Box when gymnasium is not installed (same attribute names)."""

    def __init__(self, low, high, shape, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = np.broadcast_to(np.asarray(low, dtype=dtype), shape), np.broadcast_to(np.asarray(high, dtype=dtype), shape), tuple(shape), dtype


def _box(low, high, shape):
    try:
        from gymnasium import spaces

        return spaces.Box(low=np.broadcast_to(np.float32(low), shape).copy(), high=np.broadcast_to(np.float32(high), shape).copy(), dtype=np.float32)
    except ImportError:
        return _Box(low, high, shape)


class GymVecEnv:
    """gymnasium.vector-shaped face of VecEnv, with the conventions of the reference's single-env GymWrapper (wrappers/gym_wrapper.py:45-163):
    flattened observation = the chosen keys concatenated (default `object-state` then `robot0_proprio-state`), reward range (0, reward_scale),
    the horizon reports `terminated` (the reference passes its `done` there) and `truncated` is always False.  Episodes restart on the device:
    after a terminated step `obs` is already the reset observation (gymnasium's autoreset) and info["final_observation"] holds the last record of
    the finished episode for the envs flagged in info["_final_observation"]."""

    def __init__(self, env: VecEnv, keys=None):
        self.env, self.keys = env, keys
        self.num_envs = env.n_envs
        dim = int(env.flat_obs(env.env.obs(), keys).shape[1]) if keys is not None else env.obs_dim
        lo, hi = env.action_spec
        self.single_observation_space, self.single_action_space = _box(-np.inf, np.inf, (dim,)), _box(lo[0], hi[0], (env.action_dim,))
        self.observation_space, self.action_space = _box(-np.inf, np.inf, (self.num_envs, dim)), _box(lo[0], hi[0], (self.num_envs, env.action_dim))

    def reset(self, seed=None, options=None):
        if seed is not None and not isinstance(seed, int):
            raise TypeError("Seed must be an integer type!")      # gym_wrapper.py:139-143
        # the reference's wrapper seeds numpy's global generator with it (gym_wrapper.py:136-143); the batched envs have one generator each, so a
        # seed re-keys them: env i's episode stream becomes default_rng(seed + i).  Same seed -> same episodes, another seed -> other episodes.
        return self.env.flat_obs(self.env.reset(seed=seed), self.keys), {}

    def step(self, actions):
        import torch

        obs, reward, done, info = self.env.step(actions)
        term = done.to(torch.bool)
        return (self.env.flat_obs(obs, self.keys), reward, term, torch.zeros_like(term),
                {"success": info["success"], "final_observation": self.env.flat_obs(info["terminal_obs"], self.keys), "_final_observation": term})

    def close(self):
        pass
