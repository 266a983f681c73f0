"""ctypes binding of the C-ABI HIP backend (include/rsim.h -> librsim_hip.so).

There is NO CPU fallback: if the shared library is missing, or no HIP device is visible when a batch is
created, this module raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported here.)
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import mjcf

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RSIM_LIB", os.path.join(_HERE, "librsim_hip.so"))  # RSIM_LIB: A/B a second build of the same ABI
_LIB = None

# enum rsim_field (include/rsim.h)
FIELDS = ["qpos", "qvel", "qacc_warmstart", "ctrl", "time", "cstate", "xpos", "xquat", "qM", "qfrc_bias", "qfrc_passive", "qfrc_actuator",
          "qfrc_constraint", "qacc", "cdof", "rootcom", "contact", "efc_force", "ncon", "nefc", "niter", "obs", "reward", "success", "done", "ep_step",
          "ep_index", "diverged", "overflow", "bank_stale", "terminal_obs", "sensordata", "task_object", "cap_need", "qfrc_applied", "polish"]
FIELD_ID = {n: i for i, n in enumerate(FIELDS)}
INT_FIELDS = {"polish", "ncon", "nefc", "niter", "success", "done", "ep_step", "ep_index", "diverged", "overflow", "bank_stale", "task_object", "cap_need"}
CON_REC = 24
CSTATE = 32
OBS_MAX = 128
OBS_KINDS = {"qpos": 0, "cos": 1, "sin": 2, "qvel": 3, "qacc": 4, "site_pos": 5, "body_quat": 6, "site_quat": 7, "body_pos": 8, "body_minus_site": 9, "body_minus_body": 10,
             "peg_cos": 11, "peg_t": 12, "peg_d": 13, "rel_pos": 14, "rel_quat": 15, "task_object": 16}


class RsimError(RuntimeError):
    pass


JNT_MAX = 16  # include/rsim.h RSIM_JNT_MAX


class CtrlDesc(C.Structure):
    _fields_ = [("ndof", C.c_int32), ("qpos_idx", C.c_int32 * JNT_MAX), ("dof_idx", C.c_int32 * JNT_MAX), ("act_idx", C.c_int32 * JNT_MAX), ("eef_site", C.c_int32),
                ("base_site", C.c_int32), ("kp", C.c_float * JNT_MAX), ("damping_ratio", C.c_float), ("input_min", C.c_float * JNT_MAX), ("input_max", C.c_float * JNT_MAX),
                ("output_min", C.c_float * JNT_MAX), ("output_max", C.c_float * JNT_MAX), ("uncouple_pos_ori", C.c_int32), ("nullspace_kp", C.c_float),
                ("ngrip", C.c_int32), ("grip_act", C.c_int32 * 4), ("grip_sign", C.c_float * 4), ("grip_speed", C.c_float),
                ("type", C.c_int32), ("torque_min", C.c_float * JNT_MAX), ("torque_max", C.c_float * JNT_MAX), ("impedance_mode", C.c_int32),
                ("kp_min", C.c_float * JNT_MAX), ("kp_max", C.c_float * JNT_MAX), ("damping_min", C.c_float * JNT_MAX), ("damping_max", C.c_float * JNT_MAX),
                ("interp_steps", C.c_int32), ("part_of", C.c_int32 * JNT_MAX), ("narm", C.c_int32), ("ndof2", C.c_int32), ("eef_site2", C.c_int32),
                ("base_site2", C.c_int32)]


class RsimContact(C.Structure):     # include/rsim.h rsim_contact
    _fields_ = [("dist", C.c_double), ("pos", C.c_double * 3), ("frame", C.c_double * 9), ("friction", C.c_double * 5), ("normal_force", C.c_double),
                ("geom1", C.c_int32), ("geom2", C.c_int32), ("dim", C.c_int32), ("efc_address", C.c_int32)]


# arm part-controller types with an in-kernel implementation (include/rsim.h enum rsim_ctrl_type; names = the reference's config "type" strings)
CTRL_TYPES = {"OSC_POSE": 0, "OSC_POSITION": 1, "JOINT_POSITION": 2, "JOINT_TORQUE": 3, "JOINT_VELOCITY": 4}


IMPEDANCE_MODES = {"fixed": 0, "variable": 1, "variable_kp": 2}


def control_dim(cfg: dict) -> int:
    """control_dim of the goal-update part of the action (without the variable-impedance gain entries)."""
    t = cfg.get("type", "OSC_POSE")
    return {"OSC_POSE": 6, "OSC_POSITION": 3}.get(t, len(cfg["qpos_idx"]))


def gain_dim(cfg: dict) -> int:
    """Gain entries in front of the goal update: 2n ("variable"), n ("variable_kp"), 0 ("fixed"); n = 6 for the OSC types, ndof for JOINT_POSITION."""
    mode = IMPEDANCE_MODES[cfg.get("impedance_mode", "fixed")]
    n = 6 if cfg.get("type", "OSC_POSE").startswith("OSC") else len(cfg["qpos_idx"])
    return {0: 0, 1: 2 * n, 2: n}[mode]


class TaskDesc(C.Structure):
    _fields_ = [("nobs", C.c_int32), ("obs_prog", C.c_int32 * (OBS_MAX * 3)), ("task", C.c_int32), ("object_body", C.c_int32), ("grip_site", C.c_int32),
                ("table_height", C.c_float), ("lift_margin", C.c_float), ("reward_scale", C.c_float), ("reward_shaping", C.c_int32),
                ("left_pad_geoms", C.c_uint64), ("right_pad_geoms", C.c_uint64), ("object_geoms", C.c_uint64), ("object2_body", C.c_int32),
                ("object2_geoms", C.c_uint64), ("nobj", C.c_int32), ("obj_body", C.c_int32 * 4), ("obj_geoms", C.c_uint64 * 4), ("pos_slot", C.c_int32 * 4),
                ("eef_body", C.c_int32), ("bin2_pos", C.c_float * 3), ("bin_size", C.c_float * 2), ("bin_target", C.c_float * 8), ("single_object_mode", C.c_int32)]


class DrDesc(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("density_ratio", "viscosity_ratio", "position_size", "quaternion_size", "inertia_ratio", "mass_ratio",
                                         "friction_ratio", "solref_ratio", "solimp_ratio", "frictionloss_size", "damping_size", "armature_size", "stiffness_ratio")] + \
               [(n, C.c_uint64) for n in ("body_mask", "geom_mask", "joint_mask")]


# DEFAULT_DYNAMICS_ARGS of the reference (wrappers/domain_randomization_wrapper.py:47-81); masks 0 = all bodies / geoms / joints (the wrapper's `*_names=None`)
DEFAULT_DYNAMICS_ARGS = dict(density_ratio=0.1, viscosity_ratio=0.1, position_size=0.0015, quaternion_size=0.003, inertia_ratio=0.02, mass_ratio=0.02,
                             friction_ratio=0.1, solref_ratio=0.1, solimp_ratio=0.1, frictionloss_size=0.05, damping_size=0.01, armature_size=0.01,
                             stiffness_ratio=0.1, body_mask=0, geom_mask=0, joint_mask=0)


def dr_masks(model, body_names=None, geom_names=None, joint_names=None) -> dict:
    """`body_names` / `geom_names` / `joint_names` of the reference wrapper as the bit sets of rsim_dr_desc (geoms by colliding-geom index; a geom that never
    collides has no dynamics parameter that matters and is left out).  None = all, as there."""
    flat = model.flat
    out = {}

    def bit(kind, name, limit=64):
        try:
            i = flat.name2id(kind, name)
        except KeyError:
            raise ValueError(f'dynamics randomisation: the model has no {kind} named "{name}"') from None
        if not 0 <= i < limit:
            raise ValueError(f'dynamics randomisation: {kind} "{name}" has id {i}; name subsets are 64-bit sets (ids 0..63)')
        return i

    if body_names is not None:
        out["body_mask"] = sum(1 << bit("body", n) for n in set(body_names))
    if joint_names is not None:
        out["joint_mask"] = sum(1 << bit("joint", n) for n in set(joint_names))
    if geom_names is not None:
        cg = {model._L.rsim_model_cgeom(model.ptr, bit("geom", n, 1 << 30)) for n in geom_names}
        if any(c >= 64 for c in cg):
            raise ValueError("dynamics randomisation: a named geom has colliding-geom index >= 64; name subsets are 64-bit sets")
        out["geom_mask"] = sum(1 << c for c in cg if c >= 0)
    for k, v in out.items():
        if v == 0:
            raise ValueError(f"{k}: the subset selects nothing (an empty set cannot be told from `all`; set the magnitudes to 0 instead)")
    return out


def two_arm_osc_desc(cfg: dict) -> CtrlDesc:
    """Two OSC arm parts (cfg["parts"], `robot.arms` order): the second arm's entries at offset 8 of the per-joint / per-axis arrays (include/rsim.h)."""
    a, b = cfg["parts"]
    d = ctrl_desc({**a, "part_of": [0] * len(a["qpos_idx"])})
    d.narm, d.ndof2, d.eef_site2, d.base_site2 = 2, len(b["qpos_idx"]), b["eef_site"], b["base_site"]
    if b.get("grip_act"):
        raise RsimError("two OSC arm parts: a gripper is only supported on the first arm")
    for i in range(d.ndof2):
        d.qpos_idx[8 + i], d.dof_idx[8 + i], d.act_idx[8 + i] = b["qpos_idx"][i], b["dof_idx"][i], b["act_idx"][i]
    for i in range(control_dim(b)):
        d.input_min[8 + i], d.input_max[8 + i], d.output_min[8 + i], d.output_max[8 + i] = b["input_min"][i], b["input_max"][i], b["output_min"][i], b["output_max"][i]
    for i, v in enumerate(b["kp"][:6]):
        d.kp[8 + i] = v
    return d


def ctrl_desc(cfg: dict) -> CtrlDesc:
    """Build the C struct from the dict form used by tests/golden/*.cfg.json and robosuite_amd.env."""
    if cfg.get("type", "OSC_POSE").startswith("OSC") and len(cfg.get("parts", [])) == 2 and all(p.get("type", "").startswith("OSC") for p in cfg["parts"]):
        return two_arm_osc_desc(cfg)
    d = CtrlDesc()
    n = len(cfg["qpos_idx"])
    d.ndof = n
    for i in range(n):
        d.qpos_idx[i], d.dof_idx[i], d.act_idx[i] = cfg["qpos_idx"][i], cfg["dof_idx"][i], cfg["act_idx"][i]
    d.eef_site, d.base_site = cfg.get("eef_site", 0), cfg.get("base_site", 0)
    for i, p in enumerate(cfg.get("part_of", [0] * n)):
        d.part_of[i] = p
    ctype = cfg.get("type", "OSC_POSE")
    if ctype not in CTRL_TYPES:
        raise RsimError(f"part controller type {ctype!r} has no in-kernel implementation (have {sorted(CTRL_TYPES)})")
    d.type = CTRL_TYPES[ctype]
    cdim = control_dim(cfg)
    for i, v in enumerate(cfg.get("kp", [])[:JNT_MAX]):
        d.kp[i] = v
    for k in ("input_min", "input_max", "output_min", "output_max"):
        if len(cfg[k]) != cdim:  # noqa: E501
            raise RsimError(f"controller {ctype}: {k} has {len(cfg[k])} entries, control_dim is {cdim}")
    for i in range(cdim):
        d.input_min[i], d.input_max[i] = cfg["input_min"][i], cfg["input_max"][i]
        d.output_min[i], d.output_max[i] = cfg["output_min"][i], cfg["output_max"][i]
    tl = cfg.get("torque_limits") or cfg.get("velocity_limits")
    if tl:
        for i in range(n):
            d.torque_min[i], d.torque_max[i] = tl[0][i], tl[1][i]
    d.damping_ratio = cfg.get("damping_ratio", 1.0)
    d.interp_steps = int(cfg.get("interp_steps", 0))
    d.impedance_mode = IMPEDANCE_MODES[cfg.get("impedance_mode", "fixed")]
    if d.impedance_mode:
        ng = 6 if ctype.startswith("OSC") else n
        kl, dl = cfg["kp_limits"], cfg["damping_ratio_limits"]
        for i in range(ng):
            d.kp_min[i], d.kp_max[i] = np.broadcast_to(kl[0], (ng,))[i], np.broadcast_to(kl[1], (ng,))[i]
            d.damping_min[i], d.damping_max[i] = np.broadcast_to(dl[0], (ng,))[i], np.broadcast_to(dl[1], (ng,))[i]
    d.uncouple_pos_ori = int(cfg.get("uncouple", 1))
    d.nullspace_kp = cfg.get("nullspace_kp", 10.0)
    g = cfg.get("grip_act", [])
    d.ngrip = len(g)
    for i in range(len(g)):
        d.grip_act[i] = g[i]
        d.grip_sign[i] = cfg["grip_sign"][i]
    d.grip_speed = cfg.get("grip_speed", 0.0)
    return d


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RsimError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(make -C robosuite_amd/csrc). There is no CPU fallback.")
        import torch  # noqa: F401  (must come first: librsim_hip.so binds to the HIP runtime torch ships, see csrc/Makefile)

        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.rsim_last_error.restype = C.c_char_p
        L.rsim_model_create.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(vp)]
        if hasattr(L, "rsim_model_compile"):       # (RSIM_LIB may name a build of an earlier round for an A/B: it has no compiler inside)
            L.rsim_model_compile.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(vp)]
            L.rsim_mjcf_to_blob.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_size_t)]
            L.rsim_blob_free.argtypes = [vp]
            L.rsim_blob_free.restype = None
        L.rsim_model_free.argtypes = [vp]
        L.rsim_model_int.argtypes = [vp, C.c_char_p]
        L.rsim_model_set_controller.argtypes = [vp, C.POINTER(CtrlDesc)]
        L.rsim_model_set_task.argtypes = [vp, C.POINTER(TaskDesc)]
        L.rsim_model_cgeom.argtypes = [vp, C.c_int]
        L.rsim_model_config.argtypes = [vp, C.POINTER(C.c_int * 10)]
        L.rsim_batch_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
        L.rsim_batch_free.argtypes = [vp]
        L.rsim_batch_size.argtypes = [vp]
        L.rsim_batch_limits.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.rsim_reset.argtypes = [vp, C.c_char_p]
        L.rsim_dr_save_defaults.argtypes = [vp]
        L.rsim_randomize_dynamics.argtypes = [vp, C.POINTER(DrDesc), C.c_uint64, C.c_uint64]
        L.rsim_set_episode.argtypes = [vp, C.c_int]
        L.rsim_set_reset_bank.argtypes = [vp, C.c_int, C.c_int, vp, vp]
        L.rsim_refill_reset_bank.argtypes = [vp, C.c_int, vp, vp, vp]
        L.rsim_bank_poll_begin.argtypes = [vp]
        L.rsim_bank_poll.argtypes = [vp, vp, C.c_int]
        L.rsim_refill_reset_bank_async.argtypes = [vp, C.c_int, vp, vp, vp]
        L.rsim_bank_flush.argtypes = [vp]
        L.rsim_param_offset.argtypes = [vp, C.c_char_p, C.c_int]
        for f in ("rsim_forward", "rsim_step1", "rsim_step2", "rsim_step", "rsim_sync", "rsim_observe", "rsim_run_controller", "rsim_step2_last"):
            getattr(L, f).argtypes = [vp]
        L.rsim_control_step.argtypes = [vp, vp, C.c_int]
        L.rsim_ctrl_reset.argtypes = [vp, C.c_char_p]
        L.rsim_get_array.argtypes = [vp, C.c_int, vp, C.c_size_t]
        L.rsim_set_array.argtypes = [vp, C.c_int, vp, C.c_size_t]
        L.rsim_device_ptr.restype = vp
        L.rsim_device_ptr.argtypes = [vp, C.c_int, C.POINTER(C.c_size_t)]
        L.rsim_stream.restype = vp
        L.rsim_stream.argtypes = [vp]
        L.rsim_jac_site.argtypes = [vp, C.c_int, C.c_int, vp, vp]
        L.rsim_jac_body.argtypes = [vp, C.c_int, C.c_int, vp, vp]
        L.rsim_model_param_set.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, vp, C.c_size_t]
        L.rsim_model_param_get.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, vp, C.c_size_t]
        L.rsim_profile.argtypes = [vp, C.c_int, vp, C.c_int]
        L.rsim_wavelog.argtypes = [vp, vp]
        L.rsim_pairlog.argtypes = [vp, vp]
        L.rsim_set_schedule.argtypes = [vp, C.c_int]
        L.rsim_set_stream_groups.argtypes = [vp, C.c_int]
        L.rsim_group_stream.restype = vp; L.rsim_group_stream.argtypes = [vp, C.c_int]
        L.rsim_profile_env.argtypes = [vp, C.c_int]
        L.rsim_tier_snapshot.argtypes = [vp, vp]
        L.rsim_tier_stats.argtypes = [vp, vp]
        L.rsim_tuning_defaults.restype = C.c_char_p
        L.rsim_name2id.argtypes = [vp, C.c_char_p, C.c_char_p]
        L.rsim_id2name.argtypes = [vp, C.c_char_p, C.c_int]; L.rsim_id2name.restype = C.c_char_p
        L.rsim_full_M.argtypes = [vp, C.c_int, vp]
        L.rsim_contacts.argtypes = [vp, C.c_int, C.c_int, vp]
        L.rsim_comm_unique_id.argtypes = [vp, C.c_size_t]
        L.rsim_comm_create.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
        L.rsim_allreduce_stats.argtypes = [vp, vp, C.c_int, C.c_int]
        L.rsim_comm_free.argtypes = [vp]; L.rsim_comm_free.restype = None
        _LIB = L
    return _LIB


def _chk(rc):
    if rc != 0:
        raise RsimError(lib().rsim_last_error().decode())


def tuning_sha16() -> str:
    """sha256[:16] of rsim_tuning_defaults(): the host-side dispatch / solver settings a measurement was taken under (bench.py pmc_evidence)."""
    import hashlib
    return hashlib.sha256(lib().rsim_tuning_defaults()).hexdigest()[:16]


class HipComm:
    """rsim_comm handle (include/rsim.h): RCCL communicator for the rollout-statistics reduction, for hosts without torch.distributed.
    `HipComm.unique_id()` on rank 0, hand the 128 bytes to every rank, then `HipComm(uid, rank, world, device)` on all of them."""

    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(HipComm.ID_BYTES)
        _chk(lib().rsim_comm_unique_id(C.cast(buf, C.c_void_p), HipComm.ID_BYTES))
        return buf.raw

    def __init__(self, uid: bytes, rank: int, world: int, device: int = 0):
        self.ptr = C.c_void_p()
        self.rank, self.world = int(rank), int(world)
        buf = C.create_string_buffer(bytes(uid), len(uid))
        _chk(lib().rsim_comm_create(C.cast(buf, C.c_void_p), len(uid), self.rank, self.world, int(device), C.byref(self.ptr)))

    def allreduce(self, values, op: str = "sum"):
        """Reduction of a float64 vector over all ranks (op: "sum" | "max"); returns a new array."""
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        _chk(lib().rsim_allreduce_stats(self.ptr, v.ctypes.data_as(C.c_void_p), v.size, {"sum": 0, "max": 1}[op]))
        return v

    def close(self):
        if self.ptr:
            lib().rsim_comm_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass


def compile_mjcf_blob(xml: str, asset_dir: str | None = None) -> bytes:
    """MJCF string -> model blob through the C-ABI's own compiler (include/rsim.h rsim_mjcf_to_blob, csrc/rsim_mjcf.cpp): what a binder without Python gets from
    rsim_model_compile, 10 - 20 x faster than robosuite_amd.mjcf.compile_mjcf, which remains the checker (tests/test_mjcf_cpp.py)."""
    b = xml.encode("utf-8") if isinstance(xml, str) else bytes(xml)
    p, n = C.c_void_p(), C.c_size_t()
    _chk(lib().rsim_mjcf_to_blob(b, len(b), asset_dir.encode() if asset_dir else None, C.byref(p), C.byref(n)))
    try:
        return C.string_at(p, n.value)
    finally:
        lib().rsim_blob_free(p)


class HipModel:
    """rsim_model handle (host side only; no GPU needed)."""

    @classmethod
    def from_xml_string(cls, xml: str, asset_dir: str | None = None):
        """mujoco.MjModel.from_xml_string (binding_utils.py:1077-1080) through rsim_model_compile: the MJCF string is compiled inside the library."""
        return cls(compile_mjcf_blob(xml, asset_dir))

    def __init__(self, flat_or_blob):
        self.flat = flat_or_blob if isinstance(flat_or_blob, mjcf.FlatModel) else mjcf.from_blob(flat_or_blob)
        blob = mjcf.to_blob(self.flat) if isinstance(flat_or_blob, mjcf.FlatModel) else bytes(flat_or_blob)
        self._L = lib()
        self.ptr = C.c_void_p()
        _chk(self._L.rsim_model_create(blob, len(blob), C.byref(self.ptr)))
        self.ctrl_cfg = None
        self.task_cfg = None
        self.nobs = 0

    def int(self, name):
        return self._L.rsim_model_int(self.ptr, name.encode())

    def name2id(self, kind: str, name: str) -> int:
        """mj_name2id through the C-ABI (rsim_name2id): -1 if there is no such name."""
        return self._L.rsim_name2id(self.ptr, kind.encode(), name.encode())

    def id2name(self, kind: str, i: int):
        """mj_id2name through the C-ABI (rsim_id2name): None for an unnamed object or a bad id."""
        r = self._L.rsim_id2name(self.ptr, kind.encode(), int(i))
        return None if r is None else r.decode()

    def kernel_config(self):
        """(config id, limits dict) of the compiled kernel configuration that serves this model; id -1 = unsupported size."""
        lim = (C.c_int * 10)()
        c = self._L.rsim_model_config(self.ptr, C.byref(lim))
        return c, dict(zip(("nbody", "njnt", "nv", "ncgeom", "nsite", "ncon", "nefc", "npair", "ntree", "tendons"), list(lim)))

    def set_controller(self, cfg: dict):
        d = ctrl_desc(cfg)
        _chk(self._L.rsim_model_set_controller(self.ptr, C.byref(d)))
        self.ctrl_cfg = cfg
        self.action_dim = self._L.rsim_model_int(self.ptr, b"action_dim")   # control_dim (x arm parts) + gain entries + gripper entry
        self.cstate_size = self._L.rsim_model_int(self.ptr, b"cstate_size")

    def set_task(self, task: dict):
        """task: dict(obs=[(kind, a, b), ...], task="lift", object_body, grip_site, table_height, lift_margin, reward_scale, reward_shaping,
        left_pad_geoms=[geom ids], right_pad_geoms=[...], object_geoms=[...]) -- see include/rsim.h rsim_task_desc."""
        d = TaskDesc()
        obs = task["obs"]
        if len(obs) > OBS_MAX:
            raise RsimError(f"observation record of {len(obs)} floats exceeds RSIM_OBS_MAX")
        d.nobs = len(obs)
        for i, (kind, a, b) in enumerate(obs):
            d.obs_prog[3 * i], d.obs_prog[3 * i + 1], d.obs_prog[3 * i + 2] = OBS_KINDS[kind] if isinstance(kind, str) else int(kind), int(a), int(b)
        d.task = {"none": 0, "lift": 1, "stack": 2, "peg_in_hole": 3, "pick_place": 4}[task.get("task", "none")]
        d.object2_body = int(task.get("object2_body", 0))
        d.object_body, d.grip_site = int(task.get("object_body", 0)), int(task.get("grip_site", 0))
        d.table_height, d.lift_margin = float(task.get("table_height", 0.0)), float(task.get("lift_margin", 0.04))
        d.reward_scale, d.reward_shaping = float(task.get("reward_scale", 1.0)), int(bool(task.get("reward_shaping", True)))

        def mask(geoms):
            mk = 0
            for g in geoms:
                c = self._L.rsim_model_cgeom(self.ptr, int(g))
                if c >= 0:
                    mk |= 1 << c
            return mk

        d.left_pad_geoms, d.right_pad_geoms, d.object_geoms = mask(task.get("left_pad_geoms", [])), mask(task.get("right_pad_geoms", [])), mask(task.get("object_geoms", []))
        d.object2_geoms = mask(task.get("object2_geoms", []))
        if d.task == 4:
            d.nobj, d.eef_body = len(task["obj_body"]), int(task["eef_body"])
            d.single_object_mode = int(task.get("single_object_mode", 0))
            for i in range(d.nobj):
                d.obj_body[i], d.pos_slot[i], d.obj_geoms[i] = int(task["obj_body"][i]), int(task["pos_slot"][i]), mask(task["obj_geoms"][i])
            for i in range(3):
                d.bin2_pos[i] = task["bin2_pos"][i]
            d.bin_size[0], d.bin_size[1] = task["bin_size"][0], task["bin_size"][1]
            for i in range(d.nobj):
                d.bin_target[2 * i], d.bin_target[2 * i + 1] = task["bin_target"][i][0], task["bin_target"][i][1]
        _chk(self._L.rsim_model_set_task(self.ptr, C.byref(d)))
        self.task_cfg = task
        self.nobs = len(obs)

    def __del__(self):
        try:
            if self.ptr:
                self._L.rsim_model_free(self.ptr)
        except Exception:
            pass


class _DevArray:
    """Exposes a library-owned device buffer through __cuda_array_interface__ (torch.as_tensor aliases it, zero-copy)."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2, "strides": None}
        self._owner = owner


class HipBatch:
    """B environments resident on one MI355X."""

    def __init__(self, model: HipModel, B: int, device: int = 0, per_env_params: bool = False):
        self._L = lib()
        self.model = model
        self.B = B
        self.device = device
        self.ptr = C.c_void_p()
        _chk(self._L.rsim_batch_create(model.ptr, B, device, int(per_env_params), C.byref(self.ptr)))
        mc, me = C.c_int(), C.c_int()
        self._L.rsim_batch_limits(self.ptr, C.byref(mc), C.byref(me))
        self.maxcon, self.maxefc = mc.value, me.value
        m = model.flat
        nq, nv, nu, nb = m.nq, m.nv, m.nu, m.nbody
        self.shapes = {"qpos": (B, nq), "qvel": (B, nv), "qacc_warmstart": (B, nv), "ctrl": (B, nu), "time": (B,), "cstate": (B, getattr(model, "cstate_size", CSTATE)),
                       "xpos": (B, nb, 3), "xquat": (B, nb, 4), "qM": (B, nv, nv), "qfrc_bias": (B, nv), "qfrc_passive": (B, nv),
                       "qfrc_actuator": (B, nv), "qfrc_constraint": (B, nv), "qacc": (B, nv), "cdof": (B, nv, 6), "rootcom": (B, nb, 3),
                       "contact": (B, self.maxcon, CON_REC), "efc_force": (B, self.maxefc), "ncon": (B,), "nefc": (B,), "niter": (B,),
                       "obs": (B, model.nobs), "reward": (B,), "success": (B,), "done": (B,), "ep_step": (B,), "ep_index": (B,), "diverged": (B,), "overflow": (B,), "bank_stale": (B,), "terminal_obs": (B, model.nobs),
                       "sensordata": (B, int(m.arrays["sensor_dim"].sum()) if getattr(m, "nsensor", 0) else 0), "task_object": (B,), "cap_need": (B, 2), "qfrc_applied": (B, m.nv), "polish": (B,)}

    # ---- state access (host copies) --------------------------------------------------------
    def get(self, name):
        dt = np.int32 if name in INT_FIELDS else np.float32
        out = np.empty(self.shapes[name], dtype=dt)
        _chk(self._L.rsim_get_array(self.ptr, FIELD_ID[name], out.ctypes.data, out.size))
        return out

    def set(self, name, value):
        dt = np.int32 if name in INT_FIELDS else np.float32
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dtype=dt), self.shapes[name]))
        _chk(self._L.rsim_set_array(self.ptr, FIELD_ID[name], a.ctypes.data, a.size))

    def tensor(self, name):
        """torch tensor aliasing the device buffer (no copy)."""
        import torch

        cnt = C.c_size_t()
        p = self._L.rsim_device_ptr(self.ptr, FIELD_ID[name], C.byref(cnt))
        typestr = "<i4" if name in INT_FIELDS else "<f4"
        return torch.as_tensor(_DevArray(p, self.shapes[name], typestr, self), device=f"cuda:{self.device}")

    # ---- simulation ------------------------------------------------------------------------
    def reset(self, mask=None):
        _chk(self._L.rsim_reset(self.ptr, None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8).tobytes()))

    def forward(self):
        _chk(self._L.rsim_forward(self.ptr))

    def step1(self):
        _chk(self._L.rsim_step1(self.ptr))

    def step2(self):
        _chk(self._L.rsim_step2(self.ptr))

    def step(self):
        _chk(self._L.rsim_step(self.ptr))

    def sync(self):
        _chk(self._L.rsim_sync(self.ptr))

    def step2_last(self):
        """Last step2 of a host-driven control step: + observation / reward / horizon epilogue (include/rsim.h rsim_step2_last)."""
        _chk(self._L.rsim_step2_last(self.ptr))

    def run_controller(self):
        """One evaluation of the in-kernel part controllers on the current state and controller state (include/rsim.h rsim_run_controller)."""
        _chk(self._L.rsim_run_controller(self.ptr))

    def observe(self):
        """forward() + observation / reward epilogue without advancing time (the observation `env.reset()` returns)."""
        _chk(self._L.rsim_observe(self.ptr))

    def dr_save_defaults(self):
        _chk(self._L.rsim_dr_save_defaults(self.ptr))

    def randomize_dynamics(self, seed: int, step: int, **args):
        """Re-draw the dynamics parameters of every env around the saved defaults (reference DynamicsModder.randomize)."""
        a = {**DEFAULT_DYNAMICS_ARGS, **args}
        d = DrDesc(**{k: (int(v) if k.endswith("_mask") else float(v)) for k, v in a.items()})
        _chk(self._L.rsim_randomize_dynamics(self.ptr, C.byref(d), int(seed), int(step)))

    def param_get(self, field, env0=0, nenv=None):
        """Live values of a float model array for envs [env0, env0+nenv): float64 [nenv, ...] in the compiled model's layout."""
        nenv = self.B - env0 if nenv is None else nenv
        shape = np.asarray(self.model.flat.arrays[field]).shape if field != "opt" else (10,)
        out = np.zeros((nenv, int(np.prod(shape))), dtype=np.float64)
        _chk(self._L.rsim_model_param_get(self.ptr, field.encode(), int(env0), int(nenv), out.ctypes.data, out.shape[1]))
        return out.reshape((nenv,) + tuple(shape))

    def set_episode(self, horizon: int):
        _chk(self._L.rsim_set_episode(self.ptr, int(horizon)))

    def param_offset(self, field: str, elem: int) -> int:
        return self._L.rsim_param_offset(self.ptr, field.encode(), int(elem))

    def set_reset_bank(self, qpos, patch_idx=(), patch_val=None):
        """qpos: [B, E, nq]; patch_idx: float-table offsets (param_offset); patch_val: [B, E, len(patch_idx)]."""
        q = np.asarray(qpos, dtype=np.float32)
        B, E, nq = q.shape
        idx = np.ascontiguousarray(patch_idx, dtype=np.int32)
        bank = q if len(idx) == 0 else np.concatenate([q, np.asarray(patch_val, dtype=np.float32).reshape(B, E, len(idx))], axis=2)
        bank = np.ascontiguousarray(bank, dtype=np.float32)
        _chk(self._L.rsim_set_reset_bank(self.ptr, E, len(idx), idx.ctypes.data if len(idx) else None, bank.ctypes.data))

    def refill_reset_bank(self, env, episode, qpos, patch_val=None):
        """Overwrite ring slots: reset `episode[i]` of env `env[i]` := qpos[i] (+ patch_val[i]); see include/rsim.h rsim_refill_reset_bank."""
        env = np.ascontiguousarray(env, dtype=np.int32); episode = np.ascontiguousarray(episode, dtype=np.int32)
        q = np.asarray(qpos, dtype=np.float32).reshape(len(env), -1)
        rows = q if patch_val is None or np.size(patch_val) == 0 else np.concatenate([q, np.asarray(patch_val, dtype=np.float32).reshape(len(env), -1)], axis=1)
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        _chk(self._L.rsim_refill_reset_bank(self.ptr, len(env), env.ctypes.data, episode.ctypes.data, rows.ctypes.data))

    def refill_reset_bank_async(self, env, episode, qpos, patch_val=None):
        """refill_reset_bank on the batch's side stream, through pinned staging, without waiting (include/rsim.h rsim_refill_reset_bank_async)."""
        env = np.ascontiguousarray(env, dtype=np.int32); episode = np.ascontiguousarray(episode, dtype=np.int32)
        q = np.asarray(qpos, dtype=np.float32).reshape(len(env), -1)
        rows = q if patch_val is None or np.size(patch_val) == 0 else np.concatenate([q, np.asarray(patch_val, dtype=np.float32).reshape(len(env), -1)], axis=1)
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        _chk(self._L.rsim_refill_reset_bank_async(self.ptr, len(env), env.ctypes.data, episode.ctypes.data, rows.ctypes.data))

    def bank_poll_begin(self):
        _chk(self._L.rsim_bank_poll_begin(self.ptr))

    def bank_poll(self, out: np.ndarray, wait: bool = False) -> bool:
        """True + RSIM_EP_INDEX (as of the poll) in `out` (int32 [B]) once the asynchronous copy has landed."""
        r = self._L.rsim_bank_poll(self.ptr, out.ctypes.data, int(bool(wait)))
        if r < 0:
            raise RsimError(self._L.rsim_last_error().decode())
        return r == 1

    def bank_flush(self):
        _chk(self._L.rsim_bank_flush(self.ptr))

    def ctrl_reset(self, mask=None):
        _chk(self._L.rsim_ctrl_reset(self.ptr, None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8).tobytes()))

    def control_step(self, actions, n_sub=25):
        """actions: torch CUDA float32 tensor [B, action_dim] (or an int device pointer)."""
        if isinstance(actions, int):
            ptr = actions
        else:
            if not actions.is_cuda or actions.dtype.is_floating_point is False:
                raise RsimError("actions must be a CUDA float32 tensor")
            # reference: `assert len(action) == self.action_dim` (environments/robot_env.py:586); a narrower tensor would be read out of bounds on the device
            adim = self._L.rsim_model_int(self.model.ptr, b"action_dim")
            if tuple(actions.shape) != (self.B, adim):
                raise RsimError(f"actions must have shape ({self.B}, {adim}) = (n_envs, action_dim), got {tuple(actions.shape)}")
            actions = actions.contiguous().float()
            self._hold(actions)
            ptr = actions.data_ptr()
        _chk(self._L.rsim_control_step(self.ptr, C.c_void_p(ptr), int(n_sub)))

    def stream(self):
        return self._L.rsim_stream(self.ptr)

    def _hold(self, actions):
        """The step reads `actions` on the batch's own stream(s) after control_step() has returned (include/rsim.h: the buffer must stay untouched
        until the step has run), and an env block of a stream group may lag many steps behind.  Keep a reference to every tensor until the streams
        have passed a marker recorded after it (one set of events every 32 steps), so that a tensor the caller drops right away is not handed out
        again by torch's caching allocator while a step still reads it -- whatever stream the caller allocates from.  (Tensor.record_stream would say
        the same to the allocator, but then the allocator touches the batch's streams when the tensor is freed, possibly after the batch is gone.)"""
        import collections

        import torch

        if not hasattr(self, "_held"):
            self._held, self._marks, self._nstep = collections.deque(), collections.deque(), 0
        self._nstep += 1
        self._held.append((self._nstep, actions))
        if self._nstep % 32 == 0:
            evs = []
            for s_ in self._step_streams():
                e = torch.cuda.Event()
                e.record(s_)
                evs.append(e)
            self._marks.append((self._nstep, evs))
        while self._marks and all(e.query() for e in self._marks[0][1]):
            n, _ = self._marks.popleft()
            while self._held and self._held[0][0] <= n:
                self._held.popleft()

    def _step_streams(self):
        """torch views of the HIP streams control steps run on (the main stream, or one per env block with stream groups); cached per group count."""
        import torch

        key = getattr(self, "_ngroups", 1)
        if getattr(self, "_ext_key", None) != key:
            ptrs = [self.group_stream(g) for g in range(key)] if key > 1 else [self.stream()]
            self._ext, self._ext_key = [torch.cuda.ExternalStream(p, device=f"cuda:{self.device}") for p in ptrs], key
        return self._ext

    PROFILE_SLOTS = ("load", "kin", "com", "crb", "broad", "narrow", "makec", "vel", "ctrl", "act", "solve", "euler", "store",
                     "n_sub", "n_cand", "n_con", "n_efc", "n_newton", "n_ls", "boxbox", "mpr", "plane", "n_boxbox", "n_mpr", "n_support",
                     "x0", "x1", "x2", "x3", "x4", "x5", "x6", "x7", "x8", "x9")

    def profile_env(self, env=-1):
        _chk(self._L.rsim_profile_env(self.ptr, int(env)))

    def tier_snapshot(self):
        """Capacity tier of every env for the next control step (int32 [B]: 0 native configuration, 1 the wider one); synchronises the batch's stream."""
        out = np.zeros(self.B, dtype=np.int32)
        _chk(self._L.rsim_tier_snapshot(self.ptr, out.ctypes.data))
        return out

    def tier_stats(self):
        """(env-steps the wider capacity tier stepped, of these handed over / redone in mid-step) since the batch was created; synchronises the stream."""
        out = np.zeros(2, dtype=np.uint64)
        _chk(self._L.rsim_tier_stats(self.ptr, out.ctypes.data))
        return int(out[0]), int(out[1])

    def wavelog(self):
        """Per-env {hw_id, xcc_id, t_start, t_end} of the last launch (profiling must be armed)."""
        out = np.zeros((self.B, 8), dtype=np.uint64)
        _chk(self._L.rsim_wavelog(self.ptr, out.ctypes.data))
        return out

    def set_schedule(self, longest_first=True):
        """Dispatch order of control_step: slowest envs of the previous step first (default) or identity."""
        _chk(self._L.rsim_set_schedule(self.ptr, int(bool(longest_first))))

    def set_stream_groups(self, groups: int):
        """Step the batch as `groups` env blocks on their own HIP streams (include/rsim.h rsim_set_stream_groups): a block's next control step no
        longer waits for the slowest env of the whole batch.  Same results; 1 = one launch per step."""
        _chk(self._L.rsim_set_stream_groups(self.ptr, int(groups)))
        self._ngroups = int(groups)

    def group_stream(self, g: int):
        return self._L.rsim_group_stream(self.ptr, int(g))

    def pairlog(self):
        """Per candidate pair {narrow-phase visits, support calls} since profiling was armed -> (visits[npair], supports[npair])."""
        out = np.zeros(1280, dtype=np.uint64)
        _chk(self._L.rsim_pairlog(self.ptr, out.ctypes.data))
        n = len(self.model.flat.arrays["pair_geom1"])
        return out[:n].astype(np.int64), out[640:640 + n].astype(np.int64)

    def profile(self, enable=True):
        """Read (then re-arm or disarm) the kernel's per-phase cycle accumulators -> dict."""
        out = np.zeros(len(self.PROFILE_SLOTS), dtype=np.uint64)
        _chk(self._L.rsim_profile(self.ptr, int(enable), out.ctypes.data, len(out)))
        return dict(zip(self.PROFILE_SLOTS, out.tolist()))

    def jac_site(self, env, site):
        nv = self.model.flat.nv
        jp, jr = np.zeros((3, nv)), np.zeros((3, nv))
        _chk(self._L.rsim_jac_site(self.ptr, env, site, jp.ctypes.data, jr.ctypes.data))
        return jp, jr

    def jac_body(self, env, body):
        nv = self.model.flat.nv
        jp, jr = np.zeros((3, nv)), np.zeros((3, nv))
        _chk(self._L.rsim_jac_body(self.ptr, env, body, jp.ctypes.data, jr.ctypes.data))
        return jp, jr

    def param_set(self, field, values, env0=0):
        v = np.ascontiguousarray(values, dtype=np.float64)
        if v.ndim == 1:
            v = v[None]
        v = v.reshape(v.shape[0], -1)
        _chk(self._L.rsim_model_param_set(self.ptr, field.encode(), int(env0), v.shape[0], v.ctypes.data, v.shape[1]))

    def full_M(self, env=0):
        """mj_fullM of one env through the C-ABI (rsim_full_M): float64 [nv, nv]."""
        nv = self.model.flat.nv
        M = np.empty((nv, nv), dtype=np.float64)
        _chk(self._L.rsim_full_M(self.ptr, int(env), M.ctypes.data))
        return M

    def contacts_abi(self, env=0):
        """sim.data.contact[:ncon] of one env through the C-ABI (rsim_contacts), as dicts."""
        buf = (RsimContact * self.maxcon)()
        n = self._L.rsim_contacts(self.ptr, int(env), self.maxcon, C.cast(buf, C.c_void_p))
        if n < 0:
            raise RsimError(self._L.rsim_last_error().decode())
        return [dict(dist=c.dist, pos=np.array(c.pos), frame=np.array(c.frame).reshape(3, 3), geom1=c.geom1, geom2=c.geom2, dim=c.dim, efc_address=c.efc_address,
                     normal_force=c.normal_force, friction=np.array(c.friction)) for c in buf[:n]]

    def contacts(self, env=0):
        n = int(self.get("ncon")[env])
        rec = self.get("contact")[env]
        return [dict(dist=float(r[0]), pos=r[1:4].astype(np.float64), frame=r[4:13].astype(np.float64).reshape(3, 3), geom1=int(r[13]), geom2=int(r[14]),
                     dim=int(r[15]), efc_address=int(r[16]), normal_force=float(r[17]), friction=r[18:23].astype(np.float64)) for r in rec[:n]]

    def __del__(self):
        try:
            if self.ptr:
                self._L.rsim_batch_free(self.ptr)
        except Exception:
            pass
