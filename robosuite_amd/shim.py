"""A `mujoco`-shaped module so robosuite's host layers run unchanged on this project's backends.

This is the B=1 compatibility face of the drop-in boundary: robosuite's
`utils/binding_utils.py` does `import mujoco` and touches exactly the surface listed there
(`MjModel.from_xml_string`, `MjData`, `mj_forward/step/step1/step2/resetData`, `mj_jacSite/Body/Geom`,
`mj_fullM`, `mj_id2name`, `mj_saveLastXML`, the `mjtObj/mjtJoint` enums; binding_utils.py:307,507,681-851,
1079-1107; controllers/parts/controller.py:226-227).  `install(backend_factory)` registers this module as
`sys.modules["mujoco"]`; the backend does the arithmetic (HIP C-ABI in the product; the CPU oracle in tests).

Backend protocol (duck-typed):
    backend = factory(flat_model)
    backend.model_array(name) -> writable numpy view or None (then the FlatModel's own array is used)
    backend.data_array(name)  -> writable numpy view
    backend.sync_model()      -> push host-side model edits (no-op for zero-copy backends)
    backend.forward() / step() / step1() / step2() / reset()
    backend.jac(kind, idx) -> (jacp[3,nv], jacr[3,nv]);  backend.full_M() -> [nv,nv]
    backend.ncon (int); backend.contacts() -> list of dict(geom1, geom2, dist, pos, frame)
    optional: backend.nefc (int); backend.efc_array("efc_force" | "efc_aref" | "efc_R" | "efc_type") -> array of nefc entries
"""
from __future__ import annotations

import sys
import types

import numpy as np

from . import mjcf

_BACKEND_FACTORY = None


class _Enum(int):
    pass


class mjtObj:
    mjOBJ_UNKNOWN = 0
    mjOBJ_BODY = 1
    mjOBJ_XBODY = 2
    mjOBJ_JOINT = 3
    mjOBJ_DOF = 4
    mjOBJ_GEOM = 5
    mjOBJ_SITE = 6
    mjOBJ_CAMERA = 7
    mjOBJ_LIGHT = 8
    mjOBJ_MESH = 11
    mjOBJ_TENDON = 18
    mjOBJ_ACTUATOR = 19
    mjOBJ_SENSOR = 20


class mjtJoint:
    mjJNT_FREE = 0
    mjJNT_BALL = 1
    mjJNT_SLIDE = 2
    mjJNT_HINGE = 3


class mjtGeom:
    mjGEOM_PLANE, mjGEOM_HFIELD, mjGEOM_SPHERE, mjGEOM_CAPSULE, mjGEOM_ELLIPSOID, mjGEOM_CYLINDER, mjGEOM_BOX, mjGEOM_MESH = range(8)


class _Placeholder:
    """Rendering-side enums robosuite imports but the rendering-free path never evaluates."""

    def __getattr__(self, name):
        return 0


_OBJ_KIND = {1: "body", 3: "joint", 5: "geom", 6: "site", 7: "camera", 8: "light", 11: "mesh", 18: "tendon", 19: "actuator", 20: "sensor"}

# model fields exposed as attributes (class-level names matter: binding_utils._MjModelMeta enumerates dir())
_MODEL_FIELDS = [
    "body_parentid", "body_rootid", "body_weldid", "body_mocapid", "body_jntadr", "body_jntnum", "body_dofadr", "body_dofnum",
    "body_geomadr", "body_geomnum", "body_pos", "body_quat", "body_ipos", "body_iquat", "body_mass", "body_inertia",
    "body_invweight0", "body_subtreemass", "jnt_type", "jnt_qposadr", "jnt_dofadr", "jnt_bodyid", "jnt_pos", "jnt_axis",
    "jnt_limited", "jnt_range", "jnt_stiffness", "jnt_margin", "jnt_solref", "jnt_solimp", "qpos0", "qpos_spring",
    "dof_bodyid", "dof_jntid", "dof_parentid", "dof_armature", "dof_damping", "dof_frictionloss", "dof_solref", "dof_solimp",
    "dof_invweight0", "dof_M0", "geom_type", "geom_bodyid", "geom_contype", "geom_conaffinity", "geom_condim", "geom_priority",
    "geom_group", "geom_dataid", "geom_size", "geom_pos", "geom_quat", "geom_friction", "geom_solref", "geom_solimp",
    "geom_solmix", "geom_margin", "geom_gap", "geom_rbound", "geom_rgba", "site_bodyid", "site_pos", "site_quat", "site_size",
    "site_rgba", "actuator_trnid", "actuator_gear", "actuator_gainprm", "actuator_biasprm", "actuator_ctrllimited",
    "actuator_ctrlrange", "actuator_forcelimited", "actuator_forcerange", "sensor_dim",
]
_MODEL_SIZES = ["nq", "nv", "nu", "na", "nbody", "njnt", "ngeom", "nsite", "ncam", "nlight", "nsensor", "ntendon", "nmesh", "nmocap"]
_NAME_ADRS = ["name_bodyadr", "name_jntadr", "name_geomadr", "name_siteadr", "name_lightadr", "name_camadr", "name_actuatoradr",
              "name_sensoradr", "name_tendonadr", "name_meshadr"]
_DATA_FIELDS = ["qpos", "qvel", "qacc", "qacc_warmstart", "ctrl", "qfrc_applied", "mocap_pos", "mocap_quat", "xpos", "xquat", "xmat",
                "xipos", "ximat", "geom_xpos", "geom_xmat", "site_xpos", "site_xmat", "subtree_com", "qM", "qfrc_bias",
                "qfrc_passive", "qfrc_actuator", "qfrc_constraint", "actuator_force", "sensordata"]
_DATA_SHAPES = {"mocap_pos": 3, "mocap_quat": 4, "xpos": 3, "xquat": 4, "xmat": 9, "xipos": 3, "ximat": 9, "geom_xpos": 3,
                "geom_xmat": 9, "site_xpos": 3, "site_xmat": 9, "subtree_com": 3}


class _Opt:
    """mjOption view: timestep/density/viscosity/impratio/gravity are model scalars robosuite's DR touches
    (wrappers/domain_randomization_wrapper.py:47-81)."""

    _FIELDS = ("timestep", "density", "viscosity", "impratio", "tolerance")

    def __init__(self, model):
        object.__setattr__(self, "_m", model)

    def __getattr__(self, k):
        m = object.__getattribute__(self, "_m")
        if k in self._FIELDS:
            return float(m._arr(k)[0])
        if k in ("gravity", "wind"):
            return m._arr(k)
        if k == "iterations":
            return int(m._arr(k)[0])
        raise AttributeError(k)

    def __setattr__(self, k, v):
        m = object.__getattribute__(self, "_m")
        if k in self._FIELDS or k == "iterations":
            m._arr(k)[0] = v
        elif k in ("gravity", "wind"):
            m._arr(k)[:] = v
        else:
            raise AttributeError(k)
        m._dirty = True


def _model_prop(name):
    def get(self):
        return self._arr(name)

    def set_(self, value):
        self._arr(name)[...] = value
        self._dirty = True

    return property(get, set_)


class MjModel:
    """Shim for `mujoco.MjModel` backed by a compiled :class:`mjcf.FlatModel`."""

    def __init__(self, flat: mjcf.FlatModel):
        if _BACKEND_FACTORY is None:
            raise RuntimeError("robosuite_amd.shim: no backend installed (call shim.install(factory) first)")
        self._flat = flat
        self._dirty = False
        self._backend = _BACKEND_FACTORY(flat)
        self._opt = _Opt(self)

    @property
    def opt(self):
        """mjOption view.  A class-level attribute: robosuite's MjModel wrapper builds its delegating properties from `dir(mujoco.MjModel)`
        (utils/binding_utils.py:252-268), and `controller_factory` reads `sim.model.opt.timestep` for the interpolators."""
        return self._opt

    @classmethod
    def from_xml_string(cls, xml, assets=None):
        global _LAST_XML
        flat = mjcf.compile_mjcf(xml)
        _LAST_XML = xml
        return cls(flat)

    @classmethod
    def from_xml_path(cls, path):
        with open(path) as f:
            return cls.from_xml_string(f.read())

    def _arr(self, name):
        a = self._backend.model_array(name)
        if a is None:
            a = self._flat.arrays[name]
        w = mjcf._SHAPES.get(name)
        return a.reshape(-1, w) if w else a

    def _sync(self):
        if self._dirty:
            self._backend.sync_model()
            self._dirty = False

    na = 0


for _f in _MODEL_FIELDS:
    setattr(MjModel, _f, _model_prop(_f))
for _f in _MODEL_SIZES:
    if _f != "na":
        setattr(MjModel, _f, property(lambda self, _f=_f: int(self._flat.arrays[_f][0])))
for _f in _NAME_ADRS:
    setattr(MjModel, _f, property(lambda self: None))


class _Contact:
    def __init__(self, c):
        self.geom1, self.geom2 = c["geom1"], c["geom2"]
        self.geom = np.array([c["geom1"], c["geom2"]])
        self.dist, self.pos, self.frame = c["dist"], c["pos"], np.asarray(c["frame"]).reshape(-1)
        self.dim = c.get("dim", 3)


def _data_prop(name):
    def get(self):
        a = self._backend.data_array(name)
        w = _DATA_SHAPES.get(name)
        return a.reshape(-1, w) if w else a

    def set_(self, value):
        self._backend.data_array(name)[...] = np.asarray(value).reshape(-1)

    return property(get, set_)


class MjData:
    """Shim for `mujoco.MjData`: arrays are live views into the backend's state."""

    def __init__(self, model: MjModel):
        self._model = model
        self._backend = model._backend

    @property
    def time(self):
        return float(self._backend.data_array("time")[0])

    @time.setter
    def time(self, v):
        self._backend.data_array("time")[0] = v

    @property
    def ncon(self):
        return self._backend.ncon

    @property
    def contact(self):
        return [_Contact(c) for c in self._backend.contacts()]

    @property
    def nefc(self):
        return int(self._backend.nefc)

    # constraint rows (mjData.efc_*): read by tools/gen_golden_with_mujoco.py only; a backend that does not keep them raises
    @property
    def efc_force(self):
        return self._backend.efc_array("efc_force")

    @property
    def efc_aref(self):
        return self._backend.efc_array("efc_aref")

    @property
    def efc_R(self):
        return self._backend.efc_array("efc_R")

    @property
    def efc_type(self):
        return self._backend.efc_array("efc_type")


for _f in _DATA_FIELDS:
    setattr(MjData, _f, _data_prop(_f))

_LAST_XML = None


def mj_id2name(m, objtype, i):
    return m._flat.names[_OBJ_KIND[int(objtype)]][i]


def mj_name2id(m, objtype, name):
    try:
        return m._flat.names[_OBJ_KIND[int(objtype)]].index(name)
    except ValueError:
        return -1


def mj_saveLastXML(filename, m):
    if isinstance(filename, bytes):
        filename = filename.decode()
    with open(filename, "w") as f:
        f.write(m._flat.xml)
    return 1


def mj_resetData(m, d):
    d._backend.reset()


def mj_forward(m, d):
    m._sync()
    d._backend.forward()


def mj_step(m, d, nstep=1):
    m._sync()
    for _ in range(nstep):
        d._backend.step()


def mj_step1(m, d):
    m._sync()
    d._backend.step1()


def mj_step2(m, d):
    d._backend.step2()


def _jac(kind):
    def f(m, d, jacp, jacr, idx):
        jp, jr = d._backend.jac(kind, int(idx))
        if jacp is not None:
            jacp[...] = jp.reshape(jacp.shape)
        if jacr is not None:
            jacr[...] = jr.reshape(jacr.shape)

    return f


mj_jacSite, mj_jacBody, mj_jacGeom = _jac("site"), _jac("body"), _jac("geom")


def mj_fullM(m, dst, qM):
    dst[...] = m._backend.full_M().reshape(dst.shape)


def install(backend_factory, stub_missing=True):
    """Register this module as `mujoco` (plus no-op stubs for numba/termcolor/cv2 when they are absent)."""
    global _BACKEND_FACTORY
    _BACKEND_FACTORY = backend_factory
    mod = types.ModuleType("mujoco")
    mod.__version__ = "3.3.0"
    mod.__file__ = __file__
    mod.__path__ = []
    for k in ("MjModel", "MjData", "mjtObj", "mjtJoint", "mjtGeom", "mj_id2name", "mj_name2id", "mj_saveLastXML", "mj_resetData", "mj_forward",
              "mj_step", "mj_step1", "mj_step2", "mj_jacSite", "mj_jacBody", "mj_jacGeom", "mj_fullM"):
        setattr(mod, k, globals()[k])
    for k in ("mjtCamera", "mjtFontScale", "mjtFramebuffer", "mjtCatBit", "mjtRndFlag", "mjtTexture", "mjtVisFlag"):
        setattr(mod, k, _Placeholder())
    viewer = types.ModuleType("mujoco.viewer")
    mod.viewer = viewer
    sys.modules["mujoco"] = mod
    sys.modules["mujoco.viewer"] = viewer
    if stub_missing:
        _stub_optional()
    return mod


def _stub_optional():
    import importlib.util

    def missing(name):
        try:
            return importlib.util.find_spec(name) is None
        except (ImportError, ValueError):
            return True

    if missing("numba"):
        nb = types.ModuleType("numba")
        nb.jit = lambda *a, **k: (lambda f: f)
        sys.modules["numba"] = nb
    if missing("termcolor"):
        tc = types.ModuleType("termcolor")
        tc.colored = lambda s, *a, **k: s
        sys.modules["termcolor"] = tc
    if missing("cv2"):
        sys.modules["cv2"] = types.ModuleType("cv2")
