"""MJCF -> flat model compiler (host side, numpy).

This is the model-ingest half of the drop-in boundary: robosuite assembles ONE MJCF string per
reset (reference `robosuite/environments/base.py:262-269`, `robosuite/models/base.py:149-158`) and
hands it to `mujoco.MjModel.from_xml_string` (`robosuite/utils/binding_utils.py:1077-1080`).
This module ingests that same string and produces the flat, array-only model the HIP backend
(and the test oracle) consume.  It is a from-scratch restatement of the MJCF subset robosuite emits
for the BASELINE configs; MuJoCo's compiler itself is third-party and not available here [3P].

Conventions follow the MuJoCo documentation (XML reference, "Computation" chapter):
  * ids are depth-first document order; world body = 0
  * quaternions are (w, x, y, z)
  * `inertiagrouprange`, `autolimits`, `angle`, per-element defaults as documented
"""
from __future__ import annotations

import os
import struct
import xml.etree.ElementTree as ET
from collections import OrderedDict

import numpy as np

# ------------------------------------------------------------------------------------------------
# enums (values follow MuJoCo's mjtJoint / mjtGeom / mjtObj numbering, which robosuite hard-codes
# through `mujoco.mjtJoint.*` in binding_utils.py:523-553)
# ------------------------------------------------------------------------------------------------
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = range(8)
_GEOM_TYPES = {
    "plane": GEOM_PLANE,
    "hfield": GEOM_HFIELD,
    "sphere": GEOM_SPHERE,
    "capsule": GEOM_CAPSULE,
    "ellipsoid": GEOM_ELLIPSOID,
    "cylinder": GEOM_CYLINDER,
    "box": GEOM_BOX,
    "mesh": GEOM_MESH,
}
_JNT_TYPES = {"free": JNT_FREE, "ball": JNT_BALL, "slide": JNT_SLIDE, "hinge": JNT_HINGE}
MINVAL = 1e-15

GAIN_FIXED, BIAS_NONE, BIAS_AFFINE = 0, 0, 1


class MJCFError(ValueError):
    """Raised on malformed / unsupported MJCF (mirrors MuJoCo raising on compile errors)."""


# ------------------------------------------------------------------------------------------------
# small math
# ------------------------------------------------------------------------------------------------
def _floats(s, n=None, default=None):
    if s is None:
        return None if default is None else np.array(default, dtype=np.float64)
    v = np.array([float(x) for x in s.split()], dtype=np.float64)
    if n is not None and len(v) != n:
        if default is not None and len(v) < n:
            out = np.array(default, dtype=np.float64)
            out[: len(v)] = v
            return out
        raise MJCFError(f"expected {n} numbers, got '{s}'")
    return v


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array(
        [
            aw * bw - ax * bx - ay * by - az * bz,
            aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw,
        ]
    )


def quat_normalize(q):
    n = np.linalg.norm(q)
    if n < MINVAL:
        return np.array([1.0, 0.0, 0.0, 0.0])
    return q / n


def quat2mat(q):
    w, x, y, z = q
    return np.array(
        [
            [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
            [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
            [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z],
        ]
    )


def mat2quat(R):
    """Rotation matrix -> unit quaternion (w,x,y,z)."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s])
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = np.array([(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s])
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = np.array([(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s])
    return quat_normalize(q)


def axisangle2quat(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    n = np.linalg.norm(axis)
    if n < MINVAL:
        return np.array([1.0, 0.0, 0.0, 0.0])
    axis = axis / n
    return np.concatenate([[np.cos(angle / 2)], axis * np.sin(angle / 2)])


def quat_z2vec(vec):
    """Quaternion rotating the z axis onto `vec` (MJCF `zaxis` / `fromto`)."""
    vec = np.asarray(vec, dtype=np.float64)
    n = np.linalg.norm(vec)
    if n < MINVAL:
        return np.array([1.0, 0.0, 0.0, 0.0])
    vec = vec / n
    z = np.array([0.0, 0.0, 1.0])
    axis = np.cross(z, vec)
    s = np.linalg.norm(axis)
    if s < 1e-10:
        # parallel or anti-parallel
        if vec[2] > 0:
            return np.array([1.0, 0.0, 0.0, 0.0])
        return np.array([0.0, 1.0, 0.0, 0.0])
    ang = np.arctan2(s, vec[2])
    return axisangle2quat(axis / s, ang)


# ------------------------------------------------------------------------------------------------
# meshes
# ------------------------------------------------------------------------------------------------
def load_stl(path):
    """Read a binary or ASCII STL; returns (nvert x 3 float64 vertices, ntri x 3 int faces)."""
    with open(path, "rb") as f:
        data = f.read()
    tri = None
    if len(data) >= 84:
        ntri = struct.unpack_from("<I", data, 80)[0]
        if 84 + ntri * 50 == len(data):
            rec = np.frombuffer(data, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", 9), ("a", "<u2")]), offset=84, count=ntri)
            tri = rec["v"].reshape(ntri, 3, 3).astype(np.float64)
    if tri is None:
        verts = []
        for line in data.decode("ascii", errors="ignore").splitlines():
            line = line.strip()
            if line.startswith("vertex"):
                verts.append([float(x) for x in line.split()[1:4]])
        tri = np.array(verts, dtype=np.float64).reshape(-1, 3, 3)
    flat = tri.reshape(-1, 3)
    uniq, inv = np.unique(flat, axis=0, return_inverse=True)
    return uniq, inv.reshape(-1, 3)


def load_obj(path):
    verts, faces = [], []
    with open(path, "r") as f:
        for line in f:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                verts.append([float(x) for x in p[1:4]])
            elif p[0] == "f":
                idx = [int(t.split("/")[0]) - 1 for t in p[1:]]
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
    return np.array(verts, dtype=np.float64), np.array(faces, dtype=np.int64).reshape(-1, 3)


def load_msh(path):
    """Legacy MuJoCo .msh (binary: nvertex nnormal ntexcoord nface, then float32/int32 arrays)."""
    with open(path, "rb") as f:
        data = f.read()
    nv, nn, nt, nf = struct.unpack_from("<4i", data, 0)
    off = 16
    v = np.frombuffer(data, dtype="<f4", count=3 * nv, offset=off).reshape(nv, 3).astype(np.float64)
    off += 12 * nv + 12 * nn + 8 * nt
    fc = np.frombuffer(data, dtype="<i4", count=3 * nf, offset=off).reshape(nf, 3).astype(np.int64)
    return v, fc


def load_mesh(path):
    ext = os.path.splitext(path)[1].lower()
    if ext == ".stl":
        return load_stl(path)
    if ext == ".obj":
        return load_obj(path)
    if ext == ".msh":
        return load_msh(path)
    raise MJCFError(f"unsupported mesh format: {path}")


def convex_hull(verts):
    """Convex hull vertices + triangles (scipy/qhull, the library MuJoCo's compiler also uses [3P])."""
    from scipy.spatial import ConvexHull

    hull = ConvexHull(verts)
    idx = np.unique(hull.simplices.ravel())
    remap = -np.ones(len(verts), dtype=np.int64)
    remap[idx] = np.arange(len(idx))
    hv = verts[idx]
    faces = remap[hull.simplices]
    # orient faces outward
    c = hv.mean(axis=0)
    for k, fc in enumerate(faces):
        a, b, d = hv[fc]
        nrm = np.cross(b - a, d - a)
        if np.dot(nrm, a - c) < 0:
            faces[k] = fc[::-1]
    return hv, faces


def mesh_volume_props(verts, faces):
    """Volume, centre of mass and inertia about the COM (unit density) of a closed triangle mesh."""
    vol = 0.0
    com = np.zeros(3)
    # covariance integral  ∫ x x^T dV  via signed tetrahedra with the origin
    C = np.zeros((3, 3))
    canon = np.array([[2.0, 1.0, 1.0], [1.0, 2.0, 1.0], [1.0, 1.0, 2.0]]) / 120.0
    for fc in faces:
        A = verts[fc].T  # columns are vertices
        det = np.linalg.det(A)
        vol += det / 6.0
        com += det / 24.0 * A.sum(axis=1)
        C += det * (A @ canon @ A.T)
    if abs(vol) < MINVAL:
        raise MJCFError("mesh volume is zero")
    com = com / vol
    C = C - vol * np.outer(com, com)
    inertia = np.trace(C) * np.eye(3) - C
    if vol < 0:
        vol, inertia = -vol, -inertia
    return vol, com, inertia


# ------------------------------------------------------------------------------------------------
# primitive mass properties (MuJoCo XML reference, geom/density)
# ------------------------------------------------------------------------------------------------
def _geom_volume_inertia(gtype, size, mesh=None):
    """(volume, unit-density inertia diagonal about the geom centre in the geom frame)."""
    if gtype == GEOM_SPHERE:
        r = size[0]
        v = 4.0 / 3.0 * np.pi * r**3
        return v, np.full(3, 0.4 * v * r * r)
    if gtype == GEOM_BOX:
        a, b, c = size
        v = 8 * a * b * c
        return v, v / 3.0 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    if gtype == GEOM_CYLINDER:
        r, h = size[0], size[1]
        v = np.pi * r * r * 2 * h
        ixx = v * (3 * r * r + 4 * h * h) / 12.0
        return v, np.array([ixx, ixx, v * r * r / 2.0])
    if gtype == GEOM_ELLIPSOID:
        a, b, c = size
        v = 4.0 / 3.0 * np.pi * a * b * c
        return v, v / 5.0 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    if gtype == GEOM_CAPSULE:
        r, h = size[0], size[1]
        vc = np.pi * r * r * 2 * h
        vs = 4.0 / 3.0 * np.pi * r**3
        v = vc + vs
        izz = vc * r * r / 2 + vs * 0.4 * r * r
        ixx = vc * (3 * r * r + 4 * h * h) / 12.0 + vs * (0.4 * r * r + h * h + 0.75 * r * h)
        return v, np.array([ixx, ixx, izz])
    if gtype == GEOM_MESH:
        return mesh["volume"], None
    return 0.0, np.zeros(3)


# ------------------------------------------------------------------------------------------------
# compiler
# ------------------------------------------------------------------------------------------------
class FlatModel:
    """Array-only compiled model.  `arrays` maps field name -> numpy array (int32 or float64);
    `names` maps object kind -> list of names (None for unnamed)."""

    def __init__(self):
        self.arrays: "OrderedDict[str, np.ndarray]" = OrderedDict()
        self.names = {}
        self.xml = None

    def __getattr__(self, key):
        arrays = self.__dict__.get("arrays", {})
        if key in arrays:
            a = arrays[key]
            if a.ndim == 1 and a.shape[0] == 1 and key in _SCALARS:
                return a[0].item()
            return a
        raise AttributeError(key)

    def set(self, key, value, dtype):
        a = np.ascontiguousarray(np.asarray(value, dtype=dtype))
        if a.ndim == 0:
            a = a.reshape(1)
        self.arrays[key] = a

    def name2id(self, kind, name):
        try:
            return self.names[kind].index(name)
        except ValueError:
            raise KeyError(f'no {kind} named "{name}"')

    def copy(self):
        m = FlatModel()
        m.arrays = OrderedDict((k, v.copy()) for k, v in self.arrays.items())
        m.names = {k: list(v) for k, v in self.names.items()}
        m.xml = self.xml
        return m


_SCALARS = {
    "nq", "nv", "nu", "nbody", "njnt", "ngeom", "nsite", "nmesh", "nmeshvert", "npair", "ncam", "nlight",
    "nsensor", "ntendon", "neq", "nmocap", "timestep", "density", "viscosity", "impratio", "cone", "iterations",
    "tolerance", "ncgeom", "solver",
}


def _orientation(elem, compiler):
    """Resolve quat / euler / axisangle / xyaxes / zaxis into a unit quaternion."""
    if elem.get("quat") is not None:
        return quat_normalize(_floats(elem.get("quat"), 4))
    scale = 1.0 if compiler["angle"] == "radian" else np.pi / 180.0
    if elem.get("euler") is not None:
        e = _floats(elem.get("euler"), 3) * scale
        q = np.array([1.0, 0.0, 0.0, 0.0])
        seq = compiler["eulerseq"]
        for k, ch in enumerate(seq):
            ax = np.zeros(3)
            ax["xyz".index(ch.lower())] = 1.0
            qk = axisangle2quat(ax, e[k])
            # lower-case = intrinsic (rotating frame): post-multiply; upper = extrinsic: pre-multiply
            q = quat_mul(q, qk) if ch.islower() else quat_mul(qk, q)
        return quat_normalize(q)
    if elem.get("axisangle") is not None:
        a = _floats(elem.get("axisangle"), 4)
        return axisangle2quat(a[:3], a[3] * scale)
    if elem.get("xyaxes") is not None:
        a = _floats(elem.get("xyaxes"), 6)
        x = a[:3] / np.linalg.norm(a[:3])
        y = a[3:] - np.dot(a[3:], x) * x
        y = y / np.linalg.norm(y)
        z = np.cross(x, y)
        return mat2quat(np.stack([x, y, z], axis=1))
    if elem.get("zaxis") is not None:
        return quat_z2vec(_floats(elem.get("zaxis"), 3))
    return np.array([1.0, 0.0, 0.0, 0.0])


class _Defaults:
    """Minimal <default> class support (robosuite inlines defaults before emitting XML,
    `robosuite/models/base.py:209-255`, but stock MJCF assets may still carry them)."""

    def __init__(self, root):
        self.classes = {"main": {}}
        d = root.find("default")
        if d is not None:
            self._walk(d, "main", {})

    def _walk(self, node, name, inherited):
        cur = {k: dict(v) for k, v in inherited.items()}
        for ch in node:
            if ch.tag == "default":
                continue
            cur.setdefault(ch.tag, {}).update(ch.attrib)
        self.classes[name] = cur
        for ch in node:
            if ch.tag == "default":
                self._walk(ch, ch.get("class"), cur)

    def apply(self, elem, tag, childclass):
        cls = elem.get("class") or childclass or "main"
        base = self.classes.get(cls, {}).get(tag, {})
        if not base:
            return elem
        merged = dict(base)
        merged.update(elem.attrib)
        e2 = ET.Element(elem.tag, merged)
        e2.extend(list(elem))
        return e2


def compile_mjcf(xml: str, asset_dir: str | None = None, max_hull_vert: int = 0) -> FlatModel:
    """Compile an MJCF string into a :class:`FlatModel`.

    Replaces `mujoco.MjModel.from_xml_string` (reference call sites
    `robosuite/utils/binding_utils.py:1079`, `robosuite/models/base.py:145`, `robosuite/robots/robot.py:217-223`).
    Raises :class:`MJCFError` on malformed input (MuJoCo raises on compile errors [3P]).
    """
    try:
        root = ET.fromstring(xml)
    except ET.ParseError as e:
        raise MJCFError(f"XML parse error: {e}")
    if root.tag != "mujoco":
        raise MJCFError("root element must be <mujoco>")

    comp = root.find("compiler")
    compiler = {
        "angle": "degree",
        "eulerseq": "xyz",
        "autolimits": True,
        "inertiagrouprange": (0, 5),
        "meshdir": "",
        "inertiafromgeom": "auto",
        "boundmass": 0.0,
        "boundinertia": 0.0,
    }
    if comp is not None:
        compiler["angle"] = comp.get("angle", "degree")
        compiler["eulerseq"] = comp.get("eulerseq", "xyz")
        compiler["autolimits"] = comp.get("autolimits", "true") == "true"
        if comp.get("inertiagrouprange"):
            lo, hi = comp.get("inertiagrouprange").split()
            compiler["inertiagrouprange"] = (int(lo), int(hi))
        compiler["meshdir"] = comp.get("meshdir", "")
        compiler["inertiafromgeom"] = comp.get("inertiafromgeom", "auto")
        compiler["boundmass"] = float(comp.get("boundmass", 0))
        compiler["boundinertia"] = float(comp.get("boundinertia", 0))
    ang_scale = 1.0 if compiler["angle"] == "radian" else np.pi / 180.0

    defaults = _Defaults(root)

    opt = root.find("option")
    o = opt.attrib if opt is not None else {}
    timestep = float(o.get("timestep", 0.002))
    gravity = _floats(o.get("gravity"), 3, [0, 0, -9.81])
    wind = _floats(o.get("wind"), 3, [0, 0, 0])
    density = float(o.get("density", 0))
    viscosity = float(o.get("viscosity", 0))
    impratio = float(o.get("impratio", 1))
    cone = {"pyramidal": 0, "elliptic": 1}[o.get("cone", "pyramidal")]
    iterations = int(o.get("iterations", 100))
    tolerance = float(o.get("tolerance", 1e-8))
    solver = {"PGS": 0, "CG": 1, "Newton": 1}[o.get("solver", "Newton")]  # CG solves the same primal problem
    integrator = o.get("integrator", "Euler")
    if integrator != "Euler":
        raise MJCFError(f"integrator '{integrator}' not supported (robosuite uses the default Euler)")

    # ---- assets: meshes ------------------------------------------------------------------------
    meshes = OrderedDict()
    asset = root.find("asset")
    if asset is not None:
        for me in asset.findall("mesh"):
            me = defaults.apply(me, "mesh", None)
            name = me.get("name")
            f = me.get("file")
            if name is None and f is not None:
                name = os.path.splitext(os.path.basename(f))[0]
            meshes[name] = {"file": f, "scale": _floats(me.get("scale"), 3, [1, 1, 1]), "loaded": False}

    def _load_mesh(name):
        m = meshes[name]
        if m["loaded"]:
            return m
        path = m["file"]
        if not os.path.isabs(path):
            path = os.path.join(asset_dir or "", compiler["meshdir"], path)
        if not os.path.exists(path):
            raise MJCFError(f"mesh file not found: {path}")
        v, fc = load_mesh(path)
        v = v * m["scale"]
        hv, hf = convex_hull(v)
        if max_hull_vert and len(hv) > max_hull_vert:
            raise MJCFError("hull decimation not implemented")
        vol, com, inert = mesh_volume_props(hv, hf)
        m.update(loaded=True, hull_vert=hv, hull_face=hf, volume=vol, com=com, inertia=inert)
        return m

    # ---- kinematic tree ------------------------------------------------------------------------
    bodies, joints, geoms, sites, cams, lights = [], [], [], [], [], []
    bodies.append(
        dict(name="world", parent=0, pos=np.zeros(3), quat=np.array([1.0, 0, 0, 0]), mocap=False, inertial=None, jnts=[], geoms=[])
    )

    def parse_geom(ge, bid, childclass):
        ge = defaults.apply(ge, "geom", childclass)
        gtype = _GEOM_TYPES[ge.get("type", "sphere")]
        if gtype == GEOM_HFIELD:
            raise MJCFError("hfield geoms not supported")
        g = dict(name=ge.get("name"), body=bid, type=gtype)
        size = _floats(ge.get("size"))
        pos = _floats(ge.get("pos"), 3, [0, 0, 0])
        quat = _orientation(ge, compiler)
        if ge.get("fromto") is not None:
            ft = _floats(ge.get("fromto"), 6)
            a, b = ft[:3], ft[3:]
            pos = 0.5 * (a + b)
            quat = quat_z2vec(b - a)
            half = 0.5 * np.linalg.norm(b - a)
            if gtype in (GEOM_CAPSULE, GEOM_CYLINDER):
                size = np.array([size[0], half])
            elif gtype in (GEOM_BOX, GEOM_ELLIPSOID):
                size = np.array([size[0], size[1] if len(size) > 1 else size[0], half])
        s3 = np.zeros(3)
        if size is not None:
            s3[: min(3, len(size))] = size[:3]
        if gtype == GEOM_MESH:
            mname = ge.get("mesh")
            if mname not in meshes:
                raise MJCFError(f"geom references unknown mesh '{mname}'")
            g["mesh"] = mname
        else:
            g["mesh"] = None
        g.update(
            size=s3,
            pos=pos,
            quat=quat,
            contype=int(ge.get("contype", 1)),
            conaffinity=int(ge.get("conaffinity", 1)),
            condim=int(ge.get("condim", 3)),
            group=int(ge.get("group", 0)),
            priority=int(ge.get("priority", 0)),
            friction=_floats(ge.get("friction"), 3, [1, 0.005, 0.0001]),
            solref=_floats(ge.get("solref"), 2, [0.02, 1.0]),
            solimp=_floats(ge.get("solimp"), 5, [0.9, 0.95, 0.001, 0.5, 2.0]),
            solmix=float(ge.get("solmix", 1)),
            margin=float(ge.get("margin", 0)),
            gap=float(ge.get("gap", 0)),
            density=float(ge.get("density", 1000)),
            mass=None if ge.get("mass") is None else float(ge.get("mass")),
            rgba=_floats(ge.get("rgba"), 4, [0.5, 0.5, 0.5, 1]),
        )
        geoms.append(g)
        bodies[bid]["geoms"].append(len(geoms) - 1)

    def parse_site(se, bid, childclass):
        se = defaults.apply(se, "site", childclass)
        size = _floats(se.get("size"), None, None)
        s3 = np.full(3, 0.005)
        if size is not None:
            s3[: min(3, len(size))] = size[:3]
        sites.append(
            dict(
                name=se.get("name"),
                body=bid,
                pos=_floats(se.get("pos"), 3, [0, 0, 0]),
                quat=_orientation(se, compiler),
                size=s3,
                rgba=_floats(se.get("rgba"), 4, [0.5, 0.5, 0.5, 1]),
            )
        )

    def parse_joint(je, bid, childclass):
        if je.tag == "freejoint":
            jt = JNT_FREE
            attrib = dict(je.attrib)
        else:
            je = defaults.apply(je, "joint", childclass)
            jt = _JNT_TYPES[je.get("type", "hinge")]
            attrib = je.attrib
        rng = _floats(attrib.get("range"), 2, [0, 0])
        limited_attr = attrib.get("limited", "auto")
        if limited_attr == "auto":
            limited = compiler["autolimits"] and attrib.get("range") is not None
            if not compiler["autolimits"] and attrib.get("range") is not None:
                raise MJCFError("range specified without limited and autolimits=false")
        else:
            limited = limited_attr == "true"
        if jt in (JNT_HINGE, JNT_BALL):
            rng = rng * ang_scale
        axis = _floats(attrib.get("axis"), 3, [0, 0, 1])
        n = np.linalg.norm(axis)
        axis = axis / n if n > MINVAL else np.array([0.0, 0, 1])
        ref = float(attrib.get("ref", 0)) * (ang_scale if jt == JNT_HINGE else 1.0)
        sref = float(attrib.get("springref", 0)) * (ang_scale if jt == JNT_HINGE else 1.0)
        joints.append(
            dict(
                name=attrib.get("name"),
                body=bid,
                type=jt,
                pos=_floats(attrib.get("pos"), 3, [0, 0, 0]),
                axis=axis,
                limited=limited and jt in (JNT_HINGE, JNT_SLIDE, JNT_BALL),
                range=rng,
                damping=float(attrib.get("damping", 0)),
                armature=float(attrib.get("armature", 0)),
                frictionloss=float(attrib.get("frictionloss", 0)),
                stiffness=float(attrib.get("stiffness", 0)),
                margin=float(attrib.get("margin", 0)),
                ref=ref,
                springref=sref,
                solreflimit=_floats(attrib.get("solreflimit"), 2, [0.02, 1.0]),
                solimplimit=_floats(attrib.get("solimplimit"), 5, [0.9, 0.95, 0.001, 0.5, 2.0]),
                solreffriction=_floats(attrib.get("solreffriction"), 2, [0.02, 1.0]),
                solimpfriction=_floats(attrib.get("solimpfriction"), 5, [0.9, 0.95, 0.001, 0.5, 2.0]),
            )
        )
        bodies[bid]["jnts"].append(len(joints) - 1)

    def walk(belem, bid, childclass):
        cc = belem.get("childclass") or childclass
        for ch in belem:
            if ch.tag in ("joint", "freejoint"):
                parse_joint(ch, bid, cc)
            elif ch.tag == "geom":
                parse_geom(ch, bid, cc)
            elif ch.tag == "site":
                parse_site(ch, bid, cc)
            elif ch.tag == "camera":
                cams.append(dict(name=ch.get("name"), body=bid))
            elif ch.tag == "light":
                lights.append(dict(name=ch.get("name"), body=bid))
            elif ch.tag == "inertial":
                bodies[bid]["inertial"] = ch
        for ch in belem:
            if ch.tag == "body":
                nb = len(bodies)
                bodies.append(
                    dict(
                        name=ch.get("name"),
                        parent=bid,
                        pos=_floats(ch.get("pos"), 3, [0, 0, 0]),
                        quat=_orientation(ch, compiler),
                        mocap=ch.get("mocap", "false") == "true",
                        inertial=None,
                        jnts=[],
                        geoms=[],
                    )
                )
                walk(ch, nb, cc)

    wb = root.find("worldbody")
    if wb is not None:
        walk(wb, 0, None)

    nbody, njnt, ngeom, nsite = len(bodies), len(joints), len(geoms), len(sites)

    # MuJoCo orders geoms/sites/joints by owning body id (document order within a body)
    def _reorder(items, key="body"):
        order = sorted(range(len(items)), key=lambda i: (items[i][key], i))
        return [items[i] for i in order]

    geoms = _reorder(geoms)
    sites = _reorder(sites)
    joints = _reorder(joints)
    cams = _reorder(cams)
    lights = _reorder(lights)
    for b in bodies:
        b["jnts"], b["geoms"] = [], []
    for i, j in enumerate(joints):
        bodies[j["body"]]["jnts"].append(i)
    for i, g in enumerate(geoms):
        bodies[g["body"]]["geoms"].append(i)

    # ---- joint / dof addressing ----------------------------------------------------------------
    nq = nv = 0
    for j in joints:
        j["qposadr"], j["dofadr"] = nq, nv
        nq += {JNT_FREE: 7, JNT_BALL: 4, JNT_SLIDE: 1, JNT_HINGE: 1}[j["type"]]
        nv += {JNT_FREE: 6, JNT_BALL: 3, JNT_SLIDE: 1, JNT_HINGE: 1}[j["type"]]
    for b in bodies[1:]:
        for ji in b["jnts"]:
            if joints[ji]["type"] == JNT_FREE and (b["parent"] != 0 or len(b["jnts"]) != 1):
                raise MJCFError("free joint must be the only joint of a top-level body")

    m = FlatModel()
    m.xml = xml
    I32, F64 = np.int32, np.float64

    # ---- mesh geoms: hull vertices in the geom frame -------------------------------------------
    mesh_names = list(meshes.keys())
    used_mesh = OrderedDict()
    mesh_vert = []
    mesh_vertadr, mesh_vertnum = [], []
    for g in geoms:
        if g["type"] == GEOM_MESH:
            if g["contype"] == 0 and g["conaffinity"] == 0 and not (
                compiler["inertiagrouprange"][0] <= g["group"] <= compiler["inertiagrouprange"][1] and bodies[g["body"]]["inertial"] is None
            ):
                g["dataid"] = -1  # visual-only mesh: never loaded (robosuite's checkout lacks some of them)
                continue
            if g["mesh"] not in used_mesh:
                md = _load_mesh(g["mesh"])
                used_mesh[g["mesh"]] = len(used_mesh)
                mesh_vertadr.append(sum(mesh_vertnum))
                mesh_vertnum.append(len(md["hull_vert"]))
                mesh_vert.append(md["hull_vert"])
            g["dataid"] = used_mesh[g["mesh"]]
        else:
            g["dataid"] = -1

    # ---- body inertial properties --------------------------------------------------------------
    body_mass = np.zeros(nbody)
    body_inertia = np.zeros((nbody, 3))
    body_ipos = np.zeros((nbody, 3))
    body_iquat = np.tile(np.array([1.0, 0, 0, 0]), (nbody, 1))
    glo, ghi = compiler["inertiagrouprange"]

    def geom_mass_inertia(g):
        """mass, COM offset (geom frame), inertia matrix about COM (geom frame)"""
        if g["type"] == GEOM_MESH:
            md = _load_mesh(g["mesh"])
            vol, com, inert = md["volume"], md["com"], md["inertia"]
            mass = g["mass"] if g["mass"] is not None else g["density"] * vol
            return mass, com, inert * (mass / vol)
        vol, idiag = _geom_volume_inertia(g["type"], g["size"])
        if vol <= 0:
            return 0.0, np.zeros(3), np.zeros((3, 3))
        mass = g["mass"] if g["mass"] is not None else g["density"] * vol
        return mass, np.zeros(3), np.diag(idiag) * (mass / vol)

    for bi, b in enumerate(bodies):
        ie = b["inertial"]
        use_geoms = compiler["inertiafromgeom"] == "true" or (compiler["inertiafromgeom"] == "auto" and ie is None)
        if ie is not None and not use_geoms:
            body_mass[bi] = float(ie.get("mass", 0))
            body_ipos[bi] = _floats(ie.get("pos"), 3, [0, 0, 0])
            iq = _orientation(ie, compiler)
            if ie.get("fullinertia") is not None:
                f = _floats(ie.get("fullinertia"), 6)
                Im = np.array([[f[0], f[3], f[4]], [f[3], f[1], f[5]], [f[4], f[5], f[2]]])
                w, V = np.linalg.eigh(Im)
                order = np.argsort(-w)
                w, V = w[order], V[:, order]
                if np.linalg.det(V) < 0:
                    V[:, 2] = -V[:, 2]
                body_inertia[bi] = w
                iq = quat_mul(iq, mat2quat(V))
            else:
                body_inertia[bi] = _floats(ie.get("diaginertia"), 3, [0, 0, 0])
            body_iquat[bi] = quat_normalize(iq)
        elif use_geoms and bi > 0:
            sel = [gi for gi in b["geoms"] if glo <= geoms[gi]["group"] <= ghi]
            parts = []
            for gi in sel:
                g = geoms[gi]
                mass, coff, Ig = geom_mass_inertia(g)
                if mass <= 0:
                    continue
                R = quat2mat(g["quat"])
                parts.append((mass, g["pos"] + R @ coff, R @ Ig @ R.T, g, np.allclose(coff, 0)))
            if len(parts) == 1 and parts[0][4]:
                mass, c, Iw, g, _ = parts[0]
                body_mass[bi] = mass
                body_ipos[bi] = c
                body_iquat[bi] = g["quat"]
                Rg = quat2mat(g["quat"])
                body_inertia[bi] = np.diag(Rg.T @ Iw @ Rg)
            elif parts:
                mt = sum(p[0] for p in parts)
                c = sum(p[0] * p[1] for p in parts) / mt
                It = np.zeros((3, 3))
                for mass, ci, Iw, _, _ in parts:
                    d = ci - c
                    It += Iw + mass * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
                w, V = np.linalg.eigh(It)
                order = np.argsort(-w)
                w, V = w[order], V[:, order]
                if np.linalg.det(V) < 0:
                    V[:, 2] = -V[:, 2]
                body_mass[bi], body_ipos[bi], body_inertia[bi] = mt, c, w
                body_iquat[bi] = mat2quat(V)
        if compiler["boundmass"] > 0 and bi > 0:
            body_mass[bi] = max(body_mass[bi], compiler["boundmass"])
        if compiler["boundinertia"] > 0 and bi > 0:
            body_inertia[bi] = np.maximum(body_inertia[bi], compiler["boundinertia"])

    # ---- tree bookkeeping ----------------------------------------------------------------------
    body_parentid = np.array([b["parent"] for b in bodies], dtype=I32)
    body_rootid = np.zeros(nbody, dtype=I32)
    body_weldid = np.zeros(nbody, dtype=I32)
    body_mocapid = -np.ones(nbody, dtype=I32)
    nmocap = 0
    for bi in range(1, nbody):
        p = body_parentid[bi]
        body_rootid[bi] = bi if p == 0 else body_rootid[p]
        body_weldid[bi] = bi if bodies[bi]["jnts"] else body_weldid[p]
        if bodies[bi]["mocap"]:
            if p != 0 or bodies[bi]["jnts"]:
                raise MJCFError("mocap body must be a jointless child of the world")
            body_mocapid[bi] = nmocap
            nmocap += 1
    body_jntadr = np.array([b["jnts"][0] if b["jnts"] else -1 for b in bodies], dtype=I32)
    body_jntnum = np.array([len(b["jnts"]) for b in bodies], dtype=I32)
    body_dofadr = -np.ones(nbody, dtype=I32)
    body_dofnum = np.zeros(nbody, dtype=I32)
    dof_bodyid = np.zeros(nv, dtype=I32)
    dof_jntid = np.zeros(nv, dtype=I32)
    for ji, j in enumerate(joints):
        nd = {JNT_FREE: 6, JNT_BALL: 3, JNT_SLIDE: 1, JNT_HINGE: 1}[j["type"]]
        if body_dofadr[j["body"]] < 0:
            body_dofadr[j["body"]] = j["dofadr"]
        body_dofnum[j["body"]] += nd
        dof_bodyid[j["dofadr"] : j["dofadr"] + nd] = j["body"]
        dof_jntid[j["dofadr"] : j["dofadr"] + nd] = ji
    # dof_parentid: previous dof in the same body, else last dof of the nearest ancestor with dofs
    dof_parentid = -np.ones(nv, dtype=I32)
    for d in range(nv):
        b = dof_bodyid[d]
        if d > body_dofadr[b]:
            dof_parentid[d] = d - 1
        else:
            p = body_parentid[b]
            while p > 0 and body_dofnum[p] == 0:
                p = body_parentid[p]
            if p > 0:
                dof_parentid[d] = body_dofadr[p] + body_dofnum[p] - 1
    body_geomadr = np.array([b["geoms"][0] if b["geoms"] else -1 for b in bodies], dtype=I32)
    body_geomnum = np.array([len(b["geoms"]) for b in bodies], dtype=I32)

    # ---- scalars -------------------------------------------------------------------------------
    m.set("nq", nq, I32), m.set("nv", nv, I32), m.set("nbody", nbody, I32), m.set("njnt", njnt, I32)
    m.set("ngeom", ngeom, I32), m.set("nsite", nsite, I32), m.set("ncam", len(cams), I32), m.set("nlight", len(lights), I32)
    m.set("nmocap", nmocap, I32)
    m.set("timestep", timestep, F64), m.set("gravity", gravity, F64), m.set("wind", wind, F64)
    m.set("density", density, F64), m.set("viscosity", viscosity, F64), m.set("impratio", impratio, F64)
    m.set("cone", cone, I32), m.set("iterations", iterations, I32), m.set("tolerance", tolerance, F64)
    m.set("solver", solver, I32)

    # ---- bodies --------------------------------------------------------------------------------
    m.set("body_parentid", body_parentid, I32), m.set("body_rootid", body_rootid, I32)
    m.set("body_weldid", body_weldid, I32), m.set("body_mocapid", body_mocapid, I32)
    m.set("body_jntadr", body_jntadr, I32), m.set("body_jntnum", body_jntnum, I32)
    m.set("body_dofadr", body_dofadr, I32), m.set("body_dofnum", body_dofnum, I32)
    m.set("body_geomadr", body_geomadr, I32), m.set("body_geomnum", body_geomnum, I32)
    m.set("body_pos", np.array([b["pos"] for b in bodies]), F64)
    m.set("body_quat", np.array([b["quat"] for b in bodies]), F64)
    m.set("body_ipos", body_ipos, F64), m.set("body_iquat", body_iquat, F64)
    m.set("body_mass", body_mass, F64), m.set("body_inertia", body_inertia, F64)

    # ---- joints / dofs -------------------------------------------------------------------------
    def jarr(key, dtype=F64, shape=None):
        a = np.array([j[key] for j in joints], dtype=dtype)
        if shape is not None:
            a = a.reshape((njnt,) + shape)
        return a

    m.set("jnt_type", jarr("type", I32), I32), m.set("jnt_qposadr", jarr("qposadr", I32), I32)
    m.set("jnt_dofadr", jarr("dofadr", I32), I32), m.set("jnt_bodyid", jarr("body", I32), I32)
    m.set("jnt_pos", jarr("pos", F64, (3,)), F64), m.set("jnt_axis", jarr("axis", F64, (3,)), F64)
    m.set("jnt_limited", jarr("limited", I32), I32), m.set("jnt_range", jarr("range", F64, (2,)), F64)
    m.set("jnt_stiffness", jarr("stiffness"), F64), m.set("jnt_margin", jarr("margin"), F64)
    m.set("jnt_solref", jarr("solreflimit", F64, (2,)), F64), m.set("jnt_solimp", jarr("solimplimit", F64, (5,)), F64)
    qpos0 = np.zeros(nq)
    qpos_spring = np.zeros(nq)
    for j in joints:
        a = j["qposadr"]
        if j["type"] == JNT_FREE:
            qpos0[a : a + 3] = bodies[j["body"]]["pos"]
            qpos0[a + 3 : a + 7] = bodies[j["body"]]["quat"]
            qpos_spring[a : a + 7] = qpos0[a : a + 7]
        elif j["type"] == JNT_BALL:
            qpos0[a : a + 4] = [1, 0, 0, 0]
            qpos_spring[a : a + 4] = [1, 0, 0, 0]
        else:
            qpos0[a] = j["ref"]
            qpos_spring[a] = j["springref"]
    m.set("qpos0", qpos0, F64), m.set("qpos_spring", qpos_spring, F64)
    m.set("dof_bodyid", dof_bodyid, I32), m.set("dof_jntid", dof_jntid, I32), m.set("dof_parentid", dof_parentid, I32)

    def darr(key, width=None):
        out = []
        for d in range(nv):
            out.append(joints[dof_jntid[d]][key])
        a = np.array(out, dtype=F64)
        return a.reshape(nv, width) if width else a.reshape(nv)

    m.set("dof_armature", darr("armature"), F64), m.set("dof_damping", darr("damping"), F64)
    m.set("dof_frictionloss", darr("frictionloss"), F64)
    m.set("dof_solref", darr("solreffriction", 2), F64), m.set("dof_solimp", darr("solimpfriction", 5), F64)

    # ---- geoms ---------------------------------------------------------------------------------
    def garr(key, dtype=F64, shape=None):
        a = np.array([g[key] for g in geoms], dtype=dtype)
        if shape is not None:
            a = a.reshape((ngeom,) + shape)
        return a

    m.set("geom_type", garr("type", I32), I32), m.set("geom_bodyid", garr("body", I32), I32)
    m.set("geom_contype", garr("contype", I32), I32), m.set("geom_conaffinity", garr("conaffinity", I32), I32)
    m.set("geom_condim", garr("condim", I32), I32), m.set("geom_priority", garr("priority", I32), I32)
    m.set("geom_group", garr("group", I32), I32), m.set("geom_dataid", garr("dataid", I32), I32)
    m.set("geom_size", garr("size", F64, (3,)), F64), m.set("geom_pos", garr("pos", F64, (3,)), F64)
    m.set("geom_quat", garr("quat", F64, (4,)), F64), m.set("geom_friction", garr("friction", F64, (3,)), F64)
    m.set("geom_solref", garr("solref", F64, (2,)), F64), m.set("geom_solimp", garr("solimp", F64, (5,)), F64)
    m.set("geom_solmix", garr("solmix"), F64), m.set("geom_margin", garr("margin"), F64), m.set("geom_gap", garr("gap"), F64)
    m.set("geom_rgba", garr("rgba", F64, (4,)), F64)

    # meshes (convex hull vertices, geom-local frame)
    nmesh = len(used_mesh)
    mv = np.concatenate(mesh_vert, axis=0) if mesh_vert else np.zeros((0, 3))
    m.set("nmesh", nmesh, I32), m.set("nmeshvert", len(mv), I32)
    m.set("mesh_vertadr", np.array(mesh_vertadr, dtype=I32), I32)
    m.set("mesh_vertnum", np.array(mesh_vertnum, dtype=I32), I32)
    m.set("mesh_vert", mv, F64)
    # bounding sphere (about geom_rbound_center, geom-local) used by the broadphase cull
    rbound = np.zeros(ngeom)
    rcenter = np.zeros((ngeom, 3))
    for gi, g in enumerate(geoms):
        t, s = g["type"], g["size"]
        if t == GEOM_SPHERE:
            rbound[gi] = s[0]
        elif t == GEOM_CAPSULE:
            rbound[gi] = s[0] + s[1]
        elif t == GEOM_CYLINDER:
            rbound[gi] = np.sqrt(s[0] ** 2 + s[1] ** 2)
        elif t == GEOM_ELLIPSOID:
            rbound[gi] = max(s)
        elif t == GEOM_BOX:
            rbound[gi] = np.linalg.norm(s)
        elif t == GEOM_MESH and g["dataid"] >= 0:
            hv = mesh_vert[g["dataid"]]
            c = 0.5 * (hv.min(axis=0) + hv.max(axis=0))
            rcenter[gi] = c
            rbound[gi] = np.linalg.norm(hv - c, axis=1).max()
        else:
            rbound[gi] = 0.0  # plane: infinite, handled by type
    m.set("geom_rbound", rbound, F64), m.set("geom_rcenter", rcenter, F64)

    # ---- sites ---------------------------------------------------------------------------------
    m.set("site_bodyid", np.array([s["body"] for s in sites], dtype=I32), I32)
    m.set("site_pos", np.array([s["pos"] for s in sites]).reshape(nsite, 3), F64)
    m.set("site_quat", np.array([s["quat"] for s in sites]).reshape(nsite, 4), F64)
    m.set("site_size", np.array([s["size"] for s in sites]).reshape(nsite, 3), F64)
    m.set("site_rgba", np.array([s["rgba"] for s in sites]).reshape(nsite, 4), F64)

    # ---- actuators -----------------------------------------------------------------------------
    acts = []
    act = root.find("actuator")
    jname2id = {j["name"]: i for i, j in enumerate(joints) if j["name"] is not None}
    if act is not None:
        for ae in act:
            ae = defaults.apply(ae, ae.tag, None)
            if ae.tag not in ("motor", "position", "velocity", "general"):
                raise MJCFError(f"actuator type '{ae.tag}' not supported")
            jn = ae.get("joint")
            if jn is None or jn not in jname2id:
                raise MJCFError(f"actuator '{ae.get('name')}' needs a valid joint transmission")
            if joints[jname2id[jn]]["type"] not in (JNT_HINGE, JNT_SLIDE):
                raise MJCFError("only hinge/slide joint transmissions supported")
            gear = _floats(ae.get("gear"), None, None)
            gear = 1.0 if gear is None else gear[0]
            gainprm = np.zeros(3)
            biasprm = np.zeros(3)
            biastype = BIAS_NONE
            if ae.tag == "motor":
                gainprm[0] = 1.0
            elif ae.tag == "position":
                kp = float(ae.get("kp", 1))
                kv = float(ae.get("kv", 0))
                gainprm[0] = kp
                biasprm[:] = [0, -kp, -kv]
                biastype = BIAS_AFFINE
            elif ae.tag == "velocity":
                kv = float(ae.get("kv", 1))
                gainprm[0] = kv
                biasprm[:] = [0, 0, -kv]
                biastype = BIAS_AFFINE
            else:
                gp = _floats(ae.get("gainprm"), None, None)
                bp = _floats(ae.get("biasprm"), None, None)
                gainprm[0] = 1.0
                if gp is not None:
                    gainprm[: min(3, len(gp))] = gp[:3]
                if bp is not None:
                    biasprm[: min(3, len(bp))] = bp[:3]
                if ae.get("gaintype", "fixed") != "fixed" or ae.get("dyntype", "none") != "none":
                    raise MJCFError("only fixed-gain, stateless general actuators supported")
                biastype = {"none": BIAS_NONE, "affine": BIAS_AFFINE}[ae.get("biastype", "none")]

            def _lim(flag, rng_attr):
                v = ae.get(flag, "auto")
                if v == "auto":
                    return compiler["autolimits"] and ae.get(rng_attr) is not None
                return v == "true"

            acts.append(
                dict(
                    name=ae.get("name"),
                    jnt=jname2id[jn],
                    gear=gear,
                    gainprm=gainprm,
                    biasprm=biasprm,
                    biastype=biastype,
                    ctrllimited=_lim("ctrllimited", "ctrlrange"),
                    ctrlrange=_floats(ae.get("ctrlrange"), 2, [0, 0]),
                    forcelimited=_lim("forcelimited", "forcerange"),
                    forcerange=_floats(ae.get("forcerange"), 2, [0, 0]),
                )
            )
    nu = len(acts)
    m.set("nu", nu, I32)
    m.set("actuator_trnid", np.array([a["jnt"] for a in acts], dtype=I32), I32)
    m.set("actuator_gear", np.array([a["gear"] for a in acts]), F64)
    m.set("actuator_gainprm", np.array([a["gainprm"] for a in acts]).reshape(nu, 3), F64)
    m.set("actuator_biasprm", np.array([a["biasprm"] for a in acts]).reshape(nu, 3), F64)
    m.set("actuator_biastype", np.array([a["biastype"] for a in acts], dtype=I32), I32)
    m.set("actuator_ctrllimited", np.array([a["ctrllimited"] for a in acts], dtype=I32), I32)
    m.set("actuator_ctrlrange", np.array([a["ctrlrange"] for a in acts]).reshape(nu, 2), F64)
    m.set("actuator_forcelimited", np.array([a["forcelimited"] for a in acts], dtype=I32), I32)
    m.set("actuator_forcerange", np.array([a["forcerange"] for a in acts]).reshape(nu, 2), F64)

    # ---- sensors / tendons / equality: names + dims only (force/torque sensors are §8(f)) ------
    sens = []
    se = root.find("sensor")
    if se is not None:
        for s in se:
            sens.append(dict(name=s.get("name"), type=s.tag, site=s.get("site")))
    _sdim = {"force": 3, "torque": 3, "touch": 1, "framepos": 3, "framequat": 4, "jointpos": 1, "jointvel": 1}
    m.set("nsensor", len(sens), I32)
    m.set("sensor_dim", np.array([_sdim.get(s["type"], 1) for s in sens], dtype=I32), I32)
    sname2id = {s["name"]: i for i, s in enumerate(sites)}
    m.set("sensor_objid", np.array([sname2id.get(s["site"], -1) for s in sens], dtype=I32), I32)
    m.set("sensor_type", np.array([{"force": 0, "torque": 1}.get(s["type"], -1) for s in sens], dtype=I32), I32)
    # ---- fixed tendons, tendon equality constraints (Robotiq grippers: models/assets/grippers/robotiq_gripper_140.xml:15-44) -------------
    # MuJoCo semantics [3P, docs "XML reference: tendon/fixed, equality/tendon"]: length = sum_i coef_i q_i; optional limit rows on the length;
    # an equality/tendon with one tendon constrains (length - length0) to polycoef[0] (default 0).
    tend = root.find("tendon")
    tendons = []
    jname2id = {j["name"]: i for i, j in enumerate(joints)}
    if tend is not None:
        for t in tend:
            if t.tag != "fixed":
                raise MJCFError("spatial tendons are not supported (only <tendon><fixed>)")
            sl = _floats(t.get("springlength"), None, [-1.0])     # one value = no deadband; -1 = the length at qpos0 (set in _set_const)
            sl = [sl[0], sl[0]] if len(sl) == 1 else list(sl[:2])
            wraps = [(jname2id[w.get("joint")], float(w.get("coef", "1"))) for w in t.findall("joint")]
            for jid, _ in wraps:
                if joints[jid]["type"] not in (JNT_HINGE, JNT_SLIDE):
                    raise MJCFError("fixed tendons over ball / free joints are not supported")
            rng = _floats(t.get("range"), 2, [0.0, 0.0])
            limited = t.get("limited")
            tendons.append(dict(name=t.get("name"), wraps=wraps, range=rng, stiffness=float(t.get("stiffness", "0")), damping=float(t.get("damping", "0")), lengthspring=sl, limited=1 if (limited == "true" or (limited in (None, "auto") and t.get("range") is not None and compiler["autolimits"])) else 0,
                                margin=float(t.get("margin", "0")), solref=_floats(t.get("solreflimit"), 2, [0.02, 1.0]),
                                solimp=_floats(t.get("solimplimit"), 5, [0.9, 0.95, 0.001, 0.5, 2.0]),
                                # dry friction along the tendon (Jaco fingers, jaco_three_finger_gripper.xml:17-29): one friction-loss row per tendon
                                frictionloss=float(t.get("frictionloss", "0")), solref_fri=_floats(t.get("solreffriction"), 2, [0.02, 1.0]),
                                solimp_fri=_floats(t.get("solimpfriction"), 5, [0.9, 0.95, 0.001, 0.5, 2.0])))
    ntendon = len(tendons)
    eqs = []
    eq = root.find("equality")
    tname2id = {t["name"]: i for i, t in enumerate(tendons)}
    if eq is not None:
        for e in eq:
            if e.tag != "tendon" or e.get("tendon2") is not None:
                raise MJCFError(f"equality/{e.tag} is not supported (only equality/tendon with one tendon)")
            if e.get("active", "true") != "true":
                continue
            eqs.append(dict(name=e.get("name"), tendon=tname2id[e.get("tendon1")], polycoef=_floats(e.get("polycoef"), 5, [0.0, 1.0, 0.0, 0.0, 0.0]),
                            solref=_floats(e.get("solref"), 2, [0.02, 1.0]), solimp=_floats(e.get("solimp"), 5, [0.9, 0.95, 0.001, 0.5, 2.0])))
    neq = len(eqs)
    m.set("ntendon", ntendon, I32)
    m.set("neq", neq, I32)
    wrap_adr, wrap_jnt, wrap_coef = [], [], []
    for t in tendons:
        wrap_adr.append(len(wrap_jnt))
        for jid, coef in t["wraps"]:
            wrap_jnt.append(jid); wrap_coef.append(coef)
    m.set("tendon_adr", np.array(wrap_adr, dtype=I32), I32)
    m.set("tendon_num", np.array([len(t["wraps"]) for t in tendons], dtype=I32), I32)
    m.set("wrap_objid", np.array(wrap_jnt, dtype=I32), I32)
    m.set("wrap_prm", np.array(wrap_coef, dtype=F64), F64)
    m.set("tendon_limited", np.array([t["limited"] for t in tendons], dtype=I32), I32)
    m.set("tendon_range", np.array([t["range"] for t in tendons], dtype=F64).reshape(ntendon, 2), F64)
    m.set("tendon_margin", np.array([t["margin"] for t in tendons], dtype=F64), F64)
    m.set("tendon_stiffness", np.array([t["stiffness"] for t in tendons], dtype=F64), F64)
    m.set("tendon_damping", np.array([t["damping"] for t in tendons], dtype=F64), F64)
    m.set("tendon_lengthspring", np.array([t["lengthspring"] for t in tendons], dtype=F64).reshape(ntendon, 2), F64)
    m.set("tendon_solref_lim", np.array([t["solref"] for t in tendons], dtype=F64).reshape(ntendon, 2), F64)
    m.set("tendon_solimp_lim", np.array([t["solimp"] for t in tendons], dtype=F64).reshape(ntendon, 5), F64)
    m.set("tendon_frictionloss", np.array([t["frictionloss"] for t in tendons], dtype=F64), F64)
    m.set("tendon_solref_fri", np.array([t["solref_fri"] for t in tendons], dtype=F64).reshape(ntendon, 2), F64)
    m.set("tendon_solimp_fri", np.array([t["solimp_fri"] for t in tendons], dtype=F64).reshape(ntendon, 5), F64)
    m.set("eq_obj1id", np.array([e["tendon"] for e in eqs], dtype=I32), I32)
    m.set("eq_data", np.array([e["polycoef"] for e in eqs], dtype=F64).reshape(neq, 5), F64)
    m.set("eq_solref", np.array([e["solref"] for e in eqs], dtype=F64).reshape(neq, 2), F64)
    m.set("eq_solimp", np.array([e["solimp"] for e in eqs], dtype=F64).reshape(neq, 5), F64)

    # ---- collision pair list (MuJoCo filter rules, docs "Collision detection") -----------------
    excl = set()
    con = root.find("contact")
    bname2id = {b["name"]: i for i, b in enumerate(bodies)}
    if con is not None:
        for e in con.findall("exclude"):
            a, b2 = bname2id[e.get("body1")], bname2id[e.get("body2")]
            excl.add((min(a, b2), max(a, b2)))
        if con.findall("pair"):
            raise MJCFError("explicit <contact><pair> not supported")
    pairs = []
    for g1 in range(ngeom):
        a = geoms[g1]
        if a["contype"] == 0 and a["conaffinity"] == 0:
            continue
        for g2 in range(g1 + 1, ngeom):
            b = geoms[g2]
            if not ((a["contype"] & b["conaffinity"]) or (b["contype"] & a["conaffinity"])):
                continue
            b1, b2 = a["body"], b["body"]
            if b1 == b2:
                continue
            w1, w2 = body_weldid[b1], body_weldid[b2]
            if w1 == w2:
                continue
            if (min(b1, b2), max(b1, b2)) in excl:
                continue
            wp1 = body_weldid[body_parentid[w1]]
            wp2 = body_weldid[body_parentid[w2]]
            if (w1 != 0 and w2 == wp1 and w2 != 0) or (w2 != 0 and w1 == wp2 and w1 != 0):
                continue  # parent-child filter (not applied when the parent is welded to the world)
            if a["type"] == GEOM_PLANE and b["type"] == GEOM_PLANE:
                continue
            # MuJoCo orders the pair so that type1 <= type2
            if a["type"] > b["type"]:
                pairs.append((g2, g1))
            else:
                pairs.append((g1, g2))
    m.set("npair", len(pairs), I32)
    m.set("pair_geom1", np.array([p[0] for p in pairs], dtype=I32), I32)
    m.set("pair_geom2", np.array([p[1] for p in pairs], dtype=I32), I32)

    # ---- names ---------------------------------------------------------------------------------
    m.names = {
        "body": [b["name"] for b in bodies],
        "joint": [j["name"] for j in joints],
        "geom": [g["name"] for g in geoms],
        "site": [s["name"] for s in sites],
        "camera": [c["name"] for c in cams],
        "light": [l["name"] for l in lights],
        "actuator": [a["name"] for a in acts],
        "sensor": [s["name"] for s in sens],
        "tendon": [t["name"] for t in tendons],
        "equality": [e["name"] for e in eqs],
        "mesh": mesh_names,
    }

    _set_const(m)
    return m


# ------------------------------------------------------------------------------------------------
# constants computed at qpos0 (MuJoCo `mj_setConst` [3P]: body/dof inverse weights feed the
# constraint regulariser R through diagApprox)
# ------------------------------------------------------------------------------------------------
def kinematics_np(m: FlatModel, qpos):
    """Plain numpy forward kinematics (compile-time use only). Returns xpos, xmat, xipos, anchors, axes."""
    nbody = m.nbody
    xpos = np.zeros((nbody, 3))
    xquat = np.tile(np.array([1.0, 0, 0, 0]), (nbody, 1))
    xanchor = np.zeros((m.njnt, 3))
    xaxis = np.zeros((m.njnt, 3))
    for b in range(1, nbody):
        p = m.body_parentid[b]
        jadr, jnum = m.body_jntadr[b], m.body_jntnum[b]
        if jnum == 1 and m.jnt_type[jadr] == JNT_FREE:
            a = m.jnt_qposadr[jadr]
            xpos[b] = qpos[a : a + 3]
            xquat[b] = quat_normalize(qpos[a + 3 : a + 7])
            xanchor[jadr] = xpos[b]
            xaxis[jadr] = [0, 0, 1]
            continue
        Rp = quat2mat(xquat[p])
        pos = xpos[p] + Rp @ m.body_pos[b]
        quat = quat_mul(xquat[p], m.body_quat[b])
        for j in range(jadr, jadr + jnum):
            R = quat2mat(quat)
            xanchor[j] = pos + R @ m.jnt_pos[j]
            xaxis[j] = R @ m.jnt_axis[j]
            a = m.jnt_qposadr[j]
            t = m.jnt_type[j]
            if t == JNT_HINGE:
                quat = quat_mul(quat, axisangle2quat(m.jnt_axis[j], qpos[a] - m.qpos0[a]))
                pos = xanchor[j] - quat2mat(quat) @ m.jnt_pos[j]
            elif t == JNT_SLIDE:
                pos = pos + xaxis[j] * (qpos[a] - m.qpos0[a])
            elif t == JNT_BALL:
                quat = quat_mul(quat, quat_normalize(qpos[a : a + 4]))
                pos = xanchor[j] - quat2mat(quat) @ m.jnt_pos[j]
        xpos[b] = pos
        xquat[b] = quat_normalize(quat)
    xmat = np.array([quat2mat(q) for q in xquat])
    xipos = xpos + np.einsum("bij,bj->bi", xmat, m.body_ipos)
    ximat = np.array([quat2mat(quat_mul(xquat[b], m.body_iquat[b])) for b in range(nbody)])
    return xpos, xquat, xmat, xipos, ximat, xanchor, xaxis


def body_jacobian_np(m: FlatModel, xpos, xmat, xanchor, xaxis, body, point):
    """6 x nv Jacobian [linear; angular] of `point` attached to `body` (world frame)."""
    nv = m.nv
    jacp = np.zeros((3, nv))
    jacr = np.zeros((3, nv))
    b = body
    while b > 0:
        jadr, jnum = m.body_jntadr[b], m.body_jntnum[b]
        for j in range(jadr, jadr + jnum):
            d = m.jnt_dofadr[j]
            t = m.jnt_type[j]
            if t == JNT_FREE:
                jacp[:, d : d + 3] = np.eye(3)
                R = xmat[b]
                for k in range(3):
                    jacr[:, d + 3 + k] = R[:, k]
                    jacp[:, d + 3 + k] = np.cross(R[:, k], point - xpos[b])
            elif t == JNT_BALL:
                R = xmat[b]
                for k in range(3):
                    jacr[:, d + k] = R[:, k]
                    jacp[:, d + k] = np.cross(R[:, k], point - xanchor[j])
            elif t == JNT_SLIDE:
                jacp[:, d] = xaxis[j]
            else:
                jacr[:, d] = xaxis[j]
                jacp[:, d] = np.cross(xaxis[j], point - xanchor[j])
        b = m.body_parentid[b]
    return jacp, jacr


def mass_matrix_np(m: FlatModel, qpos):
    """Joint-space inertia by the definition M = Σ_b Jᵀ diag(m, I) J + armature (compile-time only)."""
    xpos, xquat, xmat, xipos, ximat, xanchor, xaxis = kinematics_np(m, qpos)
    nv = m.nv
    M = np.zeros((nv, nv))
    for b in range(1, m.nbody):
        if m.body_mass[b] <= 0 and not np.any(m.body_inertia[b] > 0):
            continue
        jp, jr = body_jacobian_np(m, xpos, xmat, xanchor, xaxis, b, xipos[b])
        Iw = ximat[b] @ np.diag(m.body_inertia[b]) @ ximat[b].T
        M += m.body_mass[b] * jp.T @ jp + jr.T @ Iw @ jr
    M += np.diag(m.dof_armature)
    return M, (xpos, xquat, xmat, xipos, ximat, xanchor, xaxis)


def _set_const(m: FlatModel):
    nv, nbody = m.nv, m.nbody
    F64 = np.float64
    sub = m.body_mass.copy()
    for b in range(nbody - 1, 0, -1):
        sub[m.body_parentid[b]] += sub[b]
    m.set("body_subtreemass", sub, F64)
    body_invweight0 = np.zeros((nbody, 2))
    dof_invweight0 = np.zeros(nv)
    dof_M0 = np.zeros(nv)
    if nv > 0:
        M, (xpos, xquat, xmat, xipos, ximat, xanchor, xaxis) = mass_matrix_np(m, m.qpos0)
        dof_M0 = np.diag(M).copy()
        Minv = np.linalg.inv(M)
        for b in range(1, nbody):
            if m.body_weldid[b] == 0:
                continue
            jp, jr = body_jacobian_np(m, xpos, xmat, xanchor, xaxis, b, xipos[b])
            body_invweight0[b, 0] = np.trace(jp @ Minv @ jp.T) / 3.0
            body_invweight0[b, 1] = np.trace(jr @ Minv @ jr.T) / 3.0
        for j in range(m.njnt):
            d = m.jnt_dofadr[j]
            t = m.jnt_type[j]
            if t == JNT_FREE:
                dof_invweight0[d : d + 3] = np.mean(np.diag(Minv)[d : d + 3])
                dof_invweight0[d + 3 : d + 6] = np.mean(np.diag(Minv)[d + 3 : d + 6])
            elif t == JNT_BALL:
                dof_invweight0[d : d + 3] = np.mean(np.diag(Minv)[d : d + 3])
            else:
                dof_invweight0[d] = Minv[d, d]
    m.set("body_invweight0", body_invweight0, F64)
    m.set("dof_invweight0", dof_invweight0, F64)
    m.set("dof_M0", dof_M0, F64)
    # fixed tendons: length at qpos0 and inverse weight J M^-1 J^T (mj_setConst [3P])
    nt = int(m.ntendon) if "ntendon" in m.arrays else 0
    len0, tinv = np.zeros(nt), np.zeros(nt)
    if nt:
        for t in range(nt):
            J = np.zeros(nv)
            for w in range(int(m.tendon_adr[t]), int(m.tendon_adr[t]) + int(m.tendon_num[t])):
                j = int(m.wrap_objid[w])
                len0[t] += m.wrap_prm[w] * m.qpos0[m.jnt_qposadr[j]]
                J[m.jnt_dofadr[j]] += m.wrap_prm[w]
            tinv[t] = J @ Minv @ J
    m.set("tendon_length0", len0, F64)
    m.set("tendon_invweight0", tinv, F64)
    if nt and "tendon_lengthspring" in m.arrays:      # springlength -1: rest length = length at qpos0 (mj_setConst [3P])
        ls = np.array(m.tendon_lengthspring, dtype=np.float64).reshape(nt, 2)
        for t in range(nt):
            if ls[t, 0] == -1.0 and ls[t, 1] == -1.0:
                ls[t] = len0[t]
        m.set("tendon_lengthspring", ls, F64)


# ------------------------------------------------------------------------------------------------
# blob (de)serialisation: the on-disk model format == the byte string handed to the C-ABI
# ------------------------------------------------------------------------------------------------
_MAGIC = b"RSIMMDL1"


def to_blob(m: FlatModel) -> bytes:
    """Serialise: magic(8) nentries(u32) pad(u32) then entries {name[32], dtype u32 (0=i32,1=f64),
    count u32, offset u64} and 8-byte aligned payloads.  See include/rsim.h."""
    entries = []
    payload = bytearray()
    arrays = dict(m.arrays)
    # name tables (mjModel.names + name_<kind>adr, binding_utils.py:296-360): per kind one int32 entry "names:<kind>" = the UTF-8 bytes of the names, each
    # terminated by 0 (an unnamed object is the empty string), in id order -- what rsim_name2id / rsim_id2name of the C-ABI read
    for kind, names in (getattr(m, "names", None) or {}).items():
        raw = b"".join(((n or "").encode("utf-8") + b"\0") for n in names)
        arrays["names:" + kind] = np.frombuffer(raw, dtype=np.uint8).astype(np.int32)
    header_len = 16 + 48 * len(arrays)
    for name, a in arrays.items():
        if a.dtype == np.int32:
            dt = 0
        elif a.dtype == np.float64:
            dt = 1
        else:
            raise TypeError(f"{name}: unsupported dtype {a.dtype}")
        raw = np.ascontiguousarray(a).tobytes()
        off = header_len + len(payload)
        entries.append((name.encode()[:31], dt, a.size, off))
        payload += raw
        payload += b"\0" * ((-len(payload)) % 8)
    out = bytearray(_MAGIC) + struct.pack("<II", len(entries), 0)
    for name, dt, cnt, off in entries:
        out += struct.pack("<32sIIQ", name, dt, cnt, off)
    out += payload
    return bytes(out)


def from_blob(blob: bytes) -> FlatModel:
    if blob[:8] != _MAGIC:
        raise MJCFError("bad model blob magic")
    n = struct.unpack_from("<I", blob, 8)[0]
    m = FlatModel()
    for i in range(n):
        name, dt, cnt, off = struct.unpack_from("<32sIIQ", blob, 16 + 48 * i)
        name = name.rstrip(b"\0").decode()
        dtype = np.int32 if dt == 0 else np.float64
        a = np.frombuffer(blob, dtype=dtype, count=cnt, offset=off).copy()
        if name.startswith("names:"):
            m.names[name[6:]] = [(x.decode("utf-8") or None) for x in bytes(a.astype(np.uint8)).split(b"\0")[:-1]]
            continue
        m.arrays[name] = a
    _reshape(m)
    return m


_SHAPES = {
    "body_pos": 3, "body_quat": 4, "body_ipos": 3, "body_iquat": 4, "body_inertia": 3, "body_invweight0": 2,
    "jnt_pos": 3, "jnt_axis": 3, "jnt_range": 2, "jnt_solref": 2, "jnt_solimp": 5, "dof_solref": 2, "dof_solimp": 5,
    "geom_size": 3, "geom_pos": 3, "geom_quat": 4, "geom_friction": 3, "geom_solref": 2, "geom_solimp": 5,
    "geom_rgba": 4, "geom_rcenter": 3, "mesh_vert": 3, "site_pos": 3, "site_quat": 4, "site_size": 3, "site_rgba": 4,
    "actuator_gainprm": 3, "actuator_biasprm": 3, "actuator_ctrlrange": 2, "actuator_forcerange": 2,
}


def _reshape(m):
    for k, w in _SHAPES.items():
        if k in m.arrays:
            m.arrays[k] = m.arrays[k].reshape(-1, w)


def save_model(m: FlatModel, path: str):
    """Write `<path>` (blob) and `<path>.names.json` (name tables)."""
    import json

    with open(path, "wb") as f:
        f.write(to_blob(m))
    with open(path + ".names.json", "w") as f:
        json.dump(m.names, f)


def load_model(path: str) -> FlatModel:
    import json

    with open(path, "rb") as f:
        m = from_blob(f.read())
    with open(path + ".names.json") as f:
        m.names = json.load(f)
    return m
