import sys, ctypes as C
sys.path.insert(0,'/root/repo')
import numpy as np
from oracle.oracle import OracleModel, OracleData, lib
from robosuite_amd import mjcf, factory
L = lib(); L.rso_forces_at.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
g = np.load(sys.argv[1])
flat, cfg = factory.load_shipped("pickplace_iiwa")
arm, fing = np.asarray(cfg["dof_idx"]), np.asarray(cfg["grip_dof_idx"])
obj = np.setdiff1d(np.arange(flat.nv), np.concatenate([arm, fing]))
for e, pol in zip(g["envs"], g["polish"]):
    k = lambda n: g[f"e{e}_{n}"]
    om = OracleModel(k("blob").tobytes()); od = OracleData(om)
    od.qpos[:] = k("qpos"); od.qvel[:] = k("qvel"); od.qacc_warmstart[:] = k("ws"); od.ctrl[:] = k("ctrl")
    geo = k("geo")
    contacts = [dict(dist=r[0], pos=r[1:4], frame=r[4:13].reshape(3, 3)) for r in geo]
    ok = od.forward_with_contact_geometry(contacts)
    a_o = np.array(od.qacc); a_k = k("qacc").astype(np.float64); n = od.nefc; nv = od.nv
    f_o = np.array(od.efc_force[:n]); f_k = k("efc").astype(np.float64)
    c_o, g_o = od.cost(a_o, True); c_k, g_k = od.cost(a_k, True)
    fk_at = np.zeros(n); st_k = np.zeros(n, dtype=np.int32); fo_at = np.zeros(n); st_o = np.zeros(n, dtype=np.int32)
    L.rso_forces_at(od.ptr, a_k.ctypes.data, fk_at.ctypes.data, st_k.ctypes.data)
    L.rso_forces_at(od.ptr, a_o.ctypes.data, fo_at.ctypes.data, st_o.ctypes.data)
    da = a_k - a_o
    print(f"env {e} polish {pol:08d} fed {ok} nefc {n}: cost gap {(c_k-c_o)/max(1,abs(c_o)):.1e}; |grad| at kernel's a: arm {np.abs(g_k[arm]).max():.1e} grip {np.abs(g_k[fing]).max():.1e} obj {np.abs(g_k[obj]).max():.1e} (at oracle's: {np.abs(g_o).max():.1e})")
    print(f"   da: arm {np.abs(da[arm]).max():.1e} grip {np.abs(da[fing]).max():.1e} obj {np.abs(da[obj]).max():.1e} of |a| {np.abs(a_o[arm]).max():.1e} {np.abs(a_o[fing]).max():.1e} {np.abs(a_o[obj]).max():.1e}; worst obj dofs {obj[np.argsort(-np.abs(da[obj]))[:4]]} da {np.sort(np.abs(da[obj]))[-4:][::-1]}")
    print(f"   kernel's reported forces vs the oracle rows' forces AT the kernel's a: max |df| {np.abs(f_k - fk_at).max():.2e} (fmax {np.abs(f_o).max():.1f}); rows whose state differs between kernel's a and optimum: {np.nonzero(st_k != st_o)[0].tolist()} states {st_k[st_k != st_o].tolist()} -> {st_o[st_k != st_o].tolist()}")
    types = od.efc_types()
    bad = np.argsort(-np.abs(f_k - fk_at))[:3]
    print(f"   rows with the largest force mismatch: {[(int(r), types[r], float(f_k[r]), float(fk_at[r])) for r in bad]}")
print("---- oracle's own Newton warm-started at the kernel's acceleration")
for e, pol in zip(g["envs"], g["polish"]):
    k = lambda n: g[f"e{e}_{n}"]
    om = OracleModel(k("blob").tobytes()); od = OracleData(om)
    geo = k("geo"); contacts = [dict(dist=r[0], pos=r[1:4], frame=r[4:13].reshape(3, 3)) for r in geo]
    out = []
    for ws in (k("ws").astype(np.float64), k("qacc").astype(np.float64)):
        od.qpos[:] = k("qpos"); od.qvel[:] = k("qvel"); od.qacc_warmstart[:] = ws; od.ctrl[:] = k("ctrl")
        od.forward_with_contact_geometry(contacts)
        out.append((od.solver_iter, float(np.abs(np.array(od.qacc) - k("qacc")).max())))
    # cone blocks: T of each elliptic contact at the optimum and at the kernel's point
    print(f"env {e}: oracle iterations from the state's warm start {out[0][0]}, from the kernel's qacc {out[1][0]}")
