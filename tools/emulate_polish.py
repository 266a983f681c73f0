"""Offline emulation (numpy) of the kernel's polish on the dumped worst envs: the oracle's rows (J, aref, D, R, types, cone blocks), start at the KERNEL's acceleration.
Variants: unit step / exact line search; H and its factor in fp64 or fp32; cone weights in fp32."""
import sys, ctypes as C
sys.path.insert(0,'/root/repo')
import numpy as np
from oracle.oracle import OracleModel, OracleData, lib
L = lib(); L.rso_efc_id.argtypes = [C.c_void_p, C.c_int]; L.rso_efc_id.restype = C.c_int
g = np.load(sys.argv[1])
SAT, QUAD, LNEG, LPOS, CONE = range(5)

class Prob:
    def __init__(s, e):
        k = lambda n: g[f"e{e}_{n}"]
        s.om = OracleModel(k("blob").tobytes()); s.od = od = OracleData(s.om)
        od.qpos[:] = k("qpos"); od.qvel[:] = k("qvel"); od.qacc_warmstart[:] = k("ws"); od.ctrl[:] = k("ctrl")
        geo = k("geo"); od.forward_with_contact_geometry([dict(dist=r[0], pos=r[1:4], frame=r[4:13].reshape(3, 3)) for r in geo])
        s.n, s.nv = od.nefc, od.nv
        s.J = np.array(od.field("efc_J")).reshape(s.n, s.nv); s.aref = np.array(od.field("efc_aref")); s.D = np.array(od.field("efc_D")); s.R = np.array(od.field("efc_R"))
        s.fl = np.array(od.field("efc_frictionloss")); s.type = np.array(od.efc_types()); s.id = np.array([L.rso_efc_id(od.ptr, i) for i in range(s.n)])
        s.M = od.full_M(); s.fsm = np.array(od.field("qfrc_smooth")); s.asm = np.array(od.field("qacc_smooth"))
        s.a_opt = np.array(od.qacc); s.a_k = k("qacc").astype(np.float64)
        s.con = od.contacts()
        s.scale = 1.0   # relative numbers only
    def update(s, a, hess=False, f32w=False):
        jar = s.J @ a - s.aref
        n = s.n; f = np.zeros(n); st = np.zeros(n, int); cost = 0.0
        W = np.zeros((n, n)) if hess else None     # block-diagonal row-space Hessian
        i = 0
        while i < n:
            t = s.type[i]
            if t in (0, 6):
                fl, R, D, x = s.fl[i], s.R[i], s.D[i], jar[i]
                if x <= -R * fl: st[i] = LNEG; f[i] = fl; cost += fl * (-0.5 * R * fl - x)
                elif x >= R * fl: st[i] = LPOS; f[i] = -fl; cost += fl * (-0.5 * R * fl + x)
                else:
                    st[i] = QUAD; f[i] = -D * x; cost += 0.5 * D * x * x
                    if hess: W[i, i] = D
                i += 1
            elif t == 4:
                st[i] = QUAD; f[i] = -s.D[i] * jar[i]; cost += 0.5 * s.D[i] * jar[i] ** 2
                if hess: W[i, i] = s.D[i]
                i += 1
            elif t in (1, 2, 5):
                if jar[i] < 0:
                    st[i] = QUAD; f[i] = -s.D[i] * jar[i]; cost += 0.5 * s.D[i] * jar[i] ** 2
                    if hess: W[i, i] = s.D[i]
                i += 1
            else:
                c = s.con[s.id[i]]; dim = c["dim"]; fr = c["friction"]
                mu = fr[0] * np.sqrt(s.R[i + 1] / s.R[i])
                U = np.zeros(dim); U[0] = jar[i] * mu
                for j in range(1, dim): U[j] = jar[i + j] * fr[j - 1]
                T = np.sqrt((U[1:] ** 2).sum()); N = U[0]
                if N >= mu * T or (T <= 0 and N >= 0): pass
                elif mu * N + T <= 0 or (T <= 0 and N < 0):
                    for j in range(dim):
                        st[i + j] = QUAD; f[i + j] = -s.D[i + j] * jar[i + j]; cost += 0.5 * s.D[i + j] * jar[i + j] ** 2
                        if hess: W[i + j, i + j] = s.D[i + j]
                else:
                    Dm = s.D[i] / max(mu * mu * (1 + mu * mu), 1e-15); gg = N - mu * T
                    cost += 0.5 * Dm * gg * gg; f[i] = -Dm * gg * mu
                    for j in range(1, dim): f[i + j] = -f[i] / T * U[j] * fr[j - 1]
                    st[i:i + dim] = CONE
                    if hess:
                        ft = np.float32 if f32w else np.float64
                        mu_, T_, gg_ = ft(mu), ft(T), ft(gg); Uf = U.astype(ft); frf = np.asarray(fr, dtype=ft)
                        gr = np.zeros(dim, dtype=ft); gr[0] = mu_
                        for j in range(1, dim): gr[j] = -mu_ * Uf[j] * frf[j - 1] / T_
                        for j in range(dim):
                            for kk in range(dim):
                                h = gr[j] * gr[kk]
                                if j > 0 and kk > 0: h += -gg_ * mu_ * frf[j - 1] * frf[kk - 1] * ((ft(1) / T_ if j == kk else ft(0)) - Uf[j] * Uf[kk] / (T_ * T_ * T_))
                                # hcone is in U-space (scaled residuals): chain rule to row space multiplies by the scale factors of rows j, kk
                                sj = mu if j == 0 else fr[j - 1]; sk = mu if kk == 0 else fr[kk - 1]
                                W[i + j, i + kk] = float(ft(Dm) * h)
                i += dim
        ma = s.M @ a
        cost += 0.5 * (ma - s.fsm) @ (a - s.asm)
        grad = ma - s.fsm - s.J.T @ f
        return cost, grad, st, W, jar

def run(p, ls=True, f32fac=False, f32w=False, R=12, verbose=False):
    a = p.a_k.copy(); c_opt = p.update(p.a_opt)[0]
    hist = []
    for it in range(R):
        c, gk, st, W, jar = p.update(a, True, f32w)
        hist.append((c - c_opt) / max(1, abs(c_opt)))
        H = p.M + p.J.T @ W @ p.J
        if f32fac:
            Hf = H.astype(np.float32)
            try: Lc = np.linalg.cholesky(Hf.astype(np.float64)).astype(np.float32).astype(np.float64)     # fp32-rounded factor (optimistic model of an fp32 factorisation)
            except np.linalg.LinAlgError: return hist, "not PD in fp32"
            d = -np.linalg.solve(Lc.T, np.linalg.solve(Lc, gk))
        else:
            d = -np.linalg.solve(H, gk)
        if not ls:
            t = 1.0; ok = False
            for _ in range(4):
                c2 = p.update(a + t * d)[0]
                if c2 < c: ok = True; break
                t *= 0.5
            if not ok: return hist, "no descent"
            a = a + t * d
        else:
            # exact line search: bisection on phi'
            def dphi(al):
                eps = 1e-7 * max(1.0, abs(al))
                return (p.update(a + (al + eps) * d)[0] - p.update(a + (al - eps) * d)[0]) / (2 * eps)
            d0 = gk @ d
            if d0 >= 0: return hist, "not a descent direction"
            lo, hi = 0.0, 1.0
            while dphi(hi) < 0 and hi < 1e6: lo, hi = hi, 2 * hi
            for _ in range(60):
                mid = 0.5 * (lo + hi)
                if dphi(mid) < 0: lo = mid
                else: hi = mid
            a = a + 0.5 * (lo + hi) * d
    c = p.update(a)[0]; hist.append((c - c_opt) / max(1, abs(c_opt)))
    return hist, "budget"

for e in g["envs"][:8]:
    p = Prob(int(e))
    c0 = p.update(p.a_k)
    fm = lambda h: " ".join(f"{x:.0e}" for x in h[0][:13]) + " | " + h[1]
    print(f"env {e}: gap trajectory from the kernel's point")
    print("   unit step, fp64 H      :", fm(run(p, ls=False)))
    print("   line search, fp64 H    :", fm(run(p, ls=True)))
    print("   unit step, fp32 factor :", fm(run(p, ls=False, f32fac=True)))
    print("   line search, fp32 fac  :", fm(run(p, ls=True, f32fac=True)))
    print("   line search, fp32 cone weights + fp32 factor:", fm(run(p, ls=True, f32fac=True, f32w=True)))
