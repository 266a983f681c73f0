"""Sliding acceleration of the Lift cube on a tilted table (tests/test_hip_edge_cases.py::test_friction_cone_on_the_device_matches_coulomb) as a number, for A/B of
library builds: RSIM_LIB=... python tools/friction_probe.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.util import load_golden, make_hip   # noqa: E402

g, cfg, flat = load_golden("seed0_gentle")
cube, table = flat.name2id("geom", "cube_g0"), flat.name2id("geom", "table_collision")
mu = 0.3
for fac in (1.2, 2.0):
    th = np.arctan(fac * mu)
    f2 = flat.copy()
    f2.arrays["gravity"][:] = 9.81 * np.array([np.sin(th), 0.0, -np.cos(th)]); f2.arrays["density"][:] = 0; f2.arrays["viscosity"][:] = 0
    f2.arrays["geom_friction"][cube][0] = mu; f2.arrays["geom_friction"][table][0] = mu
    hm, hb = make_hip(f2, None, B=1)
    hb.set("qpos", g["states"][0][1:1 + flat.nq][None]); hb.set("qvel", 0); hb.set("qacc_warmstart", 0); hb.set("ctrl", 0)
    vx, nc = [], []
    for _ in range(150):
        hb.step1()
        c = np.zeros(flat.nu); c[:7] = hb.get("qfrc_bias")[0][:7]; c[7:9] = [0.04, -0.04]
        hb.set("ctrl", c[None])
        hb.step2()
        vx.append(float(hb.get("qvel")[0][9])); nc.append(int(hb.get("ncon")[0]))
    a = (vx[-1] - vx[-51]) / (50 * 0.002)
    print(f"fac {fac}: a {a:.5f} expected {9.81 * (np.sin(th) - mu * np.cos(th)):.5f}; a over steps 50-100 {(vx[99] - vx[49]) / 0.1:.5f}; ncon last 100 steps: {sorted(set(nc[50:]))}; vx[-1] {vx[-1]:.5f}")
