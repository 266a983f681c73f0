"""Per-step HIP-vs-oracle / HIP-vs-fixture error trace of a controller fixture (diagnostic)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from tests.util import load_golden, make_oracle, make_hip
tag = sys.argv[1]
g, cfg, flat = load_golden(tag)
nq = flat.nq
om, od, oc = make_oracle(flat, cfg)
hm, hb = make_hip(flat, cfg, B=2)
s0 = g["states"][0]
od.qpos[:] = s0[1:1 + nq]; od.qvel[:] = s0[1 + nq:]; od.qacc_warmstart[:] = 0; od.forward(); oc.reset(od)
hb.set("qpos", s0[1:1 + nq][None].repeat(2, 0)); hb.set("qvel", s0[1 + nq:][None].repeat(2, 0)); hb.set("qacc_warmstart", 0); hb.set("ctrl", 0)
hb.forward(); hb.ctrl_reset()
for t in range(len(g["actions"])):
    a = torch.tensor(np.repeat(g["actions"][t][None], 2, 0), dtype=torch.float32, device="cuda")
    hb.control_step(a, 25); oc.env_step(od, g["actions"][t], 25)
    hq, hv = hb.get("qpos")[0], hb.get("qvel")[0]
    print(t, "dq %.2e dv %.2e | vs fixture dq %.2e dv %.2e | |v| %.2f ncon %d/%d dctrl %.2e" % (
        np.abs(hq - od.qpos).max(), np.abs(hv - od.qvel).max(), np.abs(hq - g["states"][t + 1][1:1 + nq]).max(),
        np.abs(hv - g["states"][t + 1][1 + nq:]).max(), np.abs(od.qvel).max(), hb.get("ncon")[0], od.ncon,
        np.abs(hb.get("ctrl")[0] - g["ctrl"][t]).max()))
