import sys, numpy as np, torch
sys.path.insert(0,'.')
from tests.util import load_golden
from robosuite_amd.vec_env import VecEnv
g,cfg,flat=load_golden("seed0_full","pickplace_iiwa")
B=128
env=VecEnv("PickPlace",B,flat,cfg,seed=0,horizon=100,bank_episodes=2)
env.reset()
gen=torch.Generator(device="cuda"); gen.manual_seed(5)
hist=[]
for t in range(34):
    a=torch.rand(B,env.action_dim,device="cuda",generator=gen)*2-1
    b=env.env.batch
    hist.append(dict(q=b.get("qpos")[87].copy(),v=b.get("qvel")[87].copy(),ws=b.get("qacc_warmstart")[87].copy(),ctrl=b.get("ctrl")[87].copy(),cs=b.get("cstate")[87].copy(),a=a[87].cpu().numpy(),time=b.get("time")[87]))
    env.step(a)
np.savez("gpurun_out/pp_env87.npz", **{f"{k}{t}":v[k] for t,v in enumerate(hist) if t>=28 for k in v})
q=env.env.batch.get("qpos")[87]; print("after step 33 finite:", np.isfinite(q).all())
# replay step 33 substep by substep on a B=1 batch to find the first bad substep
from tests.util import make_hip
hm,hb=make_hip(flat,cfg,B=1)
h=hist[33]
hb.set("qpos",h["q"][None]); hb.set("qvel",h["v"][None]); hb.set("qacc_warmstart",h["ws"][None]); hb.set("ctrl",h["ctrl"][None]); hb.set("cstate",h["cs"][None]); hb.set("time",np.array([h["time"]]))
for sub in range(25):
    hb.control_step(torch.tensor(h["a"][None],dtype=torch.float32,device="cuda"),1) if sub==0 else None
    break
# the fused step with n_sub=1 repeatedly is not identical to 25 substeps (set_goal each call); use forward() on the saved state instead
hb.set("qpos",h["q"][None]); hb.set("qvel",h["v"][None]); hb.set("qacc_warmstart",h["ws"][None]); hb.set("ctrl",h["ctrl"][None]); hb.forward()
print("state before step 33: ncon",hb.get("ncon")[0],"nefc",hb.get("nefc")[0],"niter",hb.get("niter")[0],"max|qacc|",np.abs(hb.get("qacc")[0]).max(),"max|v|",np.abs(h["v"]).max())
for c in hb.contacts(0): print("  contact",flat.names["geom"][c["geom1"]],flat.names["geom"][c["geom2"]],"dist %.4f dim %d"%(c["dist"],c["dim"]))
