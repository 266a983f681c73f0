import sys, numpy as np, torch
sys.path.insert(0,'.')
from tests.util import load_golden, make_hip, make_oracle
g,cfg,flat=load_golden("seed0_full","stack_panda")
nq=flat.nq
om,od,oc=make_oracle(flat,cfg); hm,hb=make_hip(flat,cfg,B=2)
def rel(a,b): return float(np.abs(np.asarray(a)-np.asarray(b)).max()/max(1e-12,np.abs(np.asarray(b)).max()))
for i in (0,7,29):
    s=g["states"][i]
    od.qpos[:]=s[1:1+nq]; od.qvel[:]=s[1+nq:]; od.qacc_warmstart[:]=0; od.ctrl[:]=0; od.forward()
    hb.set("qpos",s[1:1+nq][None].repeat(2,0)); hb.set("qvel",s[1+nq:][None].repeat(2,0)); hb.set("qacc_warmstart",0); hb.set("ctrl",0); hb.forward()
    print(i,"xpos",np.abs(hb.get("xpos")[0].ravel()-od.xpos).max(),"qM",rel(hb.get("qM")[0].ravel(),od.qM),"bias",rel(hb.get("qfrc_bias")[0],od.qfrc_bias),
      "pass",np.abs(hb.get("qfrc_passive")[0]-od.qfrc_passive).max(),"ncon",hb.get("ncon")[0],od.ncon,"nefc",hb.get("nefc")[0],od.nefc,
      "qacc",np.abs(hb.get("qacc")[0]-od.qacc).max()/max(1,np.abs(od.qacc).max()), "niter", hb.get("niter")[0])
s0=g["states"][0]
od.qpos[:]=s0[1:1+nq]; od.qvel[:]=s0[1+nq:]; od.qacc_warmstart[:]=0; od.ctrl[:]=0; od.forward(); oc.reset(od)
hb.set("qpos",s0[1:1+nq][None].repeat(2,0)); hb.set("qvel",s0[1+nq:][None].repeat(2,0)); hb.set("qacc_warmstart",0); hb.set("ctrl",0); hb.forward(); hb.ctrl_reset()
for t in range(len(g["actions"])):
    a=torch.tensor(np.repeat(g["actions"][t][None],2,0),dtype=torch.float32,device="cuda")
    hb.control_step(a,25); oc.env_step(od,g["actions"][t],25)
    hq,hv=hb.get("qpos")[0],hb.get("qvel")[0]
    print(t,"dq",np.abs(hq-od.qpos).max(),"dv",np.abs(hv-od.qvel).max(),"gold dq",np.abs(hq-g["states"][t+1][1:1+nq]).max())
import time
B=4096
hm,hb=make_hip(flat,cfg,B=B)
hb.set("qpos",s0[1:1+nq][None].repeat(B,0)); hb.set("qvel",0); hb.set("qacc_warmstart",0); hb.set("ctrl",0); hb.forward(); hb.ctrl_reset()
acts=torch.rand(40,B,7,device="cuda")*2-1
for t in range(5): hb.control_step(acts[t],25)
hb.sync(); t0=time.time()
for t in range(5,35): hb.control_step(acts[t],25)
hb.sync(); dt=time.time()-t0
print("stack 4096 envs: %.2f ms/step -> %.0f env-steps/s"%(dt/30*1e3, B*30/dt), "finite", np.isfinite(hb.get("qpos")).all())
