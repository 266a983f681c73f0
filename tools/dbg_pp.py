import sys, numpy as np, torch
sys.path.insert(0,'.')
from tests.util import load_golden, make_hip, make_oracle
g,cfg,flat=load_golden("seed0_full","pickplace_iiwa")
nq=flat.nq
om,od,oc=make_oracle(flat,cfg); hm,hb=make_hip(flat,cfg,B=2)
print("cfg", hm.kernel_config())
def rel(a,b): return float(np.abs(np.asarray(a)-np.asarray(b)).max()/max(1e-12,np.abs(np.asarray(b)).max()))
for i in (0,5,19):
    s=g["states"][i]
    od.qpos[:]=s[1:1+nq]; od.qvel[:]=s[1+nq:]; od.qacc_warmstart[:]=0; od.ctrl[:]=0; od.forward()
    hb.set("qpos",s[1:1+nq][None].repeat(2,0)); hb.set("qvel",s[1+nq:][None].repeat(2,0)); hb.set("qacc_warmstart",0); hb.set("ctrl",0); hb.forward()
    print(i,"xpos",np.abs(hb.get("xpos")[0].ravel()-od.xpos).max(),"qM",rel(hb.get("qM")[0].ravel(),od.qM),"bias",rel(hb.get("qfrc_bias")[0],od.qfrc_bias),
      "ncon",hb.get("ncon")[0],od.ncon,"nefc",hb.get("nefc")[0],od.nefc,"qacc",np.abs(hb.get("qacc")[0]-od.qacc).max()/max(1,np.abs(od.qacc).max()),"niter",hb.get("niter")[0])
    d=np.abs(hb.get("qacc")[0]-od.qacc); print("   worst qacc dof",int(d.argmax()),d.max(),od.qacc[int(d.argmax())])
s0=g["states"][0]
od.qpos[:]=s0[1:1+nq]; od.qvel[:]=s0[1+nq:]; od.qacc_warmstart[:]=0; od.ctrl[:]=0; od.forward(); oc.reset(od)
hb.set("qpos",s0[1:1+nq][None].repeat(2,0)); hb.set("qvel",s0[1+nq:][None].repeat(2,0)); hb.set("qacc_warmstart",0); hb.set("ctrl",0); hb.forward(); hb.ctrl_reset()
for t in range(len(g["actions"])):
    a=torch.tensor(np.repeat(g["actions"][t][None],2,0),dtype=torch.float32,device="cuda")
    hb.control_step(a,25); oc.env_step(od,g["actions"][t],25)
    dq=np.abs(hb.get("qpos")[0]-od.qpos)
    print(t,"dq arm",dq[:7].max(),"fingers",dq[7:13].max(),"objs",dq[13:].max())
import time
B=2048
hm,hb=make_hip(flat,cfg,B=B)
hb.set("qpos",s0[1:1+nq][None].repeat(B,0)); hb.set("qvel",0); hb.set("qacc_warmstart",0); hb.set("ctrl",0); hb.forward(); hb.ctrl_reset()
acts=torch.rand(20,B,7,device="cuda")*2-1
for t in range(3): hb.control_step(acts[t],25)
hb.sync(); t0=time.time()
for t in range(3,13): hb.control_step(acts[t],25)
hb.sync(); dt=time.time()-t0
print("pickplace %d envs: %.2f ms/step -> %.0f env-steps/s"%(B,dt/10*1e3,B*10/dt),"finite",np.isfinite(hb.get("qpos")).all())
