import sys, numpy as np
sys.path.insert(0,'.')
from tests.util import load_golden, make_hip, make_oracle
g,cfg,flat=load_golden("ctl_joint_torque","peg_baxter")
om,od,_=make_oracle(flat); hm,hb=make_hip(flat,None,B=2)
rng=np.random.default_rng(7)
for it in range(40):
    q=flat.qpos0.copy()
    for j in range(flat.njnt):
        lo,hi=flat.jnt_range[j]; q[flat.jnt_qposadr[j]]=rng.uniform(lo+0.05*(hi-lo),hi-0.05*(hi-lo))
    v=0.5*rng.standard_normal(flat.nv)
    od.qpos[:]=q; od.qvel[:]=v; od.qacc_warmstart[:]=0; od.ctrl[:]=0; od.forward()
    if od.ncon==0 or od.ncon>=hb.maxcon: continue
    hb.set("qpos",q[None].repeat(2,0)); hb.set("qvel",v[None].repeat(2,0)); hb.set("qacc_warmstart",0); hb.set("ctrl",0); hb.forward()
    for a,b in zip(hb.contacts(0),od.contacts()):
        t1,t2=int(flat.geom_type[b["geom1"]]),int(flat.geom_type[b["geom2"]])
        print(it,"pair",b["geom1"],b["geom2"],"types",t1,t2,"size",flat.geom_size[b["geom1"]],flat.geom_size[b["geom2"]],"dist hip %.6f orc %.6f diff %.2e"%(a["dist"],b["dist"],a["dist"]-b["dist"]), "dpos %.2e"%np.abs(a["pos"]-b["pos"]).max(), "n.n %.6f"%float(np.dot(np.ravel(a["frame"])[:3],np.ravel(b["frame"])[:3])))
