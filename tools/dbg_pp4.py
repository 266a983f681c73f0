import sys, numpy as np, torch
sys.path.insert(0,'.')
from tests.util import load_golden
from robosuite_amd.vec_env import VecEnv
g,cfg,flat=load_golden("seed0_full","pickplace_iiwa")
B=128
env=VecEnv("PickPlace",B,flat,cfg,seed=0,horizon=100,bank_episodes=2)
env.reset()
q0=env.env.batch.get("qpos").copy()
gen=torch.Generator(device="cuda"); gen.manual_seed(5)
acts=[]; first={}
for t in range(150):
    a=torch.rand(B,env.action_dim,device="cuda",generator=gen)*2-1
    acts.append(a.cpu().numpy())
    env.step(a)
    d=env.env.batch.get("diverged")
    for e in np.nonzero(d>0)[0]:
        if int(e) not in first:
            first[int(e)]=t; print("env",e,"diverged at step",t,"ncon",env.env.batch.get("ncon")[e] if False else "")
print("diverged envs:",sorted(first.items()))
e=min(first,key=first.get) if first else 0
np.savez("gpurun_out/pp_div.npz", q0=q0[e], acts=np.array(acts)[:,e], env=e, step=first.get(e,-1))
