import sys, numpy as np, torch
sys.path.insert(0,'.')
from tests.util import load_golden
from robosuite_amd.vec_env import VecEnv
g,cfg,flat=load_golden("seed0_full","pickplace_iiwa")
B=128
outs=[]
for rep in range(2):
    env=VecEnv("PickPlace",B,flat,cfg,seed=0,horizon=100,bank_episodes=2)
    env.reset()
    gen=torch.Generator(device="cuda"); gen.manual_seed(5)
    first_bad=None
    for t in range(60):
        a=torch.rand(B,env.action_dim,device="cuda",generator=gen)*2-1
        obs,rew,done,info=env.step(a)
        q=env.env.batch.get("qpos"); v=env.env.batch.get("qvel")
        bad=~(np.isfinite(q).all(1)&np.isfinite(v).all(1)&np.isfinite(obs.cpu().numpy()).all(1))
        if bad.any() and first_bad is None:
            first_bad=(t,np.nonzero(bad)[0][:5]); e=int(np.nonzero(bad)[0][0])
            print("rep",rep,"first non-finite at step",t,"envs",np.nonzero(bad)[0][:8],"qpos finite",np.isfinite(q[e]).all(),"qvel finite",np.isfinite(v[e]).all(),"obs finite",np.isfinite(obs[e].cpu().numpy()).all())
            print("  max|qvel| prev envs:", np.nanmax(np.abs(v),axis=1)[:8].round(1))
        if t%10==9: print("rep",rep,"t",t,"max|qvel| over envs %.1f"%np.nanmax(np.abs(v)), "max finger vel %.1f"%np.nanmax(np.abs(v[:,7:13])), "ncon max", int(env.env.batch.get("ncon").max()) if False else "")
    outs.append(env.env.batch.get("qpos").copy())
print("deterministic:", np.array_equal(outs[0],outs[1],equal_nan=True))
