import sys, numpy as np, torch
sys.path.insert(0,'.')
from tests.util import load_golden, make_hip
from robosuite_amd.vec_env import VecEnv
g,cfg,flat=load_golden("seed0_full","pickplace_iiwa")
B=128
env=VecEnv("PickPlace",B,flat,cfg,seed=0,horizon=100,bank_episodes=2)
env.reset()
gen=torch.Generator(device="cuda"); gen.manual_seed(5)
E=70
for t in range(13):
    a=torch.rand(B,env.action_dim,device="cuda",generator=gen)*2-1
    env.step(a)
a=torch.rand(B,env.action_dim,device="cuda",generator=gen)*2-1
b=env.env.batch
st={k:b.get(k)[E].copy() for k in ("qpos","qvel","qacc_warmstart","ctrl","cstate")}
act=a[E].cpu().numpy()
np.savez("gpurun_out/pp_div_state.npz", act=act, **st)
hm,hb=make_hip(flat,cfg,B=1)
for k in (1,2,3,4,5,6,8,10,12,15,20,25):
    for f,v in st.items(): hb.set(f,v[None])
    hb.control_step(torch.tensor(act[None],dtype=torch.float32,device="cuda"),k)
    q,v=hb.get("qpos")[0],hb.get("qvel")[0]
    print("n_sub",k,"finite",np.isfinite(q).all(),"max|v| %.3g"%np.nanmax(np.abs(v)),"div",hb.get("diverged")[0],"fingers v",np.round(v[7:13],1))
