"""Which reference environment / robot models fit a compiled kernel configuration (CPU only: the unmodified reference assembles the MJCF, the
compiler ingests it, rsim_model_config picks the configuration).  Output is the table in DESIGN.md section 5."""
import sys, time, traceback
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import gen_golden as G
import numpy as np
from robosuite_amd import backend
suite=G.suite
cases=[("Lift","Panda"),("Lift","Sawyer"),("Lift","IIWA"),("Lift","Jaco"),("Lift","Kinova3"),("Lift","UR5e"),
       ("Stack","Panda"),("NutAssembly","Panda"),("NutAssemblySquare","Panda"),("PickPlace","Panda"),("PickPlaceCan","Panda"),("Door","Panda"),("Wipe","Panda"),("ToolHang","Panda"),
       ("TwoArmLift",["Panda","Panda"]),("TwoArmHandover",["Panda","Panda"]),("TwoArmTransport",["Panda","Panda"]),("TwoArmPegInHole",["Panda","Panda"]),("TwoArmLift","Baxter")]
for env_name, robots in cases:
    try:
        kw=dict(has_renderer=False, has_offscreen_renderer=False, use_camera_obs=False, control_freq=20, seed=0)
        if isinstance(robots,list): kw["env_configuration"]="opposed" if env_name!="TwoArmTransport" else "opposed"
        if robots=="Baxter": kw["env_configuration"]="single-robot"
        env=suite.make(env_name, robots=robots, **kw)
        flat=env.sim.model._model._flat
        hm=backend.HipModel(flat)
        cid,lim=hm.kernel_config()
        used=set(flat.arrays["pair_geom1"].tolist())|set(flat.arrays["pair_geom2"].tolist())
        jt=set(int(t) for t in flat.jnt_type)
        multi=int(max(flat.body_jntnum)) if flat.nbody else 0
        env.step(np.zeros(env.action_dim))
        print(f"{env_name:18s} {str(robots):22s} nbody {flat.nbody:3d} nv {flat.nv:3d} ncg {len(used):3d} npair {len(flat.arrays['pair_geom1']):4d} nsite {flat.nsite:3d} ntendon {int(flat.ntendon)} jointtypes {sorted(jt)} max jnt/body {multi} -> cfg {cid}")
    except Exception as e:
        print(f"{env_name:18s} {str(robots):22s} FAILED: {type(e).__name__}: {str(e)[:140]}")
