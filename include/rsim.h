/*
 * rsim.h -- C-ABI of librsim_hip.so, the MI355X backend that replaces robosuite's MjSim hot path.
 *
 * Every entry point names the reference interface it stands in for (paths relative to the robosuite
 * checkout).  All functions return 0 on success, non-zero on error (message via rsim_last_error()).
 * Handles are opaque; the library owns all device memory; host buffers are owned by the caller.
 * No callbacks into the host language.  Thread-compatible per batch handle (one HIP stream per batch).
 */
#ifndef RSIM_H
#define RSIM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rsim_model rsim_model;
typedef struct rsim_batch rsim_batch;

/* Built-in controller description: OSC_POSE arm + GRIP gripper.
 * Replaces controllers/parts/arm/osc.py:120-224 (constructor arguments) and
 * controllers/parts/gripper/simple_grip.py:62-107; index tables are what
 * robots/robot.py:300-333 (`setup_references`) resolves by name. */
/* arm part-controller types with an in-kernel implementation (names = the reference's `type` strings, controller_factory.py:93-147):
 *   OSC_POSE       arm/osc.py, control_dim 6            OSC_POSITION  arm/osc.py with use_ori=False, control_dim 3 (zero orientation delta, osc.py:255-263)
 *   JOINT_POSITION generic/joint_pos.py:200-266, control_dim ndof: goal = q + scaled delta, tau = M_arm (kp (goal - q) - kd qd) + qfrc_bias
 *   JOINT_TORQUE   generic/joint_tor.py:111-167, control_dim ndof: tau = clip(scaled action, torque_limits) + qfrc_bias
 *   JOINT_VELOCITY generic/joint_vel.py:129-209, control_dim ndof: PID on the joint-velocity error (kp per joint, ki = 0.005 kp, kd = 0.001 kp over a
 *                  5-tap mean of the error increments, integrator frozen while the part's torques saturate) + qfrc_bias, clipped to the actuator range.
 *                  The reference constructor raises in the surveyed snapshot (joint_vel.py:118 assigns to a read-only property); the law implemented
 *                  is that file's run_controller with the defect resolved as use_torque_compensation = True.
 * The action row of rsim_control_step is [control_dim arm entries, 1 gripper entry if ngrip > 0]. */
enum rsim_ctrl_type { RSIM_CTRL_OSC_POSE = 0, RSIM_CTRL_OSC_POSITION = 1, RSIM_CTRL_JOINT_POSITION = 2, RSIM_CTRL_JOINT_TORQUE = 3, RSIM_CTRL_JOINT_VELOCITY = 4 };
#define RSIM_JNT_MAX 16
typedef struct rsim_ctrl_desc {
  int32_t ndof;            /* controlled joints: <= 8 for the OSC types (one arm); <= 16 for the joint-space types, where the arms of a multi-arm
                            * robot are concatenated in `robot.arms` order (= the order CompositeController slices the action, composite_controller.py:97-103) */
  int32_t qpos_idx[RSIM_JNT_MAX];     /* Controller.qpos_index */
  int32_t dof_idx[RSIM_JNT_MAX];      /* Controller.qvel_index */
  int32_t act_idx[RSIM_JNT_MAX];      /* robot._ref_actuators_indexes_dict[arm] */
  int32_t eef_site;        /* site id of Controller.ref_name */
  int32_t base_site;       /* site id of f"{naming_prefix}{part_name}_center" (osc.py:453) */
  float kp[RSIM_JNT_MAX];  /* osc.py:176 (6 task-space gains) / joint_pos.py:150 (ndof joint gains) */
  float damping_ratio;     /* kd = 2 sqrt(kp) damping_ratio, osc.py:177, joint_pos.py:151 */
  float input_min[RSIM_JNT_MAX], input_max[RSIM_JNT_MAX], output_min[RSIM_JNT_MAX], output_max[RSIM_JNT_MAX]; /* controller.py:149-168, first control_dim entries used */
  int32_t uncouple_pos_ori; /* osc.py:476-482 */
  float nullspace_kp;       /* control_utils.py:7 (default 10) */
  int32_t ngrip;            /* gripper actuators (<= 4), 0 = no gripper */
  int32_t grip_act[4];
  float grip_sign[4];       /* PandaGripper.format_action direction, models/grippers/panda_gripper.py:55-57 */
  float grip_speed;         /* panda_gripper.py:61 */
  int32_t type;             /* enum rsim_ctrl_type: which arm part controller of controller_factory.py:73-159 */
  float torque_min[RSIM_JNT_MAX], torque_max[RSIM_JNT_MAX]; /* RSIM_CTRL_JOINT_TORQUE: torque_limits (joint_tor.py:95-96; default = actuator ctrlrange);
                                                             * RSIM_CTRL_JOINT_VELOCITY: velocity_limits (joint_vel.py:113, 147-148; all zero = none) */
  int32_t impedance_mode;             /* 0 "fixed", 1 "variable": action = [damping_ratio x n, kp x n, goal update], 2 "variable_kp": [kp x n, goal update]
                                       * (osc.py:243-253 with n = 6, joint_pos.py:204-214 with n = ndof): kp = clip(kp, kp_limits),
                                       * kd = 2 sqrt(kp) clip(damping_ratio, damping_ratio_limits) (1 in mode 2), applied from this control step on */
  float kp_min[RSIM_JNT_MAX], kp_max[RSIM_JNT_MAX], damping_min[RSIM_JNT_MAX], damping_max[RSIM_JNT_MAX];
  int32_t interp_steps;               /* 0: no interpolator; > 0: LinearInterpolator.total_steps = ceil(ramp_ratio * controller_freq / policy_freq)
                                       * (utils/traj_utils.py:25-155, controller_factory.py:87-94).  Restated as it is: set_goal moves start := previous
                                       * GOAL (not the value reached) and step := 0; each run_controller uses start + (goal - start) / (total - step)
                                       * and advances step up to total - 1.  Joint-space types interpolate their goal vector; OSC_POSITION its goal_pos,
                                       * which the reference then uses as a WORLD position although it holds base-frame coordinates (osc.py:418-423).
                                       * OSC_POSE adds the orientation interpolator (controller_factory.py:102-106, osc.py:277-283, 433-437): set_goal
                                       * stores orientation_error(goal_ori, current eef ori) as the new goal VECTOR, run_controller takes
                                       * mat2euler(slerp(quat(euler start), quat(euler goal), (step + 1) / total)) as its orientation error
                                       * (traj_utils.py:129-146: the error vectors are handled as Euler angles). */
  int32_t part_of[RSIM_JNT_MAX];      /* joint-space types: which part controller (arm) owns joint i; JOINT_POSITION multiplies by that part's own
                                       * mass-matrix block only (joint_pos.py:256-259 uses Controller.mass_matrix of the part) */
  int32_t narm;                       /* OSC types: number of arm parts, 1 (0 is read as 1) or 2.  Two = one OperationalSpaceController object per arm, as
                                       * CompositeController builds them for a bimanual robot (composite_controller.py:70-121, default_baxter.json): each
                                       * arm runs its own torque law on its own mass-matrix block (Controller.mass_matrix of the part, controller.py:226-233),
                                       * around its own "<arm>_center" origin site.  The second arm's joints / gains / action limits sit at entries
                                       * [8, 8 + ndof2) of qpos_idx / dof_idx / act_idx and [8, 14) of kp / input_* / output_*; its slice of the action row follows
                                       * the first arm's control_dim entries (and that arm's gripper entry).  damping_ratio, uncouple_pos_ori and nullspace_kp
                                       * are shared.  Not combined with impedance modes or interpolators. */
  int32_t ndof2, eef_site2, base_site2;
} rsim_ctrl_desc;

/* On-device observation / reward epilogue of the fused control step.
 * Observation record = robosuite's per-key observables concatenated in `_get_observations` order (environments/base.py:429-465),
 * one float per entry of `obs_prog`; every entry is (kind, a, b):
 *   RSIM_OBS_QPOS/COS/SIN/QVEL/QACC: joint quantity at address a          (robots/robot.py:334-394)
 *   RSIM_OBS_SITE_POS: site a, component b                                 (robot.py:412-414 eef_pos)
 *   RSIM_OBS_BODY_QUAT / RSIM_OBS_SITE_QUAT: body / site a, xyzw comp b    (robot.py:416-462; T.convert_quat / T.mat2quat)
 *   RSIM_OBS_BODY_POS: body a, component b                                 (lift.py:371-373 cube_pos)
 *   RSIM_OBS_BODY_MINUS_SITE: body a minus site (b >> 2), component b & 3  (manipulation_env.py:218-242 gripper_to_cube_pos)
 *   RSIM_OBS_BODY_MINUS_BODY: body a minus body (b >> 2), component b & 3  (stack.py:432-438 cubeA_to_cubeB = cubeB_pos - cubeA_pos)
 *   RSIM_OBS_PEG_COS / _T / _D: `angle`, `t`, `d` of TwoArmPegInHole._compute_orientation (two_arm_peg_in_hole.py:462-486, 523-560) between
 *                             object_body (peg) and object2_body (hole)
 *   RSIM_OBS_REL_POS / REL_QUAT: `{obj}_to_{arm}eef_pos / _quat` of PickPlace (manipulation_env.py:244-329, pick_place.py:639-668): pose of object a
 *                             (index into objects[]) in the gripper frame {eef_pos = grip_site, eef_quat = eef_body}, component b.  As in the
 *                             reference, the OBJECT pose is the one sampled at the previous control step (the relative sensors run before the
 *                             object's own pos / quat sensors and read them from the observation cache) while the gripper pose is current; after
 *                             rsim_observe (reset) they are zero.  pos_slot[a] = offset of `{obj}_pos` (3 floats, followed by `{obj}_quat` xyzw)
 *                             in the observation record.
 *   RSIM_OBS_TASK_OBJECT: the `obj_id` observable of PickPlace single-object mode 1 (pick_place.py:626-635): RSIM_TASK_OBJECT of the env as a float;
 *                             in that mode a = -1 in REL_POS / REL_QUAT / BODY_POS / BODY_QUAT means "the env's current object" (its root body).
 * Sampling instants follow the reference exactly: after `reset()` every Observable samples on the LAST substep of a control step,
 * i.e. positions/orientations come from that substep's step1 kinematics, qpos/qvel from after its step2 (utils/observables.py:214-259).
 * task 1: reward = Lift.reward (environments/manipulation/lift.py:224-273), success = Lift._check_success (lift.py:433-444),
 * grasp = ManipulationEnv._check_grasp on the contact list (manipulation_env.py:331-376).
 * task 2: reward = Stack.reward / staged_rewards (environments/manipulation/stack.py:224-312) with object = cubeA, object2 = cubeB
 * (reach + grasp, lift + align, stack = lifted, released and cubeA touching cubeB via check_contact, utils/sim_utils.py:8-40),
 * success = Stack._check_success (stack.py:476-484: r_stack > 0); scaled by reward_scale / 2.0.
 * task 3: reward = TwoArmPegInHole.reward (two_arm_peg_in_hole.py:240-290) with object = peg, object2 = hole: success (d < 0.06, -0.12 <= t <= 0.14,
 * cos > 0.95; :513-521) + reaching + perpendicular / parallel distance + alignment terms, scaled by reward_scale / 5.0.
 * task 4: reward = PickPlace.reward / staged_rewards / _check_success (pick_place.py:274-429, 737-762) over nobj objects (all-objects mode): objects
 * in their bins + max(reach, grasp, lift, hover) over the others, scaled by reward_scale / 4.0; success = every object in its bin
 * (single_object_mode != 0: scaled by reward_scale, success = any object in its bin). */
enum { RSIM_OBS_QPOS = 0, RSIM_OBS_COS, RSIM_OBS_SIN, RSIM_OBS_QVEL, RSIM_OBS_QACC, RSIM_OBS_SITE_POS, RSIM_OBS_BODY_QUAT, RSIM_OBS_SITE_QUAT,
       RSIM_OBS_BODY_POS, RSIM_OBS_BODY_MINUS_SITE, RSIM_OBS_BODY_MINUS_BODY, RSIM_OBS_PEG_COS, RSIM_OBS_PEG_T, RSIM_OBS_PEG_D,
       RSIM_OBS_REL_POS, RSIM_OBS_REL_QUAT, RSIM_OBS_TASK_OBJECT };
#define RSIM_OBS_MAX 128
typedef struct rsim_task_desc {
  int32_t nobs;                       /* floats in the observation record (<= RSIM_OBS_MAX) */
  int32_t obs_prog[RSIM_OBS_MAX * 3]; /* (kind, a, b) per output float */
  int32_t task;                       /* 0 = none, 1 = Lift, 2 = Stack, 3 = TwoArmPegInHole, 4 = PickPlace */
  int32_t object_body;                /* cube root body */
  int32_t grip_site;                  /* gripper.important_sites["grip_site"] */
  float table_height;                 /* model.mujoco_arena.table_offset[2] */
  float lift_margin;                  /* 0.04 */
  float reward_scale;                 /* reward_scale (1.0), applied as reward_scale / 2.25 */
  int32_t reward_shaping;
  uint64_t left_pad_geoms, right_pad_geoms, object_geoms; /* bit g set: colliding-geom index g belongs to the group (_check_grasp) */
  int32_t object2_body;               /* Stack: cubeB root body */
  uint64_t object2_geoms;             /* Stack: cubeB contact geoms */
  /* PickPlace (task 4) */
  int32_t nobj;                       /* objects (<= 4), bin i is the target of object i */
  int32_t obj_body[4];                /* object root bodies */
  uint64_t obj_geoms[4];              /* contact geoms per object (colliding-geom bit masks) */
  int32_t pos_slot[4];                /* offset of `{obj}_pos` in the observation record (see RSIM_OBS_REL_POS) */
  int32_t eef_body;                   /* body whose quaternion is `{arm}eef_quat` */
  float bin2_pos[3], bin_size[2];     /* pick_place.py:188-199 */
  float bin_target[8];                /* target_bin_placements[i][0..1] (pick_place.py:570-583) */
  int32_t single_object_mode;         /* PickPlace: 0 = all objects; 1 = one object, drawn anew at every reset (PickPlaceSingle, pick_place.py:717-722, 800-807):
                                       * the env's current object id lives in RSIM_TASK_OBJECT -- set by the host's reset, and by the on-device episode reset
                                       * from the reset-bank column whose patch index is RSIM_PATCH_TASK_OBJECT -- observation entries with a = -1
                                       * (RSIM_OBS_REL_POS / REL_QUAT: object; RSIM_OBS_BODY_POS / BODY_QUAT: the object's root body) refer to it and
                                       * RSIM_OBS_TASK_OBJECT is the `obj_id` observable (:626-635); all pos_slot entries name the one `{obj}_pos` slot;
                                       * 2 = one fixed object (PickPlaceMilk / Bread / Cereal / Can, pick_place.py:800-847): the
                                       * reward is not divided by 4 (:308-310) and success = any object in its bin (:757-759).  The other objects stay in the
                                       * model -- the host's reset moves them to (10, 10, 10) (base.py:591-602) -- and in every sum of the reward, as in the reference */
} rsim_task_desc;

/* state / derived arrays addressable through rsim_get_array / rsim_set_array / rsim_device_ptr.
 * The derived arrays RSIM_XPOS .. RSIM_NITER and RSIM_SENSORDATA are written by rsim_forward / rsim_step1 / rsim_step2 / rsim_step / rsim_run_controller /
 * rsim_step2_last / rsim_observe; the fused rsim_control_step does not produce them.  Reading one of them (rsim_get_array, rsim_device_ptr, rsim_jac_*)
 * after a fused control step first runs rsim_forward on the current state -- the sim.forward() robosuite itself issues before it reads derived
 * quantities (base.py:298-303) -- so they are never the leftovers of an earlier launch. */
enum rsim_field {
  RSIM_QPOS = 0,       /* [B,nq]  sim.data.qpos   (binding_utils.py MjData.qpos)            */
  RSIM_QVEL,           /* [B,nv]  sim.data.qvel                                              */
  RSIM_QACC_WARMSTART, /* [B,nv]                                                            */
  RSIM_CTRL,           /* [B,nu]  sim.data.ctrl   (fixed_base_robot.py:153)                 */
  RSIM_TIME,           /* [B]     sim.data.time                                              */
  RSIM_CSTATE,         /* [B,cs]  controller state, cs = rsim_model_int(m, "cstate_size"): OSC types (32) goal_pos3 goal_ori9 q0[8] grip[4] tau[8];
                        *          joint-space types (64) goal[16] - grip[4] at 20 - tau[16] at 32; JOINT_VELOCITY (192) adds last_err[16] at 48,
                        *          summed_err[16] at 64, derr ring[5][16] at 80, ring ptr / size at 160 / 161, saturated[part] at 164;
                        *          variable-impedance modes (128): current kp[16] at 96, kd[16] at 112;
                        *          interpolator (200): start[16] at 180, step at 196; OSC_POSE orientation interpolator start[3] at 184, goal[3] at 188 */
  RSIM_XPOS,           /* [B,nbody,3]  sim.data.xpos      (derived, valid after forward/step1) */
  RSIM_XQUAT,          /* [B,nbody,4]  sim.data.xquat                                        */
  RSIM_QM,             /* [B,nv,nv]    dense mass matrix (mj_fullM, controller.py:226-227)  */
  RSIM_QFRC_BIAS,      /* [B,nv]       sim.data.qfrc_bias (controller.py:303-311)           */
  RSIM_QFRC_PASSIVE,   /* [B,nv] */
  RSIM_QFRC_ACTUATOR,  /* [B,nv] */
  RSIM_QFRC_CONSTRAINT,/* [B,nv] */
  RSIM_QACC,           /* [B,nv]       sim.data.qacc                                         */
  RSIM_CDOF,           /* [B,nv,6]     motion axes about the tree COM (for Jacobians)       */
  RSIM_ROOTCOM,        /* [B,nbody,3]  subtree COM per tree root                            */
  RSIM_CONTACT,        /* [B,maxcon,24] dist pos3 frame9 geom1 geom2 dim efc_adr fn friction5 */
  RSIM_EFC_FORCE,      /* [B,maxefc] */
  RSIM_NCON,           /* [B] int32    sim.data.ncon                                         */
  RSIM_NEFC,           /* [B] int32 */
  RSIM_NITER,          /* [B] int32    solver iterations of the last substep                 */
  RSIM_OBS,            /* [B,nobs]     observation record of the last control step (float32) */
  RSIM_REWARD,         /* [B]          reward of the last control step                       */
  RSIM_SUCCESS,        /* [B] int32    _check_success() after the last control step          */
  RSIM_DONE,           /* [B] int32    timestep >= horizon after the last control step (base.py:532-548) */
  RSIM_EP_STEP,        /* [B] int32    MujocoEnv.timestep of the running episode            */
  RSIM_EP_INDEX,       /* [B] int32    number of the running episode (0, 1, 2, ...; its reset came from bank slot number % n_episodes) */
  RSIM_DIVERGED,       /* [B] int32    how often the env hit MuJoCo's bad-state guard: after a substep that leaves a non-finite or > 1e10 qpos / qvel
                        *               entry the env is put back to qpos0 with zero velocity, control and time, as mj_checkPos / mj_checkVel +
                        *               mj_resetData do [3P]; robosuite never reads the corresponding mjData warning, this counter makes it visible */
  RSIM_OVERFLOW,       /* [B] int32    contacts + constraint rows the env had to DROP since the batch was created because a substep found more than the
                        *               compiled capacity (rsim_batch_limits: 16 contacts / 64 rows in the Lift configuration, 128 rows in the largest).
                        *               MuJoCo's nconmax = 5000 (models/assets/base.xml:5) never truncates; a non-zero count marks results that can differ
                        *               from the reference for that reason */
  RSIM_BANK_STALE,     /* [B] int32    on-device resets that found a reset-bank slot the host had not refilled in time (the stale entry was used):
                        *               0 as long as rsim_refill_reset_bank keeps up, i.e. no reset is ever replayed */
  RSIM_TERMINAL_OBS,   /* [B,nobs]     observation record of the control step that ENDED an env's episode (valid where RSIM_DONE was reported); with a reset
                        *               bank installed RSIM_OBS of such an env already holds the observation MujocoEnv.reset() returns for its next
                        *               episode (gym auto-reset convention), the reward / success flags stay those of the terminal step */
  RSIM_SENSORDATA,     /* [B,nsensordata] mjData.sensordata (binding_utils.py:935-938; Robot.get_sensor_measurement, robots/robot.py:739-751): the <force> and
                        *               <torque> sensors at a site (the grippers' ft_frame) as mj_sensorAcc evaluates them -- the wrench the site body's
                        *               parent transmits to it, from the solved accelerations and contact forces, in the site frame -- written by
                        *               rsim_forward / rsim_step2 / rsim_step (the values of the last substep, before its integration).  Sensors of
                        *               other types read zero.  rsim_model_int("nsensordata") gives the row length */
  RSIM_TASK_OBJECT,    /* [B] int32    PickPlace single-object mode 1: index of the object this env's current episode uses (see rsim_task_desc.single_object_mode) */
  RSIM_CAP_NEED,       /* [B,2] int32  largest number of contacts / constraint rows any substep of the env has asked for since the batch was created (what
                        *               RSIM_OVERFLOW's drops are measured against: a value above rsim_batch_limits means that substep was truncated);
                        *               sizes the compiled capacities against a workload (bench.py reports the maxima) */
  RSIM_QFRC_APPLIED,   /* [B,nv]       mjData.qfrc_applied: user-specified generalised forces, added to the smooth forces by rsim_forward / rsim_step1 / rsim_step2 /
                        *               rsim_step (the B = 1 compatibility entries: models/grippers/gripper_tester.py:197-202 writes its gravity compensation
                        *               there); zeroed by rsim_reset.  The fused rsim_control_step does not read it -- nothing on robosuite's env.step path
                        *               writes qfrc_applied */
  RSIM_POLISH,         /* [B] int32    wide configurations, debug entries: how the fp64 polish behind the Newton iteration of the last substep ended (no MuJoCo counterpart;
                        *               solve_newton in csrc/rsim_step.hip): 10000 x passes + 100000 x floor(-log10 of the scaled fp64 gradient at the accepted point)
                        *               + 10^7 x exit (1 gradient below tolerance, 2 improvement below tolerance, 3 pass budget,
                        *               5 direction was no descent direction, 6 a step raised the objective; 0 not run) */
  RSIM_FIELD_COUNT
};
#define RSIM_PATCH_TASK_OBJECT (-1)   /* rsim_set_reset_bank patch index: this column of a reset row is the episode's RSIM_TASK_OBJECT, not a float-table entry */

const char* rsim_last_error(void);

/* Ingests a compiled model blob ("RSIMMDL1": magic, entry table {name[32], dtype, count, offset}, 8-byte aligned payloads; written by rsim_mjcf_to_blob
 * below and by robosuite_amd.mjcf.to_blob).  Host only, no GPU. */
int rsim_model_create(const void* blob, size_t len, rsim_model** out);
/* mujoco.MjModel.from_xml_string itself (binding_utils.py:1077-1080; MujocoXML.get_model, models/base.py:125-147; the per-reset rebuild of
 * environments/base.py:262-269) for a host WITHOUT Python: the MJCF string robosuite assembles goes in, a model handle comes out.  The compiler is
 * robosuite_amd/csrc/rsim_mjcf.cpp (defaults classes, bodies / joints / geoms / sites, mesh loading + convex hulls + mesh inertia, inertia from geoms,
 * actuators, fixed tendons + tendon equalities, the collision pair list, names, the constants at qpos0: subtree masses and inverse weights); the Python
 * compiler robosuite_amd/mjcf.py is kept as its checker (tests/test_mjcf_cpp.py).  asset_dir (NULL = none) is what relative mesh file names are resolved
 * against, after <compiler meshdir>; robosuite writes absolute paths.  Malformed or unsupported MJCF fails with the reason in rsim_last_error(), as
 * MuJoCo's compiler raises.  Host only, no GPU.
 * rsim_mjcf_to_blob is the same compile stopping at the blob (what rsim_model_create ingests, what robosuite_amd.mjcf.to_blob writes): *blob is
 * malloc'ed and released with rsim_blob_free -- for binders that cache compiled models or ship them to other ranks. */
int rsim_model_compile(const char* xml, size_t len, const char* asset_dir, rsim_model** out);
int rsim_mjcf_to_blob(const char* xml, size_t len, const char* asset_dir, void** blob, size_t* blob_len);
void rsim_blob_free(void* blob);
void rsim_model_free(rsim_model* m);
/* scalar / size query by blob field name ("nq", "nv", ...); returns -1 if unknown */
int rsim_model_int(const rsim_model* m, const char* name);
/* Which compiled kernel configuration serves this model: 0 = 32 bodies x 16 dofs (Lift/Panda), 1 = 32 x 32 (Stack/Panda), 2 = 64 x 16 (Baxter);
 * 3 = 64 x 64 with tendon rows (PickPlace/IIWA+Robotiq140); -1 = none (rsim_batch_create would refuse it).  limits, if not NULL, receives 10 ints: {nbody, njnt, nv, ncgeom, nsite, ncon, nefc, npair, articulated trees} maxima
 * and a flag word: bit 0 = the configuration carries tendon / equality rows (the 32 x 16 one does not: such models go to the next larger one), bit 1 = two-arm
 * controller masks, bits 2 / 3 / 4 = the build keeps its constraint Jacobian / mass matrix / contact block in a per-env buffer in global memory instead of LDS
 * (an occupancy choice of the build, sized by rsim_batch_create; nothing a caller has to act on). */
int rsim_model_config(const rsim_model* m, int* limits);
/* mj_name2id / mj_id2name as robosuite's MjModel wrapper uses them (binding_utils.py:296-360: body_name2id, joint_name2id, geom_name2id, site_name2id,
 * actuator_name2id, camera_name2id, sensor_name2id, ... and the id2name inverses).  kind = "body" | "joint" | "geom" | "site" | "actuator" | "camera" | "light" |
 * "sensor" | "tendon" | "equality" | "mesh"; the names ride in the model blob (entries "names:<kind>").  rsim_name2id: id, or -1 if there is no such name;
 * rsim_id2name: the name (owned by the model, valid until rsim_model_free), or NULL for an unnamed object / a bad id. */
int rsim_name2id(const rsim_model* m, const char* kind, const char* name);
const char* rsim_id2name(const rsim_model* m, const char* kind, int id);
/* controller_factory (controllers/parts/controller_factory.py:73-159) for the built-in arm part (rsim_ctrl_type) + GRIP pair */
int rsim_model_set_controller(rsim_model* m, const rsim_ctrl_desc* desc);
/* observation / reward epilogue of rsim_control_step (must be set before rsim_batch_create) */
int rsim_model_set_task(rsim_model* m, const rsim_task_desc* desc);
/* colliding-geom index (bit position in rsim_task_desc geom masks) of a model geom id, -1 if the geom never collides */
int rsim_model_cgeom(const rsim_model* m, int geom_id);

/* mujoco.MjData(model) x B (binding_utils.py:586-590): allocates B environments on `device`.
 * per_env_params != 0 gives every env its own copy of the float model constants (domain randomisation,
 * per-episode object sizes); 0 shares one copy. */
int rsim_batch_create(rsim_model* m, int B, int device, int per_env_params, rsim_batch** out);
void rsim_batch_free(rsim_batch* b);
int rsim_batch_size(const rsim_batch* b);
int rsim_batch_limits(const rsim_batch* b, int* maxcon, int* maxefc);

/* mj_resetData (binding_utils.py:1091): qpos = qpos0, qvel = 0, ctrl = 0, time = 0 for envs with mask[i] != 0 (NULL = all) */
int rsim_reset(rsim_batch* b, const uint8_t* host_mask);
/* mj_forward / mj_step1 / mj_step2 / mj_step (binding_utils.py:1095-1107) on every env */
int rsim_forward(rsim_batch* b);
int rsim_step1(rsim_batch* b);
int rsim_step2(rsim_batch* b);
int rsim_step(rsim_batch* b);
/* Fused fast path = MujocoEnv.step's substep loop (environments/base.py:494-504) with the built-in controllers:
 * n_sub x { step1; control(action, policy_step = first); step2 } in ONE launch.  `actions_dev` is a DEVICE pointer
 * to [B, action_dim] float32 (action_dim = control_dim(type) + (ngrip>0); rsim_model_int(m, "action_dim")). */
int rsim_control_step(rsim_batch* b, const float* actions_dev, int n_sub);
/* CompositeController.run_controller() alone (composite_controller.py:109-116 -> osc.py:403-495, joint_pos.py:238-266, joint_vel.py:129-209,
 * simple_grip.py:150-186; the clip into ctrl: fixed_base_robot.py:143-153): mj_step1-equivalent position / velocity stage on the current state, then ONE
 * evaluation of the part controllers from RSIM_CSTATE as it stands (goals, initial joints, gripper action, PID state) -- no set_goal, no actuation, no
 * integration.  Results: RSIM_CTRL and the torque slots of RSIM_CSTATE (un-clipped tau, layout at enum rsim_field). */
int rsim_run_controller(rsim_batch* b);
/* The last rsim_step2 of a control step driven from the host side -- user part controllers evaluated on [B, ...] device tensors between rsim_step1 and
 * rsim_step2 of the whole batch (robosuite_amd/controllers.py, the batched form of the Controller plugin API, controllers/parts/controller.py:35-44,
 * 140-147): the substep's actuation / solve / integration, then what MujocoEnv.step does after its loop (base.py:508-548: timestep, reward, done) and
 * the observation record, with the on-device episode restart of rsim_control_step.  RSIM_DONE tells the caller which envs restarted. */
int rsim_step2_last(rsim_batch* b);
/* Robot.reset's controller re-creation (robots/robot.py:271 -> controller.py:125-131, osc.py:520-532):
 * forward kinematics, initial_joint := q, goal := current eef pose, gripper action := 0 */
int rsim_ctrl_reset(rsim_batch* b, const uint8_t* host_mask);
int rsim_sync(rsim_batch* b);
/* forward() + observation/reward epilogue without advancing time: the observation MujocoEnv.reset returns (base.py:298-347) */
int rsim_observe(rsim_batch* b);

/* Episode bookkeeping of MujocoEnv.step (base.py:508, 532-548): every rsim_control_step increments the per-env timestep and sets
 * RSIM_DONE when it reaches `horizon` (0 = never).  With a reset bank installed, a finished env is re-initialised ON THE DEVICE at the
 * end of that same launch: qpos := bank entry, qvel/ctrl/warm start/time := 0, the listed float-table entries (per-episode model edits
 * such as the Lift cube size, lift.py:311-318) are patched, and the controller is re-created at the start of the next launch
 * (robots/robot.py:271).  The same call then produces the observation MujocoEnv.reset() returns (forward + observables on the reset state)
 * in RSIM_OBS and keeps the finished episode's last record in RSIM_TERMINAL_OBS.
 * The bank is a RING of `n_episodes` (>= 2) slots per env: episode k of an env is read from slot k % n_episodes (RSIM_EP_INDEX counts
 * episodes for ever).  rsim_set_reset_bank fills slots with episodes 0 .. n_episodes-1 (drawn by the host in the reference's RNG order);
 * rsim_refill_reset_bank overwrites consumed slots with later episodes, so that every reset of every env is a fresh draw, as the
 * reference's hard reset is (base.py:277-347).  A reset that finds a slot not holding its episode is counted in RSIM_BANK_STALE.
 * bank: HOST float32 [B, n_episodes, nq + n_patch]; patch_idx: HOST int32 [n_patch] offsets into the env's float table
 * (rsim_param_offset).  Requires per_env_params when n_patch > 0.
 * rsim_refill_reset_bank: rows[i] (HOST float32 [n, nq + n_patch]) = reset `episode[i]` of env `env[i]`. */
int rsim_set_episode(rsim_batch* b, int horizon);
int rsim_set_reset_bank(rsim_batch* b, int n_episodes, int n_patch, const int32_t* patch_idx, const float* bank);
int rsim_refill_reset_bank(rsim_batch* b, int n, const int32_t* env, const int32_t* episode, const float* rows);
/* The same upkeep without ever stalling the control steps (the reference has no counterpart: its reset is a blocking recompile, base.py:277-347).
 * All three run on a side stream of the batch and neither wait for the control steps in flight nor make them wait; they may be called from a second
 * host thread while the first one keeps stepping (one upkeep thread per batch).
 *   rsim_bank_poll_begin          starts an asynchronous copy of RSIM_EP_INDEX into pinned host memory (at most one in flight);
 *   rsim_bank_poll                1 + the counters in ep_index[B] once that copy has landed, 0 while it is in flight (wait != 0: block), -1 on error.
 *                                 The counters only grow, so a copy taken beside running control steps is at worst slightly old;
 *   rsim_refill_reset_bank_async  like rsim_refill_reset_bank, through pinned staging, returns without waiting.  Only slots whose episode the env
 *                                 has already STARTED (episode <= polled counter) may be overwritten; the slot's tag is published after its row;
 *   rsim_bank_flush               returns when every refill issued so far is in the ring (tests, shutdown). */
int rsim_bank_poll_begin(rsim_batch* b);
int rsim_bank_poll(rsim_batch* b, int32_t* ep_index, int wait);
int rsim_refill_reset_bank_async(rsim_batch* b, int n, const int32_t* env, const int32_t* episode, const float* rows);
int rsim_bank_flush(rsim_batch* b);
/* offset of element `elem` of model float array `field` ("geom_size", "body_mass", ...) inside an env's float table, -1 if unknown */
int rsim_param_offset(const rsim_batch* b, const char* field, int elem);

/* Dispatch order of rsim_control_step (no reference counterpart; results do not depend on it).  longest_first != 0 (default): the envs that took
 * longest in the previous control step (contact-rich envs stay so for many steps) are handed to the first workgroups, which shortens the
 * tail of a launch; 0: env i = workgroup i. */
int rsim_set_schedule(rsim_batch* b, int longest_first);
/* Stream groups of rsim_control_step (no reference counterpart; results do not depend on it).  A control step lasts as long as its slowest env
 * (contact-rich envs take 3-4 x the median) and the envs are independent, so with groups = G > 1 the batch is stepped as G contiguous env
 * blocks, each on its own HIP stream: block g's step t + 1 starts as soon as ITS envs have finished step t, filling the CUs that the other
 * blocks' stragglers leave idle.  The caller's view is unchanged -- one call enqueues one step of all envs; rsim_sync() and every entry point
 * that reads or writes batch state first wait for all groups; the actions buffer of a step must stay untouched until then.  1 = one launch
 * on the batch's stream (default).  groups <= 32 and <= the batch size. */
int rsim_set_stream_groups(rsim_batch* b, int groups);
void* rsim_group_stream(rsim_batch* b, int group);   /* hipStream_t the control steps of env block `group` run on (for event timing) */

/* Per-phase cycle accounting of the fused kernel (no reference counterpart: the reference has no profiling).
 * enable != 0 (re)arms and zeroes the accumulators, 0 disarms; if `out` is non-NULL the current accumulators are copied out first:
 * cycles {load kin com crb broad narrow makec vel ctrl act solve euler store} then counts {substeps candidates contacts efc newton ls}. */
int rsim_profile(rsim_batch* b, int enable, unsigned long long* out, int n_out);
/* While profiling is armed every launch also logs, per env, {HW_ID, XCC_ID, start, end (s_memrealtime ticks, 100 MHz), #MPR runs, #support
 * evaluations, #Newton iterations, #broadphase candidates}: out = HOST u64 [B,8]. */
int rsim_wavelog(rsim_batch* b, unsigned long long* out);
/* restrict the phase accumulators to one env (-1 = all envs) */
int rsim_profile_env(rsim_batch* b, int env);
/* Capacity tier of every env for the NEXT control step, HOST int32 [B]: 0 = stepped by the batch's native kernel configuration, 1 = by the wider one (MuJoCo
 * never truncates contacts -- nconmax = 5000, models/assets/base.xml:5 -- so an env that outgrows the native contact / row capacity is stepped with more).
 * Synchronises the batch's stream.  No reference counterpart (diagnostics: which envs a lockstep launch waits for). */
int rsim_tier_snapshot(rsim_batch* b, int* host_tier);
/* The host-side settings that decide what a control step dispatches and how its solver stops -- compiled-in defaults and the RSIM_* environment overrides in
 * force -- as one string (valid until the next call).  Measurement files carry its sha next to the kernel's: evidence of other settings is not this build's. */
const char* rsim_tuning_defaults(void);
/* out2[0] = env-steps (env x control step) the wider capacity tier stepped since the batch was created, out2[1] = how many of them changed tier in mid-step
 * (carried on from the substep in which they outgrew the native capacity, or redone).  Synchronises the batch's stream.  bench.py reports both. */
int rsim_tier_stats(rsim_batch* b, unsigned long long* out2);
/* per candidate pair p (model pair order): out[p] = narrow-phase visits, out[640 + p] = support-function calls (out: 1280 entries), summed over envs and launches since arming */
int rsim_pairlog(rsim_batch* b, unsigned long long* out);

/* zero-copy numpy-view replacement: copy a field to / from HOST float32 (int32 for RSIM_NCON..) buffers of `count` elements */
int rsim_get_array(rsim_batch* b, int field, void* host_dst, size_t count);
int rsim_set_array(rsim_batch* b, int field, const void* host_src, size_t count);
/* device pointer + element count of a field (for DLPack / __cuda_array_interface__ aliasing by the host language) */
void* rsim_device_ptr(rsim_batch* b, int field, size_t* count);
void* rsim_stream(rsim_batch* b);

/* mj_jacSite / mj_jacBody (binding_utils.py:681-695, 826-851) for one env: jacp, jacr are HOST float64 [3,nv], may be NULL.
 * Valid after forward/step1. */
int rsim_jac_site(rsim_batch* b, int env, int site, double* jacp, double* jacr);
int rsim_jac_body(rsim_batch* b, int env, int body, double* jacp, double* jacr);

/* mj_fullM (controllers/parts/controller.py:226-227): dense joint-space inertia of one env, HOST float64 [nv, nv].  Valid after forward / step1 (after a fused
 * control step it is brought up to the current state first, like every derived quantity). */
int rsim_full_M(rsim_batch* b, int env, double* M);
/* sim.data.contact[:ncon] of one env (binding_utils.py:1008-1035; what utils/sim_utils.py:8-40 check_contact and manipulation_env.py:331-376 _check_grasp walk):
 * fills out[0 .. n), returns n = min(ncon, max_out), -1 on error.  geom1 / geom2 are MODEL geom ids; frame = rows {normal (geom1 -> geom2), tangent 1, tangent 2};
 * efc_address = first constraint row of the contact (-1: within the margin but not active); normal_force = constraint force along the normal of the last solve. */
typedef struct rsim_contact {
  double dist, pos[3], frame[9], friction[5], normal_force;
  int32_t geom1, geom2, dim, efc_address;
} rsim_contact;
int rsim_contacts(rsim_batch* b, int env, int max_out, rsim_contact* out);

/* DynamicsModder / per-episode model edits (utils/mjmod.py:1705-1964, lift.py:311-318): overwrite a float model array
 * (blob field name, e.g. "geom_size", "body_mass", "dof_damping") for envs [env0, env0+nenv). `values` is HOST float64
 * [nenv, count_per_env] in the blob's layout.  Requires per_env_params unless nenv == B with identical rows. */
int rsim_model_param_set(rsim_batch* b, const char* field, int env0, int nenv, const double* values, size_t count_per_env);

/* read back the live value of a float model array (same field names / layout as rsim_model_param_set): HOST float64 [nenv, count_per_env] */
int rsim_model_param_get(rsim_batch* b, const char* field, int env0, int nenv, double* values, size_t count_per_env);

/* Dynamics domain randomisation on the device = DomainRandomizationWrapper(randomize_dynamics) + DynamicsModder.randomize
 * (wrappers/domain_randomization_wrapper.py:47-81,227-259; utils/mjmod.py:1540-1729).  Every call re-draws all enabled parameters of every
 * env RELATIVE TO THE SAVED DEFAULTS (never cumulatively): val = default * (1 + p u) for "ratio" entries, default + p u for "size" entries,
 * u ~ U(-1, 1), then the reference's clips (>= 0; solref in [0, 1]); body quaternions are re-normalised (mjmod.py:1811-1826).
 * Defaults are the float tables at the time of rsim_dr_save_defaults (the wrapper calls save_defaults() after every reset); the on-device
 * episode reset patches defaults and live tables alike.  Draws come from a counter-based generator keyed by (seed, step, env, parameter):
 * the reference uses the unseeded global numpy generator here (mjmod.py:1721), so only the distribution can be matched, not a stream.
 * Requires per_env_params.  A magnitude of 0 disables that parameter. */
typedef struct rsim_dr_desc {
  float density_ratio, viscosity_ratio;                                   /* 0.1, 0.1 */
  float position_size, quaternion_size, inertia_ratio, mass_ratio;        /* 0.0015, 0.003, 0.02, 0.02 */
  float friction_ratio, solref_ratio, solimp_ratio;                       /* 0.1, 0.1, 0.1 */
  float frictionloss_size, damping_size, armature_size;                   /* 0.05, 0.01, 0.01 */
  float stiffness_ratio;                                                  /* 0.1 (domain_randomization_wrapper.py:73,77).  Accepted for completeness: DynamicsModder.mod_stiffness
                                                                           * skips joints whose stiffness is 0 (mjmod.py:1907-1909) and rsim_batch_create refuses models
                                                                           * with joint springs, so there is never a joint for it to act on */
  uint64_t body_mask, geom_mask, joint_mask;                              /* `body_names` / `geom_names` / `joint_names` of the wrapper (:55,64,71) as bit sets: bit b = body id b,
                                                                           * bit g = colliding-geom index g (rsim_model_cgeom), bit j = joint id j; 0 = every body / geom / joint
                                                                           * (the wrapper's None).  Elements outside a subset keep the values they have */
} rsim_dr_desc;
int rsim_dr_save_defaults(rsim_batch* b);
int rsim_randomize_dynamics(rsim_batch* b, const rsim_dr_desc* d, uint64_t seed, uint64_t step);


/* Rollout statistics across the GPUs of a job (SURVEY section 8(b) `rsim_allreduce_stats`, 8(e)): the path's only collective.  Environments are
 * independent worlds, sharded as contiguous blocks over one process per GPU (no data-path exchange); what a training loop sums over ranks once
 * per rollout is a handful of scalars -- env-steps, reward sum, successes, overflow / diverged counts (robosuite itself has no counterpart: its
 * envs are single-process, environments/base.py:23-42).  RCCL (librccl.so, resolved at run time from the process -- the copy PyTorch-ROCm loads --
 * so the library has no link-time dependency on it) over xGMI.
 *   rsim_comm_unique_id   rank 0 draws the 128-byte id (ncclGetUniqueId) and hands it to the other ranks out of band (file, env, torch store ...)
 *   rsim_comm_create      every rank, same id: ncclCommInitRank on `device`; world == 1 is valid (a one-rank communicator)
 *   rsim_allreduce_stats  HOST float64 [n] in, the reduction over all ranks out (op 0 = sum, 1 = max); blocks until the result is on the host
 * The Python host (robosuite_amd/shard.py) uses torch.distributed for the same reduction ("nccl" = RCCL on ROCm, gloo on CPU); this entry is for
 * binders without torch. */
typedef struct rsim_comm rsim_comm;
#define RSIM_COMM_ID_BYTES 128
int rsim_comm_unique_id(void* id_out, size_t bytes);
int rsim_comm_create(const void* id, size_t bytes, int rank, int world, int device, rsim_comm** out);
int rsim_allreduce_stats(rsim_comm* c, double* inout, int n, int op);
void rsim_comm_free(rsim_comm* c);


#ifdef __cplusplus
}
#endif
#endif
