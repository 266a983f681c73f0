// rsim_internal.h -- device-side model/batch descriptors shared by the HIP kernels and the C-ABI host code.
// Not part of the public boundary (that is include/rsim.h).
#pragma once
#include <stdint.h>

// ---- packed int tables (shared by all envs) ---------------------------------------------------------------
enum {
  IO_body_parentid, IO_body_rootid, IO_body_jntadr, IO_body_jntnum, IO_body_dofadr, IO_body_dofnum, IO_body_depth,
  IO_body_mocap, IO_body_moving, IO_body_ancmask /*2 ints*/, IO_body_dofmask /*2 ints*/, IO_body_isroot,
  IO_jnt_type, IO_jnt_qposadr, IO_jnt_dofadr, IO_jnt_bodyid, IO_jnt_limited,
  IO_dof_bodyid, IO_dof_jntid, IO_dof_ancmask /*2*/, IO_dof_cvelmask /*2*/, IO_dof_zerodot,
  IO_cg_geomid, IO_cg_type, IO_cg_bodyid, IO_cg_condim, IO_cg_priority, IO_cg_meshadr, IO_cg_meshnum,
  IO_pair_g1, IO_pair_g2,
  IO_site_bodyid,
  IO_act_trnid, IO_act_biastype, IO_act_ctrllimited, IO_act_forcelimited,
  IO_tendon_adr, IO_tendon_num, IO_tendon_limited, IO_wrap_dof, IO_wrap_qadr, IO_eq_tendon,   /* fixed tendons + equality/tendon rows */
  IO_sensor_type /* 0 force, 1 torque, -1 other (reads zero) */, IO_sensor_site, IO_sensor_adr /* nsensor + 1 entries */,
  IO_COUNT
};
// ---- packed float tables (per-env stride `fstride`, 0 = shared) -------------------------------------------
enum {
  FO_body_pos, FO_body_quat, FO_body_ipos, FO_body_iquat, FO_body_mass, FO_body_inertia, FO_body_invweight0, FO_body_subtreemass,
  FO_jnt_pos, FO_jnt_axis, FO_jnt_range, FO_jnt_margin, FO_jnt_solref, FO_jnt_solimp, FO_qpos0,
  FO_dof_armature, FO_dof_damping, FO_dof_frictionloss, FO_dof_solref, FO_dof_solimp, FO_dof_invweight0,
  FO_cg_size, FO_cg_pos, FO_cg_quat, FO_cg_friction, FO_cg_solref, FO_cg_solimp, FO_cg_solmix, FO_cg_margin, FO_cg_gap,
  FO_cg_rbound, FO_cg_rcenter, FO_cg_aabb /* mesh geoms: centre3 + half3 of the hull's box in the geom frame */,
  FO_cg_capsule /* mesh geoms: bounding capsule p3 q3 R in the geom frame (R < 0: none) + pad */,
  FO_site_pos, FO_site_quat,
  FO_act_gear, FO_act_gainprm, FO_act_biasprm, FO_act_ctrlrange, FO_act_forcerange,
  FO_opt /* timestep, gx,gy,gz, density, viscosity, impratio, windx,windy,windz */,
  FO_wrap_prm, FO_tendon_range, FO_tendon_margin, FO_tendon_solref, FO_tendon_solimp, FO_tendon_len0, FO_tendon_invw,
  FO_eq_data0 /* polycoef[0] */, FO_eq_solref, FO_eq_solimp,
  FO_tendon_stiffness, FO_tendon_damping, FO_tendon_lspring /* 2 per tendon */,
  FO_tendon_fl /* frictionloss */, FO_tendon_solref_fri /* 2 */, FO_tendon_solimp_fri /* 5 */,
  FO_COUNT
};

// ---- per-lane packed int constants ("lane table", 64 ints per row): what lane l needs in each of its roles --------------
enum {
  LT_part,   /* body b = lane: pointer-jump partners of kinematics rounds 0..3 (8 bits each) */
  LT_part4,  /* round-4 partner (trees deeper than 16) */
  LT_binfo,  /* body: jtype(4b, 15 = no joint) | qposadr<<4 | dofadr<<12 | rootid<<20 | moving<<28 */
  LT_bdofs,  /* body: bit k set if dof k moves the body */
  LT_dinfo,  /* dof i = lane: body | zerodot<<8 | limited<<9 | qposadr<<10 | jntid<<18 | jtype<<26 */
  LT_ginfo,  /* colliding geom g = lane: body | type<<8 */
  LT_sinfo,  /* site k = lane: body */
  LT_ainfo,  /* actuator a = lane: dof | qposadr<<8 | biastype<<16 | ctrllimited<<18 | forcelimited<<19 */
  LT_pair0, LT_pair1, LT_pair2, /* candidate pair p = lane + 64 t: g1 | g2<<8 | valid<<16 */
  LT_ghull,  /* colliding geom g = lane: first slot of its hull in the LDS-resident vertex pool, -1 = scan from global memory */
  LT_mfbits, /* 0/1 operands of the tree-incidence MFMAs: sub[8] | bodydof[2][4] | dcv[4] | m1[4] | m2[4] */
  LT_pair3, LT_pair4, /* candidate pairs 192..319 (configurations with more than three rows of pairs) */
  LT_pair5, LT_pair6, LT_pair7, LT_pair8, LT_pair9, /* candidate pairs 320..639 */
  LT_COUNT
};
#define RSIM_MAXDYNROOT 8
#define RSIM_PAIR_MAX 640    /* candidate pairs of the largest kernel configuration (per-pair profile counters) */
#define RSIM_HULL_POOL 512   /* hull vertices kept resident in LDS (distal links / gripper first) */
#define RSIM_ARM_MAX 8
#define RSIM_GRIP_MAX 4

// built-in controller configuration (OSC_POSE arm + GRIP gripper), SURVEY section 8 rows a9-a19
struct DCtrl {
  int enabled;
  int ndof;
  int qpos_idx[RSIM_JNT_MAX], dof_idx[RSIM_JNT_MAX], act_idx[RSIM_JNT_MAX];
  int eef_site, base_site;
  int narm, ndof2, eef_site2, base_site2;   // OSC types: a second arm part (its joints / gains / limits sit at offset RSIM_ARM_MAX of the 16-entry arrays, its state at RSIM_CS_SIZE)
  float kp[RSIM_JNT_MAX], kd[RSIM_JNT_MAX], in_min[RSIM_JNT_MAX], in_max[RSIM_JNT_MAX], out_min[RSIM_JNT_MAX], out_max[RSIM_JNT_MAX];
  float tl_lo[RSIM_JNT_MAX], tl_hi[RSIM_JNT_MAX];
  int part_of[RSIM_JNT_MAX];
  int interp_steps;     // LinearInterpolator.total_steps, 0 = none
  int imp_mode, nimp;   // impedance mode (0 fixed, 1 variable, 2 variable_kp) and number of gains in the action (6 / ndof)
  float kp_min[RSIM_JNT_MAX], kp_max[RSIM_JNT_MAX], dr_min[RSIM_JNT_MAX], dr_max[RSIM_JNT_MAX];
  int type, cdim;   // rsim_ctrl_type, control_dim of the arm part(s)
  int cs_size;      // floats of per-env controller state (RSIM_CS_SIZE for the OSC types, RSIM_CS_SIZE_JOINT for the joint-space ones)
  int uncouple;
  float nullspace_kp;
  int ngrip;
  int grip_act[RSIM_GRIP_MAX];
  float grip_sign[RSIM_GRIP_MAX];
  float grip_speed;
  int action_dim;
};
// controller state block per env (floats): goal_pos[3] goal_ori[9] q0[8] grip_action[4] torques[8]
#define RSIM_CS_GOALQ 0    /* joint-space parts: goal_qpos / goal_torque[8] share the task-space goal slots */
#define RSIM_CS_GOALPOS 0
#define RSIM_CS_GOALORI 3
#define RSIM_CS_Q0 12
#define RSIM_CS_GRIP 20
#define RSIM_CS_TAU 24
#define RSIM_CS_SIZE 32
/* joint-space parts (up to RSIM_JNT_MAX joints): goal[16] at 0, grip[4] at 20 (shared with the OSC layout), tau[16] at 32 */
#define RSIM_CS_TAU_JOINT 32
#define RSIM_CS_SIZE_JOINT 64
/* JOINT_VELOCITY PID state (joint_vel.py:105-110): last_err[16], summed_err[16], RingBuffer(5) of error increments, its ptr / size, saturated flag per part */
#define RSIM_CS_JV_LASTERR 48
#define RSIM_CS_JV_SUMMED 64
#define RSIM_CS_JV_RING 80
#define RSIM_CS_JV_PTR 160
#define RSIM_CS_JV_SIZE 161
#define RSIM_CS_JV_SAT 164
#define RSIM_CS_KP 96          /* variable-impedance modes: gains in force (set by set_goal, read by run_controller) */
#define RSIM_CS_KD 112
#define RSIM_CS_SIZE_VARIMP 128
#define RSIM_CS_SIZE_JVEL 192
#define RSIM_CS_ISTART 180     /* LinearInterpolator: start[16], step */
#define RSIM_CS_ISTART_ORI (RSIM_CS_ISTART + 4)    /* OSC_POSE orientation interpolator: start[3], goal[3] (error vectors) */
#define RSIM_CS_IGOAL_ORI (RSIM_CS_ISTART + 8)
#define RSIM_CS_ISTEP 196
#define RSIM_CS_SIZE_INTERP 200
#ifndef RSIM_CS_MAX
#define RSIM_CS_MAX 200
#endif
#define RSIM_CS_LDS 64         /* controller-state floats staged in LDS by the step kernel (the OSC and plain joint-space layouts entirely); slots beyond stay in global memory */

// observation / reward epilogue (include/rsim.h rsim_task_desc), device form
struct DTask {
  int enabled, nobs, task, object_body, grip_site, reward_shaping;
  float table_height, lift_margin, reward_scale;
  unsigned long long left_pad, right_pad, object_geoms, object2_geoms;
  int object2_body;
  int nobj, obj_body[4], pos_slot[4], eef_body;
  unsigned long long obj_geoms[4];
  float bin2_pos[3], bin_size[2], bin_target[8];
  int single_mode;       // PickPlace single-object modes: reward not divided by the object count, success = any object in its bin
  const int* obs_prog;   // device [nobs][3]
};

struct DModel {
  int nq, nv, nu, nbody, njnt, ncg, nsite, npair, maxdepth, nroot;
  int ntendon, neq;   // fixed tendons, equality/tendon constraints
  int nsensor, nsensordata;
  int nv_damped;       // dofs 0 .. nv_damped - 1 cover every kinematic tree that has a damped joint (implicit-damping Euler factors only those)
  int iterations, ls_iterations, cone, solver;
  float tolerance, meaninertia;
  float mpr_cone;      // half-angle (rad) of the cone of three directions around the previous contact normal from which a pair with a smooth shape restarts its portal
                       // (rsim_step.hip convex_convex, warm-start flag 3); 0: such pairs start cold every substep
  float near_gain;     // ... by this fraction of its duration (0.5: as if it took 1.5 x)
  float near_thresh;   // (m) an env whose closest separated convex pair is nearer than this is dispatched as if it took 1.5 x its duration (rsim_step.hip step_body); 0: off
  float bp_reach;      // broadphase active pair list: bounding-sphere gap (m) up to which a pair is listed; 0: every pair every substep
  float newton_ns, newton_na, newton_ng, newton_ls;
  int newton_refine;   // wide configurations: at most this many polish passes behind the fp32 Newton iteration (fp64 residuals / states / objective / gradient, solve_newton), 0 = none
  float newton_polish_tol;   // the passes end when the scaled fp64 gradient or improvement falls below tolerance x this (1: MuJoCo's own criteria)
  float newton_polish_gate;  // < 0: the polish searches the line from its first pass on (A/B); default 0: plain Newton step first
  int newton_exact;    // 1: a Newton step that leaves every row's state where it was (and no row on the cone) ends the solve -- the objective is quadratic on that piece, the step is its minimiser
  int newton_wide;     // wide (nv > 16) configurations: 0 = neither fp32 rule, 1 = step rule and line-search exit as in the one-tile ones, 2 = line-search exit only   // fp32 stopping rules of the Newton solver (solve_newton): relative / absolute step floor, gradient noise factor
  const int* it;
  const float* ft;
  const float* ft0;            // shared copy of the float table (read for every field no env has overridden)
  unsigned long long fenv;     // bit f set: float-table field f has per-env values (read from ft + env * fstride)
  const float* mesh_vert;
  int fstride;
  const int* lt;            // lane table [LT_COUNT][64]
  int kin_rounds;           // pointer-jumping rounds = ceil(log2(maxdepth))
  int ndynroot, dynroot[RSIM_MAXDYNROOT];  // tree roots that carry dofs (subtree COM needed)
  int maxcondim;
  int io[IO_COUNT];
  int fo[FO_COUNT];
  DCtrl ctrl;
  DTask task;
};

// per-contact record written for the host (floats): dist, pos3, frame9, g1, g2, dim, efc_adr, fn, friction5
#define RSIM_CON_REC 24

struct DBatch {
  int B;
  float *qpos, *qvel, *qacc_ws, *ctrl, *time, *cstate;
  // compat / debug outputs (may be null)
  float *xpos, *xquat, *qM, *qfrc_bias, *qfrc_passive, *qfrc_actuator, *qfrc_constraint, *qacc, *cdof, *rootcom, *contact, *efc_force;
  int *ncon, *nefc, *niter, *diverged;
  float *obs, *reward;   // [B, nobs], [B]
  int* success;          // [B]
  // episode bookkeeping / on-device reset
  int *done, *ep_step, *ep_index, *needs_reset;   // [B]
  int horizon, bank_E, bank_P;
  const float* bank;     // [B][bank_E][nq + bank_P]: ring of pre-drawn resets, slot = episode number % bank_E
  const int* bank_tag;   // [B][bank_E] episode number each slot holds
  int* bank_stale;       // [B] resets that found a slot the host had not refilled in time
  float* term_obs;       // [B, nobs] observation record of the control step that ended an episode (valid where done was reported)
  const int* patch_idx;  // [bank_P] offsets into the env's float table
  float* ft_rw;          // writable alias of the float tables (per-env patches)
  float* ft_base;        // saved defaults of the float tables (domain randomisation), may be null
  // longest-job-first dispatch: workgroup i steps env order[i]; cost[env] = shader-clock ticks the env's wavefront took in the previous launch
  void* cm;              // constant block (Cmem of the kernel configuration) built from the shared float table by k_prepare
  void* cm_env;          // per-env constant blocks [B] (used once a float-table field has per-env values)
  long long cm_stride;   // bytes between the blocks of consecutive envs, 0 = no env has its own block yet
  int* overflow;         // [B] contacts + constraint rows dropped for lack of capacity (null = not counted)
  // capacity tiers (rsim_api.cpp launch()): tier_cur[env] (read-only during a control step) says which configuration steps the env, every pass that
  // commits an env's step writes tier_next[env]; the host swaps the two after the step.  wlist / wcount: the env list a wide pass walks; wlist2 /
  // wcount2: the redo list pass 0 appends to.  tier_con / tier_efc: the NATIVE capacities (the wide pass decides against them when an env may go back)
  const int* tier_cur;
  int* tier_next;
  const int* wlist; const int* wcount;
  int* wlist2; int* wcount2;
  int tier_pass;         // -1 (or tier_cur == null): no tiers; 0: native pass; 1 / 2: wide pass over wlist
  int tier_con, tier_efc;
  unsigned long long* tstat;   // [2] or null: env-steps the wider capacity tier stepped (in whole or from mid-step on) | of these, env-steps handed over / redone in mid-step (rsim_tier_stats)
  int tier_up_con, tier_up_efc;   // an env moves up once a substep came within this many contacts / rows of the native capacity (and back down 2 / 8 below that)
  int* cap_need;         // [B][2] largest number of contacts / constraint rows any substep of the env asked for (RSIM_CAP_NEED), null = not tracked
  int mprc_portal;       // 0: keep only the (exact) separating-direction warm start
  int* task_object;      // [B] PickPlace single-object mode 1: the object of the env's current episode (RSIM_TASK_OBJECT)
  float* sensordata;     // [B][nsensordata] (debug build of the kernel: rsim_step.hip sensor_acc)
  int* bpl;              // [B][5][64] or null: broadphase pair list (rsim_step.hip collision(): sphere centres at build time, packed pair constants, pair indices)
  int* polish;           // [B] RSIM_POLISH: how the fp64 polish of the last substep's solve ended (debug entries only)
  float* qfrc_applied;   // [B][nv] mjData.qfrc_applied: added to the smooth forces by the debug form of the kernel (rsim_forward / step1 / step2 / step); the fused control step ignores it
  float* jg;             // RSIM_JGLOBAL builds: per-env scratch in global memory, [B][jg_stride] floats: the constraint Jacobian, NEFC * (NV + 1), then (RSIM_MGLOBAL) the mass matrix, NV * (NV + 1); null otherwise
  long long jg_stride;   // floats per env = the largest need of the configurations that step this batch (native and wide tier): ONE stride for all of them --
                         // with per-configuration strides the wide pass's env i overlapped the native pass's envs 2 i, 2 i + 1 while both passes were running
  float* mprc;           // [B][npair][12] or null: the separating direction (x, y, z, valid) each candidate pair's last convex narrow-phase run ended on
                         // (warm start of the next substep's run, see convex_convex); zeroed whenever the host writes positions
  const int* order;      // [B] or null (identity)
  unsigned* cost;        // [B] or null
  unsigned long long* prof;  // optional [RP_COUNT] phase-cycle / event accumulators (null = off)
  int prof_env;              // >= 0: only this env adds to the phase accumulators
  // stream groups (rsim_set_stream_groups): this launch covers envs [env0, env0 + nenv) -- its own workgroup count, its own slice of
  // order[] (indices relative to env0) -- while the arrays above stay those of the whole batch; nenv = 0: all B envs
  int env0, nenv;
};

// Newton solver, fp32 stopping rules (0 = rule off; experiments: RSIM_NEWTON_NS / _NA / _NG in the environment when the batch is created).
// Defaults from tools/newton_sweep.py (profiles/r03_d_newton_sweep.txt, r03_e_newton_sweep.txt): a step that moves no acceleration component by more
// than 1e-5 of its value + 1e-5 ends the solve.  On 78 reached states (the Newton-heaviest of a launch included) forces and accelerations against the
// oracle are unchanged to the printed digits (5.7e-4 / 3.3e-4 of the env's largest, same as without the rule), the p99 of the iterations per control step
// drops from 126 to 85 and the maximum from 238 to 140.  The gradient-noise rule (NG) made no difference and stays off.
#ifndef RSIM_NEWTON_LS
#define RSIM_NEWTON_LS 1.0f    // line search: a correction of alpha below this multiple of the step rule's threshold ends it (0: relative 1e-6 only)
#endif
#ifndef RSIM_NEWTON_NS
#ifndef RSIM_MPR_CONE
#define RSIM_MPR_CONE 0.0f   // rad: restart cone of smooth-shape contacts (rsim_step.hip convex_convex, flag 3); 0 = off (the default: see the note there -- +5 % on Lift, but another path to the contact than the cold run the oracle takes)
#endif
#ifndef RSIM_NEAR_THRESH
#define RSIM_NEAR_THRESH 0.002f   // dispatch-order hint for contacts about to start (rsim_step.hip step_body); swept 0.0005 .. 0.05 on three configurations: profiles/r06_v_near_contact_hint.txt
#endif
#ifndef RSIM_NEAR_GAIN
#define RSIM_NEAR_GAIN 0.5f
#endif
#ifndef RSIM_BP_REACH
#define RSIM_BP_REACH 0.04f   // broadphase active pair list (rsim_step.hip collision()): listed up to this bounding-sphere gap; valid while no geom centre moved reach / 2
#endif
#define RSIM_NEWTON_NS 1e-5f
#define RSIM_NEWTON_NA 1e-5f
#define RSIM_NEWTON_NG 0.f
#endif

// profile slots (cycles of s_memtime summed over envs and substeps, then event counters)
enum { RP_LOAD, RP_KIN, RP_COM, RP_CRB, RP_BROAD, RP_NARROW, RP_MAKEC, RP_VEL, RP_CTRL, RP_ACT, RP_SOLVE, RP_EULER, RP_STORE,
       RP_N_SUB, RP_N_CAND, RP_N_CON, RP_N_EFC, RP_N_NEWTON, RP_N_LS, RP_BOXBOX, RP_MPR, RP_PLANE, RP_N_BOXBOX, RP_N_MPR, RP_N_SUPPORT,
       RP_X0, RP_X1, RP_X2, RP_X3, RP_X4, RP_X5, RP_X6, RP_X7, RP_X8, RP_X9, /* ad-hoc sub-phase cycle slots */ RP_COUNT };

// device form of rsim_dr_desc
struct DDr { float density, viscosity, pos, quat, inertia, mass, friction, solref, solimp, frictionloss, damping, armature;
             unsigned long long body_mask, geom_mask, joint_mask; };   // bit set = this body / colliding geom / joint is randomised (rsim_dr_desc: 0 = all)

// flags for the step kernel
enum {
  RF_POSVEL = 1,     // position + velocity stages (always needed)
  RF_CTRL = 2,       // run the built-in controller every substep (writes ctrl)
  RF_SETGOAL = 4,    // first substep: controller.set_goal(action)
  RF_ACTSOLVE = 8,   // actuation + acceleration + constraint solve
  RF_INTEGRATE = 16, // Euler integration, advance time, warm start
  RF_DEBUG = 32,     // write compat/debug arrays
  RF_OBS = 64,       // observation / reward epilogue after the last substep
  RF_EPISODE = 128,  // episode step counter, done flag, on-device reset from the bank
  RF_RESET_ONLY = 256,  // only the envs whose needs_reset flag is set (the reset-observation pass after a control step)
  RF_NOSTORE = 512,     // debug form only: leave the state arrays (qpos .. time, warm start, controller state) as they are -- the refresh of the derived arrays on read
};
