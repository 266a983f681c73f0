/*
 * rsim_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU oracle (serial, fp64) for the robosuite hot path
 *   MjSim.step1/step2/forward  +  Controller.run_controller()  +  env.step() substep loop.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path (robosuite_amd/csrc, HIP) never links or calls it.
 *
 * PARITY STATUS
 *   - controller half (OSC / GRIP): restates the reference's own Python
 *       robosuite/controllers/parts/arm/osc.py:225-495, utils/control_utils.py:7-111,
 *       controllers/parts/controller.py:149-232, controllers/parts/gripper/simple_grip.py:110-186,
 *       models/grippers/panda_gripper.py:43-58, robots/fixed_base_robot.py:121-153
 *     plus the other arm parts generic/joint_pos.py:200-266, generic/joint_tor.py:111-167, generic/joint_vel.py:129-209, the
 *     variable-impedance action layouts (osc.py:243-253, joint_pos.py:204-214) and utils/traj_utils.py:25-155 (LinearInterpolator, incl. the
 *     Euler / slerp orientation interpolator of OSC_POSE, osc.py:277-283, 433-437),
 *     and is PINNED against golden vectors produced by importing that Python (tests/golden/).
 *   - physics half: the reference delegates to the third-party `mujoco` wheel (setup.py:18,
 *     >=3.3.0,<3.10) which is absent here and has no source under /root/reference.  This file
 *     restates MuJoCo's published algorithm ("Computation" chapter of its documentation: kinematics,
 *     CRBA, RNE, soft-constraint model with solref/solimp impedance, elliptic cones, fixed tendons with equality/tendon,
 *     tendon-limit and tendon friction-loss rows and their passive spring / damper, the primal Newton solver with exact line search (and PGS on the dual as a cross-check),
 *     semi-implicit Euler with implicit joint damping) anchored on the reference call sites
 *     utils/binding_utils.py:1089-1107 and environments/base.py:467-521.
 *     ==> physics parity with MuJoCo is UNPINNED (no MuJoCo binary, no golden vectors in the
 *     reference's tests); it is validated by analytic identities and by closed forms of the
 *     documented model and of mechanics in tests/test_oracle.py: textbook mass matrix and bias of the planar two-link arm, gravity bias =
 *     gradient of the potential energy, energy and momentum in free flight, the Coulomb stick threshold, solref -> (b, k), the impedance d(r),
 *     R = (1 - d) / d x diagApprox, aref; resting depths of a contact and of a joint limit from r = a0 (1 - d) / (k d^2); soft friction-loss
 *     creep and breakaway; implicit joint damping; forces on the elliptic cone when slipping; narrow phase against elementary geometry; force /
 *     torque sensors from statics and from momentum rates; primal Newton against dual PGS.  tests/test_hip_known_answers.py holds the KERNEL to the
 *     same closed forms directly.
 *
 * Collision narrow-phase (box-box manifold, MPR for convex pairs) is this project's own design of
 * the same contract (contacts = {pos, frame, dist} for penetrating geom pairs passing MuJoCo's
 * documented filters); it is not a transcription of MuJoCo's functions.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MINVAL 1e-15
#define MAXCON 64
#define MAXEFC 320
#define PI 3.14159265358979323846

enum { JNT_FREE = 0, JNT_BALL = 1, JNT_SLIDE = 2, JNT_HINGE = 3 };
enum { G_PLANE = 0, G_HFIELD, G_SPHERE, G_CAPSULE, G_ELLIPSOID, G_CYLINDER, G_BOX, G_MESH };
enum { C_FRICTION_DOF = 0, C_LIMIT_JOINT = 1, C_CONTACT_FRICTIONLESS = 2, C_CONTACT_ELLIPTIC = 3,
       C_EQUALITY = 4 /* bilateral: always quadratic */, C_LIMIT_TENDON = 5 /* as a joint limit, on a fixed tendon's length */,
       C_FRICTION_TENDON = 6 /* friction loss along a fixed tendon: the cost of C_FRICTION_DOF on the tendon's coefficient row */ };

/* ------------------------------------------------------------------------------------------- */
/* model blob                                                                                  */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
  char name[32];
  uint32_t dtype, count;
  uint64_t offset;
} blob_entry;

typedef struct {
  unsigned char *blob;
  size_t len;
  int nentries;
  blob_entry *entries;
  /* sizes */
  int nq, nv, nu, nbody, njnt, ngeom, nsite, nmesh, npair, nmocap, cone, iterations, solver;
  int nsensor, *sensor_type, *sensor_objid, *sensor_dim;   /* force (0) / torque (1) sensors at a site; anything else reads zero */
  double timestep, density, viscosity, impratio, tolerance, meaninertia;
  double *gravity, *wind;
  int *body_parentid, *body_rootid, *body_weldid, *body_mocapid, *body_jntadr, *body_jntnum, *body_dofadr, *body_dofnum;
  double *body_pos, *body_quat, *body_ipos, *body_iquat, *body_mass, *body_inertia, *body_invweight0, *body_subtreemass;
  int *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid, *jnt_limited;
  double *jnt_pos, *jnt_axis, *jnt_range, *jnt_margin, *jnt_solref, *jnt_solimp, *jnt_stiffness, *qpos0, *qpos_spring;
  int *dof_bodyid, *dof_jntid, *dof_parentid;
  double *dof_armature, *dof_damping, *dof_frictionloss, *dof_solref, *dof_solimp, *dof_invweight0, *dof_M0;
  int *geom_type, *geom_bodyid, *geom_contype, *geom_conaffinity, *geom_condim, *geom_priority, *geom_dataid;
  double *geom_size, *geom_pos, *geom_quat, *geom_friction, *geom_solref, *geom_solimp, *geom_solmix, *geom_margin, *geom_gap, *geom_rbound, *geom_rcenter;
  int *mesh_vertadr, *mesh_vertnum;
  double *mesh_vert;
  int *site_bodyid;
  double *site_pos, *site_quat;
  int *actuator_trnid, *actuator_biastype, *actuator_ctrllimited, *actuator_forcelimited;
  double *actuator_gear, *actuator_gainprm, *actuator_biasprm, *actuator_ctrlrange, *actuator_forcerange;
  int *pair_geom1, *pair_geom2;
  /* fixed tendons (length = sum coef q) and equality/tendon constraints; absent in models compiled before they existed */
  int ntendon, neq;
  int *tendon_adr, *tendon_num, *wrap_objid, *tendon_limited, *eq_obj1id;
  double *wrap_prm, *tendon_range, *tendon_margin, *tendon_solref_lim, *tendon_solimp_lim, *tendon_length0, *tendon_invweight0, *eq_data, *eq_solref, *eq_solimp;
  double *tendon_stiffness, *tendon_damping, *tendon_lengthspring;   /* spring-damper on the tendon length (may be absent) */
  double *tendon_frictionloss, *tendon_solref_fri, *tendon_solimp_fri; /* dry friction along the tendon (may be absent) */
} rso_model;

static void *blob_find(rso_model *m, const char *name, int *count) {
  for (int i = 0; i < m->nentries; i++)
    if (strncmp(m->entries[i].name, name, 32) == 0) {
      if (count) *count = (int)m->entries[i].count;
      return m->blob + m->entries[i].offset;
    }
  if (count) *count = 0;
  return NULL;
}
static int blob_i(rso_model *m, const char *n) { int *p = (int *)blob_find(m, n, NULL); return p ? p[0] : 0; }
static double blob_d(rso_model *m, const char *n) { double *p = (double *)blob_find(m, n, NULL); return p ? p[0] : 0.0; }

rso_model *rso_model_create(const void *blob, size_t len) {
  if (len < 16 || memcmp(blob, "RSIMMDL1", 8) != 0) return NULL;
  rso_model *m = (rso_model *)calloc(1, sizeof(rso_model));
  m->blob = (unsigned char *)malloc(len);
  memcpy(m->blob, blob, len);
  m->len = len;
  m->nentries = (int)*(uint32_t *)(m->blob + 8);
  m->entries = (blob_entry *)(m->blob + 16);
#define GI(f) m->f = blob_i(m, #f)
#define GD(f) m->f = blob_d(m, #f)
#define PI_(f) m->f = (int *)blob_find(m, #f, NULL)
#define PD_(f) m->f = (double *)blob_find(m, #f, NULL)
  GI(nq); GI(nv); GI(nu); GI(nbody); GI(njnt); GI(ngeom); GI(nsite); GI(nmesh); GI(npair); GI(nmocap); GI(cone); GI(iterations); GI(solver);
  GD(timestep); GD(density); GD(viscosity); GD(impratio); GD(tolerance);
  PD_(gravity); PD_(wind);
  PI_(body_parentid); PI_(body_rootid); PI_(body_weldid); PI_(body_mocapid); PI_(body_jntadr); PI_(body_jntnum); PI_(body_dofadr); PI_(body_dofnum);
  PD_(body_pos); PD_(body_quat); PD_(body_ipos); PD_(body_iquat); PD_(body_mass); PD_(body_inertia); PD_(body_invweight0); PD_(body_subtreemass);
  PI_(jnt_type); PI_(jnt_qposadr); PI_(jnt_dofadr); PI_(jnt_bodyid); PI_(jnt_limited);
  PD_(jnt_pos); PD_(jnt_axis); PD_(jnt_range); PD_(jnt_margin); PD_(jnt_solref); PD_(jnt_solimp); PD_(jnt_stiffness); PD_(qpos0); PD_(qpos_spring);
  PI_(dof_bodyid); PI_(dof_jntid); PI_(dof_parentid);
  PD_(dof_armature); PD_(dof_damping); PD_(dof_frictionloss); PD_(dof_solref); PD_(dof_solimp); PD_(dof_invweight0); PD_(dof_M0);
  PI_(geom_type); PI_(geom_bodyid); PI_(geom_contype); PI_(geom_conaffinity); PI_(geom_condim); PI_(geom_priority); PI_(geom_dataid);
  PD_(geom_size); PD_(geom_pos); PD_(geom_quat); PD_(geom_friction); PD_(geom_solref); PD_(geom_solimp); PD_(geom_solmix); PD_(geom_margin); PD_(geom_gap); PD_(geom_rbound); PD_(geom_rcenter);
  PI_(mesh_vertadr); PI_(mesh_vertnum); PD_(mesh_vert);
  PI_(site_bodyid); PD_(site_pos); PD_(site_quat);
  PI_(actuator_trnid); PI_(actuator_biastype); PI_(actuator_ctrllimited); PI_(actuator_forcelimited);
  PD_(actuator_gear); PD_(actuator_gainprm); PD_(actuator_biasprm); PD_(actuator_ctrlrange); PD_(actuator_forcerange);
  PI_(pair_geom1); PI_(pair_geom2);
  GI(ntendon); GI(neq); GI(nsensor);
  PI_(sensor_type); PI_(sensor_objid); PI_(sensor_dim);
  PI_(tendon_adr); PI_(tendon_num); PI_(wrap_objid); PI_(tendon_limited); PI_(eq_obj1id);
  PD_(wrap_prm); PD_(tendon_range); PD_(tendon_margin); PD_(tendon_solref_lim); PD_(tendon_solimp_lim); PD_(tendon_length0); PD_(tendon_invweight0);
  PD_(eq_data); PD_(eq_solref); PD_(eq_solimp);
  PD_(tendon_stiffness); PD_(tendon_damping); PD_(tendon_lengthspring); PD_(tendon_frictionloss); PD_(tendon_solref_fri); PD_(tendon_solimp_fri);
  /* mean diagonal inertia at qpos0 (MuJoCo stat.meaninertia [3P]) */
  double s = 0;
  for (int i = 0; i < m->nv; i++) s += m->dof_M0[i];
  m->meaninertia = m->nv > 0 ? s / m->nv : 1.0;
  if (m->meaninertia < MINVAL) m->meaninertia = 1.0;
  return m;
}
void rso_model_free(rso_model *m) {
  if (!m) return;
  free(m->blob);
  free(m);
}
/* raw access to a model array (tests mutate model constants for domain randomisation) */
void *rso_model_field(rso_model *m, const char *name, int *count, int *dtype) {
  for (int i = 0; i < m->nentries; i++)
    if (strncmp(m->entries[i].name, name, 32) == 0) {
      *count = (int)m->entries[i].count;
      *dtype = (int)m->entries[i].dtype;
      return m->blob + m->entries[i].offset;
    }
  *count = 0;
  return NULL;
}

/* ------------------------------------------------------------------------------------------- */
/* per-env data                                                                                */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
  double dist, pos[3], frame[9], friction[5], solref[2], solimp[5], mu, includemargin;
  int dim, geom1, geom2, efc_address;
} rso_contact;

typedef struct {
  rso_model *m;
  double time;
  double *qpos, *qvel, *qacc, *qacc_warmstart, *ctrl, *qfrc_applied, *mocap_pos, *mocap_quat;
  double *xpos, *xquat, *xmat, *xipos, *ximat, *geom_xpos, *geom_xmat, *site_xpos, *site_xmat, *xanchor, *xaxis;
  double *subtree_com, *cinert, *crb, *cdof, *cdof_dot, *cvel, *cacc, *cfrc;
  double *cfrc_int, *cfrc_ext, *sensordata;   /* sensor_acc() */
  int nsensordata;
  double *qM, *qL, *qLD; /* dense M, chol(M), chol(M + h*D) */
  double *qfrc_bias, *qfrc_passive, *qfrc_actuator, *qfrc_smooth, *qacc_smooth, *qfrc_constraint, *actuator_force;
  int ncon, nefc;
  rso_contact contact[MAXCON];
  int efc_type[MAXEFC], efc_id[MAXEFC];
  double *efc_J; /* MAXEFC x nv */
  double efc_pos[MAXEFC], efc_margin[MAXEFC], efc_D[MAXEFC], efc_R[MAXEFC], efc_aref[MAXEFC], efc_vel[MAXEFC];
  double efc_frictionloss[MAXEFC], efc_force[MAXEFC], efc_KBIP[MAXEFC * 4], efc_diagApprox[MAXEFC], efc_b[MAXEFC];
  double *efc_AR; /* MAXEFC x MAXEFC */
  double *efc_MinvJT; /* nv x MAXEFC */
  int solver_iter;
  int round_rows;  /* test infrastructure (rso_set_round_rows): the constraint solve sees its INPUTS rounded to float32 -- what a kernel that stores them in fp32 is handed at best */
  int con_overflow;
} rso_data;

static double *dalloc(size_t n) { return (double *)calloc(n ? n : 1, sizeof(double)); }

void rso_reset(rso_data *d);

rso_data *rso_data_create(rso_model *m) {
  rso_data *d = (rso_data *)calloc(1, sizeof(rso_data));
  d->m = m;
  int nv = m->nv, nb = m->nbody;
  d->qpos = dalloc(m->nq); d->qvel = dalloc(nv); d->qacc = dalloc(nv); d->qacc_warmstart = dalloc(nv);
  d->ctrl = dalloc(m->nu); d->qfrc_applied = dalloc(nv); d->mocap_pos = dalloc(3 * m->nmocap); d->mocap_quat = dalloc(4 * m->nmocap);
  d->xpos = dalloc(3 * nb); d->xquat = dalloc(4 * nb); d->xmat = dalloc(9 * nb); d->xipos = dalloc(3 * nb); d->ximat = dalloc(9 * nb);
  d->geom_xpos = dalloc(3 * m->ngeom); d->geom_xmat = dalloc(9 * m->ngeom); d->site_xpos = dalloc(3 * m->nsite); d->site_xmat = dalloc(9 * m->nsite);
  d->xanchor = dalloc(3 * m->njnt); d->xaxis = dalloc(3 * m->njnt);
  d->subtree_com = dalloc(3 * nb); d->cinert = dalloc(10 * nb); d->crb = dalloc(10 * nb); d->cdof = dalloc(6 * nv); d->cdof_dot = dalloc(6 * nv);
  d->cvel = dalloc(6 * nb); d->cacc = dalloc(6 * nb); d->cfrc = dalloc(6 * nb);
  d->cfrc_int = dalloc(6 * nb); d->cfrc_ext = dalloc(6 * nb);
  for (int i = 0; i < m->nsensor; i++) d->nsensordata += m->sensor_dim[i];
  d->sensordata = dalloc(d->nsensordata);
  d->qM = dalloc(nv * nv); d->qL = dalloc(nv * nv); d->qLD = dalloc(nv * nv);
  d->qfrc_bias = dalloc(nv); d->qfrc_passive = dalloc(nv); d->qfrc_actuator = dalloc(nv); d->qfrc_smooth = dalloc(nv); d->qacc_smooth = dalloc(nv);
  d->qfrc_constraint = dalloc(nv); d->actuator_force = dalloc(m->nu);
  d->efc_J = dalloc((size_t)MAXEFC * nv); d->efc_AR = dalloc((size_t)MAXEFC * MAXEFC); d->efc_MinvJT = dalloc((size_t)MAXEFC * nv);
  rso_reset(d);
  return d;
}
void rso_data_free(rso_data *d) {
  if (!d) return;
  double **p[] = {&d->qpos, &d->qvel, &d->qacc, &d->qacc_warmstart, &d->ctrl, &d->qfrc_applied, &d->mocap_pos, &d->mocap_quat, &d->xpos, &d->xquat, &d->xmat,
                  &d->xipos, &d->ximat, &d->geom_xpos, &d->geom_xmat, &d->site_xpos, &d->site_xmat, &d->xanchor, &d->xaxis, &d->subtree_com, &d->cinert,
                  &d->crb, &d->cdof, &d->cdof_dot, &d->cvel, &d->cacc, &d->cfrc, &d->qM, &d->qL, &d->qLD, &d->qfrc_bias, &d->qfrc_passive, &d->qfrc_actuator,
                  &d->qfrc_smooth, &d->qacc_smooth, &d->qfrc_constraint, &d->actuator_force, &d->efc_J, &d->efc_AR, &d->efc_MinvJT, &d->cfrc_int, &d->cfrc_ext,
                  &d->sensordata};
  for (size_t i = 0; i < sizeof(p) / sizeof(p[0]); i++) free(*p[i]);
  free(d);
}

/* mj_resetData [3P]: qpos = qpos0, everything else zero; mocap pose from body pos/quat */
void rso_reset(rso_data *d) {
  rso_model *m = d->m;
  memcpy(d->qpos, m->qpos0, sizeof(double) * m->nq);
  memset(d->qvel, 0, sizeof(double) * m->nv);
  memset(d->qacc, 0, sizeof(double) * m->nv);
  memset(d->qacc_warmstart, 0, sizeof(double) * m->nv);
  memset(d->ctrl, 0, sizeof(double) * m->nu);
  memset(d->qfrc_applied, 0, sizeof(double) * m->nv);
  for (int b = 0; b < m->nbody; b++)
    if (m->body_mocapid[b] >= 0) {
      memcpy(d->mocap_pos + 3 * m->body_mocapid[b], m->body_pos + 3 * b, 3 * sizeof(double));
      memcpy(d->mocap_quat + 4 * m->body_mocapid[b], m->body_quat + 4 * b, 4 * sizeof(double));
    }
  d->time = 0;
  d->ncon = d->nefc = 0;
}

/* ------------------------------------------------------------------------------------------- */
/* small math                                                                                  */
/* ------------------------------------------------------------------------------------------- */
static inline double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(double *r, const double *a, const double *b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline double norm3(const double *a) { return sqrt(dot3(a, a)); }
static inline double normalize3(double *a) {
  double n = norm3(a);
  if (n < MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; return 0; }
  a[0] /= n; a[1] /= n; a[2] /= n;
  return n;
}
static inline void quat_mul(double *r, const double *a, const double *b) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static inline void quat_norm(double *q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static inline void quat2mat(double *R, const double *q) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = w * w + x * x - y * y - z * z; R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = w * w - x * x + y * y - z * z; R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = w * w - x * x - y * y + z * z;
}
static inline void mat_vec3(double *r, const double *R, const double *v) {
  double x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2], y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2], z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void matT_vec3(double *r, const double *R, const double *v) {
  double x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2], y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2], z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void axisangle_quat(double *q, const double *axis, double angle) {
  double s = sin(0.5 * angle);
  q[0] = cos(0.5 * angle); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
/* spatial (6D) helpers: vectors are [angular(3); linear(3)] about the tree's subtree COM */
static inline void cross_motion(double *r, const double *v, const double *s) {
  double t[3];
  cross3(r, v, s);          /* w x s_ang */
  cross3(r + 3, v, s + 3);  /* w x s_lin */
  cross3(t, v + 3, s);      /* v x s_ang */
  r[3] += t[0]; r[4] += t[1]; r[5] += t[2];
}
static inline void cross_force(double *r, const double *v, const double *f) {
  double t[3];
  cross3(r, v, f);          /* w x f_ang */
  cross3(t, v + 3, f + 3);  /* v x f_lin */
  r[0] += t[0]; r[1] += t[1]; r[2] += t[2];
  cross3(r + 3, v, f + 3);  /* w x f_lin */
}
/* 10-vector inertia {Ixx,Iyy,Izz,Ixy,Ixz,Iyz, m*cx,m*cy,m*cz, m} times motion vector -> force vector */
static inline void mul_inert_vec(double *r, const double *I, const double *v) {
  r[0] = I[0] * v[0] + I[3] * v[1] + I[4] * v[2] - I[8] * v[4] + I[7] * v[5];
  r[1] = I[3] * v[0] + I[1] * v[1] + I[5] * v[2] + I[8] * v[3] - I[6] * v[5];
  r[2] = I[4] * v[0] + I[5] * v[1] + I[2] * v[2] - I[7] * v[3] + I[6] * v[4];
  r[3] = I[8] * v[1] - I[7] * v[2] + I[9] * v[3];
  r[4] = I[6] * v[2] - I[8] * v[0] + I[9] * v[4];
  r[5] = I[7] * v[0] - I[6] * v[1] + I[9] * v[5];
}
static inline double dot6(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5]; }

/* dense Cholesky (lower) in place on an n x n copy; returns 0 ok */
static int chol_factor(double *L, const double *A, int n) {
  memcpy(L, A, sizeof(double) * n * n);
  for (int j = 0; j < n; j++) {
    double s = L[j * n + j];
    for (int k = 0; k < j; k++) s -= L[j * n + k] * L[j * n + k];
    if (s < MINVAL) s = MINVAL;
    double dj = sqrt(s);
    L[j * n + j] = dj;
    for (int i = j + 1; i < n; i++) {
      double t = L[i * n + j];
      for (int k = 0; k < j; k++) t -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = t / dj;
    }
  }
  return 0;
}
static void chol_solve(const double *L, double *x, int n) {
  for (int i = 0; i < n; i++) {
    double s = x[i];
    for (int k = 0; k < i; k++) s -= L[i * n + k] * x[k];
    x[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = x[i];
    for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k];
    x[i] = s / L[i * n + i];
  }
}

/* ------------------------------------------------------------------------------------------- */
/* kinematics (mj_kinematics, mj_comPos [3P])                                                  */
/* ------------------------------------------------------------------------------------------- */
static void kinematics(rso_data *d) {
  rso_model *m = d->m;
  d->xpos[0] = d->xpos[1] = d->xpos[2] = 0;
  d->xquat[0] = 1; d->xquat[1] = d->xquat[2] = d->xquat[3] = 0;
  quat2mat(d->xmat, d->xquat);
  for (int b = 1; b < m->nbody; b++) {
    double pos[3], quat[4], R[9];
    int p = m->body_parentid[b], jadr = m->body_jntadr[b], jnum = m->body_jntnum[b];
    if (m->body_mocapid[b] >= 0) {
      memcpy(pos, d->mocap_pos + 3 * m->body_mocapid[b], sizeof(pos));
      memcpy(quat, d->mocap_quat + 4 * m->body_mocapid[b], sizeof(quat));
      quat_norm(quat);
    } else if (jnum == 1 && m->jnt_type[jadr] == JNT_FREE) {
      int a = m->jnt_qposadr[jadr];
      memcpy(pos, d->qpos + a, sizeof(pos));
      memcpy(quat, d->qpos + a + 3, sizeof(quat));
      quat_norm(quat);
      memcpy(d->xanchor + 3 * jadr, pos, sizeof(pos));
      d->xaxis[3 * jadr] = 0; d->xaxis[3 * jadr + 1] = 0; d->xaxis[3 * jadr + 2] = 1;
    } else {
      mat_vec3(pos, d->xmat + 9 * p, m->body_pos + 3 * b);
      for (int k = 0; k < 3; k++) pos[k] += d->xpos[3 * p + k];
      quat_mul(quat, d->xquat + 4 * p, m->body_quat + 4 * b);
      for (int j = jadr; j < jadr + jnum; j++) {
        quat2mat(R, quat);
        double anchor[3], axis[3];
        mat_vec3(anchor, R, m->jnt_pos + 3 * j);
        for (int k = 0; k < 3; k++) anchor[k] += pos[k];
        mat_vec3(axis, R, m->jnt_axis + 3 * j);
        memcpy(d->xanchor + 3 * j, anchor, sizeof(anchor));
        memcpy(d->xaxis + 3 * j, axis, sizeof(axis));
        int a = m->jnt_qposadr[j];
        if (m->jnt_type[j] == JNT_SLIDE) {
          double q = d->qpos[a] - m->qpos0[a];
          for (int k = 0; k < 3; k++) pos[k] += axis[k] * q;
        } else {
          double ql[4], qn[4], off[3];
          if (m->jnt_type[j] == JNT_HINGE) axisangle_quat(ql, m->jnt_axis + 3 * j, d->qpos[a] - m->qpos0[a]);
          else { memcpy(ql, d->qpos + a, sizeof(ql)); quat_norm(ql); }
          quat_mul(qn, quat, ql);
          memcpy(quat, qn, sizeof(quat));
          quat2mat(R, quat);
          mat_vec3(off, R, m->jnt_pos + 3 * j);
          for (int k = 0; k < 3; k++) pos[k] = anchor[k] - off[k];
        }
      }
      quat_norm(quat);
    }
    memcpy(d->xpos + 3 * b, pos, sizeof(pos));
    memcpy(d->xquat + 4 * b, quat, sizeof(quat));
    quat2mat(d->xmat + 9 * b, quat);
  }
  for (int b = 0; b < m->nbody; b++) {
    double q[4];
    mat_vec3(d->xipos + 3 * b, d->xmat + 9 * b, m->body_ipos + 3 * b);
    for (int k = 0; k < 3; k++) d->xipos[3 * b + k] += d->xpos[3 * b + k];
    quat_mul(q, d->xquat + 4 * b, m->body_iquat + 4 * b);
    quat2mat(d->ximat + 9 * b, q);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_bodyid[g];
    double q[4];
    mat_vec3(d->geom_xpos + 3 * g, d->xmat + 9 * b, m->geom_pos + 3 * g);
    for (int k = 0; k < 3; k++) d->geom_xpos[3 * g + k] += d->xpos[3 * b + k];
    quat_mul(q, d->xquat + 4 * b, m->geom_quat + 4 * g);
    quat_norm(q);
    quat2mat(d->geom_xmat + 9 * g, q);
  }
  for (int s = 0; s < m->nsite; s++) {
    int b = m->site_bodyid[s];
    double q[4];
    mat_vec3(d->site_xpos + 3 * s, d->xmat + 9 * b, m->site_pos + 3 * s);
    for (int k = 0; k < 3; k++) d->site_xpos[3 * s + k] += d->xpos[3 * b + k];
    quat_mul(q, d->xquat + 4 * b, m->site_quat + 4 * s);
    quat_norm(q);
    quat2mat(d->site_xmat + 9 * s, q);
  }
}

static void com_pos(rso_data *d) {
  rso_model *m = d->m;
  int nb = m->nbody;
  /* subtree COM */
  double *acc = dalloc(3 * nb);
  for (int b = 0; b < nb; b++)
    for (int k = 0; k < 3; k++) acc[3 * b + k] = m->body_mass[b] * d->xipos[3 * b + k];
  for (int b = nb - 1; b > 0; b--)
    for (int k = 0; k < 3; k++) acc[3 * m->body_parentid[b] + k] += acc[3 * b + k];
  for (int b = 0; b < nb; b++) {
    if (m->body_subtreemass[b] < MINVAL) memcpy(d->subtree_com + 3 * b, d->xipos + 3 * b, 3 * sizeof(double));
    else
      for (int k = 0; k < 3; k++) d->subtree_com[3 * b + k] = acc[3 * b + k] / m->body_subtreemass[b];
  }
  free(acc);
  /* cinert: body inertia about the tree COM, world orientation */
  for (int b = 0; b < nb; b++) {
    const double *R = d->ximat + 9 * b, *I = m->body_inertia + 3 * b, *c = d->subtree_com + 3 * m->body_rootid[b];
    double mass = m->body_mass[b], off[3], *ci = d->cinert + 10 * b;
    for (int k = 0; k < 3; k++) off[k] = d->xipos[3 * b + k] - c[k];
    double Iw[9];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Iw[3 * i + j] = R[3 * i] * I[0] * R[3 * j] + R[3 * i + 1] * I[1] * R[3 * j + 1] + R[3 * i + 2] * I[2] * R[3 * j + 2];
    double d2 = dot3(off, off);
    ci[0] = Iw[0] + mass * (d2 - off[0] * off[0]);
    ci[1] = Iw[4] + mass * (d2 - off[1] * off[1]);
    ci[2] = Iw[8] + mass * (d2 - off[2] * off[2]);
    ci[3] = Iw[1] - mass * off[0] * off[1];
    ci[4] = Iw[2] - mass * off[0] * off[2];
    ci[5] = Iw[5] - mass * off[1] * off[2];
    ci[6] = mass * off[0]; ci[7] = mass * off[1]; ci[8] = mass * off[2]; ci[9] = mass;
  }
  /* cdof */
  for (int j = 0; j < m->njnt; j++) {
    int b = m->jnt_bodyid[j], da = m->jnt_dofadr[j];
    const double *c = d->subtree_com + 3 * m->body_rootid[b];
    double off[3];
    for (int k = 0; k < 3; k++) off[k] = c[k] - d->xanchor[3 * j + k];
    if (m->jnt_type[j] == JNT_FREE) {
      for (int k = 0; k < 3; k++) {
        double *cd = d->cdof + 6 * (da + k);
        memset(cd, 0, 6 * sizeof(double));
        cd[3 + k] = 1;
      }
      for (int k = 0; k < 3; k++) {
        double *cd = d->cdof + 6 * (da + 3 + k), ax[3] = {d->xmat[9 * b + k], d->xmat[9 * b + 3 + k], d->xmat[9 * b + 6 + k]};
        memcpy(cd, ax, sizeof(ax));
        cross3(cd + 3, ax, off);
      }
    } else if (m->jnt_type[j] == JNT_BALL) {
      for (int k = 0; k < 3; k++) {
        double *cd = d->cdof + 6 * (da + k), ax[3] = {d->xmat[9 * b + k], d->xmat[9 * b + 3 + k], d->xmat[9 * b + 6 + k]};
        memcpy(cd, ax, sizeof(ax));
        cross3(cd + 3, ax, off);
      }
    } else if (m->jnt_type[j] == JNT_SLIDE) {
      double *cd = d->cdof + 6 * da;
      cd[0] = cd[1] = cd[2] = 0;
      memcpy(cd + 3, d->xaxis + 3 * j, 3 * sizeof(double));
    } else {
      double *cd = d->cdof + 6 * da;
      memcpy(cd, d->xaxis + 3 * j, 3 * sizeof(double));
      cross3(cd + 3, d->xaxis + 3 * j, off);
    }
  }
}

/* composite rigid body algorithm -> dense M, then Cholesky (mj_crb + mj_factorM [3P]) */
static void crb(rso_data *d) {
  rso_model *m = d->m;
  int nv = m->nv;
  memcpy(d->crb, d->cinert, sizeof(double) * 10 * m->nbody);
  for (int b = m->nbody - 1; b > 0; b--)
    if (m->body_parentid[b] > 0)
      for (int k = 0; k < 10; k++) d->crb[10 * m->body_parentid[b] + k] += d->crb[10 * b + k];
  memset(d->qM, 0, sizeof(double) * nv * nv);
  for (int i = 0; i < nv; i++) {
    double buf[6];
    mul_inert_vec(buf, d->crb + 10 * m->dof_bodyid[i], d->cdof + 6 * i);
    for (int j = i; j >= 0; j = m->dof_parentid[j]) {
      double v = dot6(d->cdof + 6 * j, buf);
      d->qM[i * nv + j] = d->qM[j * nv + i] = v;
    }
    d->qM[i * nv + i] += m->dof_armature[i];
  }
  chol_factor(d->qL, d->qM, nv);
}

/* Jacobian of a world point attached to `body` (mj_jac [3P]): jacp, jacr are 3 x nv or NULL */
static void jac_point(rso_data *d, double *jacp, double *jacr, const double *point, int body) {
  rso_model *m = d->m;
  int nv = m->nv;
  if (jacp) memset(jacp, 0, sizeof(double) * 3 * nv);
  if (jacr) memset(jacr, 0, sizeof(double) * 3 * nv);
  while (body > 0 && m->body_dofnum[body] == 0) body = m->body_parentid[body];
  if (body <= 0) return;
  double off[3];
  const double *c = d->subtree_com + 3 * m->body_rootid[body];
  for (int k = 0; k < 3; k++) off[k] = point[k] - c[k];
  for (int i = m->body_dofadr[body] + m->body_dofnum[body] - 1; i >= 0; i = m->dof_parentid[i]) {
    const double *cd = d->cdof + 6 * i;
    if (jacr)
      for (int k = 0; k < 3; k++) jacr[k * nv + i] = cd[k];
    if (jacp) {
      double t[3];
      cross3(t, cd, off);
      for (int k = 0; k < 3; k++) jacp[k * nv + i] = cd[3 + k] + t[k];
    }
  }
}

/* ------------------------------------------------------------------------------------------- */
/* velocity stage (mj_comVel, mj_passive, mj_rne [3P])                                         */
/* ------------------------------------------------------------------------------------------- */
static void com_vel(rso_data *d) {
  rso_model *m = d->m;
  memset(d->cvel, 0, 6 * sizeof(double));
  for (int b = 1; b < m->nbody; b++) {
    double cv[6];
    memcpy(cv, d->cvel + 6 * m->body_parentid[b], sizeof(cv));
    for (int j = m->body_jntadr[b]; j < m->body_jntadr[b] + m->body_jntnum[b]; j++) {
      int da = m->jnt_dofadr[j];
      if (m->jnt_type[j] == JNT_FREE) {
        for (int k = 0; k < 3; k++) {
          memset(d->cdof_dot + 6 * (da + k), 0, 6 * sizeof(double));
          for (int r = 0; r < 6; r++) cv[r] += d->cdof[6 * (da + k) + r] * d->qvel[da + k];
        }
        for (int k = 3; k < 6; k++) cross_motion(d->cdof_dot + 6 * (da + k), cv, d->cdof + 6 * (da + k));
        for (int k = 3; k < 6; k++)
          for (int r = 0; r < 6; r++) cv[r] += d->cdof[6 * (da + k) + r] * d->qvel[da + k];
      } else if (m->jnt_type[j] == JNT_BALL) {
        for (int k = 0; k < 3; k++) cross_motion(d->cdof_dot + 6 * (da + k), cv, d->cdof + 6 * (da + k));
        for (int k = 0; k < 3; k++)
          for (int r = 0; r < 6; r++) cv[r] += d->cdof[6 * (da + k) + r] * d->qvel[da + k];
      } else {
        cross_motion(d->cdof_dot + 6 * da, cv, d->cdof + 6 * da);
        for (int r = 0; r < 6; r++) cv[r] += d->cdof[6 * da + r] * d->qvel[da];
      }
    }
    memcpy(d->cvel + 6 * b, cv, sizeof(cv));
  }
}

static void apply_ft(rso_data *d, const double *force, const double *torque, const double *point, int body, double *qfrc) {
  int nv = d->m->nv;
  double *jp = dalloc(3 * nv), *jr = dalloc(3 * nv);
  jac_point(d, jp, jr, point, body);
  for (int i = 0; i < nv; i++)
    for (int k = 0; k < 3; k++) qfrc[i] += jp[k * nv + i] * force[k] + jr[k * nv + i] * torque[k];
  free(jp); free(jr);
}

static void passive(rso_data *d) {
  rso_model *m = d->m;
  for (int i = 0; i < m->nv; i++) d->qfrc_passive[i] = -m->dof_damping[i] * d->qvel[i];
  /* joint springs */
  for (int j = 0; j < m->njnt; j++)
    if (m->jnt_stiffness[j] > 0 && (m->jnt_type[j] == JNT_HINGE || m->jnt_type[j] == JNT_SLIDE))
      d->qfrc_passive[m->jnt_dofadr[j]] -= m->jnt_stiffness[j] * (d->qpos[m->jnt_qposadr[j]] - m->qpos_spring[m->jnt_qposadr[j]]);
  /* spring-damper on fixed-tendon lengths (mj_passive [3P]: deadband [lengthspring0, lengthspring1], damping on the length rate) */
  if (m->tendon_stiffness)
    for (int t = 0; t < m->ntendon; t++) {
      double k = m->tendon_stiffness[t], b = m->tendon_damping ? m->tendon_damping[t] : 0;
      if (k <= 0 && b <= 0) continue;
      double len = 0, vel = 0, frc = 0;
      for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) {
        int j = m->wrap_objid[w];
        len += m->wrap_prm[w] * d->qpos[m->jnt_qposadr[j]];
        vel += m->wrap_prm[w] * d->qvel[m->jnt_dofadr[j]];
      }
      double lo = m->tendon_lengthspring[2 * t], hi = m->tendon_lengthspring[2 * t + 1];
      if (len > hi) frc = k * (hi - len); else if (len < lo) frc = k * (lo - len);
      frc -= b * vel;
      for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) d->qfrc_passive[m->jnt_dofadr[m->wrap_objid[w]]] += m->wrap_prm[w] * frc;
    }
  /* inertia-box fluid model (density/viscosity of the medium, base.xml:4) */
  if (m->density > 0 || m->viscosity > 0) {
    for (int b = 1; b < m->nbody; b++) {
      double mass = m->body_mass[b];
      if (mass < MINVAL) continue;
      const double *I = m->body_inertia + 3 * b, *R = d->ximat + 9 * b;
      double box[3], lvel[6], lfrc[6] = {0, 0, 0, 0, 0, 0}, gv[6], off[3], t[3];
      box[0] = sqrt(fmax(MINVAL, I[1] + I[2] - I[0]) / mass * 6.0);
      box[1] = sqrt(fmax(MINVAL, I[0] + I[2] - I[1]) / mass * 6.0);
      box[2] = sqrt(fmax(MINVAL, I[0] + I[1] - I[2]) / mass * 6.0);
      /* velocity at the body COM, expressed in the body inertial frame */
      memcpy(gv, d->cvel + 6 * b, sizeof(gv));
      for (int k = 0; k < 3; k++) off[k] = d->xipos[3 * b + k] - d->subtree_com[3 * m->body_rootid[b] + k];
      cross3(t, gv, off);
      for (int k = 0; k < 3; k++) gv[3 + k] += t[k] - m->wind[k];
      matT_vec3(lvel, R, gv);
      matT_vec3(lvel + 3, R, gv + 3);
      if (m->viscosity > 0) {
        double diam = (box[0] + box[1] + box[2]) / 3.0;
        for (int k = 0; k < 3; k++) lfrc[k] = -PI * diam * diam * diam * m->viscosity * lvel[k];
        for (int k = 0; k < 3; k++) lfrc[3 + k] = -3.0 * PI * diam * m->viscosity * lvel[3 + k];
      }
      if (m->density > 0) {
        lfrc[3] -= 0.5 * m->density * box[1] * box[2] * fabs(lvel[3]) * lvel[3];
        lfrc[4] -= 0.5 * m->density * box[0] * box[2] * fabs(lvel[4]) * lvel[4];
        lfrc[5] -= 0.5 * m->density * box[0] * box[1] * fabs(lvel[5]) * lvel[5];
        lfrc[0] -= m->density * box[0] * (pow(box[1], 4) + pow(box[2], 4)) * fabs(lvel[0]) * lvel[0] / 64.0;
        lfrc[1] -= m->density * box[1] * (pow(box[0], 4) + pow(box[2], 4)) * fabs(lvel[1]) * lvel[1] / 64.0;
        lfrc[2] -= m->density * box[2] * (pow(box[0], 4) + pow(box[1], 4)) * fabs(lvel[2]) * lvel[2] / 64.0;
      }
      double gt[3], gf[3];
      mat_vec3(gt, R, lfrc);
      mat_vec3(gf, R, lfrc + 3);
      apply_ft(d, gf, gt, d->xipos + 3 * b, b, d->qfrc_passive);
    }
  }
}

/* RNE with zero acceleration -> qfrc_bias (Coriolis/centrifugal + gravity) */
static void rne_bias(rso_data *d) {
  rso_model *m = d->m;
  memset(d->cacc, 0, 6 * sizeof(double));
  for (int k = 0; k < 3; k++) d->cacc[3 + k] = -m->gravity[k];
  memset(d->cfrc, 0, 6 * sizeof(double));
  for (int b = 1; b < m->nbody; b++) {
    double *ca = d->cacc + 6 * b, t1[6], t2[6];
    memcpy(ca, d->cacc + 6 * m->body_parentid[b], 6 * sizeof(double));
    for (int i = m->body_dofadr[b]; i >= 0 && i < m->body_dofadr[b] + m->body_dofnum[b]; i++)
      for (int r = 0; r < 6; r++) ca[r] += d->cdof_dot[6 * i + r] * d->qvel[i];
    mul_inert_vec(t1, d->cinert + 10 * b, ca);
    mul_inert_vec(t2, d->cinert + 10 * b, d->cvel + 6 * b);
    cross_force(d->cfrc + 6 * b, d->cvel + 6 * b, t2);
    for (int r = 0; r < 6; r++) d->cfrc[6 * b + r] += t1[r];
  }
  for (int b = m->nbody - 1; b > 0; b--)
    if (m->body_parentid[b] > 0)
      for (int r = 0; r < 6; r++) d->cfrc[6 * m->body_parentid[b] + r] += d->cfrc[6 * b + r];
  for (int i = 0; i < m->nv; i++) d->qfrc_bias[i] = dot6(d->cdof + 6 * i, d->cfrc + 6 * m->dof_bodyid[i]);
}

/* ------------------------------------------------------------------------------------------- */
/* collision                                                                                   */
/* ------------------------------------------------------------------------------------------- */
static void make_frame(double *frame) {
  normalize3(frame);
  double *y = frame + 3, *z = frame + 6;
  y[0] = y[1] = y[2] = 0;
  if (frame[1] < 0.5 && frame[1] > -0.5) y[1] = 1; else y[2] = 1;
  double t = dot3(frame, y);
  for (int k = 0; k < 3; k++) y[k] -= t * frame[k];
  normalize3(y);
  cross3(z, frame, y);
}

typedef struct { double dist, pos[3], normal[3]; } raw_contact;

/* support point of a convex geom in world direction dir (unit not required) */
static void support(rso_data *d, int g, const double *dir, double *out) {
  rso_model *m = d->m;
  const double *R = d->geom_xmat + 9 * g, *p = d->geom_xpos + 3 * g, *s = m->geom_size + 3 * g;
  double ld[3], lp[3] = {0, 0, 0};
  matT_vec3(ld, R, dir);
  switch (m->geom_type[g]) {
    case G_SPHERE: {
      double n = norm3(ld);
      if (n > MINVAL) for (int k = 0; k < 3; k++) lp[k] = ld[k] / n * s[0];
    } break;
    case G_BOX:
      for (int k = 0; k < 3; k++) lp[k] = ld[k] >= 0 ? s[k] : -s[k];
      break;
    case G_CYLINDER: {
      double n = sqrt(ld[0] * ld[0] + ld[1] * ld[1]);
      if (n > MINVAL) { lp[0] = ld[0] / n * s[0]; lp[1] = ld[1] / n * s[0]; }
      lp[2] = ld[2] >= 0 ? s[1] : -s[1];
    } break;
    case G_CAPSULE: {
      double n = norm3(ld);
      if (n > MINVAL) for (int k = 0; k < 3; k++) lp[k] = ld[k] / n * s[0];
      lp[2] += ld[2] >= 0 ? s[1] : -s[1];
    } break;
    case G_ELLIPSOID: {
      double t[3] = {ld[0] * s[0], ld[1] * s[1], ld[2] * s[2]};
      double n = norm3(t);
      if (n > MINVAL) for (int k = 0; k < 3; k++) lp[k] = t[k] / n * s[k];
    } break;
    case G_MESH: {
      int id = m->geom_dataid[g], adr = m->mesh_vertadr[id], num = m->mesh_vertnum[id], best = 0;
      double bv = -1e300;
      for (int i = 0; i < num; i++) {
        double v = dot3(m->mesh_vert + 3 * (adr + i), ld);
        if (v > bv) { bv = v; best = i; }
      }
      memcpy(lp, m->mesh_vert + 3 * (adr + best), sizeof(lp));
    } break;
    default: break;
  }
  mat_vec3(out, R, lp);
  for (int k = 0; k < 3; k++) out[k] += p[k];
}

static void geom_center(rso_data *d, int g, double *c) {
  mat_vec3(c, d->geom_xmat + 9 * g, d->m->geom_rcenter + 3 * g);
  for (int k = 0; k < 3; k++) c[k] += d->geom_xpos[3 * g + k];
}

/* plane (g1) vs box (g2): up to 4 deepest-first corner contacts */
static int plane_box(rso_data *d, int g1, int g2, double margin, raw_contact *out) {
  rso_model *m = d->m;
  const double *n = d->geom_xmat + 9 * g1; /* plane normal = z column */
  double nrm[3] = {n[2], n[5], n[8]};
  const double *R = d->geom_xmat + 9 * g2, *s = m->geom_size + 3 * g2;
  int cnt = 0;
  for (int c = 0; c < 8 && cnt < 4; c++) {
    double lp[3] = {(c & 1) ? s[0] : -s[0], (c & 2) ? s[1] : -s[1], (c & 4) ? s[2] : -s[2]}, wp[3], rel[3];
    mat_vec3(wp, R, lp);
    for (int k = 0; k < 3; k++) { wp[k] += d->geom_xpos[3 * g2 + k]; rel[k] = wp[k] - d->geom_xpos[3 * g1 + k]; }
    double dist = dot3(rel, nrm);
    if (dist > margin) continue;
    out[cnt].dist = dist;
    for (int k = 0; k < 3; k++) { out[cnt].pos[k] = wp[k] - 0.5 * dist * nrm[k]; out[cnt].normal[k] = nrm[k]; }
    cnt++;
  }
  return cnt;
}

/* plane vs generic convex: deepest support point */
static int plane_convex(rso_data *d, int g1, int g2, double margin, raw_contact *out) {
  const double *n = d->geom_xmat + 9 * g1;
  double nrm[3] = {n[2], n[5], n[8]}, neg[3] = {-n[2], -n[5], -n[8]}, sp[3], rel[3];
  support(d, g2, neg, sp);
  for (int k = 0; k < 3; k++) rel[k] = sp[k] - d->geom_xpos[3 * g1 + k];
  double dist = dot3(rel, nrm);
  if (dist > margin) return 0;
  out[0].dist = dist;
  for (int k = 0; k < 3; k++) { out[0].pos[k] = sp[k] - 0.5 * dist * nrm[k]; out[0].normal[k] = nrm[k]; }
  return 1;
}

/* clip polygon (n verts, 2D in ref-face coordinates + carried 3D) against |u|<=hu, |v|<=hv */
static int clip_poly(double (*poly)[3], int n, int axis, double h, double sign) {
  double tmp[16][3];
  int cnt = 0;
  for (int i = 0; i < n; i++) {
    double *a = poly[i], *b = poly[(i + 1) % n];
    double da = sign * a[axis] - h, db = sign * b[axis] - h;
    if (da <= 0) { memcpy(tmp[cnt++], a, 3 * sizeof(double)); }
    if ((da < 0 && db > 0) || (da > 0 && db < 0)) {
      double t = da / (da - db);
      for (int k = 0; k < 3; k++) tmp[cnt][k] = a[k] + t * (b[k] - a[k]);
      cnt++;
    }
    if (cnt >= 15) break;
  }
  memcpy(poly, tmp, sizeof(double) * 3 * cnt);
  return cnt;
}

/* box (g1) vs box (g2): SAT + face clipping / edge-edge.  Normal points from g1 to g2. */
static int box_box(rso_data *d, int g1, int g2, double margin, raw_contact *out) {
  rso_model *m = d->m;
  const double *pa = d->geom_xpos + 3 * g1, *pb = d->geom_xpos + 3 * g2, *Ra = d->geom_xmat + 9 * g1, *Rb = d->geom_xmat + 9 * g2;
  const double *ha = m->geom_size + 3 * g1, *hb = m->geom_size + 3 * g2;
  double A[3][3], B[3][3], dab[3];
  for (int i = 0; i < 3; i++)
    for (int k = 0; k < 3; k++) { A[i][k] = Ra[3 * k + i]; B[i][k] = Rb[3 * k + i]; }
  for (int k = 0; k < 3; k++) dab[k] = pb[k] - pa[k];
  double best_face = -1e300, best_edge = -1e300, sA = -1e300, sB = -1e300;
  int face_id = -1, edge_id = -1, idA = 0, idB = 3;
  double edge_axis[3] = {0, 0, 0};
  for (int i = 0; i < 6; i++) {
    const double *L = i < 3 ? A[i] : B[i - 3];
    double ra = 0, rb = 0;
    for (int k = 0; k < 3; k++) { ra += ha[k] * fabs(dot3(L, A[k])); rb += hb[k] * fabs(dot3(L, B[k])); }
    double s = fabs(dot3(L, dab)) - ra - rb;
    if (s > margin) return 0;
    if (i < 3) { if (s > sA) { sA = s; idA = i; } }
    else if (s > sB) { sB = s; idB = i; }
  }
  /* faces of box 1 are preferred unless a face of box 2 is clearly better (stable under rounding) */
  if (sB > sA + 1e-6) { best_face = sB; face_id = idB; } else { best_face = sA; face_id = idA; }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double L[3];
      cross3(L, A[i], B[j]);
      double n = norm3(L);
      if (n < 1e-6) continue;
      for (int k = 0; k < 3; k++) L[k] /= n;
      double ra = 0, rb = 0;
      for (int k = 0; k < 3; k++) { ra += ha[k] * fabs(dot3(L, A[k])); rb += hb[k] * fabs(dot3(L, B[k])); }
      double s = fabs(dot3(L, dab)) - ra - rb;
      if (s > margin) return 0;
      if (s > best_edge) { best_edge = s; edge_id = 3 * i + j; memcpy(edge_axis, L, sizeof(L)); }
    }
  /* edge contact only if it is a clearly better separating direction than the best face */
  if (edge_id >= 0 && best_edge > 0.95 * best_face + 1e-5 && best_edge > best_face + 1e-5) {
    int i = edge_id / 3, j = edge_id % 3;
    double n[3];
    memcpy(n, edge_axis, sizeof(n));
    if (dot3(n, dab) < 0) for (int k = 0; k < 3; k++) n[k] = -n[k];
    /* supporting edges: point on A furthest along n, on B furthest along -n, edge directions A[i], B[j] */
    double qa[3], qb[3];
    for (int k = 0; k < 3; k++) { qa[k] = pa[k]; qb[k] = pb[k]; }
    for (int a = 0; a < 3; a++) {
      if (a != i) { double sg = dot3(n, A[a]) > 0 ? 1 : -1; for (int k = 0; k < 3; k++) qa[k] += sg * ha[a] * A[a][k]; }
      if (a != j) { double sg = dot3(n, B[a]) > 0 ? -1 : 1; for (int k = 0; k < 3; k++) qb[k] += sg * hb[a] * B[a][k]; }
    }
    /* closest points between lines qa + s*A[i], qb + t*B[j] */
    double r[3], ab = dot3(A[i], B[j]);
    for (int k = 0; k < 3; k++) r[k] = qb[k] - qa[k];
    double den = 1 - ab * ab, ra_ = dot3(r, A[i]), rb_ = dot3(r, B[j]);
    double s = 0, t = 0;
    if (den > 1e-12) { s = (ra_ - ab * rb_) / den; t = (ab * ra_ - rb_) / den; }
    s = fmax(-ha[i], fmin(ha[i], s));
    t = fmax(-hb[j], fmin(hb[j], t));
    out[0].dist = best_edge;
    for (int k = 0; k < 3; k++) {
      out[0].pos[k] = 0.5 * ((qa[k] + s * A[i][k]) + (qb[k] + t * B[j][k]));
      out[0].normal[k] = n[k];
    }
    return 1;
  }
  /* face contact: reference box owns the axis */
  int ref_is_a = face_id < 3, ax = face_id % 3;
  const double *pr = ref_is_a ? pa : pb, *pi_ = ref_is_a ? pb : pa, *hr = ref_is_a ? ha : hb, *hi = ref_is_a ? hb : ha;
  double(*Rr)[3] = ref_is_a ? A : B, (*Ri)[3] = ref_is_a ? B : A;
  double n[3], dri[3];
  for (int k = 0; k < 3; k++) dri[k] = pi_[k] - pr[k];
  double sg = dot3(Rr[ax], dri) >= 0 ? 1 : -1;
  for (int k = 0; k < 3; k++) n[k] = sg * Rr[ax][k]; /* from reference toward incident */
  /* incident face: most anti-parallel to n */
  int iax = 0;
  double bestd = -1;
  for (int k = 0; k < 3; k++) { double v = fabs(dot3(Ri[k], n)); if (v > bestd) { bestd = v; iax = k; } }
  double isg = dot3(Ri[iax], n) > 0 ? -1 : 1;
  int u = (iax + 1) % 3, v = (iax + 2) % 3, ru = (ax + 1) % 3, rv = (ax + 2) % 3;
  double poly[16][3];
  int np = 0;
  static const double cs[4][2] = {{1, 1}, {-1, 1}, {-1, -1}, {1, -1}};
  for (int c = 0; c < 4; c++) {
    double w[3], rel[3];
    for (int k = 0; k < 3; k++) w[k] = pi_[k] + isg * hi[iax] * Ri[iax][k] + cs[c][0] * hi[u] * Ri[u][k] + cs[c][1] * hi[v] * Ri[v][k];
    for (int k = 0; k < 3; k++) rel[k] = w[k] - pr[k];
    /* coordinates in the reference face frame: (along ru, along rv, height above face along n) */
    poly[np][0] = dot3(rel, Rr[ru]); poly[np][1] = dot3(rel, Rr[rv]); poly[np][2] = dot3(rel, n) - hr[ax];
    np++;
  }
  np = clip_poly(poly, np, 0, hr[ru], 1); if (np) np = clip_poly(poly, np, 0, hr[ru], -1);
  if (np) np = clip_poly(poly, np, 1, hr[rv], 1); if (np) np = clip_poly(poly, np, 1, hr[rv], -1);
  int cnt = 0;
  for (int i = 0; i < np && cnt < 8; i++) {
    double dist = poly[i][2];
    if (dist > margin) continue;
    out[cnt].dist = dist;
    for (int k = 0; k < 3; k++) {
      double w = pr[k] + poly[i][0] * Rr[ru][k] + poly[i][1] * Rr[rv][k] + (hr[ax] + poly[i][2]) * n[k];
      out[cnt].pos[k] = w - 0.5 * dist * n[k];
      out[cnt].normal[k] = ref_is_a ? n[k] : -n[k];
    }
    cnt++;
  }
  return cnt;
}

/* Minkowski-difference support S(dir) = sup1(dir) - sup2(-dir), with witness points */
static void mink_support(rso_data *d, int g1, int g2, const double *dir, double *v, double *p1, double *p2) {
  double nd[3] = {-dir[0], -dir[1], -dir[2]};
  support(d, g1, dir, p1);
  support(d, g2, nd, p2);
  for (int k = 0; k < 3; k++) v[k] = p1[k] - p2[k];
}

/* closest point on triangle (a,b,c) to the origin (Ericson, Real-Time Collision Detection 5.1.5) */
static void tri_closest_origin(const double *a, const double *b, const double *c, double *out, double *bary) {
  double ab[3], ac[3], ap[3], bp[3], cp[3];
  for (int k = 0; k < 3; k++) { ab[k] = b[k] - a[k]; ac[k] = c[k] - a[k]; ap[k] = -a[k]; bp[k] = -b[k]; cp[k] = -c[k]; }
  double d1 = dot3(ab, ap), d2 = dot3(ac, ap);
  if (d1 <= 0 && d2 <= 0) { bary[0] = 1; bary[1] = 0; bary[2] = 0; goto done; }
  double d3 = dot3(ab, bp), d4 = dot3(ac, bp);
  if (d3 >= 0 && d4 <= d3) { bary[0] = 0; bary[1] = 1; bary[2] = 0; goto done; }
  double vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) { double v = d1 / (d1 - d3); bary[0] = 1 - v; bary[1] = v; bary[2] = 0; goto done; }
  double d5 = dot3(ab, cp), d6 = dot3(ac, cp);
  if (d6 >= 0 && d5 <= d6) { bary[0] = 0; bary[1] = 0; bary[2] = 1; goto done; }
  double vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) { double w = d2 / (d2 - d6); bary[0] = 1 - w; bary[1] = 0; bary[2] = w; goto done; }
  double va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { double w = (d4 - d3) / ((d4 - d3) + (d5 - d6)); bary[0] = 0; bary[1] = 1 - w; bary[2] = w; goto done; }
  {
    double den = 1.0 / (va + vb + vc), v = vb * den, w = vc * den;
    bary[0] = 1 - v - w; bary[1] = v; bary[2] = w;
  }
done:
  for (int k = 0; k < 3; k++) out[k] = bary[0] * a[k] + bary[1] * b[k] + bary[2] * c[k];
}

/* Minkowski Portal Refinement penetration query for two convex geoms (g1, g2). */
static int convex_convex(rso_data *d, int g1, int g2, double margin, raw_contact *out) {
  const double tol = 1e-6;
  double c1[3], c2[3], v0[3], v1[3], v2[3], v3[3], v4[3], p11[3], p12[3], p21[3], p22[3], p31[3], p32[3], p41[3], p42[3];
  double dir[3], t[3], va[3], vb[3];
  (void)margin; /* margin = 0 for all robosuite geoms: penetrating contacts only */
  geom_center(d, g1, c1);
  geom_center(d, g2, c2);
  for (int k = 0; k < 3; k++) v0[k] = c1[k] - c2[k];
  if (norm3(v0) < 1e-9) { v0[0] = 1e-5; }
  for (int k = 0; k < 3; k++) dir[k] = -v0[k];
  normalize3(dir);
  mink_support(d, g1, g2, dir, v1, p11, p12);
  if (dot3(v1, dir) <= 0) return 0;
  cross3(dir, v0, v1);
  if (norm3(dir) < 1e-12) {
    /* origin lies on the ray v0 -> v1: penetration straight along it */
    double n[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
    normalize3(n);
    out[0].dist = -dot3(v1, n);
    for (int k = 0; k < 3; k++) { out[0].normal[k] = n[k]; out[0].pos[k] = 0.5 * (p11[k] + p12[k]); }
    return 1;
  }
  normalize3(dir);
  mink_support(d, g1, g2, dir, v2, p21, p22);
  if (dot3(v2, dir) <= 0) return 0;
  for (int k = 0; k < 3; k++) { va[k] = v1[k] - v0[k]; vb[k] = v2[k] - v0[k]; }
  cross3(dir, va, vb);
  if (dot3(dir, v0) > 0) {
    for (int k = 0; k < 3; k++) { double s; s = v1[k]; v1[k] = v2[k]; v2[k] = s; s = p11[k]; p11[k] = p21[k]; p21[k] = s; s = p12[k]; p12[k] = p22[k]; p22[k] = s; dir[k] = -dir[k]; }
  }
  /* portal discovery */
  for (int it = 0;; it++) {
    if (it > 100) return 0;
    if (normalize3(dir) == 0) return 0;
    mink_support(d, g1, g2, dir, v3, p31, p32);
    if (dot3(v3, dir) <= 0) return 0;
    cross3(t, v1, v3);
    if (dot3(t, v0) < -1e-14) {
      memcpy(v2, v3, sizeof(v3)); memcpy(p21, p31, sizeof(v3)); memcpy(p22, p32, sizeof(v3));
      for (int k = 0; k < 3; k++) { va[k] = v1[k] - v0[k]; vb[k] = v3[k] - v0[k]; }
      cross3(dir, va, vb);
      continue;
    }
    cross3(t, v3, v2);
    if (dot3(t, v0) < -1e-14) {
      memcpy(v1, v3, sizeof(v3)); memcpy(p11, p31, sizeof(v3)); memcpy(p12, p32, sizeof(v3));
      for (int k = 0; k < 3; k++) { va[k] = v3[k] - v0[k]; vb[k] = v2[k] - v0[k]; }
      cross3(dir, va, vb);
      continue;
    }
    break;
  }
  /* portal refinement */
  int hit = 0;
  for (int it = 0; it < 128; it++) {
    for (int k = 0; k < 3; k++) { va[k] = v2[k] - v1[k]; vb[k] = v3[k] - v1[k]; }
    cross3(dir, va, vb);
    if (normalize3(dir) == 0) break;
    if (dot3(dir, v1) >= 0) hit = 1;
    mink_support(d, g1, g2, dir, v4, p41, p42);
    double dv4 = dot3(v4, dir);
    if (dv4 < 0 && !hit) return 0;
    double delta = dv4 - dot3(v3, dir);
    if (delta <= tol || it == 127) break;
    if (!hit && dv4 < 0) return 0;
    cross3(t, v4, v0);
    if (dot3(v1, t) > 0) {
      if (dot3(v2, t) > 0) { memcpy(v1, v4, sizeof(v4)); memcpy(p11, p41, sizeof(v4)); memcpy(p12, p42, sizeof(v4)); }
      else { memcpy(v3, v4, sizeof(v4)); memcpy(p31, p41, sizeof(v4)); memcpy(p32, p42, sizeof(v4)); }
    } else {
      if (dot3(v3, t) > 0) { memcpy(v2, v4, sizeof(v4)); memcpy(p21, p41, sizeof(v4)); memcpy(p22, p42, sizeof(v4)); }
      else { memcpy(v1, v4, sizeof(v4)); memcpy(p11, p41, sizeof(v4)); memcpy(p12, p42, sizeof(v4)); }
    }
  }
  if (!hit) return 0;
  double cp[3], bary[3];
  tri_closest_origin(v1, v2, v3, cp, bary);
  double depth = norm3(cp);
  double n[3];
  if (depth > 1e-12) for (int k = 0; k < 3; k++) n[k] = cp[k] / depth;
  else memcpy(n, dir, sizeof(n));
  out[0].dist = -depth;
  for (int k = 0; k < 3; k++) {
    double w1 = bary[0] * p11[k] + bary[1] * p21[k] + bary[2] * p31[k];
    double w2 = bary[0] * p12[k] + bary[1] * p22[k] + bary[2] * p32[k];
    out[0].pos[k] = 0.5 * (w1 + w2);
    out[0].normal[k] = n[k];
  }
  return 1;
}

static double mixd(double a, double b, double mix) { return mix * a + (1 - mix) * b; }

static void collision(rso_data *d) {
  rso_model *m = d->m;
  d->ncon = 0;
  d->con_overflow = 0;
  for (int p = 0; p < m->npair; p++) {
    int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p];
    int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
    double margin = fmax(m->geom_margin[g1], m->geom_margin[g2]);
    double gap = fmax(m->geom_gap[g1], m->geom_gap[g2]);
    /* broadphase: bounding spheres (plane: signed distance of the sphere centre) */
    double c2[3];
    geom_center(d, g2, c2);
    if (t1 == G_PLANE) {
      const double *R = d->geom_xmat + 9 * g1;
      double nrm[3] = {R[2], R[5], R[8]}, rel[3];
      for (int k = 0; k < 3; k++) rel[k] = c2[k] - d->geom_xpos[3 * g1 + k];
      if (dot3(rel, nrm) - m->geom_rbound[g2] > margin) continue;
    } else {
      double c1[3], rel[3];
      geom_center(d, g1, c1);
      for (int k = 0; k < 3; k++) rel[k] = c2[k] - c1[k];
      double bound = m->geom_rbound[g1] + m->geom_rbound[g2] + margin;
      if (dot3(rel, rel) > bound * bound) continue;
    }
    raw_contact rc[8];
    int n = 0;
    if (t1 == G_PLANE && t2 == G_BOX) n = plane_box(d, g1, g2, margin, rc);
    else if (t1 == G_PLANE) n = plane_convex(d, g1, g2, margin, rc);
    else if (t1 == G_BOX && t2 == G_BOX) n = box_box(d, g1, g2, margin, rc);
    else n = convex_convex(d, g1, g2, margin, rc);
    for (int i = 0; i < n; i++) {
      if (d->ncon >= MAXCON) { d->con_overflow = 1; break; }
      rso_contact *c = &d->contact[d->ncon++];
      memset(c, 0, sizeof(*c));
      c->dist = rc[i].dist;
      memcpy(c->pos, rc[i].pos, sizeof(c->pos));
      memcpy(c->frame, rc[i].normal, 3 * sizeof(double));
      make_frame(c->frame);
      c->geom1 = g1; c->geom2 = g2;
      c->includemargin = margin - gap;
      /* contact parameter mixing (mj_contactParam [3P]) */
      int p1 = m->geom_priority[g1], p2 = m->geom_priority[g2];
      double mix;
      const double *f1 = m->geom_friction + 3 * g1, *f2 = m->geom_friction + 3 * g2;
      double fr[3];
      if (p1 != p2) {
        int gp = p1 > p2 ? g1 : g2;
        c->dim = m->geom_condim[gp];
        memcpy(c->solref, m->geom_solref + 2 * gp, sizeof(c->solref));
        memcpy(c->solimp, m->geom_solimp + 5 * gp, sizeof(c->solimp));
        memcpy(fr, m->geom_friction + 3 * gp, sizeof(fr));
      } else {
        c->dim = m->geom_condim[g1] > m->geom_condim[g2] ? m->geom_condim[g1] : m->geom_condim[g2];
        double s1 = m->geom_solmix[g1], s2 = m->geom_solmix[g2];
        if (s1 >= MINVAL && s2 >= MINVAL) mix = s1 / (s1 + s2);
        else if (s1 < MINVAL && s2 < MINVAL) mix = 0.5;
        else mix = s1 < MINVAL ? 0.0 : 1.0;
        const double *r1 = m->geom_solref + 2 * g1, *r2 = m->geom_solref + 2 * g2;
        if (r1[0] > 0 && r2[0] > 0) { c->solref[0] = mixd(r1[0], r2[0], mix); c->solref[1] = mixd(r1[1], r2[1], mix); }
        else { c->solref[0] = fmin(r1[0], r2[0]); c->solref[1] = fmin(r1[1], r2[1]); }
        for (int k = 0; k < 5; k++) c->solimp[k] = mixd(m->geom_solimp[5 * g1 + k], m->geom_solimp[5 * g2 + k], mix);
        for (int k = 0; k < 3; k++) fr[k] = fmax(f1[k], f2[k]);
      }
      c->friction[0] = c->friction[1] = fr[0];
      c->friction[2] = fr[1];
      c->friction[3] = c->friction[4] = fr[2];
    }
  }
}

/* ------------------------------------------------------------------------------------------- */
/* constraints (mj_makeConstraint, mj_makeImpedance, mj_referenceConstraint [3P])              */
/* ------------------------------------------------------------------------------------------- */
static double impedance(const double *solimp, double x_abs) {
  double dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  dmin = fmin(0.9999, fmax(0.0001, dmin));
  dmax = fmin(0.9999, fmax(0.0001, dmax));
  width = fmax(MINVAL, width);
  mid = fmin(0.9999, fmax(0.0001, mid));
  power = fmax(1.0, power);
  double x = x_abs / width, y;
  if (x >= 1) return dmax;
  if (x <= 0) return dmin;
  if (power == 1) y = x;
  else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
  else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
  return dmin + y * (dmax - dmin);
}

static void add_row(rso_data *d, int type, int id, double pos, double margin, double frictionloss, const double *solref, const double *solimp, double diag) {
  rso_model *m = d->m;
  int i = d->nefc++;
  d->efc_type[i] = type; d->efc_id[i] = id; d->efc_pos[i] = pos; d->efc_margin[i] = margin; d->efc_frictionloss[i] = frictionloss;
  d->efc_diagApprox[i] = diag;
  double imp = impedance(solimp, fabs(pos - margin));
  double dmax = fmin(0.9999, fmax(0.0001, solimp[1]));
  double K, B;
  if (solref[0] > 0) {
    double tc = fmax(solref[0], 2 * m->timestep), dr = solref[1];
    B = 2 / fmax(MINVAL, dmax * tc);
    K = 1 / fmax(MINVAL, dmax * dmax * tc * tc * dr * dr);
  } else {
    K = -solref[0] / fmax(MINVAL, dmax * dmax);
    B = -solref[1] / fmax(MINVAL, dmax);
  }
  d->efc_KBIP[4 * i] = K; d->efc_KBIP[4 * i + 1] = B; d->efc_KBIP[4 * i + 2] = imp; d->efc_KBIP[4 * i + 3] = 0;
  d->efc_R[i] = fmax(MINVAL, (1 - imp) / imp * diag);
}

static void make_constraint(rso_data *d) {
  rso_model *m = d->m;
  int nv = m->nv;
  d->nefc = 0;
  /* equality constraints come first (mj_makeConstraint order [3P]: equality, friction loss, limits, contacts); only equality/tendon over one
   * fixed tendon exists here: residual = (length - length0) - polycoef[0], Jacobian = the tendon's coefficient row */
  for (int e = 0; e < m->neq; e++) {
    int t = m->eq_obj1id[e], r = d->nefc;
    double *J = d->efc_J + (size_t)r * nv, len = 0;
    memset(J, 0, sizeof(double) * nv);
    for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) {
      int j = m->wrap_objid[w];
      len += m->wrap_prm[w] * d->qpos[m->jnt_qposadr[j]];
      J[m->jnt_dofadr[j]] += m->wrap_prm[w];
    }
    add_row(d, C_EQUALITY, e, len - m->tendon_length0[t] - m->eq_data[5 * e], 0, 0, m->eq_solref + 2 * e, m->eq_solimp + 5 * e, m->tendon_invweight0[t]);
  }
  /* dof friction loss */
  for (int i = 0; i < nv; i++)
    if (m->dof_frictionloss[i] > 0) {
      int r = d->nefc;
      memset(d->efc_J + (size_t)r * nv, 0, sizeof(double) * nv);
      d->efc_J[(size_t)r * nv + i] = 1;
      add_row(d, C_FRICTION_DOF, i, 0, 0, m->dof_frictionloss[i], m->dof_solref + 2 * i, m->dof_solimp + 5 * i, m->dof_invweight0[i]);
    }
  /* tendon friction loss (mj_instantiateFriction [3P]: dofs first, then tendons; diagApprox = tendon_invweight0, solreffriction / solimpfriction) */
  if (m->tendon_frictionloss)
    for (int t = 0; t < m->ntendon; t++)
      if (m->tendon_frictionloss[t] > 0) {
        int r = d->nefc;
        double *J = d->efc_J + (size_t)r * nv;
        memset(J, 0, sizeof(double) * nv);
        for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) J[m->jnt_dofadr[m->wrap_objid[w]]] += m->wrap_prm[w];
        add_row(d, C_FRICTION_TENDON, t, 0, 0, m->tendon_frictionloss[t], m->tendon_solref_fri + 2 * t, m->tendon_solimp_fri + 5 * t, m->tendon_invweight0[t]);
      }
  /* joint limits (hinge / slide) */
  for (int j = 0; j < m->njnt; j++) {
    if (!m->jnt_limited[j] || (m->jnt_type[j] != JNT_HINGE && m->jnt_type[j] != JNT_SLIDE)) continue;
    double q = d->qpos[m->jnt_qposadr[j]], margin = m->jnt_margin[j];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side < 0 ? q - m->jnt_range[2 * j] : m->jnt_range[2 * j + 1] - q;
      if (dist < margin) {
        int r = d->nefc;
        memset(d->efc_J + (size_t)r * nv, 0, sizeof(double) * nv);
        d->efc_J[(size_t)r * nv + m->jnt_dofadr[j]] = -side;
        add_row(d, C_LIMIT_JOINT, j, dist, margin, 0, m->jnt_solref + 2 * j, m->jnt_solimp + 5 * j, m->dof_invweight0[m->jnt_dofadr[j]]);
      }
    }
  }
  /* tendon limits */
  for (int t = 0; t < m->ntendon; t++) {
    if (!m->tendon_limited[t]) continue;
    double len = 0, margin = m->tendon_margin[t];
    for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) len += m->wrap_prm[w] * d->qpos[m->jnt_qposadr[m->wrap_objid[w]]];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side < 0 ? len - m->tendon_range[2 * t] : m->tendon_range[2 * t + 1] - len;
      if (dist < margin) {
        int r = d->nefc;
        double *J = d->efc_J + (size_t)r * nv;
        memset(J, 0, sizeof(double) * nv);
        for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) J[m->jnt_dofadr[m->wrap_objid[w]]] += -side * m->wrap_prm[w];
        add_row(d, C_LIMIT_TENDON, t, dist, margin, 0, m->tendon_solref_lim + 2 * t, m->tendon_solimp_lim + 5 * t, m->tendon_invweight0[t]);
      }
    }
  }
  /* contacts */
  double *jp1 = dalloc(3 * nv), *jr1 = dalloc(3 * nv), *jp2 = dalloc(3 * nv), *jr2 = dalloc(3 * nv);
  for (int c = 0; c < d->ncon; c++) {
    rso_contact *con = &d->contact[c];
    con->efc_address = -1;
    if (con->dist >= con->includemargin) continue;
    int dim = con->dim;
    if (d->nefc + dim > MAXEFC) { d->con_overflow = 1; break; }
    int b1 = m->geom_bodyid[con->geom1], b2 = m->geom_bodyid[con->geom2];
    jac_point(d, jp1, jr1, con->pos, b1);
    jac_point(d, jp2, jr2, con->pos, b2);
    double tran = m->body_invweight0[2 * b1] + m->body_invweight0[2 * b2];
    double rot = m->body_invweight0[2 * b1 + 1] + m->body_invweight0[2 * b2 + 1];
    con->efc_address = d->nefc;
    int first = d->nefc;
    for (int k = 0; k < dim; k++) {
      int r = d->nefc;
      double *J = d->efc_J + (size_t)r * nv;
      const double *ax = con->frame + 3 * (k < 3 ? k : k - 3);
      for (int i = 0; i < nv; i++) {
        if (k < 3) J[i] = ax[0] * (jp2[i] - jp1[i]) + ax[1] * (jp2[nv + i] - jp1[nv + i]) + ax[2] * (jp2[2 * nv + i] - jp1[2 * nv + i]);
        else J[i] = ax[0] * (jr2[i] - jr1[i]) + ax[1] * (jr2[nv + i] - jr1[nv + i]) + ax[2] * (jr2[2 * nv + i] - jr1[2 * nv + i]);
      }
      add_row(d, dim == 1 ? C_CONTACT_FRICTIONLESS : C_CONTACT_ELLIPTIC, c, k == 0 ? con->dist : 0, k == 0 ? con->includemargin : 0, 0, con->solref,
              con->solimp, k < 3 ? tran : rot);
    }
    if (dim > 1) {
      /* elliptic cone: friction regularisers derive from the normal one (impratio, mu_j^2 R_j = const) */
      d->efc_R[first + 1] = d->efc_R[first] / fmax(MINVAL, m->impratio);
      for (int k = 2; k < dim; k++)
        d->efc_R[first + k] = d->efc_R[first + 1] * con->friction[0] * con->friction[0] / fmax(MINVAL, con->friction[k - 1] * con->friction[k - 1]);
      con->mu = con->friction[0] * sqrt(d->efc_R[first + 1] / d->efc_R[first]);
    } else con->mu = 0;
  }
  free(jp1); free(jr1); free(jp2); free(jr2);
  /* D, vel, aref */
  for (int i = 0; i < d->nefc; i++) {
    d->efc_D[i] = 1.0 / d->efc_R[i];
    double v = 0;
    for (int k = 0; k < nv; k++) v += d->efc_J[(size_t)i * nv + k] * d->qvel[k];
    d->efc_vel[i] = v;
    d->efc_aref[i] = -d->efc_KBIP[4 * i + 1] * v - d->efc_KBIP[4 * i] * d->efc_KBIP[4 * i + 2] * (d->efc_pos[i] - d->efc_margin[i]);
  }
}

/* ------------------------------------------------------------------------------------------- */
/* PGS on the dual problem                                                                     */
/* ------------------------------------------------------------------------------------------- */
static int solve_small(double *A, double *b, int n) { /* Gaussian elimination with partial pivoting; A n x n row-major, b -> x */
  for (int c = 0; c < n; c++) {
    int p = c;
    for (int r = c + 1; r < n; r++) if (fabs(A[r * n + c]) > fabs(A[p * n + c])) p = r;
    if (fabs(A[p * n + c]) < MINVAL) return -1;
    if (p != c) { for (int k = 0; k < n; k++) { double t = A[c * n + k]; A[c * n + k] = A[p * n + k]; A[p * n + k] = t; } double t = b[c]; b[c] = b[p]; b[p] = t; }
    for (int r = c + 1; r < n; r++) {
      double f = A[r * n + c] / A[c * n + c];
      for (int k = c; k < n; k++) A[r * n + k] -= f * A[c * n + k];
      b[r] -= f * b[c];
    }
  }
  for (int r = n - 1; r >= 0; r--) {
    double s = b[r];
    for (int k = r + 1; k < n; k++) s -= A[r * n + k] * b[k];
    b[r] = s / A[r * n + r];
  }
  return 0;
}

/* min 0.5 x'Ax + x'b  s.t.  sum (x_i/d_i)^2 <= r^2 ; n <= 5 */
static void qcqp(const double *A, const double *b, const double *dd, double r, int n, double *x) {
  double As[25], bs[5], M[25], y[5], y2[5];
  for (int i = 0; i < n; i++) { bs[i] = b[i] * dd[i]; for (int j = 0; j < n; j++) As[i * n + j] = A[i * n + j] * dd[i] * dd[j]; }
  double la = 0, r2 = r * r;
  for (int it = 0; it < 20; it++) {
    for (int i = 0; i < n * n; i++) M[i] = As[i];
    for (int i = 0; i < n; i++) { M[i * n + i] += la; y[i] = -bs[i]; }
    if (solve_small(M, y, n)) { for (int i = 0; i < n; i++) y[i] = 0; break; }
    double val = -r2;
    for (int i = 0; i < n; i++) val += y[i] * y[i];
    if (val < 1e-10) break;
    /* derivative of |y|^2 wrt la: -2 y' (As+la I)^-1 y */
    for (int i = 0; i < n * n; i++) M[i] = As[i];
    for (int i = 0; i < n; i++) { M[i * n + i] += la; y2[i] = y[i]; }
    if (solve_small(M, y2, n)) break;
    double deriv = 0;
    for (int i = 0; i < n; i++) deriv -= 2 * y[i] * y2[i];
    double delta = -val / deriv;
    if (delta < 1e-10) break;
    la += delta;
  }
  /* final clamp onto the ellipsoid if Newton stopped early */
  double nn = 0;
  for (int i = 0; i < n; i++) nn += y[i] * y[i];
  if (nn > r2 && nn > 0) { double s = r / sqrt(nn); for (int i = 0; i < n; i++) y[i] *= s; }
  for (int i = 0; i < n; i++) x[i] = y[i] * dd[i];
}

static double dual_cost(rso_data *d, const double *f) {
  int n = d->nefc;
  double c = 0;
  for (int i = 0; i < n; i++) {
    double s = d->efc_b[i];
    for (int j = 0; j < n; j++) s += 0.5 * d->efc_AR[(size_t)i * MAXEFC + j] * f[j];
    c += f[i] * s;
  }
  return c;
}

/* primal force function used for warm start (mj_constraintUpdate [3P]) */
static void primal_force(rso_data *d, const double *jar, double *f) {
  for (int i = 0; i < d->nefc; i++) {
    switch (d->efc_type[i]) {
      case C_FRICTION_TENDON:
      case C_FRICTION_DOF: {
        double v = -d->efc_D[i] * jar[i], fl = d->efc_frictionloss[i];
        f[i] = v > fl ? fl : (v < -fl ? -fl : v);
      } break;
      case C_EQUALITY:
        f[i] = -d->efc_D[i] * jar[i];
        break;
      case C_LIMIT_JOINT:
      case C_LIMIT_TENDON:
      case C_CONTACT_FRICTIONLESS:
        f[i] = jar[i] < 0 ? -d->efc_D[i] * jar[i] : 0;
        break;
      case C_CONTACT_ELLIPTIC: {
        rso_contact *con = &d->contact[d->efc_id[i]];
        int dim = con->dim;
        double mu = con->mu, U[6], T = 0;
        U[0] = jar[i] * mu;
        for (int j = 1; j < dim; j++) { U[j] = jar[i + j] * con->friction[j - 1]; T += U[j] * U[j]; }
        T = sqrt(T);
        double N = U[0];
        if (N >= mu * T || (T <= 0 && N >= 0)) { for (int j = 0; j < dim; j++) f[i + j] = 0; }
        else if (mu * N + T <= 0 || (T <= 0 && N < 0)) { for (int j = 0; j < dim; j++) f[i + j] = -d->efc_D[i + j] * jar[i + j]; }
        else {
          double Dm = d->efc_D[i] / fmax(mu * mu * (1 + mu * mu), MINVAL), NT = N - mu * T;
          f[i] = -Dm * NT * mu;
          for (int j = 1; j < dim; j++) f[i + j] = -f[i] / T * U[j] * con->friction[j - 1];
        }
        i += dim - 1;
      } break;
    }
  }
}

static void solve_pgs(rso_data *d) {
  rso_model *m = d->m;
  int nv = m->nv, n = d->nefc;
  /* MinvJT columns, AR = J Minv J' + R, b = J qacc_smooth - aref */
  double *col = dalloc(nv);
  for (int i = 0; i < n; i++) {
    memcpy(col, d->efc_J + (size_t)i * nv, sizeof(double) * nv);
    chol_solve(d->qL, col, nv);
    for (int k = 0; k < nv; k++) d->efc_MinvJT[(size_t)k * MAXEFC + i] = col[k];
  }
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) {
      double s = 0;
      for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * d->efc_MinvJT[(size_t)k * MAXEFC + j];
      d->efc_AR[(size_t)i * MAXEFC + j] = s;
    }
    d->efc_AR[(size_t)i * MAXEFC + i] += d->efc_R[i];
    double s = 0;
    for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * d->qacc_smooth[k];
    d->efc_b[i] = s - d->efc_aref[i];
  }
  /* warm start from previous acceleration */
  double *f = d->efc_force, jar[MAXEFC];
  for (int i = 0; i < n; i++) {
    double s = 0;
    for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * d->qacc_warmstart[k];
    jar[i] = s - d->efc_aref[i];
  }
  primal_force(d, jar, f);
  if (dual_cost(d, f) > 0) memset(f, 0, sizeof(double) * n);
  double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
  int iter;
  for (iter = 0; iter < m->iterations; iter++) {
    double improvement = 0;
    for (int i = 0; i < n; i++) {
      int type = d->efc_type[i];
      int dim = type == C_CONTACT_ELLIPTIC ? d->contact[d->efc_id[i]].dim : 1;
      double res[6], old[6], Athis[36];
      for (int k = 0; k < dim; k++) {
        double s = d->efc_b[i + k];
        for (int j = 0; j < n; j++) s += d->efc_AR[(size_t)(i + k) * MAXEFC + j] * f[j];
        res[k] = s;
        old[k] = f[i + k];
        for (int l = 0; l < dim; l++) Athis[k * dim + l] = d->efc_AR[(size_t)(i + k) * MAXEFC + i + l];
      }
      if (dim == 1) {
        double v = f[i] - res[0] / Athis[0];
        if (type == C_FRICTION_DOF || type == C_FRICTION_TENDON) { double fl = d->efc_frictionloss[i]; v = v > fl ? fl : (v < -fl ? -fl : v); }
        else if (type == C_EQUALITY) { /* bilateral: unbounded */ }
        else if (v < 0) v = 0;
        f[i] = v;
      } else {
        rso_contact *con = &d->contact[d->efc_id[i]];
        /* (1) normal / ray update */
        if (f[i] < MINVAL) {
          double v = f[i] - res[0] / Athis[0];
          f[i] = v < 0 ? 0 : v;
          for (int k = 1; k < dim; k++) f[i + k] = 0;
        } else {
          double denom = 0, vr = 0;
          for (int k = 0; k < dim; k++) { vr += old[k] * res[k]; for (int l = 0; l < dim; l++) denom += old[k] * Athis[k * dim + l] * old[l]; }
          if (denom >= MINVAL) {
            double x = -vr / denom;
            if (f[i] + x * old[0] < 0) x = -f[i] / old[0];
            for (int k = 0; k < dim; k++) f[i + k] += x * old[k];
          }
        }
        /* (2) friction QCQP with the normal force fixed */
        if (f[i] < MINVAL) { for (int k = 1; k < dim; k++) f[i + k] = 0; }
        else {
          int nf = dim - 1;
          double Ac[25], bc[5], y[5];
          for (int k = 0; k < nf; k++) {
            bc[k] = res[k + 1] + Athis[(k + 1) * dim] * (f[i] - old[0]);
            for (int l = 0; l < nf; l++) { Ac[k * nf + l] = Athis[(k + 1) * dim + l + 1]; bc[k] -= Ac[k * nf + l] * old[l + 1]; }
          }
          qcqp(Ac, bc, con->friction, f[i], nf, y);
          for (int k = 0; k < nf; k++) f[i + k + 1] = y[k];
        }
      }
      /* cost change 0.5 d'A d + d'res */
      double change = 0;
      for (int k = 0; k < dim; k++) {
        double dk = f[i + k] - old[k];
        change += dk * res[k];
        for (int l = 0; l < dim; l++) change += 0.5 * dk * Athis[k * dim + l] * (f[i + l] - old[l]);
      }
      improvement -= change;
      i += dim - 1;
    }
    if (improvement * scale < m->tolerance) { iter++; break; }
  }
  d->solver_iter = iter;
  for (int k = 0; k < nv; k++) {
    double s = 0, a = d->qacc_smooth[k];
    for (int i = 0; i < n; i++) { s += d->efc_J[(size_t)i * nv + k] * f[i]; a += d->efc_MinvJT[(size_t)k * MAXEFC + i] * f[i]; }
    d->qfrc_constraint[k] = s;
    d->qacc[k] = a;
  }
  free(col);
}


/* ------------------------------------------------------------------------------------------- */
/* Newton on the primal problem (MuJoCo's default solver [3P]; robosuite never overrides it)   */
/*   min_a  0.5 (a - a_smooth)' M (a - a_smooth) + sum_i s_i(J a - aref)                      */
/* ------------------------------------------------------------------------------------------- */
enum { ST_SATISFIED = 0, ST_QUADRATIC = 1, ST_LINEARNEG = 2, ST_LINEARPOS = 3, ST_CONE = 4 };

/* cost, force and state of every constraint row for a given jar; optional cone Hessians (36 per contact) */
static double constraint_update(rso_data *d, const double *jar, double *force, int *state, double *hcone) {
  double cost = 0;
  for (int i = 0; i < d->nefc; i++) {
    double D = d->efc_D[i], R = d->efc_R[i];
    switch (d->efc_type[i]) {
      case C_FRICTION_TENDON:
      case C_FRICTION_DOF: {
        double fl = d->efc_frictionloss[i];
        if (jar[i] <= -R * fl) { state[i] = ST_LINEARNEG; force[i] = fl; cost += fl * (-0.5 * R * fl - jar[i]); }
        else if (jar[i] >= R * fl) { state[i] = ST_LINEARPOS; force[i] = -fl; cost += fl * (-0.5 * R * fl + jar[i]); }
        else { state[i] = ST_QUADRATIC; force[i] = -D * jar[i]; cost += 0.5 * D * jar[i] * jar[i]; }
      } break;
      case C_EQUALITY:
        state[i] = ST_QUADRATIC; force[i] = -D * jar[i]; cost += 0.5 * D * jar[i] * jar[i];
        break;
      case C_LIMIT_JOINT:
      case C_LIMIT_TENDON:
      case C_CONTACT_FRICTIONLESS:
        if (jar[i] < 0) { state[i] = ST_QUADRATIC; force[i] = -D * jar[i]; cost += 0.5 * D * jar[i] * jar[i]; }
        else { state[i] = ST_SATISFIED; force[i] = 0; }
        break;
      case C_CONTACT_ELLIPTIC: {
        rso_contact *con = &d->contact[d->efc_id[i]];
        int dim = con->dim;
        double mu = con->mu, U[6], T = 0;
        U[0] = jar[i] * mu;
        for (int j = 1; j < dim; j++) { U[j] = jar[i + j] * con->friction[j - 1]; T += U[j] * U[j]; }
        T = sqrt(T);
        double N = U[0];
        if (N >= mu * T || (T <= 0 && N >= 0)) {
          for (int j = 0; j < dim; j++) { force[i + j] = 0; state[i + j] = ST_SATISFIED; }
        } else if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
          for (int j = 0; j < dim; j++) { force[i + j] = -d->efc_D[i + j] * jar[i + j]; state[i + j] = ST_QUADRATIC; cost += 0.5 * d->efc_D[i + j] * jar[i + j] * jar[i + j]; }
        } else {
          double Dm = D / fmax(mu * mu * (1 + mu * mu), MINVAL), g = N - mu * T;
          cost += 0.5 * Dm * g * g;
          force[i] = -Dm * g * mu;
          for (int j = 1; j < dim; j++) force[i + j] = -force[i] / T * U[j] * con->friction[j - 1];
          for (int j = 0; j < dim; j++) state[i + j] = ST_CONE;
          if (hcone) {
            double *H = hcone + 36 * d->efc_id[i], gr[6];
            gr[0] = mu;
            for (int j = 1; j < dim; j++) gr[j] = -mu * U[j] * con->friction[j - 1] / T;
            for (int j = 0; j < dim; j++)
              for (int k = 0; k < dim; k++) {
                double h = gr[j] * gr[k];
                if (j > 0 && k > 0) {
                  double fj = con->friction[j - 1], fk = con->friction[k - 1];
                  h += -g * mu * fj * fk * ((j == k ? 1.0 / T : 0.0) - U[j] * U[k] / (T * T * T));
                }
                H[j * 6 + k] = Dm * h;
              }
          }
        }
        i += dim - 1;
      } break;
    }
  }
  return cost;
}

/* value and derivatives of the total cost along a + alpha*search */
static void ls_eval(rso_data *d, const double *jar, const double *jv, const double *quadGauss, double alpha, double *p, double *dp, double *ddp) {
  double c = quadGauss[0] + alpha * quadGauss[1] + alpha * alpha * quadGauss[2], c1 = quadGauss[1] + 2 * alpha * quadGauss[2], c2 = 2 * quadGauss[2];
  for (int i = 0; i < d->nefc; i++) {
    double D = d->efc_D[i], R = d->efc_R[i], x = jar[i] + alpha * jv[i], v = jv[i];
    switch (d->efc_type[i]) {
      case C_FRICTION_TENDON:
      case C_FRICTION_DOF: {
        double fl = d->efc_frictionloss[i];
        if (x <= -R * fl) { c += fl * (-0.5 * R * fl - x); c1 -= fl * v; }
        else if (x >= R * fl) { c += fl * (-0.5 * R * fl + x); c1 += fl * v; }
        else { c += 0.5 * D * x * x; c1 += D * x * v; c2 += D * v * v; }
      } break;
      case C_EQUALITY:
        c += 0.5 * D * x * x; c1 += D * x * v; c2 += D * v * v;
        break;
      case C_LIMIT_JOINT:
      case C_LIMIT_TENDON:
      case C_CONTACT_FRICTIONLESS:
        if (x < 0) { c += 0.5 * D * x * x; c1 += D * x * v; c2 += D * v * v; }
        break;
      case C_CONTACT_ELLIPTIC: {
        rso_contact *con = &d->contact[d->efc_id[i]];
        int dim = con->dim;
        double mu = con->mu, U[6], V[6], T = 0, UV = 0, VV = 0;
        U[0] = x * mu; V[0] = v * mu;
        for (int j = 1; j < dim; j++) {
          U[j] = (jar[i + j] + alpha * jv[i + j]) * con->friction[j - 1];
          V[j] = jv[i + j] * con->friction[j - 1];
          T += U[j] * U[j]; UV += U[j] * V[j]; VV += V[j] * V[j];
        }
        T = sqrt(T);
        double N = U[0];
        if (N >= mu * T || (T <= 0 && N >= 0)) { /* nothing */ }
        else if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
          for (int j = 0; j < dim; j++) {
            double xj = jar[i + j] + alpha * jv[i + j], vj = jv[i + j], Dj = d->efc_D[i + j];
            c += 0.5 * Dj * xj * xj; c1 += Dj * xj * vj; c2 += Dj * vj * vj;
          }
        } else {
          double Dm = D / fmax(mu * mu * (1 + mu * mu), MINVAL), g = N - mu * T;
          double g1 = V[0] - mu * UV / T, g2 = -mu * (VV / T - UV * UV / (T * T * T));
          c += 0.5 * Dm * g * g; c1 += Dm * g * g1; c2 += Dm * (g1 * g1 + g * g2);
        }
        i += dim - 1;
      } break;
    }
  }
  *p = c; *dp = c1; *ddp = c2;
}

/* RSO_DEBUG=1: say why the Newton iteration ended (stderr) */
static int rso_debug(void) { static int v = -1; if (v < 0) v = getenv("RSO_DEBUG") ? 1 : 0; return v; }

static void solve_newton(rso_data *d) {
  rso_model *m = d->m;
  int nv = m->nv, n = d->nefc;
  double *a = d->qacc, *Ma = dalloc(nv), *grad = dalloc(nv), *search = dalloc(nv), *H = dalloc(nv * nv), *Lh = dalloc(nv * nv), *Mv = dalloc(nv);
  double jar[MAXEFC], jv[MAXEFC], *f = d->efc_force, *hcone = dalloc(36 * (d->ncon > 0 ? d->ncon : 1));
  int state[MAXEFC];
  double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
  /* warm start: previous acceleration unless the unconstrained one is cheaper */
  double cost_ws, cost_sm;
  {
    for (int i = 0; i < n; i++) { double s = 0; for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * d->qacc_smooth[k]; jar[i] = s - d->efc_aref[i]; }
    cost_sm = constraint_update(d, jar, f, state, NULL);
    for (int i = 0; i < n; i++) { double s = 0; for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * d->qacc_warmstart[k]; jar[i] = s - d->efc_aref[i]; }
    cost_ws = constraint_update(d, jar, f, state, NULL);
    for (int i = 0; i < nv; i++) { double s = 0; for (int k = 0; k < nv; k++) s += d->qM[i * nv + k] * (d->qacc_warmstart[k] - d->qacc_smooth[k]); Ma[i] = s; }
    for (int i = 0; i < nv; i++) cost_ws += 0.5 * Ma[i] * (d->qacc_warmstart[i] - d->qacc_smooth[i]);
    memcpy(a, cost_ws < cost_sm ? d->qacc_warmstart : d->qacc_smooth, sizeof(double) * nv);
  }
  int iter = 0;
  double cost = 0;
  for (;; ) {
    /* state at the current point */
    for (int i = 0; i < n; i++) { double s = 0; for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * a[k]; jar[i] = s - d->efc_aref[i]; }
    cost = constraint_update(d, jar, f, state, hcone);
    double gauss = 0;
    for (int i = 0; i < nv; i++) { double s = 0; for (int k = 0; k < nv; k++) s += d->qM[i * nv + k] * a[k]; Ma[i] = s; }
    for (int i = 0; i < nv; i++) gauss += 0.5 * (Ma[i] - d->qfrc_smooth[i]) * (a[i] - d->qacc_smooth[i]);
    cost += gauss;
    double gn = 0;
    for (int k = 0; k < nv; k++) {
      double s = Ma[k] - d->qfrc_smooth[k];
      for (int i = 0; i < n; i++) s -= d->efc_J[(size_t)i * nv + k] * f[i];
      grad[k] = s; gn += s * s;
    }
    if (iter >= m->iterations || scale * sqrt(gn) < m->tolerance) { if (rso_debug()) fprintf(stderr, "[rso newton] iter %d exit: gradient %.3e (scaled) cost %.6e\n", iter, scale * sqrt(gn), cost); break; }
    /* Hessian */
    memcpy(H, d->qM, sizeof(double) * nv * nv);
    for (int i = 0; i < n; i++) {
      if (state[i] == ST_QUADRATIC) {
        const double *J = d->efc_J + (size_t)i * nv;
        for (int r = 0; r < nv; r++) if (J[r] != 0) for (int c = 0; c < nv; c++) H[r * nv + c] += d->efc_D[i] * J[r] * J[c];
      } else if (state[i] == ST_CONE) {
        int dim = d->contact[d->efc_id[i]].dim;
        const double *hc = hcone + 36 * d->efc_id[i];
        for (int j = 0; j < dim; j++)
          for (int k = 0; k < dim; k++) {
            const double *Jj = d->efc_J + (size_t)(i + j) * nv, *Jk = d->efc_J + (size_t)(i + k) * nv;
            double h = hc[j * 6 + k];
            for (int r = 0; r < nv; r++) if (Jj[r] != 0) for (int c = 0; c < nv; c++) H[r * nv + c] += h * Jj[r] * Jk[c];
          }
        i += dim - 1;
      }
    }
    chol_factor(Lh, H, nv);
    for (int k = 0; k < nv; k++) search[k] = -grad[k];
    chol_solve(Lh, search, nv);
    /* line search */
    for (int i = 0; i < n; i++) { double s = 0; for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * search[k]; jv[i] = s; }
    double quadGauss[3] = {gauss, 0, 0}, snorm = 0;
    for (int i = 0; i < nv; i++) { double s = 0; for (int k = 0; k < nv; k++) s += d->qM[i * nv + k] * search[k]; Mv[i] = s; }
    for (int i = 0; i < nv; i++) { quadGauss[1] += search[i] * (Ma[i] - d->qfrc_smooth[i]); quadGauss[2] += 0.5 * search[i] * Mv[i]; snorm += search[i] * search[i]; }
    snorm = sqrt(snorm);
    if (snorm < MINVAL) { if (rso_debug()) fprintf(stderr, "[rso newton] iter %d exit: zero search\n", iter); break; }
    double gtol = m->tolerance * 0.01 * snorm / scale; /* tolerance * ls_tolerance * |search| * meaninertia * nv */
    double p0, d0, h0, p, dp, hp, lo = 0, hi = -1, alpha;
    ls_eval(d, jar, jv, quadGauss, 0, &p0, &d0, &h0);
    if (d0 >= 0 || h0 <= 0) { if (rso_debug()) fprintf(stderr, "[rso newton] iter %d exit: not a descent direction d0 %.3e h0 %.3e cost %.6e\n", iter, d0, h0, cost); break; }
    alpha = -d0 / h0;
    double last_step = alpha;
    for (int ls = 0; ls < 50; ls++) {
      ls_eval(d, jar, jv, quadGauss, alpha, &p, &dp, &hp);
      if (fabs(dp) < gtol) break;
      if (dp < 0) lo = alpha; else hi = alpha;
      double next = hp > 0 ? alpha - dp / hp : -1;
      if (hi < 0) { if (next <= lo) next = 2 * alpha + 1e-12; }
      else {
        /* bracketed: a Newton candidate that is out of the bracket, or that moves alpha by more than half of what the step before moved it, is replaced by the
         * midpoint (the rule of Numerical Recipes' rtsafe).  The derivative of this objective is monotone but far from linear (rows switch on along the
         * line), and plain safeguarded Newton can creep in from both ends in large alternating jumps -- 50 evaluations without reaching the minimiser on
         * stacked-cube states, after which the step was rejected for not lowering the cost (found in round 4: profiles/r04_x10_line_search.txt).  MuJoCo's
         * own search evaluates the midpoint next to both Newton candidates for the same reason (engine_solver.c PrimalSearch [3P]). */
        if (next <= lo || next >= hi || (ls >= 8 && fabs(next - alpha) > 0.5 * last_step)) next = 0.5 * (lo + hi);   /* ls >= 8: an ordinary search is over by then, and takes the path it always took */
      }
      last_step = fabs(next - alpha);
      alpha = next;
    }
    ls_eval(d, jar, jv, quadGauss, alpha, &p, &dp, &hp);
    if (!(p < p0)) { if (rso_debug()) fprintf(stderr, "[rso newton] iter %d exit: line search found no lower point p0 %.9e p %.9e alpha %.3e d0 %.3e h0 %.3e\n", iter, p0, p, alpha, d0, h0); break; }
    for (int k = 0; k < nv; k++) a[k] += alpha * search[k];
    iter++;
    if (scale * (p0 - p) < m->tolerance) {
      /* converged on improvement: refresh forces at the final point */
      for (int i = 0; i < n; i++) { double s = 0; for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * a[k]; jar[i] = s - d->efc_aref[i]; }
      constraint_update(d, jar, f, state, NULL);
      break;
    }
  }
  d->solver_iter = iter;
  for (int k = 0; k < nv; k++) { double s = 0; for (int i = 0; i < n; i++) s += d->efc_J[(size_t)i * nv + k] * f[i]; d->qfrc_constraint[k] = s; }
  free(Ma); free(grad); free(search); free(H); free(Lh); free(Mv); free(hcone);
}

/* Conditioning probe (tests/test_full_size_parity.py): with d->round_rows set, every input of the constraint solve -- Jacobian rows, reference accelerations,
 * regularisers, friction coefficients, the mass matrix and the smooth forces / accelerations -- is rounded to the nearest float32 before the fp64 solver runs
 * on it.  The difference to the plain solve is what storing those quantities in single precision costs ON THIS STATE even with exact arithmetic downstream:
 * the floor under any fp32 kernel's deviation from this oracle, and the yardstick the PickPlace full-size test measures the kernel against per env. */
static void round_inputs_f32(rso_data *d) {
  int nv = d->m->nv, n = d->nefc;
#define RF(x) ((x) = (double)(float)(x))
  for (size_t i = 0; i < (size_t)n * nv; i++) RF(d->efc_J[i]);
  for (int i = 0; i < n; i++) { RF(d->efc_aref[i]); RF(d->efc_R[i]); d->efc_D[i] = 1.0 / d->efc_R[i]; }
  for (int c = 0; c < d->ncon; c++) { RF(d->contact[c].mu); for (int k = 0; k < 5; k++) RF(d->contact[c].friction[k]); }
  for (int i = 0; i < nv * nv; i++) RF(d->qM[i]);
  for (int i = 0; i < nv; i++) { RF(d->qfrc_smooth[i]); RF(d->qacc_smooth[i]); RF(d->qacc_warmstart[i]); }
#undef RF
}
void rso_set_round_rows(rso_data *d, int on) { d->round_rows = on; }

static void fwd_constraint(rso_data *d) {
  int nv = d->m->nv;
  memset(d->qfrc_constraint, 0, sizeof(double) * nv);
  d->solver_iter = 0;
  if (d->round_rows && d->nefc > 0) round_inputs_f32(d);
  if (d->nefc == 0) { memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv); return; }
  if (d->m->solver == 0) solve_pgs(d); else solve_newton(d);
}

/* ------------------------------------------------------------------------------------------- */
/* actuation, acceleration, integration                                                        */
/* ------------------------------------------------------------------------------------------- */
static void fwd_actuation(rso_data *d) {
  rso_model *m = d->m;
  memset(d->qfrc_actuator, 0, sizeof(double) * m->nv);
  for (int a = 0; a < m->nu; a++) {
    int j = m->actuator_trnid[a], da = m->jnt_dofadr[j], qa = m->jnt_qposadr[j];
    double ctrl = d->ctrl[a], gear = m->actuator_gear[a];
    if (m->actuator_ctrllimited[a]) ctrl = fmax(m->actuator_ctrlrange[2 * a], fmin(m->actuator_ctrlrange[2 * a + 1], ctrl));
    double length = gear * d->qpos[qa], velocity = gear * d->qvel[da];
    double force = m->actuator_gainprm[3 * a] * ctrl;
    if (m->actuator_biastype[a] == 1) force += m->actuator_biasprm[3 * a] + m->actuator_biasprm[3 * a + 1] * length + m->actuator_biasprm[3 * a + 2] * velocity;
    if (m->actuator_forcelimited[a]) force = fmax(m->actuator_forcerange[2 * a], fmin(m->actuator_forcerange[2 * a + 1], force));
    d->actuator_force[a] = force;
    d->qfrc_actuator[da] += gear * force;
  }
}

static void fwd_acceleration(rso_data *d) {
  int nv = d->m->nv;
  for (int i = 0; i < nv; i++) d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_applied[i] + d->qfrc_actuator[i];
  memcpy(d->qacc_smooth, d->qfrc_smooth, sizeof(double) * nv);
  chol_solve(d->qL, d->qacc_smooth, nv);
}

static void euler(rso_data *d) {
  rso_model *m = d->m;
  int nv = m->nv;
  double h = m->timestep, *qa = dalloc(nv);
  int damped = 0;
  for (int i = 0; i < nv; i++) if (m->dof_damping[i] > 0) damped = 1;
  if (damped) {
    double *MhB = dalloc(nv * nv);
    memcpy(MhB, d->qM, sizeof(double) * nv * nv);
    for (int i = 0; i < nv; i++) { MhB[i * nv + i] += h * m->dof_damping[i]; qa[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i]; }
    chol_factor(d->qLD, MhB, nv);
    chol_solve(d->qLD, qa, nv);
    free(MhB);
  } else memcpy(qa, d->qacc, sizeof(double) * nv);
  for (int i = 0; i < nv; i++) d->qvel[i] += h * qa[i];
  for (int j = 0; j < m->njnt; j++) {
    int pa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == JNT_FREE || m->jnt_type[j] == JNT_BALL) {
      if (m->jnt_type[j] == JNT_FREE) { for (int k = 0; k < 3; k++) d->qpos[pa + k] += h * d->qvel[da + k]; pa += 3; da += 3; }
      double w[3] = {d->qvel[da], d->qvel[da + 1], d->qvel[da + 2]}, ang = norm3(w) * h;
      if (ang > MINVAL) {
        double ax[3] = {w[0], w[1], w[2]}, dq[4], q[4];
        normalize3(ax);
        axisangle_quat(dq, ax, ang);
        quat_mul(q, d->qpos + pa, dq);
        quat_norm(q);
        memcpy(d->qpos + pa, q, sizeof(q));
      }
    } else d->qpos[pa] += h * d->qvel[da];
  }
  d->time += h;
  memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * nv);
  free(qa);
}

/* Acceleration-stage sensors: mj_sensorAcc -> mj_rnePostConstraint [3P] for the two sensor types robosuite's grippers define
 * (models/assets/grippers/xml: <force site="ft_frame"/>, <torque site="ft_frame"/>; read by robots/robot.py:739-751, 795-815).
 *   cfrc_ext[b]  = wrench the contacts apply to body b (contact force in the contact frame -> world, -on geom1's body, +on geom2's), about the
 *                  subtree COM of b's kinematic tree (the frame of every c-quantity here);
 *   cacc[b]      = cacc[parent] + cdof_dot qvel + cdof qacc, cacc[world] = -gravity  (the acceleration the solver ended on, not zero as in rne_bias);
 *   cfrc_int[b]  = sum over b's subtree of (cinert cacc + cvel x* (cinert cvel) - cfrc_ext): what b's parent exerts on b through their joint;
 *   force sensor = linear part of cfrc_int[site body] in the site frame; torque sensor = its moment about the site, in the site frame.
 * Equality (connect / weld) forces would enter cfrc_ext too; the models of this path have none.  Evaluated by forward() and step2() before the
 * integrator moves the state, as mj_forward / mj_step2 do. */
static void sensor_acc(rso_data *d) {
  rso_model *m = d->m;
  if (m->nsensor == 0) return;
  int nb = m->nbody;
  memset(d->cfrc_ext, 0, sizeof(double) * 6 * nb);
  for (int i = 0; i < d->ncon; i++) {
    rso_contact *c = &d->contact[i];
    if (c->efc_address < 0) continue;
    const double *f = d->efc_force + c->efc_address;
    double fw[3] = {0, 0, 0}, tw[3] = {0, 0, 0};
    for (int k = 0; k < 3 && k < c->dim; k++) for (int r = 0; r < 3; r++) fw[r] += f[k] * c->frame[3 * k + r];
    for (int k = 3; k < c->dim; k++) for (int r = 0; r < 3; r++) tw[r] += f[k] * c->frame[3 * (k - 3) + r];
    for (int side = 0; side < 2; side++) {
      int b = m->geom_bodyid[side ? c->geom2 : c->geom1];
      if (b == 0) continue;
      double off[3], t[3], sgn = side ? 1.0 : -1.0;
      for (int k = 0; k < 3; k++) off[k] = c->pos[k] - d->subtree_com[3 * m->body_rootid[b] + k];
      cross3(t, off, fw);
      for (int k = 0; k < 3; k++) { d->cfrc_ext[6 * b + k] += sgn * (tw[k] + t[k]); d->cfrc_ext[6 * b + 3 + k] += sgn * fw[k]; }
    }
  }
  memset(d->cacc, 0, 6 * sizeof(double));
  for (int k = 0; k < 3; k++) d->cacc[3 + k] = -m->gravity[k];
  memset(d->cfrc_int, 0, 6 * sizeof(double));
  for (int b = 1; b < nb; b++) {
    double *ca = d->cacc + 6 * b, t1[6], t2[6], t3[6];
    memcpy(ca, d->cacc + 6 * m->body_parentid[b], 6 * sizeof(double));
    for (int i = m->body_dofadr[b]; i >= 0 && i < m->body_dofadr[b] + m->body_dofnum[b]; i++)
      for (int r = 0; r < 6; r++) ca[r] += d->cdof_dot[6 * i + r] * d->qvel[i] + d->cdof[6 * i + r] * d->qacc[i];
    mul_inert_vec(t1, d->cinert + 10 * b, ca);
    mul_inert_vec(t2, d->cinert + 10 * b, d->cvel + 6 * b);
    cross_force(t3, d->cvel + 6 * b, t2);
    for (int r = 0; r < 6; r++) d->cfrc_int[6 * b + r] = t1[r] + t3[r] - d->cfrc_ext[6 * b + r];
  }
  for (int b = nb - 1; b > 0; b--)
    if (m->body_parentid[b] > 0)
      for (int r = 0; r < 6; r++) d->cfrc_int[6 * m->body_parentid[b] + r] += d->cfrc_int[6 * b + r];
  int adr = 0;
  for (int i = 0; i < m->nsensor; i++) {
    int dim = m->sensor_dim[i], type = m->sensor_type[i], site = m->sensor_objid[i];
    for (int k = 0; k < dim; k++) d->sensordata[adr + k] = 0;
    if ((type == 0 || type == 1) && site >= 0 && dim == 3) {
      int b = m->site_bodyid[site];
      const double *w = d->cfrc_int + 6 * b, *R = d->site_xmat + 9 * site;
      if (type == 0) matT_vec3(d->sensordata + adr, R, w + 3);
      else {
        double off[3], t[3], tq[3];
        for (int k = 0; k < 3; k++) off[k] = d->site_xpos[3 * site + k] - d->subtree_com[3 * m->body_rootid[b] + k];
        cross3(t, off, w + 3);
        for (int k = 0; k < 3; k++) tq[k] = w[k] - t[k];
        matT_vec3(d->sensordata + adr, R, tq);
      }
    }
    adr += dim;
  }
}
int rso_nsensordata(rso_data *d) { return d->nsensordata; }
const double *rso_sensordata(rso_data *d) { return d->sensordata; }

/* mj_step1 / mj_step2 / mj_forward / mj_step (utils/binding_utils.py:1089-1107) */
static void fwd_position(rso_data *d) { kinematics(d); com_pos(d); crb(d); collision(d); make_constraint(d); }
static void fwd_velocity(rso_data *d) {
  com_vel(d); passive(d); rne_bias(d);
  /* efc_vel / aref depend on qvel only through make_constraint, already evaluated with current qvel */
}
void rso_step1(rso_data *d) { fwd_position(d); fwd_velocity(d); }
void rso_step2(rso_data *d) { fwd_actuation(d); fwd_acceleration(d); fwd_constraint(d); sensor_acc(d); euler(d); }
void rso_forward(rso_data *d) { fwd_position(d); fwd_velocity(d); fwd_actuation(d); fwd_acceleration(d); fwd_constraint(d); sensor_acc(d); }
/* The solver's own objective at an acceleration `a` of the caller's, for the constraint rows of the last forward():
 *   cost(a) = 1/2 (a - a_smooth)' M (a - a_smooth) + sum_i s_i(J a - aref), and (if grad != NULL) its gradient M a - qfrc_smooth - J' f(a).
 * Test infrastructure: two accelerations are compared in the metric the solver minimises -- the minimiser is unique, but where the Hessian is nearly
 * flat (a 1e-4 kg m^2 object under stiff contacts) accelerations far apart have costs equal to working precision. */
double rso_cost(rso_data *d, const double *a, double *grad) {
  rso_model *m = d->m;
  const int nv = m->nv, n = d->nefc;
  double *jar = (double *)malloc(sizeof(double) * (n ? n : 1)), *f = (double *)malloc(sizeof(double) * (n ? n : 1));
  int *state = (int *)malloc(sizeof(int) * (n ? n : 1));
  for (int i = 0; i < n; i++) {
    double sj = 0;
    for (int k = 0; k < nv; k++) sj += d->efc_J[(size_t)i * nv + k] * a[k];
    jar[i] = sj - d->efc_aref[i];
  }
  double cost = constraint_update(d, jar, f, state, NULL);
  for (int i = 0; i < nv; i++) {
    double ma = 0;
    for (int k = 0; k < nv; k++) ma += d->qM[(size_t)i * nv + k] * a[k];
    cost += 0.5 * (ma - d->qfrc_smooth[i]) * (a[i] - d->qacc_smooth[i]);
    if (grad) {
      double g = ma - d->qfrc_smooth[i];
      for (int r = 0; r < n; r++) g -= d->efc_J[(size_t)r * nv + i] * f[r];
      grad[i] = g;
    }
  }
  free(jar); free(f); free(state);
  return cost;
}

/* Constraint forces and row states (0 satisfied, 1 quadratic, 2 linear-, 3 linear+, 4 cone) the rows of the last forward() give at acceleration `a` (test tooling:
 * where does a kernel's solution differ -- in the rows' data or in the minimiser?) */
void rso_forces_at(rso_data *d, const double *a, double *f, int *state) {
  int nv = d->m->nv, n = d->nefc;
  double *jar = dalloc(n);
  for (int i = 0; i < n; i++) { double sj = 0; for (int k = 0; k < nv; k++) sj += d->efc_J[(size_t)i * nv + k] * a[k]; jar[i] = sj - d->efc_aref[i]; }
  constraint_update(d, jar, f, state, NULL);
  free(jar);
}

/* forward() with the contact GEOMETRY given by the caller: the collision pass runs as usual (pairs, dimensions, mixed friction / solref / solimp),
 * then contact i takes dist = geo[13 i], pos = geo[13 i + 1 .. 3], frame = geo[13 i + 4 .. 12] before the constraint rows are built.  Test
 * infrastructure for the solver / dynamics half of a parity check: fed the kernel's own contact list, everything downstream of the narrow phase is
 * compared on identical inputs (the narrow phases themselves are compared separately).  n must equal the oracle's own contact count; returns 0 on
 * success, -1 (and leaves the plain forward() result) otherwise. */
int rso_forward_with_contact_geometry(rso_data *d, int n, const double *geo) {
  kinematics(d); com_pos(d); crb(d); collision(d);
  if (n != d->ncon) { make_constraint(d); fwd_velocity(d); fwd_actuation(d); fwd_acceleration(d); fwd_constraint(d); sensor_acc(d); return -1; }
  for (int i = 0; i < n; i++) {
    rso_contact *c = &d->contact[i];
    c->dist = geo[13 * i];
    memcpy(c->pos, geo + 13 * i + 1, 3 * sizeof(double));
    memcpy(c->frame, geo + 13 * i + 4, 9 * sizeof(double));
  }
  make_constraint(d); fwd_velocity(d); fwd_actuation(d); fwd_acceleration(d); fwd_constraint(d); sensor_acc(d);
  return 0;
}
void rso_step(rso_data *d) { rso_step1(d); rso_step2(d); }

/* mj_jacSite (utils/binding_utils.py:826-851): jacp, jacr 3 x nv row-major, either may be NULL */
void rso_jac_site(rso_data *d, int site, double *jacp, double *jacr) { jac_point(d, jacp, jacr, d->site_xpos + 3 * site, d->m->site_bodyid[site]); }
void rso_jac_body(rso_data *d, int body, double *jacp, double *jacr) { jac_point(d, jacp, jacr, d->xpos + 3 * body, body); }
void rso_jac_geom(rso_data *d, int geom, double *jacp, double *jacr) { jac_point(d, jacp, jacr, d->geom_xpos + 3 * geom, d->m->geom_bodyid[geom]); }
/* mj_fullM (controllers/parts/controller.py:226-227): dense nv x nv */
void rso_full_M(rso_data *d, double *dst) { memcpy(dst, d->qM, sizeof(double) * d->m->nv * d->m->nv); }

/* generic array access for the python harness */
double *rso_data_field(rso_data *d, const char *name, int *count) {
  rso_model *m = d->m;
#define F(n, c) if (!strcmp(name, #n)) { *count = (c); return d->n; }
  F(qpos, m->nq) F(qvel, m->nv) F(qacc, m->nv) F(qacc_warmstart, m->nv) F(ctrl, m->nu) F(qfrc_applied, m->nv)
  F(mocap_pos, 3 * m->nmocap) F(mocap_quat, 4 * m->nmocap)
  F(xpos, 3 * m->nbody) F(xquat, 4 * m->nbody) F(xmat, 9 * m->nbody) F(xipos, 3 * m->nbody) F(ximat, 9 * m->nbody)
  F(geom_xpos, 3 * m->ngeom) F(geom_xmat, 9 * m->ngeom) F(site_xpos, 3 * m->nsite) F(site_xmat, 9 * m->nsite)
  F(subtree_com, 3 * m->nbody) F(cinert, 10 * m->nbody) F(cdof, 6 * m->nv) F(cvel, 6 * m->nbody) F(cdof_dot, 6 * m->nv)
  F(qM, m->nv * m->nv) F(qfrc_bias, m->nv) F(qfrc_passive, m->nv) F(qfrc_actuator, m->nv) F(qfrc_smooth, m->nv) F(qacc_smooth, m->nv)
  F(qfrc_constraint, m->nv) F(actuator_force, m->nu) F(efc_J, d->nefc * m->nv)
  F(sensordata, d->nsensordata) F(cfrc_int, 6 * m->nbody) F(cfrc_ext, 6 * m->nbody) F(cacc, 6 * m->nbody)
#undef F
#define FA(n, c) if (!strcmp(name, #n)) { *count = (c); return d->n; }
  FA(efc_frictionloss, d->nefc) FA(efc_pos, d->nefc) FA(efc_R, d->nefc) FA(efc_D, d->nefc) FA(efc_aref, d->nefc) FA(efc_force, d->nefc) FA(efc_vel, d->nefc) FA(efc_b, d->nefc)
#undef FA
  if (!strcmp(name, "time")) { *count = 1; return &d->time; }
  *count = 0;
  return NULL;
}
int rso_ncon(rso_data *d) { return d->ncon; }
int rso_nefc(rso_data *d) { return d->nefc; }
int rso_solver_iter(rso_data *d) { return d->solver_iter; }
/* contact i -> out[0]=dist, [1..3]=pos, [4..12]=frame, [13]=geom1, [14]=geom2, [15]=dim, [16]=efc_address, [17]=normal force, [18..22]=friction */
void rso_contact_get(rso_data *d, int i, double *out) {
  rso_contact *c = &d->contact[i];
  out[0] = c->dist;
  memcpy(out + 1, c->pos, 3 * sizeof(double));
  memcpy(out + 4, c->frame, 9 * sizeof(double));
  out[13] = c->geom1; out[14] = c->geom2; out[15] = c->dim; out[16] = c->efc_address;
  out[17] = c->efc_address >= 0 ? d->efc_force[c->efc_address] : 0;
  memcpy(out + 18, c->friction, 5 * sizeof(double));
}
int rso_efc_type(rso_data *d, int i) { return d->efc_type[i]; }
int rso_efc_id(rso_data *d, int i) { return d->efc_id[i]; }

/* ------------------------------------------------------------------------------------------- */
/* controllers: OSC_POSE arm + GRIP gripper (restates the reference Python, see file header)   */
/* ------------------------------------------------------------------------------------------- */
#define ARM_MAX 8
typedef struct {
  int ndof;                 /* arm dofs (7) */
  int qpos_idx[ARM_MAX], dof_idx[ARM_MAX], act_idx[ARM_MAX];
  int eef_site, base_site;
  double kp[6], kd[6];
  double in_min[ARM_MAX], in_max[ARM_MAX], out_min[ARM_MAX], out_max[ARM_MAX];
  int uncouple;
  /* part-controller type: 0 OSC_POSE (osc.py), 1 OSC_POSITION (osc.py use_ori=False), 2 JOINT_POSITION (generic/joint_pos.py),
   * 3 JOINT_TORQUE (generic/joint_tor.py), 4 JOINT_VELOCITY (generic/joint_vel.py, with the constructor defect of the surveyed snapshot
   * resolved as use_torque_compensation = True); cdim = control_dim of the arm part */
  int type, cdim;
  double jkp[ARM_MAX], jkd[ARM_MAX], tl_lo[ARM_MAX], tl_hi[ARM_MAX], goal_j[ARM_MAX];
  /* JOINT_VELOCITY PID state (joint_vel.py:105-110): RingBuffer(dim, 5) of error increments (utils/buffers.py:24-91) */
  double last_err[ARM_MAX], summed_err[ARM_MAX], ring[5][ARM_MAX];
  int ring_ptr, ring_size, saturated;
  /* impedance mode (osc.py:243-253, joint_pos.py:204-214): 0 fixed, 1 variable, 2 variable_kp; limits per gain */
  int imp_mode;
  double kp_min[ARM_MAX], kp_max[ARM_MAX], dr_min[ARM_MAX], dr_max[ARM_MAX];
  /* LinearInterpolator (utils/traj_utils.py:25-155): total steps (0 = none), start vector, step counter; the goal is goal_j / goal_pos */
  int interp_total, interp_step;
  double interp_start[ARM_MAX];
  /* OSC_POSE orientation interpolator (controller_factory.py:102-106: a deepcopy in 'euler' mode): start / goal are orientation-error
   * vectors that get_interpolated_goal treats as Euler angles (osc.py:277-283, traj_utils.py:129-146) */
  double interp_ori_start[3], interp_ori_goal[3];
  double nullspace_kp;
  /* gripper */
  int ngrip;               /* number of gripper actuators (2) */
  int grip_act[4];
  double grip_sign[4];     /* format_action direction per actuator: [-1, +1] (panda_gripper.py:55-57) */
  double grip_speed;       /* 0.2 */
  /* state */
  double goal_pos[3], goal_ori[9], initial_joint[ARM_MAX], grip_action[4], grip_goal[4];
  double torques[ARM_MAX];
} rso_ctrl;

rso_ctrl *rso_ctrl_create(void) { return (rso_ctrl *)calloc(1, sizeof(rso_ctrl)); }
void rso_ctrl_free(rso_ctrl *c) { free(c); }
void rso_ctrl_config(rso_ctrl *c, int ndof, const int *qpos_idx, const int *dof_idx, const int *act_idx, int eef_site, int base_site, const double *kp,
                     double damping_ratio, const double *in_min, const double *in_max, const double *out_min, const double *out_max, int uncouple,
                     int ngrip, const int *grip_act, const double *grip_sign, double grip_speed) {
  c->ndof = ndof;
  for (int i = 0; i < ndof; i++) { c->qpos_idx[i] = qpos_idx[i]; c->dof_idx[i] = dof_idx[i]; c->act_idx[i] = act_idx[i]; }
  c->eef_site = eef_site; c->base_site = base_site;
  for (int i = 0; i < 6; i++) {
    c->kp[i] = kp[i]; c->kd[i] = 2 * sqrt(kp[i]) * damping_ratio; /* osc.py:177-178 */
    c->in_min[i] = in_min[i]; c->in_max[i] = in_max[i]; c->out_min[i] = out_min[i]; c->out_max[i] = out_max[i];
  }
  c->uncouple = uncouple;
  c->nullspace_kp = 10; /* control_utils.py:7 default joint_kp */
  c->ngrip = ngrip;
  for (int i = 0; i < ngrip; i++) { c->grip_act[i] = grip_act[i]; c->grip_sign[i] = grip_sign[i]; }
  c->grip_speed = grip_speed;
  c->type = 0; c->cdim = 6;
}

/* the other arm part-controller types of controller_factory.py:73-159: per-joint gains / scaling for the joint-space ones
 * (joint_pos.py:135-175: kd = 2 sqrt(kp) damping_ratio; joint_tor.py:95-96: torque_limits default to the actuator ctrlrange) */
void rso_ctrl_set_type(rso_ctrl *c, int type, int cdim, const double *jkp, double damping_ratio, const double *in_min, const double *in_max,
                       const double *out_min, const double *out_max, const double *tl_lo, const double *tl_hi) {
  c->type = type; c->cdim = cdim;
  for (int i = 0; i < cdim; i++) { c->in_min[i] = in_min[i]; c->in_max[i] = in_max[i]; c->out_min[i] = out_min[i]; c->out_max[i] = out_max[i]; }
  if (type == 2) for (int i = 0; i < c->ndof; i++) { c->jkp[i] = jkp[i]; c->jkd[i] = 2 * sqrt(jkp[i]) * damping_ratio; }
  if (type == 4) for (int i = 0; i < c->ndof; i++) c->jkp[i] = jkp[i];   /* joint_vel.py:96-103: kp (x (high - low) when scalar), ki = 0.005 kp, kd = 0.001 kp */
  if (type == 3 || type == 4) for (int i = 0; i < c->ndof; i++) { c->tl_lo[i] = tl_lo[i]; c->tl_hi[i] = tl_hi[i]; }  /* torque / velocity limits */
}

void rso_ctrl_set_interpolator(rso_ctrl *c, int total_steps) {
  c->interp_total = total_steps; c->interp_step = 0; memset(c->interp_start, 0, sizeof(c->interp_start));
  memset(c->interp_ori_start, 0, sizeof(c->interp_ori_start)); memset(c->interp_ori_goal, 0, sizeof(c->interp_ori_goal));
}

/* LinearInterpolator.get_interpolated_goal for one component set; advances the step counter once per call */
static void interp_get(rso_ctrl *c, const double *goal, int n, double *out) {
  for (int i = 0; i < n; i++) out[i] = c->interp_start[i] + (goal[i] - c->interp_start[i]) / (double)(c->interp_total - c->interp_step);
  if (c->interp_step < c->interp_total - 1) c->interp_step++;
}

void rso_ctrl_set_impedance(rso_ctrl *c, int mode, const double *kp_min, const double *kp_max, const double *dr_min, const double *dr_max) {
  c->imp_mode = mode;
  int n = c->type >= 2 ? c->ndof : 6;
  for (int i = 0; i < n; i++) { c->kp_min[i] = kp_min[i]; c->kp_max[i] = kp_max[i]; c->dr_min[i] = dr_min[i]; c->dr_max[i] = dr_max[i]; }
}

void rso_osc_goal(const double *scaled, const double *ep, const double *eR, const double *op, const double *oR, double *goal_pos, double *goal_ori);

/* Controller.reset at env reset: initial_joint (controller.py:128-130), goals = current world pose (osc.py:520-532),
 * gripper current_action = 0 (robots/robot.py:289) */
void rso_ctrl_reset(rso_ctrl *c, rso_data *d) {
  for (int i = 0; i < c->ndof; i++) c->initial_joint[i] = d->qpos[c->qpos_idx[i]];
  memcpy(c->goal_pos, d->site_xpos + 3 * c->eef_site, sizeof(c->goal_pos));
  memcpy(c->goal_ori, d->site_xmat + 9 * c->eef_site, sizeof(c->goal_ori));
  for (int i = 0; i < 4; i++) { c->grip_action[i] = 0; c->grip_goal[i] = 0; }
  /* joint_pos.py:268-276 reset_goal: goal_qpos = joint_pos; joint_tor.py:170-178: goal_torque = 0 */
  for (int i = 0; i < c->ndof; i++) c->goal_j[i] = c->type == 2 ? d->qpos[c->qpos_idx[i]] : 0.0;
  memset(c->last_err, 0, sizeof(c->last_err)); memset(c->summed_err, 0, sizeof(c->summed_err)); memset(c->ring, 0, sizeof(c->ring));
  c->ring_ptr = 4; c->ring_size = 0; c->saturated = 0;
  memset(c->interp_start, 0, sizeof(c->interp_start)); c->interp_step = 0;   /* fresh interpolator, then set_goal(reset goal): start = zeros */
  memset(c->interp_ori_start, 0, sizeof(c->interp_ori_start)); memset(c->interp_ori_goal, 0, sizeof(c->interp_ori_goal));   /* osc.py:538-544: error of ref vs itself */
}

/* float32 quat2mat of the reference (transform_utils.py:461-487 casts to float32; under NumPy>=2 the
 * whole expression evaluates in float32 because python scalars are weakly typed) */
static void quat2mat_f32(const double *q_xyzw, double *R) {
  float q[4] = {(float)q_xyzw[3], (float)q_xyzw[0], (float)q_xyzw[1], (float)q_xyzw[2]};
  float n = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (n < 8.881784197001252e-16f) { memset(R, 0, 9 * sizeof(double)); R[0] = R[4] = R[8] = 1; return; } /* EPS = float64 eps * 4 */
  float inv = 2.0f / n;
  double s = sqrt((double)inv);
  for (int k = 0; k < 4; k++) q[k] = (float)(q[k] * s);
  float q2[4][4];
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) q2[i][j] = q[i] * q[j];
  R[0] = (float)(1.0f - q2[2][2] - q2[3][3]); R[1] = (float)(q2[1][2] - q2[3][0]); R[2] = (float)(q2[1][3] + q2[2][0]);
  R[3] = (float)(q2[1][2] + q2[3][0]); R[4] = (float)(1.0f - q2[1][1] - q2[3][3]); R[5] = (float)(q2[2][3] - q2[1][0]);
  R[6] = (float)(q2[1][3] - q2[2][0]); R[7] = (float)(q2[2][3] + q2[1][0]); R[8] = (float)(1.0f - q2[1][1] - q2[2][2]);
}

static void mat3_mul(double *r, const double *a, const double *b) {
  double t[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  memcpy(r, t, sizeof(t));
}
static void mat3T_mul(double *r, const double *a, const double *b) { /* a^T b */
  double t[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t[3 * i + j] = a[i] * b[j] + a[3 + i] * b[3 + j] + a[6 + i] * b[6 + j];
  memcpy(r, t, sizeof(t));
}

static void orientation_error(const double *desired, const double *current, double *err);

/* ---- orientation interpolator of OSC_POSE: LinearInterpolator in 'euler' mode (traj_utils.py:129-146) ------------------------------------
 * x = mat2quat(euler2mat(start)); g = mat2quat(euler2mat(goal)); q = quat_slerp(x, g, (step + 1) / total); out = mat2euler(quat2mat(q)) */
static void euler2mat(const double *e, double *R) { /* transform_utils.py:358-391 */
  double ai = -e[2], aj = -e[1], ak = -e[0];
  double si = sin(ai), sj = sin(aj), sk = sin(ak), ci = cos(ai), cj = cos(aj), ck = cos(ak);
  double cc = ci * ck, cs = ci * sk, sc = si * ck, ss = si * sk;
  R[8] = cj * ck; R[7] = sj * sc - cs; R[6] = sj * cc + ss;
  R[5] = cj * sk; R[4] = sj * ss + cc; R[3] = sj * cs - sc;
  R[2] = -sj;     R[1] = cj * si;      R[0] = cj * ci;
}
/* transform_utils.py:316-355 takes the eigenvector of the largest eigenvalue of the 4 x 4 matrix K built from the float32 matrix, i.e. the
 * unit quaternion of the rotation, made unique by w >= 0.  Restated with the branch-on-largest-component extraction (same quaternion for a
 * rotation matrix; the reference's float32 eigh noise of ~1e-7 is inside the test tolerance).  Output (x, y, z, w). */
static void mat2quat_f32(const double *Rd, double *q) {
  float m[9];
  for (int i = 0; i < 9; i++) m[i] = (float)Rd[i];
  double tr = (double)m[0] + m[4] + m[8], w, x, y, z;
  if (tr > 0) { double s = 2 * sqrt(1 + tr); w = 0.25 * s; x = (m[7] - m[5]) / s; y = (m[2] - m[6]) / s; z = (m[3] - m[1]) / s; }
  else if (m[0] > m[4] && m[0] > m[8]) { double s = 2 * sqrt(1 + m[0] - m[4] - m[8]); w = (m[7] - m[5]) / s; x = 0.25 * s; y = (m[1] + m[3]) / s; z = (m[2] + m[6]) / s; }
  else if (m[4] > m[8]) { double s = 2 * sqrt(1 + m[4] - m[0] - m[8]); w = (m[2] - m[6]) / s; x = (m[1] + m[3]) / s; y = 0.25 * s; z = (m[5] + m[7]) / s; }
  else { double s = 2 * sqrt(1 + m[8] - m[0] - m[4]); w = (m[3] - m[1]) / s; x = (m[2] + m[6]) / s; y = (m[5] + m[7]) / s; z = 0.25 * s; }
  double n = sqrt(w * w + x * x + y * y + z * z), sg = w < 0 ? -1.0 : 1.0;
  q[0] = sg * x / n; q[1] = sg * y / n; q[2] = sg * z / n; q[3] = sg * w / n;
}
static void quat_slerp(const double *a, const double *b, double fraction, double *out) { /* transform_utils.py:151-201, shortestpath = True */
  const double EPS = 8.881784197001252e-16;
  double q0[4], q1[4], n0 = 0, n1 = 0, d = 0;
  for (int i = 0; i < 4; i++) { n0 += a[i] * a[i]; n1 += b[i] * b[i]; }
  for (int i = 0; i < 4; i++) { q0[i] = a[i] / sqrt(n0); q1[i] = b[i] / sqrt(n1); d += q0[i] * q1[i]; }
  if (fraction == 0.0) { memcpy(out, q0, sizeof(q0)); return; }
  if (fraction == 1.0) { memcpy(out, q1, sizeof(q1)); return; }
  if (fabs(fabs(d) - 1.0) < EPS) { memcpy(out, q0, sizeof(q0)); return; }
  if (d < 0.0) { d = -d; for (int i = 0; i < 4; i++) q1[i] = -q1[i]; }
  double angle = acos(fmax(-1.0, fmin(1.0, d)));
  if (fabs(angle) < EPS) { memcpy(out, q0, sizeof(q0)); return; }
  double isin = 1.0 / sin(angle), w0 = sin((1.0 - fraction) * angle) * isin, w1 = sin(fraction * angle) * isin;
  for (int i = 0; i < 4; i++) out[i] = q0[i] * w0 + q1[i] * w1;
}
static void mat2euler_sxyz(const double *Rd, double *e) { /* transform_utils.py:394-440, axes 'sxyz' = (0, 0, 0, 0): i, j, k = 0, 1, 2; float32 input */
  float M[9];
  for (int i = 0; i < 9; i++) M[i] = (float)Rd[i];
  double cy = sqrt((double)(float)(M[0] * M[0] + M[3] * M[3]));
  if (cy > 8.881784197001252e-16) { e[0] = atan2(M[7], M[8]); e[1] = atan2(-M[6], cy); e[2] = atan2(M[3], M[0]); }
  else { e[0] = atan2(-M[5], M[4]); e[1] = atan2(-M[6], cy); e[2] = 0; }
}
static void interp_ori_get(const rso_ctrl *c, int step, double *out) {
  double R[9], x[4], g[4], q[4];
  euler2mat(c->interp_ori_start, R); mat2quat_f32(R, x);
  euler2mat(c->interp_ori_goal, R); mat2quat_f32(R, g);
  quat_slerp(x, g, (double)(step + 1) / (double)c->interp_total, q);
  quat2mat_f32(q, R);
  mat2euler_sxyz(R, out);
}

/* set_goal at a policy step: OSC (osc.py:225-283, 306-401, mode "achieved", frame "base", delta input) + gripper
 * (composite_controller.py:97-103 -> panda_gripper.py:43-58 -> simple_grip.py:110-148) */
void rso_ctrl_set_goal(rso_ctrl *c, rso_data *d, const double *action) {
  if (c->imp_mode) { /* [damping_ratio x n, kp x n, goal update] (mode 1) or [kp x n, goal update] (mode 2) */
    int n = c->type >= 2 ? c->ndof : 6;
    double *kp = c->type >= 2 ? c->jkp : c->kp, *kd = c->type >= 2 ? c->jkd : c->kd;
    for (int i = 0; i < n; i++) {
      kp[i] = fmax(c->kp_min[i], fmin(c->kp_max[i], action[(c->imp_mode == 1 ? n : 0) + i]));
      kd[i] = 2 * sqrt(kp[i]) * (c->imp_mode == 1 ? fmax(c->dr_min[i], fmin(c->dr_max[i], action[i])) : 1.0);
    }
    action += n * (c->imp_mode == 1 ? 2 : 1);
  }
  double scaled[ARM_MAX] = {0};
  for (int i = 0; i < c->cdim; i++) { /* controller.py:149-168 */
    double scale = fabs(c->out_max[i] - c->out_min[i]) / fabs(c->in_max[i] - c->in_min[i]);
    double a = fmax(c->in_min[i], fmin(c->in_max[i], action[i]));
    scaled[i] = (a - 0.5 * (c->in_max[i] + c->in_min[i])) * scale + 0.5 * (c->out_max[i] + c->out_min[i]);
  }
  if (c->interp_total) {     /* LinearInterpolator.set_goal: start := previous goal, step := 0 (traj_utils.py:101-116) */
    if (c->type >= 2) memcpy(c->interp_start, c->goal_j, sizeof(double) * c->ndof);
    else memcpy(c->interp_start, c->goal_pos, sizeof(double) * 3);
    c->interp_step = 0;
  }
  if (c->type == 2) {        /* joint_pos.py:200-236: goal_qpos = joint_pos + scaled delta (no position limits) */
    for (int i = 0; i < c->ndof; i++) c->goal_j[i] = d->qpos[c->qpos_idx[i]] + scaled[i];
  } else if (c->type == 3 || c->type == 4) { /* joint_tor.py:111-128 / joint_vel.py:145-148: goal = clip(scale_action(a), torque / velocity limits) */
    for (int i = 0; i < c->ndof; i++) c->goal_j[i] = fmax(c->tl_lo[i], fmin(c->tl_hi[i], scaled[i]));
  } else {                   /* osc.py:255-263: OSC_POSITION passes a zero orientation delta (scaled[3..5] stay 0) */
    rso_osc_goal(scaled, d->site_xpos + 3 * c->eef_site, d->site_xmat + 9 * c->eef_site, d->site_xpos + 3 * c->base_site, d->site_xmat + 9 * c->base_site,
                 c->goal_pos, c->goal_ori);
    if (c->interp_total && c->type == 0) {   /* osc.py:277-283: ori_ref = current eef orientation; goal = error of the (base-frame) goal_ori against it */
      memcpy(c->interp_ori_start, c->interp_ori_goal, sizeof(c->interp_ori_start));
      orientation_error(c->goal_ori, d->site_xmat + 9 * c->eef_site, c->interp_ori_goal);
    }
  }
  /* gripper */
  if (c->ngrip > 0) {
    double a = action[c->cdim], sg = a > 0 ? 1.0 : (a < 0 ? -1.0 : 0.0);
    for (int i = 0; i < c->ngrip; i++) {
      c->grip_action[i] = fmax(-1.0, fmin(1.0, c->grip_action[i] + c->grip_sign[i] * c->grip_speed * sg));
      c->grip_goal[i] = c->grip_action[i];
    }
  }
}

/* symmetric pseudo-inverse via Jacobi eigen-decomposition, numpy.linalg.pinv semantics (rcond = 1e-15) */
static void sym_pinv(const double *A, double *P, int n) {
  double a[36], V[36];
  memcpy(a, A, sizeof(double) * n * n);
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) V[i * n + j] = i == j;
  for (int sweep = 0; sweep < 100; sweep++) {
    double off = 0;
    for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) off += a[i * n + j] * a[i * n + j];
    if (off < 1e-300) break;
    for (int p = 0; p < n; p++)
      for (int q = p + 1; q < n; q++) {
        if (fabs(a[p * n + q]) < 1e-300) continue;
        double th = (a[q * n + q] - a[p * n + p]) / (2 * a[p * n + q]);
        double t = (th >= 0 ? 1 : -1) / (fabs(th) + sqrt(th * th + 1)), cs = 1 / sqrt(t * t + 1), sn = t * cs;
        for (int k = 0; k < n; k++) { double akp = a[k * n + p], akq = a[k * n + q]; a[k * n + p] = cs * akp - sn * akq; a[k * n + q] = sn * akp + cs * akq; }
        for (int k = 0; k < n; k++) { double apk = a[p * n + k], aqk = a[q * n + k]; a[p * n + k] = cs * apk - sn * aqk; a[q * n + k] = sn * apk + cs * aqk; }
        for (int k = 0; k < n; k++) { double vkp = V[k * n + p], vkq = V[k * n + q]; V[k * n + p] = cs * vkp - sn * vkq; V[k * n + q] = sn * vkp + cs * vkq; }
      }
  }
  double wmax = 0;
  for (int i = 0; i < n; i++) wmax = fmax(wmax, fabs(a[i * n + i]));
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      double s = 0;
      for (int k = 0; k < n; k++) { double w = a[k * n + k]; if (fabs(w) > 1e-15 * wmax) s += V[i * n + k] * V[j * n + k] / w; }
      P[i * n + j] = s;
    }
}

/* OSC torque law on explicit inputs (osc.py:403-495 + control_utils.py:7-111); pre-clip torques out[n].
 *   ep/eR eef site pose, ev[6] eef lin+ang velocity, op/oR base (origin) site pose, bv[6] base site velocity,
 *   goal_pos/goal_ori in the origin frame, J 6 x n (rows: jacp then jacr, arm columns), M n x n arm sub-block of the
 *   full mass matrix, bias = qfrc_bias[arm], q/qd arm joint pos/vel, q0 nullspace reference (initial_joint). */
/* orientation_error(desired, current) = 1/2 sum_i current[:, i] x desired[:, i]  (control_utils.py:85-111) */
static void orientation_error(const double *desired, const double *current, double *err) {
  err[0] = err[1] = err[2] = 0;
  for (int col = 0; col < 3; col++) {
    double rc[3] = {current[col], current[3 + col], current[6 + col]}, rd[3] = {desired[col], desired[3 + col], desired[6 + col]}, t[3];
    cross3(t, rc, rd);
    for (int k = 0; k < 3; k++) err[k] += 0.5 * t[k];
  }
}

/* oerr_in != NULL: the orientation error comes from the orientation interpolator (osc.py:433-437) instead of goal_ori */
static void osc_torques_impl(const double *kp, const double *kd, const double *ep, const double *eR, const double *ev, const double *op, const double *oR,
                             const double *bv, const double *goal_pos, const double *goal_ori, const double *oerr_in, const double *J, const double *M,
                             const double *bias, const double *q, const double *qd, const double *q0, double nullspace_kp, int uncouple, int n, double *out);
void rso_osc_torques(const double *kp, const double *kd, const double *ep, const double *eR, const double *ev, const double *op, const double *oR,
                     const double *bv, const double *goal_pos, const double *goal_ori, const double *J, const double *M, const double *bias,
                     const double *q, const double *qd, const double *q0, double nullspace_kp, int uncouple, int n, double *out) {
  osc_torques_impl(kp, kd, ep, eR, ev, op, oR, bv, goal_pos, goal_ori, NULL, J, M, bias, q, qd, q0, nullspace_kp, uncouple, n, out);
}
static void osc_torques_impl(const double *kp, const double *kd, const double *ep, const double *eR, const double *ev, const double *op, const double *oR,
                             const double *bv, const double *goal_pos, const double *goal_ori, const double *oerr_in, const double *J, const double *M,
                             const double *bias, const double *q, const double *qd, const double *q0, double nullspace_kp, int uncouple, int n, double *out) {
  double Minv[ARM_MAX * ARM_MAX];
  double dpos[3], dori[9], perr[3], oerr[3] = {0, 0, 0};
  mat_vec3(dpos, oR, goal_pos);
  for (int k = 0; k < 3; k++) { dpos[k] += op[k]; perr[k] = dpos[k] - ep[k]; }
  mat3_mul(dori, oR, goal_ori);
  if (oerr_in) memcpy(oerr, oerr_in, sizeof(oerr));
  else orientation_error(dori, eR, oerr);
  double F[3], T[3];
  for (int k = 0; k < 3; k++) {
    F[k] = perr[k] * kp[k] + (-(ev[k] - bv[k])) * kd[k];
    T[k] = oerr[k] * kp[3 + k] + (-(ev[3 + k] - bv[3 + k])) * kd[3 + k];
  }
  /* opspace_matrices (control_utils.py:43-82) */
  {
    double A[ARM_MAX * ARM_MAX], b[ARM_MAX];
    for (int col = 0; col < n; col++) { /* np.linalg.inv: solve for unit vectors */
      memcpy(A, M, sizeof(double) * n * n);
      for (int k = 0; k < n; k++) b[k] = k == col;
      solve_small(A, b, n);
      for (int k = 0; k < n; k++) Minv[k * n + col] = b[k];
    }
  }
  double MiJT[ARM_MAX * 6], lfi[36], lpi[9], loi[9], lf[36], lp[9], lo[9];
  for (int i = 0; i < n; i++) for (int r = 0; r < 6; r++) { double s = 0; for (int k = 0; k < n; k++) s += Minv[i * n + k] * J[r * n + k]; MiJT[i * 6 + r] = s; }
  for (int r = 0; r < 6; r++) for (int s2 = 0; s2 < 6; s2++) { double s = 0; for (int k = 0; k < n; k++) s += J[r * n + k] * MiJT[k * 6 + s2]; lfi[r * 6 + s2] = s; }
  for (int r = 0; r < 3; r++) for (int s2 = 0; s2 < 3; s2++) { lpi[r * 3 + s2] = lfi[r * 6 + s2]; loi[r * 3 + s2] = lfi[(3 + r) * 6 + 3 + s2]; }
  sym_pinv(lfi, lf, 6); sym_pinv(lpi, lp, 3); sym_pinv(loi, lo, 3);
  double wrench[6];
  if (uncouple) { mat_vec3(wrench, lp, F); mat_vec3(wrench + 3, lo, T); }
  else { double w[6] = {F[0], F[1], F[2], T[0], T[1], T[2]}; for (int r = 0; r < 6; r++) { double s = 0; for (int k = 0; k < 6; k++) s += lf[r * 6 + k] * w[k]; wrench[r] = s; } }
  /* nullspace: N = I - Jbar J, Jbar = Minv J^T lambda_full; torques += N^T M (kp (q0-q) - kv qd) */
  double Jbar[ARM_MAX * 6], N[ARM_MAX * ARM_MAX], pt[ARM_MAX], tmp[ARM_MAX];
  for (int i = 0; i < n; i++) for (int r = 0; r < 6; r++) { double s = 0; for (int k = 0; k < 6; k++) s += MiJT[i * 6 + k] * lf[k * 6 + r]; Jbar[i * 6 + r] = s; }
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { double s = 0; for (int r = 0; r < 6; r++) s += Jbar[i * 6 + r] * J[r * n + j]; N[i * n + j] = (i == j) - s; }
  double kv = sqrt(nullspace_kp) * 2;
  for (int i = 0; i < n; i++) tmp[i] = nullspace_kp * (q0[i] - q[i]) - kv * qd[i];
  for (int i = 0; i < n; i++) { double s = 0; for (int k = 0; k < n; k++) s += M[i * n + k] * tmp[k]; pt[i] = s; }
  for (int i = 0; i < n; i++) {
    double tq = bias[i];
    for (int r = 0; r < 6; r++) tq += J[r * n + i] * wrench[r];
    for (int k = 0; k < n; k++) tq += N[k * n + i] * pt[k];
    out[i] = tq;
  }
}

/* OSC goal update on explicit inputs (osc.py:306-401; "achieved" mode, "base" frame): scaled[6] = scale_action(action) */
void rso_osc_goal(const double *scaled, const double *ep, const double *eR, const double *op, const double *oR, double *goal_pos, double *goal_ori) {
  double rel[3], loc[3];
  for (int k = 0; k < 3; k++) rel[k] = ep[k] - op[k];
  matT_vec3(loc, oR, rel); /* world_to_origin_frame */
  for (int k = 0; k < 3; k++) goal_pos[k] = loc[k] + scaled[k];
  double ang = norm3(scaled + 3), qe[4] = {0, 0, 0, 1}, Rerr[9], cur[9];
  if (!(fabs(ang) <= 1e-9 * fmax(fabs(ang), 0.0))) { /* math.isclose(angle, 0.0): only exact zero */
    double s = sin(0.5 * ang);
    qe[0] = scaled[3] / ang * s; qe[1] = scaled[4] / ang * s; qe[2] = scaled[5] / ang * s; qe[3] = cos(0.5 * ang);
  }
  quat2mat_f32(qe, Rerr);
  mat3T_mul(cur, oR, eR);
  mat3_mul(goal_ori, Rerr, cur);
}

/* run_controller for one substep: gathers from sim data, writes clipped ctrl (controller.py:170-232, fixed_base_robot.py:143-153) */
void rso_ctrl_run(rso_ctrl *c, rso_data *d) {
  rso_model *m = d->m;
  int nv = m->nv, n = c->ndof;
  double *jp = dalloc(3 * nv), *jr = dalloc(3 * nv), *bjp = dalloc(3 * nv), *bjr = dalloc(3 * nv);
  rso_jac_site(d, c->eef_site, jp, jr);
  rso_jac_site(d, c->base_site, bjp, bjr);
  double J[6 * ARM_MAX], M[ARM_MAX * ARM_MAX], q[ARM_MAX], qd[ARM_MAX], bias[ARM_MAX];
  double ev[6] = {0}, bv[6] = {0};
  for (int r = 0; r < 3; r++)
    for (int k = 0; k < nv; k++) { ev[r] += jp[r * nv + k] * d->qvel[k]; ev[3 + r] += jr[r * nv + k] * d->qvel[k]; bv[r] += bjp[r * nv + k] * d->qvel[k]; bv[3 + r] += bjr[r * nv + k] * d->qvel[k]; }
  for (int i = 0; i < n; i++) {
    q[i] = d->qpos[c->qpos_idx[i]]; qd[i] = d->qvel[c->dof_idx[i]]; bias[i] = d->qfrc_bias[c->dof_idx[i]];
    for (int r = 0; r < 3; r++) { J[r * n + i] = jp[r * nv + c->dof_idx[i]]; J[(3 + r) * n + i] = jr[r * nv + c->dof_idx[i]]; }
    for (int j = 0; j < n; j++) M[i * n + j] = d->qM[c->dof_idx[i] * nv + c->dof_idx[j]];
  }
  double gj[ARM_MAX];
  memcpy(gj, c->goal_j, sizeof(gj));
  if (c->interp_total && c->type >= 2) interp_get(c, c->goal_j, n, gj);
  if (c->type == 2) {        /* joint_pos.py:238-266: M_arm (kp (goal - q) - kd qd) + qfrc_bias[arm] */
    for (int i = 0; i < n; i++) {
      double t = bias[i];
      for (int j = 0; j < n; j++) t += M[i * n + j] * (c->jkp[j] * (gj[j] - q[j]) - c->jkd[j] * qd[j]);
      c->torques[i] = t;
    }
  } else if (c->type == 3) { /* joint_tor.py:130-167: goal_torque + qfrc_bias[arm] */
    for (int i = 0; i < n; i++) c->torques[i] = gj[i] + bias[i];
  } else if (c->type == 4) { /* joint_vel.py:166-198 */
    c->ring_ptr = (c->ring_ptr + 1) % 5;
    if (c->ring_size < 5) c->ring_size++;
    int sat = 0;
    for (int i = 0; i < n; i++) {
      double err = gj[i] - qd[i], derr = err - c->last_err[i];
      c->last_err[i] = err;
      c->ring[c->ring_ptr][i] = derr;
      if (!c->saturated) c->summed_err[i] += err;
      double avg = 0;
      for (int k = 0; k < c->ring_size; k++) avg += c->ring[k][i];
      avg /= c->ring_size;
      double t = c->jkp[i] * err + 0.005 * c->jkp[i] * c->summed_err[i] + 0.001 * c->jkp[i] * avg + bias[i];
      int a = c->act_idx[i];
      double cl = fmax(m->actuator_ctrlrange[2 * a], fmin(m->actuator_ctrlrange[2 * a + 1], t));   /* clip_torques, controller.py:264-274 */
      if (cl != t) sat = 1;
      c->torques[i] = cl;
    }
    c->saturated = sat;
  } else {
    /* osc.py:418-423: with an interpolator the ramped goal values are taken as the desired WORLD position (although set_goal stores base-frame
     * coordinates); expressed here as the base-frame goal that maps onto that world point */
    double gp[3] = {c->goal_pos[0], c->goal_pos[1], c->goal_pos[2]};
    double oerr[3];
    const int use_ori = c->interp_total && c->type == 0;
    if (use_ori) interp_ori_get(c, c->interp_step, oerr);   /* both interpolators count the same steps (deepcopy, stepped once per run each) */
    if (c->interp_total) {
      double des[3], rel[3];
      const double *op = d->site_xpos + 3 * c->base_site, *oR = d->site_xmat + 9 * c->base_site;
      interp_get(c, c->goal_pos, 3, des);
      for (int k = 0; k < 3; k++) rel[k] = des[k] - op[k];
      matT_vec3(gp, oR, rel);
    }
    osc_torques_impl(c->kp, c->kd, d->site_xpos + 3 * c->eef_site, d->site_xmat + 9 * c->eef_site, ev, d->site_xpos + 3 * c->base_site,
                     d->site_xmat + 9 * c->base_site, bv, gp, c->goal_ori, use_ori ? oerr : NULL, J, M, bias, q, qd, c->initial_joint, c->nullspace_kp,
                     c->uncouple, n, c->torques);
  }
  for (int i = 0; i < n; i++) {
    int a = c->act_idx[i];
    d->ctrl[a] = fmax(m->actuator_ctrlrange[2 * a], fmin(m->actuator_ctrlrange[2 * a + 1], c->torques[i]));
  }
  /* gripper: ctrl = bias + weight * goal, clipped (simple_grip.py:150-186) */
  for (int i = 0; i < c->ngrip; i++) {
    int a = c->grip_act[i];
    double lo_ = m->actuator_ctrlrange[2 * a], hi_ = m->actuator_ctrlrange[2 * a + 1];
    double v = 0.5 * (hi_ + lo_) + 0.5 * (hi_ - lo_) * c->grip_goal[i];
    d->ctrl[a] = fmax(lo_, fmin(hi_, v));
  }
  free(jp); free(jr); free(bjp); free(bjr);
}
double *rso_ctrl_torques(rso_ctrl *c) { return c->torques; }
double *rso_ctrl_goal(rso_ctrl *c) { return c->goal_pos; } /* goal_pos[3] followed by goal_ori[9] */

/* one env.step(): n_sub x { step1, control(policy_step = i==0), step2 }  (environments/base.py:494-504) */
void rso_env_step(rso_ctrl *c, rso_data *d, const double *action, int n_sub) {
  for (int i = 0; i < n_sub; i++) {
    rso_step1(d);
    if (i == 0) rso_ctrl_set_goal(c, d, action);
    rso_ctrl_run(c, d);
    rso_step2(d);
  }
}
