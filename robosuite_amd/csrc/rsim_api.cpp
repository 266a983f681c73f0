// rsim_api.cpp -- host side of the C-ABI (include/rsim.h): model blob ingest, table packing, batch memory, launches.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and enums only: the functions are resolved with dlsym from the librccl.so the process already holds
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/rsim.h"
#include "rsim_internal.h"

// one set of launchers per compiled kernel configuration (rsim_step.hip is built once per RSIM_CFG)
#define RSIM_NCFG 5
#define RSIM_MAX_GROUPS 32
extern "C" int rsim_launch_step_cfg0(const DModel* m, const DBatch* b, const float* actions, int n_sub, int flags, hipStream_t stream);
extern "C" int rsim_launch_ctrl_reset_cfg0(const DModel* m, const DBatch* b, const unsigned char* mask, hipStream_t stream);
extern "C" int rsim_limits_cfg0(int* lim);
extern "C" int rsim_launch_step_cfg1(const DModel* m, const DBatch* b, const float* actions, int n_sub, int flags, hipStream_t stream);
extern "C" int rsim_launch_ctrl_reset_cfg1(const DModel* m, const DBatch* b, const unsigned char* mask, hipStream_t stream);
extern "C" int rsim_limits_cfg1(int* lim);
extern "C" int rsim_launch_step_cfg2(const DModel* m, const DBatch* b, const float* actions, int n_sub, int flags, hipStream_t stream);
extern "C" int rsim_launch_ctrl_reset_cfg2(const DModel* m, const DBatch* b, const unsigned char* mask, hipStream_t stream);
extern "C" int rsim_limits_cfg2(int* lim);
extern "C" int rsim_launch_step_cfg3(const DModel* m, const DBatch* b, const float* actions, int n_sub, int flags, hipStream_t stream);
extern "C" int rsim_launch_step_cfg4(const DModel* m, const DBatch* b, const float* actions, int n_sub, int flags, hipStream_t stream);
extern "C" int rsim_launch_ctrl_reset_cfg3(const DModel* m, const DBatch* b, const unsigned char* mask, hipStream_t stream);
extern "C" int rsim_launch_ctrl_reset_cfg4(const DModel* m, const DBatch* b, const unsigned char* mask, hipStream_t stream);
extern "C" int rsim_limits_cfg3(int* lim);
extern "C" int rsim_limits_cfg4(int* lim);
// configuration 5 serves no model of its own: it is the capacity tier above configuration 3 (64 contacts x 256 rows, rsim_step.hip)
#define RSIM_NCFG_ALL 8
#define RSIM_TIER_DECL(c) \
  extern "C" int rsim_limits_cfg##c(int* lim); extern "C" int rsim_cmem_bytes_cfg##c(void); \
  extern "C" int rsim_launch_prepare_cfg##c(const DModel* m, const DBatch* b, int nblocks, int reset_only, hipStream_t stream); \
  extern "C" int rsim_launch_step_list_cfg##c(const DModel* m, const DBatch* b, const float* actions, int n_sub, int flags, int grid, hipStream_t stream);
RSIM_TIER_DECL(5) RSIM_TIER_DECL(6) RSIM_TIER_DECL(7)
extern "C" int rsim_launch_step_list_cfg3(const DModel* m, const DBatch* b, const float* actions, int n_sub, int flags, int grid, hipStream_t stream);
extern "C" int rsim_launch_tier_list(const int* tier, int* list, int* count, int* zero_next, int env0, int n, hipStream_t stream);
extern "C" int rsim_limits_w_cfg0(int* lim);   // limits of the wide body compiled into configuration 0's / 1's control-step kernel (fused tier); 0: this build has none
extern "C" int rsim_limits_w_cfg1(int* lim);
extern "C" int rsim_limits_w_cfg2(int* lim);
typedef int (*limits_w_fn)(int*);
static const limits_w_fn k_limits_w[3] = {rsim_limits_w_cfg0, rsim_limits_w_cfg1, rsim_limits_w_cfg2};
typedef int (*step_list_fn)(const DModel*, const DBatch*, const float*, int, int, int, hipStream_t);
typedef int (*step_fn)(const DModel*, const DBatch*, const float*, int, int, hipStream_t);
typedef int (*creset_fn)(const DModel*, const DBatch*, const unsigned char*, hipStream_t);
typedef int (*limits_fn)(int*);
static const step_fn k_step_launch[RSIM_NCFG] = {rsim_launch_step_cfg0, rsim_launch_step_cfg1, rsim_launch_step_cfg2, rsim_launch_step_cfg3, rsim_launch_step_cfg4};
static const creset_fn k_creset_launch[RSIM_NCFG] = {rsim_launch_ctrl_reset_cfg0, rsim_launch_ctrl_reset_cfg1, rsim_launch_ctrl_reset_cfg2, rsim_launch_ctrl_reset_cfg3, rsim_launch_ctrl_reset_cfg4};
extern "C" int rsim_launch_prepare_cfg0(const DModel* m, const DBatch* b, int nblocks, int reset_only, hipStream_t stream);
extern "C" int rsim_cmem_bytes_cfg0(void);
extern "C" int rsim_launch_prepare_cfg1(const DModel* m, const DBatch* b, int nblocks, int reset_only, hipStream_t stream);
extern "C" int rsim_cmem_bytes_cfg1(void);
extern "C" int rsim_launch_prepare_cfg2(const DModel* m, const DBatch* b, int nblocks, int reset_only, hipStream_t stream);
extern "C" int rsim_cmem_bytes_cfg2(void);
extern "C" int rsim_launch_prepare_cfg3(const DModel* m, const DBatch* b, int nblocks, int reset_only, hipStream_t stream);
extern "C" int rsim_launch_prepare_cfg4(const DModel* m, const DBatch* b, int nblocks, int reset_only, hipStream_t stream);
extern "C" int rsim_cmem_bytes_cfg3(void);
extern "C" int rsim_cmem_bytes_cfg4(void);
extern "C" int rsim_launch_reset_obs_cfg0(const DModel* m, const DBatch* b, hipStream_t stream);
extern "C" int rsim_launch_reset_obs_cfg1(const DModel* m, const DBatch* b, hipStream_t stream);
extern "C" int rsim_launch_reset_obs_cfg2(const DModel* m, const DBatch* b, hipStream_t stream);
extern "C" int rsim_launch_reset_obs_cfg3(const DModel* m, const DBatch* b, hipStream_t stream);
extern "C" int rsim_launch_reset_obs_cfg4(const DModel* m, const DBatch* b, hipStream_t stream);
typedef int (*resetobs_fn)(const DModel*, const DBatch*, hipStream_t);
static const resetobs_fn k_reset_obs_launch[RSIM_NCFG] = {rsim_launch_reset_obs_cfg0, rsim_launch_reset_obs_cfg1, rsim_launch_reset_obs_cfg2, rsim_launch_reset_obs_cfg3, rsim_launch_reset_obs_cfg4};
typedef int (*prepare_fn)(const DModel*, const DBatch*, int, int, hipStream_t);
typedef int (*cmem_fn)(void);
static const prepare_fn k_prepare_launch[RSIM_NCFG] = {rsim_launch_prepare_cfg0, rsim_launch_prepare_cfg1, rsim_launch_prepare_cfg2, rsim_launch_prepare_cfg3, rsim_launch_prepare_cfg4};
static const cmem_fn k_cmem_bytes[RSIM_NCFG] = {rsim_cmem_bytes_cfg0, rsim_cmem_bytes_cfg1, rsim_cmem_bytes_cfg2, rsim_cmem_bytes_cfg3, rsim_cmem_bytes_cfg4};
static const limits_fn k_limits[RSIM_NCFG_ALL] = {rsim_limits_cfg0, rsim_limits_cfg1, rsim_limits_cfg2, rsim_limits_cfg3, rsim_limits_cfg4, rsim_limits_cfg5, rsim_limits_cfg6, rsim_limits_cfg7};
// the configurations that can serve as the upper capacity tier of a batch, walked in this order: the same lane roles and dense algebra with more
// contacts / rows first (6 above 0, 7 above 1 -- an env on its tier then runs about as fast as on its native configuration, which matters because the
// envs that need the tier are the slowest of a launch), then the wide ones
static const int k_tier_cfgs[4] = {6, 7, 3, 5};
static step_list_fn step_list_launch(int c) {
  return c == 3 ? rsim_launch_step_list_cfg3 : (c == 5 ? rsim_launch_step_list_cfg5 : (c == 6 ? rsim_launch_step_list_cfg6 : (c == 7 ? rsim_launch_step_list_cfg7 : nullptr)));
}
static prepare_fn prepare_launch_any(int c);
static int cmem_bytes_any(int c);
static prepare_fn prepare_launch_any(int c) { return c == 5 ? rsim_launch_prepare_cfg5 : (c == 6 ? rsim_launch_prepare_cfg6 : (c == 7 ? rsim_launch_prepare_cfg7 : k_prepare_launch[c])); }
static int cmem_bytes_any(int c) { return c == 5 ? rsim_cmem_bytes_cfg5() : (c == 6 ? rsim_cmem_bytes_cfg6() : (c == 7 ? rsim_cmem_bytes_cfg7() : k_cmem_bytes[c]())); }
extern "C" int rsim_launch_order(const unsigned* cost, int* order, int B, hipStream_t stream);
extern "C" int rsim_launch_bank_scatter(float* bank, int* tag, const int* env, const int* episode, const float* rows, int n, int E, int W, hipStream_t stream);
extern "C" int rsim_launch_randomize(const DModel* m, const DBatch* b, const DDr* d, unsigned long long seed, unsigned long long step, hipStream_t stream);

struct rsim_model;
static int param_offset_impl(const rsim_model* m, const char* field, int elem);
static thread_local char g_err[512] = "";
static int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return 1;
}
// for the other translation units of the library (rsim_mjcf.cpp): set the message rsim_last_error() returns.  Not part of include/rsim.h.
extern "C" int rsim_set_error(const char* msg) { return fail("%s", msg ? msg : "error"); }
extern "C" const char* rsim_last_error(void) { return g_err; }
#define HIPCHK(x)                                                                  \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) return fail("%s: %s", #x, hipGetErrorString(e_));        \
  } while (0)

typedef unsigned long long u64;

struct Entry { int dtype; size_t count; const void* ptr; };

struct rsim_model {
  std::vector<unsigned char> blob;
  std::map<std::string, Entry> f;
  int nq, nv, nu, nbody, njnt, ngeom, nsite, npair;
  // derived host tables
  std::vector<int> cg;        // colliding geom list (original geom ids)
  std::vector<int> geom2cg;   // original id -> cg index or -1
  std::vector<int> itab;
  std::vector<float> ftab;
  std::vector<float> mesh_vert;
  int io[IO_COUNT], fo[FO_COUNT], fcount[FO_COUNT];
  int maxdepth, nroot;
  std::vector<u64> body_dofmask;
  std::vector<int> lanetab;   // [LT_COUNT][64]
  int kin_rounds, ndynroot, dynroot[RSIM_MAXDYNROOT], maxcondim, multijoint;
  int ntendon, neq, nsensor, nsensordata;
  DCtrl ctrl;
  rsim_task_desc task;
  int has_task;
  float meaninertia;
  std::map<std::string, std::vector<std::string>> names;   // kind -> names in id order (blob entries "names:<kind>"); empty string = unnamed
  const int* I(const char* n) const { auto it = f.find(n); return it == f.end() ? nullptr : (const int*)it->second.ptr; }
  const double* D(const char* n) const { auto it = f.find(n); return it == f.end() ? nullptr : (const double*)it->second.ptr; }
  size_t count(const char* n) const { auto it = f.find(n); return it == f.end() ? 0 : it->second.count; }
};

struct rsim_batch {
  rsim_model* m;
  int B, device, per_env;
  int derived_stale;   // the last launch was a fused control step: the MuJoCo-shaped derived arrays (RSIM_XPOS .. RSIM_NITER, RSIM_SENSORDATA) still hold an older state
  hipStream_t stream;
  int* d_it;
  int* d_lt;
  int* d_obsprog;
  float* d_bank;
  int* d_bank_tag;
  int* d_patch;
  float* d_ft_base;
  float* d_ft;
  float* d_ft0;
  float* d_mesh;
  unsigned char* d_mask;
  DModel dm;
  DBatch db;
  void* fptr[RSIM_FIELD_COUNT];
  size_t fcount[RSIM_FIELD_COUNT];
  int fis_int[RSIM_FIELD_COUNT];
  int lim[10];
  int cfg;   // compiled kernel configuration serving this model (smallest that fits)
  int cs;    // floats of controller state per env (fixed when the batch is created)
  int* d_order;       // longest-job-first dispatch order of rsim_control_step
  unsigned* d_cost;
  // constant blocks (the kernel configuration's Cmem): one shared block, or one per env once a float-table field has per-env values
  void* d_cm;
  size_t cm_bytes;
  int cm_dirty;       // a model parameter / the controller changed since the blocks were built
  DCtrl cm_ctrl;      // controller the blocks were built for
  // Capacity tiers of the fused control step.  MuJoCo never drops a contact (nconmax = 5000, models/assets/base.xml:5); a kernel configuration has a
  // fixed contact / row capacity that decides its LDS footprint and with it the occupancy of the whole batch.  Instead of sizing every env for the
  // worst substep of the worst env, an env whose substep needs more than the native capacity is stepped -- for as long as it does -- by a WIDER
  // configuration (cfg_w: the same source compiled with more contacts / rows) from the same DModel / DBatch: envs near the native capacity move up
  // between steps (tier_next), an env that runs out of capacity in mid-step commits nothing and is redone by the wide configuration inside the same
  // rsim_control_step (redo list).  -1: no tier above this batch's configuration.
  int cfg_w;
  int share_cm;       // the tier's kernel reads the NATIVE constant blocks: its configuration differs from the batch's in contact / row capacity only, and the block layout
                      // (Cmem) depends on neither -- no wide blocks to build before a wide pass (round 6: PickPlace's 256-row tier rebuilt its listed envs' blocks
                      // three times per control step under per-step DR, 3 ms a launch, one of them on the critical path behind the native pass)
  int fused;          // the tier above the batch's configuration is compiled into its control-step kernel (limits bit 5): an env that needs it is stepped -- or carried on
                      // from the substep in which it outgrew the native capacity -- by the wide body inside its own workgroup; no list, no wide launch, no redo
  int lim_w[10];
  void* d_cm_w;       // constant blocks of the wide configuration: one shared, one per env (built on demand for the envs a wide pass steps)
  size_t cm_bytes_w;
  int* d_tier[2];     // [B] each: tier of every env for the current / the next control step (swapped after every step)
  int tier_flip;
  int* d_wlist[2];    // [B] each: envs of the wide pass (tier 1), redo list
  int* d_wcount;      // [2][2 + 2 * RSIM_MAX_GROUPS]: list lengths, double-buffered by the parity of the control step (whole batch: 0, 1; env block g: 2 + 2 g, 3 + 2 g);
                      // the tier-list kernel of a step zeroes the lengths of the next one
  hipStream_t wstream;
  hipEvent_t wfork, wjoin;
  int tier_mode;      // 0 (default): the wide pass beside the native one, on its own stream; 1 (RSIM_TIER_MODE=1, measurements): before it, on the batch's stream
  int have_cost;      // d_cost holds the costs of a previous control step
  // one-launch-per-step batches: the dispatch order of step t + 1 is sorted from the costs of step t - 1 on a side stream WHILE step t runs (envs that are
  // slow stay slow for hundreds of steps, so costs one step old order as well), which takes the sort and its two kernel boundaries off the critical
  // path of a control step.  Double-buffered: step t reads order2[t & 1], writes cost2[t & 1]; the sort beside it reads cost2[(t - 1) & 1], writes order2[(t + 1) & 1].
  int* d_order2[2];
  unsigned* d_cost2[2];
  hipStream_t ostream;
  int order_fresh;    // 1: the dispatch order of step t is sorted on the batch's stream from the costs of step t - 1 (RSIM_ORDER_FRESH, default); 0: beside step t - 1 from the costs of step t - 2
  hipEvent_t step_done[2], ord_done[2];
  int ord_valid[2];   // order2[k] holds a dispatch order (its event has been recorded)
  long nstep;         // one-launch control steps issued since the schedule was (re)started
  int schedule;       // 1 = reorder before every control step (default), 0 = identity order
  // stream groups: control steps of env block g run on gstream[g]; `forked` = the group streams hold work the main stream has not waited for
  int groups, ngroups, forked;   // streams created, groups in use (1 = everything on the main stream)
  hipStream_t gstream[RSIM_MAX_GROUPS];
  hipEvent_t gev[RSIM_MAX_GROUPS], mev;
  // asynchronous reset-bank upkeep (rsim_bank_poll_begin / _poll / rsim_refill_reset_bank_async): a side stream of its own, pinned staging
  hipStream_t bstream;
  hipEvent_t bev;
  int* h_epidx;            // pinned [B]: episode counters as of the last poll
  int bank_poll_pending;
  struct Stage { void* host; void* dev; size_t bytes; hipEvent_t done; int busy; } bstage[4];   // pinned + device staging ring of the async refills
  int bstage_next;
  // host cache for jacobians
  long gen, cache_gen;
  int cache_env;
  std::vector<float> h_cdof, h_rootcom, h_xpos, h_xquat;
};

static int join_groups(rsim_batch* b);
extern "C" int rsim_bank_flush(rsim_batch* b);
// ------------------------------------------------------------------------------------------------------------
extern "C" int rsim_model_create(const void* blob, size_t len, rsim_model** out) {
  if (!blob || len < 16 || memcmp(blob, "RSIMMDL1", 8) != 0) return fail("rsim_model_create: bad blob magic");
  rsim_model* m = new rsim_model();
  m->blob.assign((const unsigned char*)blob, (const unsigned char*)blob + len);
  const unsigned char* p = m->blob.data();
  uint32_t n = *(const uint32_t*)(p + 8);
  if (16 + 48 * (size_t)n > len) { delete m; return fail("rsim_model_create: truncated header"); }
  for (uint32_t i = 0; i < n; i++) {
    const unsigned char* e = p + 16 + 48 * i;
    char name[33];
    memcpy(name, e, 32);
    name[32] = 0;
    uint32_t dtype = *(const uint32_t*)(e + 32), cnt = *(const uint32_t*)(e + 36);
    uint64_t off = *(const uint64_t*)(e + 40);
    size_t bytes = (size_t)cnt * (dtype == 0 ? 4 : 8);
    if (off + bytes > len) { delete m; return fail("rsim_model_create: field %s out of range", name); }
    m->f[name] = Entry{(int)dtype, cnt, p + off};
  }
  for (auto& kv : m->f) {
    if (kv.first.compare(0, 6, "names:") != 0 || kv.second.dtype != 0) continue;
    std::vector<std::string>& out = m->names[kv.first.substr(6)];
    std::string cur;
    const int* c = (const int*)kv.second.ptr;
    for (size_t i = 0; i < kv.second.count; i++) { if (c[i] == 0) { out.push_back(cur); cur.clear(); } else cur.push_back((char)c[i]); }
  }
  auto geti = [&](const char* k) { const int* v = m->I(k); return v ? v[0] : 0; };
  m->nq = geti("nq"); m->nv = geti("nv"); m->nu = geti("nu"); m->nbody = geti("nbody"); m->njnt = geti("njnt");
  m->ngeom = geti("ngeom"); m->nsite = geti("nsite"); m->npair = geti("npair");
  const char* required[] = {"body_parentid", "body_rootid", "body_weldid", "body_jntadr", "body_jntnum", "body_dofadr", "body_dofnum", "body_pos",
                            "body_quat", "body_ipos", "body_iquat", "body_mass", "body_inertia", "body_invweight0", "body_subtreemass", "jnt_type",
                            "jnt_qposadr", "jnt_dofadr", "jnt_bodyid", "jnt_limited", "jnt_pos", "jnt_axis", "jnt_range", "jnt_margin", "jnt_solref",
                            "jnt_solimp", "qpos0", "dof_bodyid", "dof_jntid", "dof_parentid", "dof_armature", "dof_damping", "dof_frictionloss",
                            "dof_solref", "dof_solimp", "dof_invweight0", "dof_M0", "geom_type", "geom_bodyid", "geom_condim", "geom_priority",
                            "geom_dataid", "geom_size", "geom_pos", "geom_quat", "geom_friction", "geom_solref", "geom_solimp", "geom_solmix",
                            "geom_margin", "geom_gap", "geom_rbound", "geom_rcenter", "site_bodyid", "site_pos", "site_quat", "actuator_trnid",
                            "actuator_gear", "actuator_gainprm", "actuator_biasprm", "actuator_biastype", "actuator_ctrllimited", "actuator_ctrlrange",
                            "actuator_forcelimited", "actuator_forcerange", "pair_geom1", "pair_geom2", "gravity", "wind", "timestep"};
  for (const char* r : required)
    if (m->f.find(r) == m->f.end()) { delete m; return fail("rsim_model_create: blob lacks field '%s'", r); }
  if (m->nbody > 64 || m->nv > 64) { delete m; return fail("rsim_model_create: nbody/nv > 64 unsupported (ancestor bit-masks)"); }
  memset(&m->ctrl, 0, sizeof(m->ctrl));
  memset(&m->task, 0, sizeof(m->task));
  m->has_task = 0;

  const int nb = m->nbody, nv = m->nv, nj = m->njnt;
  const int *parent = m->I("body_parentid"), *weld = m->I("body_weldid"), *jadr = m->I("body_jntadr"), *jnum = m->I("body_jntnum");
  const int *bdofadr = m->I("body_dofadr"), *bdofnum = m->I("body_dofnum"), *jtype = m->I("jnt_type"), *jdof = m->I("jnt_dofadr");
  const int *dofbody = m->I("dof_bodyid"), *dofjnt = m->I("dof_jntid"), *dofpar = m->I("dof_parentid");
  // colliding geoms = those referenced by a pair
  m->geom2cg.assign(m->ngeom, -1);
  {
    std::vector<char> used(m->ngeom, 0);
    for (int p2 = 0; p2 < m->npair; p2++) { used[m->I("pair_geom1")[p2]] = 1; used[m->I("pair_geom2")[p2]] = 1; }
    for (int g = 0; g < m->ngeom; g++) if (used[g]) { m->geom2cg[g] = (int)m->cg.size(); m->cg.push_back(g); }
  }
  const int ncg = (int)m->cg.size();
  // masks, depth
  std::vector<int> depth(nb, 0), isroot(nb, 0), moving(nb, 0);
  std::vector<u64> anc(nb, 0), bdm(nb, 0), danc(nv, 0), dcv(nv, 0);
  std::vector<int> zerodot(nv, 0);
  m->maxdepth = 0; m->nroot = 0;
  for (int b = 1; b < nb; b++) {
    depth[b] = depth[parent[b]] + 1;
    if (depth[b] > m->maxdepth) m->maxdepth = depth[b];
    isroot[b] = parent[b] == 0;
    m->nroot += isroot[b];
    moving[b] = weld[b] != 0;
    anc[b] = anc[parent[b]] | (1ull << b);
    bdm[b] = bdm[parent[b]];
    for (int k = 0; k < bdofnum[b]; k++) bdm[b] |= 1ull << (bdofadr[b] + k);
  }
  for (int i = 0; i < nv; i++) {
    danc[i] = (dofpar[i] >= 0 ? danc[dofpar[i]] : 0) | (1ull << i);
    int b = dofbody[i], j = dofjnt[i];
    u64 mk = bdm[parent[b]];
    for (int jj = jadr[b]; jj < j; jj++) {
      int nd = jtype[jj] == 0 ? 6 : (jtype[jj] == 1 ? 3 : 1);
      for (int k = 0; k < nd; k++) mk |= 1ull << (jdof[jj] + k);
    }
    if (jtype[j] == 0) {
      if (i >= jdof[j] + 3) for (int k = 0; k < 3; k++) mk |= 1ull << (jdof[j] + k);
      else zerodot[i] = 1;
    }
    dcv[i] = mk;
  }
  m->body_dofmask = bdm;
  // ---- lane table: per-lane packed constants of the fused kernel (rsim_internal.h LT_*)
  {
    m->lanetab.assign((size_t)LT_COUNT * 64, 0);
    auto LT = [&](int row, int lane) -> int& { return m->lanetab[(size_t)row * 64 + lane]; };
    m->multijoint = 0;
    for (int b = 0; b < nb && b < 64; b++) {
      int p0 = parent[b];
      auto jump = [&](int x, int times) { for (int t = 0; t < times; t++) x = parent[x]; return x; };
      int pr[5];
      for (int r = 0; r < 5; r++) pr[r] = jump(b, 1 << r);
      (void)p0;
      LT(LT_part, b) = pr[0] | (pr[1] << 8) | (pr[2] << 16) | (pr[3] << 24);
      LT(LT_part4, b) = pr[4];
      int jt = 15, qa = 0, da = 0;
      if (jnum[b] >= 1) { jt = jtype[jadr[b]]; qa = m->I("jnt_qposadr")[jadr[b]]; da = jdof[jadr[b]]; }
      if (jnum[b] > 1) m->multijoint = 1;
      LT(LT_binfo, b) = jt | (qa << 4) | (da << 12) | (m->I("body_rootid")[b] << 20) | (moving[b] << 28);
      LT(LT_bdofs, b) = (int)(uint32_t)bdm[b];
    }
    for (int i = 0; i < nv && i < 64; i++) {
      int j = dofjnt[i], t = jtype[j];
      int lim = (t == 2 || t == 3) ? m->I("jnt_limited")[j] : 0;
      LT(LT_dinfo, i) = dofbody[i] | (zerodot[i] << 8) | (lim << 9) | (m->I("jnt_qposadr")[j] << 10) | (j << 18) | (t << 26);
    }
    for (int c = 0; c < ncg && c < 64; c++) LT(LT_ginfo, c) = m->I("geom_bodyid")[m->cg[c]] | (m->I("geom_type")[m->cg[c]] << 8);
    for (int k = 0; k < m->nsite && k < 64; k++) LT(LT_sinfo, k) = m->I("site_bodyid")[k];
    {  // LDS-resident hull pool: deepest bodies first (gripper, distal links collide most), whole hulls only
      std::vector<int> order;
      for (int c = 0; c < ncg && c < 64; c++) { LT(LT_ghull, c) = -1; if (m->I("geom_type")[m->cg[c]] == 7) order.push_back(c); }
      std::stable_sort(order.begin(), order.end(), [&](int a, int b2) { return depth[m->I("geom_bodyid")[m->cg[a]]] > depth[m->I("geom_bodyid")[m->cg[b2]]]; });
      int used = 0;
      for (int c : order) {
        int num = m->I("mesh_vertnum")[m->I("geom_dataid")[m->cg[c]]];
        if (used + num <= RSIM_HULL_POOL) { LT(LT_ghull, c) = used; used += num; }
      }
    }
    for (int a = 0; a < m->nu && a < 64; a++) {
      int j = m->I("actuator_trnid")[a];
      LT(LT_ainfo, a) = jdof[j] | (m->I("jnt_qposadr")[j] << 8) | (m->I("actuator_biastype")[a] << 16) | (m->I("actuator_ctrllimited")[a] << 18) |
                        (m->I("actuator_forcelimited")[a] << 19);
    }
    for (int p2 = 0; p2 < m->npair && p2 < RSIM_PAIR_MAX; p2++)
      LT(p2 < 192 ? LT_pair0 + p2 / 64 : (p2 < 320 ? LT_pair3 + (p2 - 192) / 64 : LT_pair5 + (p2 - 320) / 64), p2 % 64) = m->geom2cg[m->I("pair_geom1")[p2]] | (m->geom2cg[m->I("pair_geom2")[p2]] << 8) | (1 << 16);
    for (int l = 0; l < 64; l++) {
      unsigned bits = 0;
      const int q = l / 16, r = l % 16;
      for (int c = 0; c < 8; c++) {  // sub[c]: body d = 4c+q in subtree(body(dof r))
        int d = 4 * c + q;
        if (r < nv && d < nb && ((anc[d] >> dofbody[r]) & 1ull)) bits |= 1u << c;
      }
      for (int rb = 0; rb < 2; rb++)
        for (int c = 0; c < 4; c++) {  // bodydof[rb][c]: dof k = 4c+q moves body 16rb+r
          int k = 4 * c + q, b = 16 * rb + r;
          if (k < nv && b < nb && ((bdm[b] >> k) & 1ull)) bits |= 1u << (8 + 4 * rb + c);
        }
      for (int c = 0; c < 4; c++) {  // dcv[c]: dof k = 4c+q precedes dof r in the velocity recursion
        int k = 4 * c + q;
        if (k < nv && r < nv && ((dcv[r] >> k) & 1ull)) bits |= 1u << (16 + c);
      }
      for (int v = 0; v < 4; v++) {  // M(i = 4q+v, j = r): m1 = j ancestor-or-self of i, m2 = i strict ancestor of j
        int i = 4 * q + v, j = r;
        if (i < nv && j < nv) {
          if ((danc[i] >> j) & 1ull) bits |= 1u << (20 + v);
          if (i != j && ((danc[j] >> i) & 1ull)) bits |= 1u << (24 + v);
        }
      }
      LT(LT_mfbits, l) = (int)bits;
    }
    int md = m->maxdepth > 1 ? m->maxdepth : 1;
    m->kin_rounds = 0;
    while ((1 << m->kin_rounds) < md) m->kin_rounds++;
    m->ndynroot = 0;
    for (int b = 1; b < nb; b++) {
      if (parent[b] != 0) continue;
      bool dyn = false;
      for (int i = 0; i < nv; i++) if (m->I("body_rootid")[dofbody[i]] == b) dyn = true;
      if (dyn) { if (m->ndynroot < RSIM_MAXDYNROOT) m->dynroot[m->ndynroot] = b; m->ndynroot++; }
    }
    m->maxcondim = 1;
    for (int c = 0; c < ncg; c++) if (m->I("geom_condim")[m->cg[c]] > m->maxcondim) m->maxcondim = m->I("geom_condim")[m->cg[c]];
  }
  // ---- int table
  auto& it = m->itab;
  auto push = [&](int id, const std::vector<int>& v) { m->io[id] = (int)it.size(); it.insert(it.end(), v.begin(), v.end()); };
  auto vec = [&](const char* k, size_t n_) { const int* p2 = m->I(k); return std::vector<int>(p2, p2 + n_); };
  auto split = [&](const std::vector<u64>& v) { std::vector<int> o; for (u64 x : v) { o.push_back((int)(uint32_t)(x & 0xffffffffull)); o.push_back((int)(uint32_t)(x >> 32)); } return o; };
  push(IO_body_parentid, vec("body_parentid", nb)); push(IO_body_rootid, vec("body_rootid", nb)); push(IO_body_jntadr, vec("body_jntadr", nb));
  push(IO_body_jntnum, vec("body_jntnum", nb)); push(IO_body_dofadr, vec("body_dofadr", nb)); push(IO_body_dofnum, vec("body_dofnum", nb));
  push(IO_body_depth, depth); push(IO_body_mocap, std::vector<int>(nb, 0)); push(IO_body_moving, moving); push(IO_body_ancmask, split(anc));
  push(IO_body_dofmask, split(bdm)); push(IO_body_isroot, isroot);
  push(IO_jnt_type, vec("jnt_type", nj)); push(IO_jnt_qposadr, vec("jnt_qposadr", nj)); push(IO_jnt_dofadr, vec("jnt_dofadr", nj));
  push(IO_jnt_bodyid, vec("jnt_bodyid", nj)); push(IO_jnt_limited, vec("jnt_limited", nj));
  push(IO_dof_bodyid, vec("dof_bodyid", nv)); push(IO_dof_jntid, vec("dof_jntid", nv)); push(IO_dof_ancmask, split(danc)); push(IO_dof_cvelmask, split(dcv));
  push(IO_dof_zerodot, zerodot);
  {
    std::vector<int> gid(ncg), gt(ncg), gb(ncg), gc(ncg), gp(ncg), ma(ncg, 0), mn(ncg, 0);
    for (int c = 0; c < ncg; c++) {
      int g = m->cg[c];
      gid[c] = g; gt[c] = m->I("geom_type")[g]; gb[c] = m->I("geom_bodyid")[g]; gc[c] = m->I("geom_condim")[g]; gp[c] = m->I("geom_priority")[g];
      int did = m->I("geom_dataid")[g];
      if (gt[c] == 7) {
        if (did < 0) { delete m; return fail("mesh geom %d without mesh data", g); }
        ma[c] = m->I("mesh_vertadr")[did]; mn[c] = m->I("mesh_vertnum")[did];
      }
      if (gt[c] == 1) { delete m; return fail("hfield geoms unsupported"); }
    }
    push(IO_cg_geomid, gid); push(IO_cg_type, gt); push(IO_cg_bodyid, gb); push(IO_cg_condim, gc); push(IO_cg_priority, gp); push(IO_cg_meshadr, ma);
    push(IO_cg_meshnum, mn);
    std::vector<int> p1(m->npair), p2(m->npair);
    for (int p3 = 0; p3 < m->npair; p3++) { p1[p3] = m->geom2cg[m->I("pair_geom1")[p3]]; p2[p3] = m->geom2cg[m->I("pair_geom2")[p3]]; }
    push(IO_pair_g1, p1); push(IO_pair_g2, p2);
  }
  push(IO_site_bodyid, vec("site_bodyid", m->nsite));
  push(IO_act_trnid, vec("actuator_trnid", m->nu)); push(IO_act_biastype, vec("actuator_biastype", m->nu));
  push(IO_act_ctrllimited, vec("actuator_ctrllimited", m->nu)); push(IO_act_forcelimited, vec("actuator_forcelimited", m->nu));
  // fixed tendons / equality-tendon rows (absent in blobs compiled before they existed)
  m->ntendon = m->I("ntendon") ? m->I("ntendon")[0] : 0;
  m->neq = m->I("neq") ? m->I("neq")[0] : 0;
  if (m->ntendon > 64 || m->neq > 64) { delete m; return fail("rsim_model_create: more than 64 tendons / equality constraints"); }
  {
    const int nw = (int)m->count("wrap_objid");
    std::vector<int> wd(nw), wq(nw);
    for (int w = 0; w < nw; w++) { int j = m->I("wrap_objid")[w]; wd[w] = jdof[j]; wq[w] = m->I("jnt_qposadr")[j]; }
    for (int t = 0; t < m->ntendon; t++) if (m->I("tendon_num")[t] > 4) { delete m; return fail("rsim_model_create: fixed tendon over more than 4 joints"); }
    push(IO_tendon_adr, vec("tendon_adr", m->ntendon)); push(IO_tendon_num, vec("tendon_num", m->ntendon)); push(IO_tendon_limited, vec("tendon_limited", m->ntendon));
    push(IO_wrap_dof, wd); push(IO_wrap_qadr, wq); push(IO_eq_tendon, vec("eq_obj1id", m->neq));
  }
  // force / torque sensors at a site (gripper XMLs: <force site="ft_frame"/>, <torque site="ft_frame"/>); any other type reads zero
  {
    m->nsensor = m->I("nsensor") ? m->I("nsensor")[0] : 0;
    if (m->nsensor > 64) { delete m; return fail("rsim_model_create: more than 64 sensors"); }
    std::vector<int> st(m->nsensor), ss(m->nsensor), sa(m->nsensor + 1, 0);
    for (int i = 0; i < m->nsensor; i++) {
      st[i] = m->I("sensor_type")[i]; ss[i] = m->I("sensor_objid")[i]; sa[i + 1] = sa[i] + m->I("sensor_dim")[i];
    }
    m->nsensordata = sa[m->nsensor];
    push(IO_sensor_type, st); push(IO_sensor_site, ss); push(IO_sensor_adr, sa);
  }
  // ---- float table
  auto& ft = m->ftab;
  auto pushf = [&](int id, const char* k, size_t n_) {
    m->fo[id] = (int)ft.size(); m->fcount[id] = (int)n_;
    const double* p2 = m->D(k);
    for (size_t i = 0; i < n_; i++) ft.push_back((float)p2[i]);
  };
  auto pushg = [&](int id, const char* k, int w) {  // geom field restricted to the colliding list
    m->fo[id] = (int)ft.size(); m->fcount[id] = ncg * w;
    const double* p2 = m->D(k);
    for (int c = 0; c < ncg; c++) for (int q = 0; q < w; q++) ft.push_back((float)p2[(size_t)m->cg[c] * w + q]);
  };
  pushf(FO_body_pos, "body_pos", 3 * nb); pushf(FO_body_quat, "body_quat", 4 * nb); pushf(FO_body_ipos, "body_ipos", 3 * nb); pushf(FO_body_iquat, "body_iquat", 4 * nb);
  pushf(FO_body_mass, "body_mass", nb); pushf(FO_body_inertia, "body_inertia", 3 * nb); pushf(FO_body_invweight0, "body_invweight0", 2 * nb);
  pushf(FO_body_subtreemass, "body_subtreemass", nb);
  pushf(FO_jnt_pos, "jnt_pos", 3 * nj); pushf(FO_jnt_axis, "jnt_axis", 3 * nj); pushf(FO_jnt_range, "jnt_range", 2 * nj); pushf(FO_jnt_margin, "jnt_margin", nj);
  pushf(FO_jnt_solref, "jnt_solref", 2 * nj); pushf(FO_jnt_solimp, "jnt_solimp", 5 * nj); pushf(FO_qpos0, "qpos0", m->nq);
  pushf(FO_dof_armature, "dof_armature", nv); pushf(FO_dof_damping, "dof_damping", nv); pushf(FO_dof_frictionloss, "dof_frictionloss", nv);
  pushf(FO_dof_solref, "dof_solref", 2 * nv); pushf(FO_dof_solimp, "dof_solimp", 5 * nv); pushf(FO_dof_invweight0, "dof_invweight0", nv);
  pushg(FO_cg_size, "geom_size", 3); pushg(FO_cg_pos, "geom_pos", 3); pushg(FO_cg_quat, "geom_quat", 4); pushg(FO_cg_friction, "geom_friction", 3);
  pushg(FO_cg_solref, "geom_solref", 2); pushg(FO_cg_solimp, "geom_solimp", 5); pushg(FO_cg_solmix, "geom_solmix", 1); pushg(FO_cg_margin, "geom_margin", 1);
  pushg(FO_cg_gap, "geom_gap", 1); pushg(FO_cg_rbound, "geom_rbound", 1); pushg(FO_cg_rcenter, "geom_rcenter", 3);
  {  // geom-frame bounding boxes of mesh hulls (broadphase OBB test); primitives derive theirs from geom_size in the kernel
    m->fo[FO_cg_aabb] = (int)ft.size(); m->fcount[FO_cg_aabb] = ncg * 6;
    const double* mv = m->D("mesh_vert");
    for (int c = 0; c < ncg; c++) {
      int g = m->cg[c];
      double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
      if (m->I("geom_type")[g] == 7) {
        int did = m->I("geom_dataid")[g], adr = m->I("mesh_vertadr")[did], num = m->I("mesh_vertnum")[did];
        for (int k = 0; k < 3; k++) { lo[k] = 1e30; hi[k] = -1e30; }
        for (int v = 0; v < num; v++) for (int k = 0; k < 3; k++) { double x = mv[3 * (size_t)(adr + v) + k]; if (x < lo[k]) lo[k] = x; if (x > hi[k]) hi[k] = x; }
      }
      for (int k = 0; k < 3; k++) ft.push_back((float)(0.5 * (lo[k] + hi[k])));
      for (int k = 0; k < 3; k++) ft.push_back((float)(0.5 * (hi[k] - lo[k]) * 1.000001 + 1e-7));
    }
  }
  {  // bounding capsules of mesh hulls (broadphase: adjacent arm links overlap as boxes but not as capsules)
    m->fo[FO_cg_capsule] = (int)ft.size(); m->fcount[FO_cg_capsule] = ncg * 8;
    const double* mv = m->D("mesh_vert");
    for (int c = 0; c < ncg; c++) {
      int g = m->cg[c];
      double cap[8] = {0, 0, 0, 0, 0, 0, -1, 0};
      if (m->I("geom_type")[g] == 7) {
        int did = m->I("geom_dataid")[g], adr = m->I("mesh_vertadr")[did], num = m->I("mesh_vertnum")[did];
        const double* V = mv + 3 * (size_t)adr;
        double cen[3] = {0, 0, 0};
        for (int v = 0; v < num; v++) for (int k = 0; k < 3; k++) cen[k] += V[3 * v + k] / num;
        // candidate axes: principal axes of the vertex cloud (Jacobi eigenvectors of the covariance) + the frame axes
        double C[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, E[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        for (int v = 0; v < num; v++) for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C[i][j] += (V[3 * v + i] - cen[i]) * (V[3 * v + j] - cen[j]);
        for (int sweep = 0; sweep < 30; sweep++)
          for (int p2 = 0; p2 < 2; p2++) for (int q2 = p2 + 1; q2 < 3; q2++) {
            if (fabs(C[p2][q2]) < 1e-18) continue;
            double th = 0.5 * atan2(2 * C[p2][q2], C[q2][q2] - C[p2][p2]), cs = cos(th), sn = sin(th);
            for (int k = 0; k < 3; k++) { double a = C[k][p2], b2 = C[k][q2]; C[k][p2] = cs * a - sn * b2; C[k][q2] = sn * a + cs * b2; }
            for (int k = 0; k < 3; k++) { double a = C[p2][k], b2 = C[q2][k]; C[p2][k] = cs * a - sn * b2; C[q2][k] = sn * a + cs * b2; }
            for (int k = 0; k < 3; k++) { double a = E[k][p2], b2 = E[k][q2]; E[k][p2] = cs * a - sn * b2; E[k][q2] = sn * a + cs * b2; }
          }
        double bestvol = 1e300;
        for (int cand = 0; cand < 6; cand++) {
          double ax[3];
          for (int k = 0; k < 3; k++) ax[k] = cand < 3 ? E[k][cand] : (k == cand - 3 ? 1.0 : 0.0);
          double n = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
          if (n < 1e-12) continue;
          for (int k = 0; k < 3; k++) ax[k] /= n;
          double rmax = 0;
          for (int v = 0; v < num; v++) {
            double d[3] = {V[3 * v] - cen[0], V[3 * v + 1] - cen[1], V[3 * v + 2] - cen[2]}, t = d[0] * ax[0] + d[1] * ax[1] + d[2] * ax[2];
            double pp = sqrt(fmax(0.0, d[0] * d[0] + d[1] * d[1] + d[2] * d[2] - t * t));
            if (pp > rmax) rmax = pp;
          }
          double R = rmax * 1.02 + 1e-4, lo = 1e300, hi = -1e300;  // every vertex within R of the segment [lo, hi] on the axis
          for (int v = 0; v < num; v++) {
            double d[3] = {V[3 * v] - cen[0], V[3 * v + 1] - cen[1], V[3 * v + 2] - cen[2]}, t = d[0] * ax[0] + d[1] * ax[1] + d[2] * ax[2];
            double pp2 = fmax(0.0, d[0] * d[0] + d[1] * d[1] + d[2] * d[2] - t * t), reach = sqrt(fmax(0.0, R * R - pp2));
            if (t + reach < lo) lo = t + reach;
            if (t - reach > hi) hi = t - reach;
          }
          if (lo > hi) lo = hi = 0.5 * (lo + hi);
          double vol = 3.14159265358979 * R * R * (hi - lo) + 4.18879020478639 * R * R * R;
          if (vol < bestvol) {
            bestvol = vol;
            for (int k = 0; k < 3; k++) { cap[k] = cen[k] + lo * ax[k]; cap[3 + k] = cen[k] + hi * ax[k]; }
            cap[6] = R;
          }
        }
      }
      for (int k = 0; k < 8; k++) ft.push_back((float)cap[k]);
    }
  }
  pushf(FO_site_pos, "site_pos", 3 * m->nsite); pushf(FO_site_quat, "site_quat", 4 * m->nsite);
  pushf(FO_act_gear, "actuator_gear", m->nu); pushf(FO_act_gainprm, "actuator_gainprm", 3 * m->nu); pushf(FO_act_biasprm, "actuator_biasprm", 3 * m->nu);
  pushf(FO_act_ctrlrange, "actuator_ctrlrange", 2 * m->nu); pushf(FO_act_forcerange, "actuator_forcerange", 2 * m->nu);
  {
    const int nw = (int)m->count("wrap_prm"), nt = m->ntendon, ne = m->neq;
    pushf(FO_wrap_prm, "wrap_prm", nw); pushf(FO_tendon_range, "tendon_range", 2 * nt); pushf(FO_tendon_margin, "tendon_margin", nt);
    pushf(FO_tendon_solref, "tendon_solref_lim", 2 * nt); pushf(FO_tendon_solimp, "tendon_solimp_lim", 5 * nt);
    pushf(FO_tendon_len0, "tendon_length0", nt); pushf(FO_tendon_invw, "tendon_invweight0", nt);
    m->fo[FO_eq_data0] = (int)ft.size(); m->fcount[FO_eq_data0] = ne;
    for (int e = 0; e < ne; e++) ft.push_back((float)m->D("eq_data")[5 * e]);
    pushf(FO_eq_solref, "eq_solref", 2 * ne); pushf(FO_eq_solimp, "eq_solimp", 5 * ne);
    auto pushopt = [&](int id, const char* k, size_t n_, double dflt) {   // fields absent in blobs compiled before they existed
      m->fo[id] = (int)ft.size(); m->fcount[id] = (int)n_;
      const double* p2 = m->D(k);
      for (size_t i = 0; i < n_; i++) ft.push_back(p2 ? (float)p2[i] : (float)dflt);
    };
    pushopt(FO_tendon_stiffness, "tendon_stiffness", nt, 0.0); pushopt(FO_tendon_damping, "tendon_damping", nt, 0.0);
    pushopt(FO_tendon_lspring, "tendon_lengthspring", 2 * nt, 0.0);
    pushopt(FO_tendon_fl, "tendon_frictionloss", nt, 0.0); pushopt(FO_tendon_solref_fri, "tendon_solref_fri", 2 * nt, 0.0);
    pushopt(FO_tendon_solimp_fri, "tendon_solimp_fri", 5 * nt, 0.0);
  }
  m->fo[FO_opt] = (int)ft.size(); m->fcount[FO_opt] = 10;
  {
    auto d1 = [&](const char* k) { const double* v = m->D(k); return v ? (float)v[0] : 0.f; };
    ft.push_back(d1("timestep"));
    for (int k = 0; k < 3; k++) ft.push_back((float)m->D("gravity")[k]);
    ft.push_back(d1("density")); ft.push_back(d1("viscosity")); ft.push_back(d1("impratio"));
    for (int k = 0; k < 3; k++) ft.push_back((float)m->D("wind")[k]);
  }
  while (ft.size() % 16) ft.push_back(0.f);
  // mesh vertices
  {
    size_t nmv = m->count("mesh_vert");
    const double* mv = m->D("mesh_vert");
    m->mesh_vert.resize(nmv ? nmv : 3);
    for (size_t i = 0; i < nmv; i++) m->mesh_vert[i] = (float)mv[i];
  }
  double s = 0;
  for (int i = 0; i < nv; i++) s += m->D("dof_M0")[i];
  m->meaninertia = nv > 0 ? (float)(s / nv) : 1.f;
  if (m->meaninertia < 1e-15f) m->meaninertia = 1.f;
  *out = m;
  return 0;
}

extern "C" void rsim_model_free(rsim_model* m) { delete m; }

static int pick_config(const rsim_model* m, int* lim_out);
extern "C" int rsim_model_int(const rsim_model* m, const char* name) {
  if (!strcmp(name, "ncgeom")) return (int)m->cg.size();
  if (!strcmp(name, "cstate_size")) return m->ctrl.enabled ? m->ctrl.cs_size : RSIM_CS_SIZE;
  if (!strcmp(name, "action_dim")) return m->ctrl.enabled ? m->ctrl.action_dim : 0;
  if (!strcmp(name, "nsensordata")) return m->nsensordata;
  if (!strcmp(name, "float_table_size")) return (int)m->ftab.size();                              // floats of one env's model float table
  if (!strcmp(name, "constant_block_bytes")) { const int c = pick_config(m, nullptr); return c < 0 ? -1 : k_cmem_bytes[c](); }   // one env's constant block
  const int* v = m->I(name);
  return v ? v[0] : -1;
}

extern "C" int rsim_model_set_controller(rsim_model* m, const rsim_ctrl_desc* d) {
  if (d->type < RSIM_CTRL_OSC_POSE || d->type > RSIM_CTRL_JOINT_VELOCITY) return fail("controller: unknown part-controller type %d", d->type);
  const bool jointspace = d->type >= RSIM_CTRL_JOINT_POSITION;
  if (d->ndof < 1 || d->ndof > (jointspace ? RSIM_JNT_MAX : RSIM_ARM_MAX) || d->ngrip < 0 || d->ngrip > RSIM_GRIP_MAX)
    return fail("controller: bad ndof/ngrip (%d joints: at most %d for this controller type)", d->ndof, jointspace ? RSIM_JNT_MAX : RSIM_ARM_MAX);
  DCtrl& c = m->ctrl;
  memset(&c, 0, sizeof(c));
  c.enabled = 1; c.ndof = d->ndof;
  for (int i = 0; i < d->ndof; i++) {
    if (d->qpos_idx[i] < 0 || d->qpos_idx[i] >= m->nq || d->dof_idx[i] < 0 || d->dof_idx[i] >= m->nv || d->act_idx[i] < 0 || d->act_idx[i] >= m->nu)
      return fail("controller: index out of range");
    c.qpos_idx[i] = d->qpos_idx[i]; c.dof_idx[i] = d->dof_idx[i]; c.act_idx[i] = d->act_idx[i];
  }
  if (!jointspace && (d->eef_site < 0 || d->eef_site >= m->nsite || d->base_site < 0 || d->base_site >= m->nsite)) return fail("controller: bad site id");
  c.eef_site = jointspace ? 0 : d->eef_site; c.base_site = jointspace ? 0 : d->base_site;
  c.type = d->type;
  c.narm = d->narm == 2 ? 2 : 1;
  if (d->narm < 0 || d->narm > 2) return fail("controller: narm = %d (1 or 2 arm parts)", d->narm);
  if (c.narm == 2) {
    if (jointspace) return fail("controller: joint-space types concatenate the arms into one part (ndof <= 16, part_of[]), narm must be 1");
    if (d->impedance_mode || d->interp_steps) return fail("controller: two OSC arm parts cannot be combined with variable impedance or an interpolator");
    if (d->ndof2 < 1 || d->ndof2 > RSIM_ARM_MAX) return fail("controller: bad ndof2 (%d)", d->ndof2);
    if (d->eef_site2 < 0 || d->eef_site2 >= m->nsite || d->base_site2 < 0 || d->base_site2 >= m->nsite) return fail("controller: bad site id (second arm)");
    c.ndof2 = d->ndof2; c.eef_site2 = d->eef_site2; c.base_site2 = d->base_site2;
    for (int i = 0; i < d->ndof2; i++) {
      const int k = RSIM_ARM_MAX + i;
      if (d->qpos_idx[k] < 0 || d->qpos_idx[k] >= m->nq || d->dof_idx[k] < 0 || d->dof_idx[k] >= m->nv || d->act_idx[k] < 0 || d->act_idx[k] >= m->nu)
        return fail("controller: index out of range (second arm)");
      c.qpos_idx[k] = d->qpos_idx[k]; c.dof_idx[k] = d->dof_idx[k]; c.act_idx[k] = d->act_idx[k];
    }
  }
  c.cs_size = d->type == RSIM_CTRL_JOINT_VELOCITY ? RSIM_CS_SIZE_JVEL : (jointspace ? RSIM_CS_SIZE_JOINT : RSIM_CS_SIZE);
  if (d->impedance_mode < 0 || d->impedance_mode > 2) return fail("controller: impedance_mode %d", d->impedance_mode);
  if (d->impedance_mode && (d->type == RSIM_CTRL_JOINT_TORQUE || d->type == RSIM_CTRL_JOINT_VELOCITY)) return fail("controller: this part type has no variable-impedance mode");
  c.imp_mode = d->impedance_mode;
  c.nimp = d->impedance_mode ? (jointspace ? d->ndof : 6) : 0;
  if (c.imp_mode) c.cs_size = RSIM_CS_SIZE_VARIMP;
  if (d->interp_steps < 0 || d->interp_steps > 1000) return fail("controller: interp_steps %d", d->interp_steps);
  c.interp_steps = d->interp_steps;
  if (c.interp_steps) c.cs_size = RSIM_CS_SIZE_INTERP;
  if (c.narm == 2) c.cs_size = 2 * RSIM_CS_SIZE;   // one OSC state block per arm
  for (int i = 0; i < c.nimp; i++) {
    if (!(d->kp_max[i] >= d->kp_min[i]) || !(d->kp_min[i] >= 0.f) || !(d->damping_max[i] >= d->damping_min[i])) return fail("controller: bad kp / damping_ratio limits");
    c.kp_min[i] = d->kp_min[i]; c.kp_max[i] = d->kp_max[i]; c.dr_min[i] = d->damping_min[i]; c.dr_max[i] = d->damping_max[i];
  }
  for (int i = 0; i < d->ndof; i++) {
    if (d->part_of[i] < 0 || d->part_of[i] > 3) return fail("controller: part_of[%d] = %d (at most 4 parts)", i, d->part_of[i]);
    c.part_of[i] = d->part_of[i];
  }
  c.cdim = d->type == RSIM_CTRL_OSC_POSE ? 6 : d->type == RSIM_CTRL_OSC_POSITION ? 3 : d->ndof;
  const int ngain = (d->type == RSIM_CTRL_JOINT_POSITION || d->type == RSIM_CTRL_JOINT_VELOCITY) ? d->ndof : (d->type == RSIM_CTRL_JOINT_TORQUE ? 0 : 6);
  for (int i = 0; i < ngain; i++) {
    if (!(d->kp[i] >= 0.f)) return fail("controller: negative / NaN kp");
    c.kp[i] = d->kp[i]; c.kd[i] = 2.f * sqrtf(d->kp[i]) * d->damping_ratio;
  }
  for (int i = 0; i < c.cdim; i++) {
    if (!(d->input_max[i] > d->input_min[i])) return fail("controller: input_max <= input_min");
    c.in_min[i] = d->input_min[i]; c.in_max[i] = d->input_max[i]; c.out_min[i] = d->output_min[i]; c.out_max[i] = d->output_max[i];
  }
  if (c.narm == 2) {
    for (int i = 0; i < 6; i++) {
      const int k = RSIM_ARM_MAX + i;
      if (!(d->kp[k] >= 0.f)) return fail("controller: negative / NaN kp (second arm)");
      c.kp[k] = d->kp[k]; c.kd[k] = 2.f * sqrtf(d->kp[k]) * d->damping_ratio;
    }
    for (int i = 0; i < c.cdim; i++) {
      const int k = RSIM_ARM_MAX + i;
      if (!(d->input_max[k] > d->input_min[k])) return fail("controller: input_max <= input_min (second arm)");
      c.in_min[k] = d->input_min[k]; c.in_max[k] = d->input_max[k]; c.out_min[k] = d->output_min[k]; c.out_max[k] = d->output_max[k];
    }
  }
  {
    bool given = false;   // torque_limits=None -> actuator ctrlrange (joint_tor.py:95-96, controller.py:313-322)
    for (int i = 0; i < d->ndof; i++) given |= d->torque_min[i] != 0.f || d->torque_max[i] != 0.f;
    const double* cr = m->D("actuator_ctrlrange");
    for (int i = 0; i < d->ndof; i++) {
      if (d->type == RSIM_CTRL_JOINT_VELOCITY) {   // velocity_limits=None: no clipping
        c.tl_lo[i] = given ? d->torque_min[i] : -3.0e38f; c.tl_hi[i] = given ? d->torque_max[i] : 3.0e38f;
      } else {
        c.tl_lo[i] = given ? d->torque_min[i] : (float)cr[2 * d->act_idx[i]];
        c.tl_hi[i] = given ? d->torque_max[i] : (float)cr[2 * d->act_idx[i] + 1];
      }
    }
  }
  c.uncouple = d->uncouple_pos_ori; c.nullspace_kp = d->nullspace_kp > 0 ? d->nullspace_kp : 10.f;
  c.ngrip = d->ngrip;
  for (int i = 0; i < d->ngrip; i++) {
    if (d->grip_act[i] < 0 || d->grip_act[i] >= m->nu) return fail("controller: bad gripper actuator");
    c.grip_act[i] = d->grip_act[i]; c.grip_sign[i] = d->grip_sign[i];
  }
  c.grip_speed = d->grip_speed;
  c.action_dim = c.cdim * c.narm + c.nimp * (c.imp_mode == 1 ? 2 : 1) + (d->ngrip > 0 ? 1 : 0);
  return 0;
}

extern "C" int rsim_model_cgeom(const rsim_model* m, int geom_id) {
  if (geom_id < 0 || geom_id >= m->ngeom) return -1;
  return m->geom2cg[geom_id];
}

extern "C" int rsim_model_set_task(rsim_model* m, const rsim_task_desc* d) {
  if (d->nobs < 0 || d->nobs > RSIM_OBS_MAX) return fail("task: nobs out of range");
  for (int i = 0; i < d->nobs; i++) {
    int kind = d->obs_prog[3 * i], a = d->obs_prog[3 * i + 1], b2 = d->obs_prog[3 * i + 2];
    bool ok = true;
    switch (kind) {
      case RSIM_OBS_QPOS: case RSIM_OBS_COS: case RSIM_OBS_SIN: ok = a >= 0 && a < m->nq; break;
      case RSIM_OBS_QVEL: case RSIM_OBS_QACC: ok = a >= 0 && a < m->nv; break;
      case RSIM_OBS_SITE_POS: ok = a >= 0 && a < m->nsite && b2 >= 0 && b2 < 3; break;
      case RSIM_OBS_SITE_QUAT: ok = a >= 0 && a < m->nsite && b2 >= 0 && b2 < 4; break;
      case RSIM_OBS_BODY_QUAT: ok = ((a >= 0 && a < m->nbody) || (a == -1 && d->task == 4 && d->single_object_mode == 1)) && b2 >= 0 && b2 < 4; break;
      case RSIM_OBS_BODY_POS: ok = ((a >= 0 && a < m->nbody) || (a == -1 && d->task == 4 && d->single_object_mode == 1)) && b2 >= 0 && b2 < 3; break;
      case RSIM_OBS_BODY_MINUS_SITE: ok = a >= 0 && a < m->nbody && (b2 & 3) < 3 && (b2 >> 2) >= 0 && (b2 >> 2) < m->nsite; break;
      case RSIM_OBS_BODY_MINUS_BODY: ok = a >= 0 && a < m->nbody && (b2 & 3) < 3 && (b2 >> 2) >= 0 && (b2 >> 2) < m->nbody; break;
      case RSIM_OBS_PEG_COS: case RSIM_OBS_PEG_T: case RSIM_OBS_PEG_D: ok = d->task == 3; break;
      case RSIM_OBS_REL_POS: ok = d->task == 4 && ((a >= 0 && a < d->nobj) || (a == -1 && d->single_object_mode == 1)) && b2 >= 0 && b2 < 3; break;
      case RSIM_OBS_REL_QUAT: ok = d->task == 4 && ((a >= 0 && a < d->nobj) || (a == -1 && d->single_object_mode == 1)) && b2 >= 0 && b2 < 4; break;
      case RSIM_OBS_TASK_OBJECT: ok = d->task == 4 && d->single_object_mode == 1; break;
      default: ok = false;
    }
    if (!ok) return fail("task: observation entry %d (kind %d, a %d, b %d) is invalid for this model", i, kind, a, b2);
  }
  if (d->task < 0 || d->task > 4) return fail("task: unknown task id %d", d->task);
  if (d->task == 4) {
    if (d->single_object_mode < 0 || d->single_object_mode > 2) return fail("task: PickPlace single_object_mode %d (0 = all objects, 1 = one object drawn per episode, 2 = one fixed object)", d->single_object_mode);
    if (d->nobj < 1 || d->nobj > 4 || d->eef_body < 0 || d->eef_body >= m->nbody || d->grip_site < 0 || d->grip_site >= m->nsite) return fail("task: bad PickPlace description");
    for (int i = 0; i < d->nobj; i++)
      if (d->obj_body[i] < 0 || d->obj_body[i] >= m->nbody || d->pos_slot[i] < 0 || d->pos_slot[i] + 7 > d->nobs) return fail("task: bad PickPlace object %d", i);
  }
  if (d->task >= 1 && d->task <= 3 && (d->object_body < 0 || d->object_body >= m->nbody)) return fail("task: bad object body id");
  if ((d->task == 1 || d->task == 2) && (d->grip_site < 0 || d->grip_site >= m->nsite)) return fail("task: bad grip site id");
  if ((d->task == 2 || d->task == 3) && (d->object2_body < 0 || d->object2_body >= m->nbody)) return fail("task: bad second object body id");
  m->task = *d;
  m->has_task = 1;
  return 0;
}

static int join_groups(rsim_batch* b);
// ------------------------------------------------------------------------------------------------------------
template <class T>
static int dalloc(T** p, size_t n) {
  HIPCHK(hipMalloc((void**)p, (n ? n : 1) * sizeof(T)));
  HIPCHK(hipMemset(*p, 0, (n ? n : 1) * sizeof(T)));
  return 0;
}

static bool config_holds(const rsim_model* m, const int* lim) {
  const int ncg = (int)m->cg.size();
  const bool tendons = m->ntendon > 0 || m->neq > 0;
  return !(m->nbody > lim[0] || m->njnt > lim[1] || m->nv > lim[2] || m->nq > lim[2] + 8 || m->nu > 16 || ncg > lim[3] || m->nsite > lim[4] || m->npair > lim[7] ||
           m->ndynroot > lim[8] || (tendons && !(lim[9] & 1)) || (m->ctrl.enabled && m->ctrl.narm == 2 && !(lim[9] & 2)));
}
// smallest compiled kernel configuration whose lane roles hold the model, -1 if none does; lim (10 ints, may be null) receives its limits
static int pick_config(const rsim_model* m, int* lim_out) {
  for (int c = 0; c < RSIM_NCFG; c++) {
    int lim[10];
    k_limits[c](lim);
    if (config_holds(m, lim)) {
      if (lim_out) memcpy(lim_out, lim, sizeof(lim));
      return c;
    }
  }
  return -1;
}
// the configuration that steps the envs of a batch of configuration `cfg` which outgrow its contact / row capacity: the first tier configuration
// that holds the model and more contacts or rows; -1 if there is none (or RSIM_NO_TIERS is set: drops are then counted in RSIM_OVERFLOW, as before round 4)
static int pick_wide(const rsim_model* m, int cfg, const int* lim, int* lim_out) {
  if (getenv("RSIM_NO_TIERS")) return -1;
  for (int c : k_tier_cfgs) {
    int lw[10];
    k_limits[c](lw);
    if (c != cfg && config_holds(m, lw) && lw[5] >= lim[5] && lw[6] >= lim[6] && (lw[5] > lim[5] || lw[6] > lim[6])) { memcpy(lim_out, lw, sizeof(lw)); return c; }
  }
  return -1;
}
extern "C" int rsim_model_config(const rsim_model* m, int* limits) { return pick_config(m, limits); }

// mj_name2id / mj_id2name (binding_utils.py:296-360: body_name2id, joint_name2id, geom_name2id, site_name2id, actuator_name2id, ... and their inverses)
extern "C" int rsim_name2id(const rsim_model* m, const char* kind, const char* name) {
  if (!kind || !name || !*name) return -1;
  auto it = m->names.find(kind);
  if (it == m->names.end()) { fail("rsim_name2id: the model blob carries no names of kind '%s'", kind); return -1; }
  for (size_t i = 0; i < it->second.size(); i++) if (it->second[i] == name) return (int)i;
  return -1;
}
extern "C" const char* rsim_id2name(const rsim_model* m, const char* kind, int id) {
  auto it = m->names.find(kind ? kind : "");
  if (it == m->names.end() || id < 0 || id >= (int)it->second.size() || it->second[id].empty()) return nullptr;
  return it->second[id].c_str();
}

extern "C" int rsim_batch_create(rsim_model* m, int B, int device, int per_env, rsim_batch** out) {
  if (B < 1) return fail("rsim_batch_create: B < 1");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("rsim_batch_create: no HIP device visible (the HIP backend has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail("rsim_batch_create: device %d out of range (%d visible)", device, ndev);
  rsim_batch* b = new rsim_batch();
  memset(&b->dm, 0, sizeof(b->dm));
  memset(&b->db, 0, sizeof(b->db));
  b->m = m; b->B = B; b->device = device; b->per_env = per_env ? 1 : 0;
  b->gen = 1; b->cache_gen = 0; b->cache_env = -1;
  b->db.prof_env = -1;
  b->d_bank = nullptr; b->d_bank_tag = nullptr; b->d_patch = nullptr; b->d_ft_base = nullptr;
  b->bstream = nullptr; b->bev = nullptr; b->h_epidx = nullptr; b->bank_poll_pending = 0; b->bstage_next = 0;
  memset(b->bstage, 0, sizeof(b->bstage));
  const int ncg = (int)m->cg.size();
  b->cfg = pick_config(m, b->lim);
  if (b->cfg < 0) {
    int lim[10];
    k_limits[RSIM_NCFG - 1](lim);
    int r = fail("rsim_batch_create: model (nbody %d njnt %d nv %d ncgeom %d nsite %d npair %d, %d articulated trees) exceeds the largest compiled kernel configuration "
                 "(%d %d %d %d %d .. %d, %d trees)", m->nbody, m->njnt, m->nv, ncg, m->nsite, m->npair, m->ndynroot, lim[0], lim[1], lim[2], lim[3], lim[4], lim[7], lim[8]);
    delete b;
    return r;
  }
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipStreamCreate(&b->stream));
  if (dalloc(&b->d_it, m->itab.size())) return 1;
  HIPCHK(hipMemcpy(b->d_it, m->itab.data(), m->itab.size() * sizeof(int), hipMemcpyHostToDevice));
  size_t fs = m->ftab.size();
  size_t copies = b->per_env ? (size_t)B : 1;
  if (dalloc(&b->d_ft, fs * copies)) return 1;
  for (size_t c = 0; c < copies; c++) HIPCHK(hipMemcpy(b->d_ft + c * fs, m->ftab.data(), fs * sizeof(float), hipMemcpyHostToDevice));
  if (dalloc(&b->d_ft0, fs)) return 1;
  HIPCHK(hipMemcpy(b->d_ft0, m->ftab.data(), fs * sizeof(float), hipMemcpyHostToDevice));
  if (dalloc(&b->d_mesh, m->mesh_vert.size())) return 1;
  HIPCHK(hipMemcpy(b->d_mesh, m->mesh_vert.data(), m->mesh_vert.size() * sizeof(float), hipMemcpyHostToDevice));
  if (dalloc(&b->d_mask, (size_t)B)) return 1;
  b->cm_bytes = (size_t)k_cmem_bytes[b->cfg]();
  if (dalloc((char**)&b->d_cm, b->cm_bytes * (1 + (b->per_env ? (size_t)B : 0)))) return 1;   // block 0: shared; blocks 1..B: per env
  b->db.cm = b->d_cm; b->db.cm_env = (char*)b->d_cm + b->cm_bytes; b->db.cm_stride = 0; b->cm_dirty = 1;
  memset(&b->cm_ctrl, 0, sizeof(b->cm_ctrl));
  b->d_order = nullptr; b->d_cost = nullptr; b->schedule = 1; b->have_cost = 0;
  if (dalloc(&b->d_order, (size_t)B)) return 1;
  if (dalloc(&b->d_cost, (size_t)B)) return 1;
  b->ostream = nullptr; b->nstep = 0;
  { const char* e = getenv("RSIM_ORDER_FRESH"); b->order_fresh = e ? atoi(e) : 1; }
  for (int k = 0; k < 2; k++) {
    b->d_order2[k] = nullptr; b->d_cost2[k] = nullptr; b->step_done[k] = nullptr; b->ord_done[k] = nullptr; b->ord_valid[k] = 0;
    if (dalloc(&b->d_order2[k], (size_t)B) || dalloc(&b->d_cost2[k], (size_t)B)) return 1;
  }
  if (dalloc(&b->db.needs_reset, (size_t)B)) return 1;
  b->db.mprc = nullptr;
  b->db.jg = nullptr;
  if (!getenv("RSIM_NO_MPR_WARMSTART") && m->npair > 0 && dalloc(&b->db.mprc, (size_t)B * m->npair * 12)) return 1;
  b->db.mprc_portal = getenv("RSIM_NO_MPR_PORTAL_WARMSTART") ? 0 : 1;
  b->db.bpl = nullptr;
  if (m->npair > 0 && m->cg.size() <= 64 && dalloc(&b->db.bpl, (size_t)B * 320)) return 1;   // broadphase pair list (lane = colliding geom: 64 at most)
  b->db.ft_rw = b->d_ft;
  DModel& dm = b->dm;
  dm.nq = m->nq; dm.nv = m->nv; dm.nu = m->nu; dm.nbody = m->nbody; dm.njnt = m->njnt; dm.ncg = ncg; dm.nsite = m->nsite; dm.npair = m->npair;
  dm.maxdepth = m->maxdepth; dm.nroot = m->nroot;
  dm.ntendon = m->ntendon; dm.neq = m->neq; dm.nsensor = m->nsensor; dm.nsensordata = m->nsensordata;
  {
    // dofs are numbered tree by tree (depth first): the prefix that holds every tree with a damped joint
    const int* dbody = m->I("dof_bodyid"); const int* root = m->I("body_rootid");
    int last = -1;
    for (int i = 0; i < m->nv; i++) if (m->D("dof_damping")[i] != 0.0) last = i;
    int nd = last + 1;
    while (nd < m->nv && last >= 0 && root[dbody[nd]] == root[dbody[last]]) nd++;
    dm.nv_damped = getenv("RSIM_EULER_FULL") ? m->nv : nd;
  }
  dm.iterations = m->I("iterations") ? m->I("iterations")[0] : 100;
  dm.ls_iterations = 50; dm.cone = m->I("cone") ? m->I("cone")[0] : 1; dm.solver = 1;
  dm.tolerance = m->D("tolerance") ? (float)m->D("tolerance")[0] : 1e-8f;
  dm.meaninertia = m->meaninertia;
  dm.newton_ns = getenv("RSIM_NEWTON_NS") ? (float)atof(getenv("RSIM_NEWTON_NS")) : RSIM_NEWTON_NS;
  dm.newton_na = getenv("RSIM_NEWTON_NA") ? (float)atof(getenv("RSIM_NEWTON_NA")) : RSIM_NEWTON_NA;
  dm.mpr_cone = getenv("RSIM_MPR_CONE") ? (float)atof(getenv("RSIM_MPR_CONE")) : RSIM_MPR_CONE;
  dm.near_gain = getenv("RSIM_NEAR_GAIN") ? (float)atof(getenv("RSIM_NEAR_GAIN")) : RSIM_NEAR_GAIN;
  dm.near_thresh = getenv("RSIM_NEAR_THRESH") ? (float)atof(getenv("RSIM_NEAR_THRESH")) : RSIM_NEAR_THRESH;
  dm.bp_reach = getenv("RSIM_BP_REACH") ? (float)atof(getenv("RSIM_BP_REACH")) : RSIM_BP_REACH;
  dm.newton_wide = getenv("RSIM_NEWTON_WIDE") ? atoi(getenv("RSIM_NEWTON_WIDE")) : 1;
  dm.newton_ls = getenv("RSIM_NEWTON_LS") ? (float)atof(getenv("RSIM_NEWTON_LS")) : RSIM_NEWTON_LS;
  // refinement passes with an fp64 gradient: on for the models of the 64 x 48 class and beyond (PickPlace: mesh objects and Robotiq links of 1e-5 .. 4e-3 kg m^2
  // under condim-4 contacts at the refsafe limit); Stack-class models reach 2e-5 of the oracle without it and would pay 13 % for it
  dm.newton_refine = getenv("RSIM_NEWTON_REFINE") ? atoi(getenv("RSIM_NEWTON_REFINE")) : (b->cfg >= 3 ? 16 : 0);
  if (dm.newton_refine > 0 && b->cfg < 3 && getenv("RSIM_NEWTON_REFINE"))
    fprintf(stderr, "[rsim] RSIM_NEWTON_REFINE=%d ignored: the fp64 polish is compiled into the 64 x 48 / 64 x 64 configurations and their tiers only (this batch runs configuration %d)\n", dm.newton_refine, b->cfg);
  dm.newton_polish_tol = getenv("RSIM_POLISH_TOL") ? (float)atof(getenv("RSIM_POLISH_TOL")) : 1.0f;
  dm.newton_polish_gate = getenv("RSIM_POLISH_GATE") ? (float)atof(getenv("RSIM_POLISH_GATE")) : 0.0f;
  dm.newton_exact = getenv("RSIM_NEWTON_EXACT") ? atoi(getenv("RSIM_NEWTON_EXACT")) : 1;
  dm.newton_ng = getenv("RSIM_NEWTON_NG") ? (float)atof(getenv("RSIM_NEWTON_NG")) : RSIM_NEWTON_NG;
  if (m->multijoint) { int r = fail("rsim_batch_create: bodies with more than one joint are not supported by the fused kernel"); delete b; return r; }
  if (m->D("jnt_stiffness")) for (int j = 0; j < m->njnt; j++) if (m->D("jnt_stiffness")[j] != 0.0) {
    int r = fail("rsim_batch_create: joint %d has a spring (stiffness %g): passive joint springs are not implemented in the fused kernel (tendon springs are)", j, m->D("jnt_stiffness")[j]);
    delete b; return r;
  }

  if (m->maxcondim > 4) { int r = fail("rsim_batch_create: condim %d contacts are not supported by the compiled kernel configuration (max 4)", m->maxcondim); delete b; return r; }
  for (int j = 0; j < m->njnt; j++) if (m->I("jnt_type")[j] == 1) { int r = fail("rsim_batch_create: ball joints are not supported by the fused kernel"); delete b; return r; }
  if (dalloc(&b->d_lt, m->lanetab.size())) return 1;
  HIPCHK(hipMemcpy(b->d_lt, m->lanetab.data(), m->lanetab.size() * sizeof(int), hipMemcpyHostToDevice));
  dm.lt = b->d_lt; dm.kin_rounds = m->kin_rounds; dm.ndynroot = m->ndynroot; dm.maxcondim = m->maxcondim;
  for (int r = 0; r < RSIM_MAXDYNROOT; r++) dm.dynroot[r] = r < m->ndynroot ? m->dynroot[r] : 0;
  dm.ft0 = b->d_ft0; dm.fenv = 0;
  dm.it = b->d_it; dm.ft = b->d_ft; dm.mesh_vert = b->d_mesh; dm.fstride = b->per_env ? (int)fs : 0;
  memcpy(dm.io, m->io, sizeof(dm.io));
  memcpy(dm.fo, m->fo, sizeof(dm.fo));
  dm.ctrl = m->ctrl;
  b->cs = m->ctrl.enabled ? m->ctrl.cs_size : RSIM_CS_SIZE;
  dm.ctrl.cs_size = b->cs;
  // capacity tiers: only for controllers whose state lives in LDS for the whole launch (a step that is handed over must not have written anything)
  b->cfg_w = b->cs <= RSIM_CS_LDS ? pick_wide(m, b->cfg, b->lim, b->lim_w) : -1;
  b->fused = 0; b->share_cm = 0;
  if (b->cfg >= 0 && b->cfg <= 2 && (b->lim[9] & 32) && b->cs <= RSIM_CS_LDS && !getenv("RSIM_NO_TIERS")) {
    int lw[10];
    if (k_limits_w[b->cfg](lw) && config_holds(m, lw)) { memcpy(b->lim_w, lw, sizeof(lw)); b->fused = 1; if (b->cfg_w < 0) b->cfg_w = b->cfg; }   // (cfg_w only marks "tiered" from here on: a fused batch launches no tier kernel)
  }
  {
    // builds that keep the constraint Jacobian in global memory (RSIM_JGLOBAL: limits bit 2) get their per-env buffer, [B][NEFC * (NV + 1)] floats
    // limits bit 3 (RSIM_MGLOBAL): the mass matrix behind J in the same buffer, NV * (NV + 1) floats more.  One stride for the native and the wide configuration.
    // limits bit 4 (RSIM_CGLOBAL): the contact block (frames and material parameters, 22 floats per contact) behind M.
    auto jg_need = [](const int* lim) -> size_t {
      if (!(lim[9] & 4)) return 0;
      return (size_t)lim[6] * (size_t)(lim[2] + 1) + ((lim[9] & 8) ? (size_t)lim[2] * (size_t)(lim[2] + 1) : 0) + ((lim[9] & 16) ? (size_t)lim[5] * 22 : 0);
    };
    size_t jgf = jg_need(b->lim);
    if (b->cfg_w >= 0) jgf = std::max(jgf, jg_need(b->lim_w));
    b->db.jg_stride = (long long)jgf;
    if (jgf && dalloc(&b->db.jg, (size_t)B * jgf)) return 1;
  }
  b->d_cm_w = nullptr; b->d_tier[0] = b->d_tier[1] = nullptr; b->d_wlist[0] = b->d_wlist[1] = nullptr; b->d_wcount = nullptr; b->wstream = nullptr; b->tier_flip = 0;
  if (b->cfg_w >= 0) {
    b->cm_bytes_w = (size_t)cmem_bytes_any(b->cfg_w);
    b->share_cm = b->cm_bytes_w == b->cm_bytes && b->lim_w[0] == b->lim[0] && b->lim_w[1] == b->lim[1] && b->lim_w[2] == b->lim[2] && b->lim_w[3] == b->lim[3] &&
                  b->lim_w[4] == b->lim[4] && b->lim_w[7] == b->lim[7] && !getenv("RSIM_NO_SHARE_CM");
    if (dalloc((char**)&b->d_cm_w, b->cm_bytes_w * (1 + (b->per_env ? (size_t)B : 0)))) return 1;
    for (int k = 0; k < 2; k++) if (dalloc(&b->d_tier[k], (size_t)B) || dalloc(&b->d_wlist[k], (size_t)B)) return 1;
    if (dalloc(&b->d_wcount, (size_t)2 * (2 + 2 * RSIM_MAX_GROUPS))) return 1;
    if (dalloc(&b->db.tstat, (size_t)2)) return 1;
    HIPCHK(hipMemset(b->db.tstat, 0, 2 * sizeof(unsigned long long)));
    {
      // the wide pass runs beside the native one on a stream of the highest priority (its few workgroups are the slowest envs of the step)
      int lo = 0, hi = 0;
      HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
      HIPCHK(hipStreamCreateWithPriority(&b->wstream, hipStreamNonBlocking, hi));
    }
    b->tier_mode = getenv("RSIM_TIER_MODE") ? atoi(getenv("RSIM_TIER_MODE")) : 0;
    HIPCHK(hipEventCreateWithFlags(&b->wfork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&b->wjoin, hipEventDisableTiming));
  }
  b->d_obsprog = nullptr;
  memset(&dm.task, 0, sizeof(dm.task));
  if (m->has_task) {
    const rsim_task_desc& t = m->task;
    if (dalloc(&b->d_obsprog, (size_t)RSIM_OBS_MAX * 3)) return 1;
    HIPCHK(hipMemcpy(b->d_obsprog, t.obs_prog, sizeof(t.obs_prog), hipMemcpyHostToDevice));
    dm.task.enabled = 1; dm.task.nobs = t.nobs; dm.task.task = t.task; dm.task.object_body = t.object_body; dm.task.grip_site = t.grip_site;
    dm.task.reward_shaping = t.reward_shaping; dm.task.table_height = t.table_height; dm.task.lift_margin = t.lift_margin; dm.task.reward_scale = t.reward_scale;
    dm.task.left_pad = t.left_pad_geoms; dm.task.right_pad = t.right_pad_geoms; dm.task.object_geoms = t.object_geoms; dm.task.obs_prog = b->d_obsprog;
    dm.task.object2_body = t.object2_body; dm.task.object2_geoms = t.object2_geoms;
    dm.task.nobj = t.nobj; dm.task.eef_body = t.eef_body; dm.task.single_mode = t.single_object_mode;
    for (int i = 0; i < 4; i++) { dm.task.obj_body[i] = t.obj_body[i]; dm.task.pos_slot[i] = t.pos_slot[i]; dm.task.obj_geoms[i] = t.obj_geoms[i]; }
    for (int i = 0; i < 3; i++) dm.task.bin2_pos[i] = t.bin2_pos[i];
    for (int i = 0; i < 2; i++) dm.task.bin_size[i] = t.bin_size[i];
    for (int i = 0; i < 8; i++) dm.task.bin_target[i] = t.bin_target[i];
  }
  DBatch& db = b->db;
  db.B = B;
  const int nq = m->nq, nv = m->nv, nu = m->nu, nb = m->nbody, NCON = b->lim[5], NEFC = b->lim[6];
  struct { int id; void** p; size_t n; int is_int; } fields[] = {
      {RSIM_QPOS, (void**)&db.qpos, (size_t)B * nq, 0}, {RSIM_QVEL, (void**)&db.qvel, (size_t)B * nv, 0}, {RSIM_QACC_WARMSTART, (void**)&db.qacc_ws, (size_t)B * nv, 0},
      {RSIM_CTRL, (void**)&db.ctrl, (size_t)B * nu, 0}, {RSIM_TIME, (void**)&db.time, (size_t)B, 0}, {RSIM_CSTATE, (void**)&db.cstate, (size_t)B * b->cs, 0},
      {RSIM_XPOS, (void**)&db.xpos, (size_t)B * nb * 3, 0}, {RSIM_XQUAT, (void**)&db.xquat, (size_t)B * nb * 4, 0}, {RSIM_QM, (void**)&db.qM, (size_t)B * nv * nv, 0},
      {RSIM_QFRC_BIAS, (void**)&db.qfrc_bias, (size_t)B * nv, 0}, {RSIM_QFRC_PASSIVE, (void**)&db.qfrc_passive, (size_t)B * nv, 0},
      {RSIM_QFRC_ACTUATOR, (void**)&db.qfrc_actuator, (size_t)B * nv, 0}, {RSIM_QFRC_CONSTRAINT, (void**)&db.qfrc_constraint, (size_t)B * nv, 0},
      {RSIM_QACC, (void**)&db.qacc, (size_t)B * nv, 0}, {RSIM_CDOF, (void**)&db.cdof, (size_t)B * nv * 6, 0}, {RSIM_ROOTCOM, (void**)&db.rootcom, (size_t)B * nb * 3, 0},
      {RSIM_CONTACT, (void**)&db.contact, (size_t)B * NCON * RSIM_CON_REC, 0}, {RSIM_EFC_FORCE, (void**)&db.efc_force, (size_t)B * NEFC, 0},
      {RSIM_NCON, (void**)&db.ncon, (size_t)B, 1}, {RSIM_NEFC, (void**)&db.nefc, (size_t)B, 1}, {RSIM_NITER, (void**)&db.niter, (size_t)B, 1},
      {RSIM_OBS, (void**)&db.obs, (size_t)B * (m->has_task ? m->task.nobs : 0), 0}, {RSIM_REWARD, (void**)&db.reward, (size_t)B, 0},
      {RSIM_SUCCESS, (void**)&db.success, (size_t)B, 1}, {RSIM_DONE, (void**)&db.done, (size_t)B, 1}, {RSIM_EP_STEP, (void**)&db.ep_step, (size_t)B, 1},
      {RSIM_EP_INDEX, (void**)&db.ep_index, (size_t)B, 1}, {RSIM_DIVERGED, (void**)&db.diverged, (size_t)B, 1}, {RSIM_OVERFLOW, (void**)&db.overflow, (size_t)B, 1},
      {RSIM_BANK_STALE, (void**)&db.bank_stale, (size_t)B, 1}, {RSIM_TERMINAL_OBS, (void**)&db.term_obs, (size_t)B * (m->has_task ? m->task.nobs : 0), 0},
      {RSIM_SENSORDATA, (void**)&db.sensordata, (size_t)B * m->nsensordata, 0}, {RSIM_TASK_OBJECT, (void**)&db.task_object, (size_t)B, 1},
      {RSIM_CAP_NEED, (void**)&db.cap_need, (size_t)B * 2, 1}, {RSIM_QFRC_APPLIED, (void**)&db.qfrc_applied, (size_t)B * nv, 0},
      {RSIM_POLISH, (void**)&db.polish, (size_t)B, 1}};
  for (auto& fd : fields) {
    if (dalloc((float**)fd.p, fd.n)) return 1;
    b->fptr[fd.id] = *fd.p; b->fcount[fd.id] = fd.n; b->fis_int[fd.id] = fd.is_int;
  }
  *out = b;
  return rsim_reset(b, nullptr);
}

extern "C" void rsim_batch_free(rsim_batch* b) {
  if (!b) return;
  hipSetDevice(b->device);
  join_groups(b);
  hipStreamSynchronize(b->stream);
  for (int g = 0; g < b->groups; g++) { hipStreamDestroy(b->gstream[g]); hipEventDestroy(b->gev[g]); }
  if (b->mev) hipEventDestroy(b->mev);
  for (int i = 0; i < RSIM_FIELD_COUNT; i++) if (b->fptr[i]) hipFree(b->fptr[i]);
  hipFree(b->d_cm);
  if (b->cfg_w >= 0) {
    hipStreamSynchronize(b->wstream); hipStreamDestroy(b->wstream); hipEventDestroy(b->wfork); hipEventDestroy(b->wjoin);
    hipFree(b->d_cm_w); hipFree(b->d_wcount); hipFree(b->db.tstat);
    for (int k = 0; k < 2; k++) { hipFree(b->d_tier[k]); hipFree(b->d_wlist[k]); }
  }
  if (b->db.mprc) hipFree(b->db.mprc);
  if (b->db.jg) hipFree(b->db.jg);
  if (b->db.bpl) hipFree(b->db.bpl);
  hipFree(b->d_it); hipFree(b->d_lt); hipFree(b->d_ft); hipFree(b->d_ft0); if (b->d_obsprog) hipFree(b->d_obsprog);
  if (b->d_bank) hipFree(b->d_bank); if (b->d_bank_tag) hipFree(b->d_bank_tag); if (b->d_patch) hipFree(b->d_patch); hipFree(b->db.needs_reset); if (b->d_ft_base) hipFree(b->d_ft_base); hipFree(b->d_mesh); hipFree(b->d_mask);
  if (b->d_order) hipFree(b->d_order); if (b->d_cost) hipFree(b->d_cost);
  if (b->ostream) { hipStreamSynchronize(b->ostream); hipStreamDestroy(b->ostream); }
  for (int k = 0; k < 2; k++) {
    if (b->d_order2[k]) hipFree(b->d_order2[k]); if (b->d_cost2[k]) hipFree(b->d_cost2[k]);
    if (b->step_done[k]) hipEventDestroy(b->step_done[k]); if (b->ord_done[k]) hipEventDestroy(b->ord_done[k]);
  }
  if (b->db.prof) hipFree(b->db.prof);
  if (b->bstream) { hipStreamSynchronize(b->bstream); hipStreamDestroy(b->bstream); }
  if (b->bev) hipEventDestroy(b->bev);
  if (b->h_epidx) hipHostFree(b->h_epidx);
  for (auto& st : b->bstage) { if (st.host) hipHostFree(st.host); if (st.dev) hipFree(st.dev); if (st.done) hipEventDestroy(st.done); }
  hipStreamDestroy(b->stream);
  delete b;
}
extern "C" int rsim_batch_size(const rsim_batch* b) { return b->B; }
extern "C" int rsim_batch_limits(const rsim_batch* b, int* maxcon, int* maxefc) { *maxcon = b->lim[5]; *maxefc = b->lim[6]; return 0; }
extern "C" void* rsim_stream(rsim_batch* b) { return (void*)b->stream; }
extern "C" void* rsim_group_stream(rsim_batch* b, int g) { return (b->ngroups > 1 && g >= 0 && g < b->ngroups) ? (void*)b->gstream[g] : (void*)b->stream; }
extern "C" int rsim_sync(rsim_batch* b) { if (join_groups(b)) return 1; HIPCHK(hipSetDevice(b->device)); HIPCHK(hipStreamSynchronize(b->stream)); return 0; }

extern "C" int rsim_reset(rsim_batch* b, const uint8_t* mask) { if (join_groups(b)) return 1;
  rsim_model* m = b->m;
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  const int nq = m->nq, nv = m->nv, nu = m->nu, B = b->B;
  std::vector<float> q0(nq);
  for (int i = 0; i < nq; i++) q0[i] = (float)m->D("qpos0")[i];
  std::vector<float> zv(nv > nu ? nv : nu, 0.f);
  if (!mask) {
    std::vector<float> all((size_t)B * nq);
    for (int e = 0; e < B; e++) memcpy(&all[(size_t)e * nq], q0.data(), nq * sizeof(float));
    HIPCHK(hipMemcpy(b->db.qpos, all.data(), all.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(b->db.qvel, 0, (size_t)B * nv * sizeof(float)));
    HIPCHK(hipMemset(b->db.qacc_ws, 0, (size_t)B * nv * sizeof(float)));
    HIPCHK(hipMemset(b->db.ctrl, 0, (size_t)B * nu * sizeof(float)));
    HIPCHK(hipMemset(b->db.qfrc_applied, 0, (size_t)B * nv * sizeof(float)));
    HIPCHK(hipMemset(b->db.time, 0, (size_t)B * sizeof(float)));
    HIPCHK(hipMemset(b->db.cstate, 0, (size_t)B * b->cs * sizeof(float)));
    HIPCHK(hipMemset(b->db.ep_step, 0, (size_t)B * sizeof(int))); HIPCHK(hipMemset(b->db.ep_index, 0, (size_t)B * sizeof(int)));
    HIPCHK(hipMemset(b->db.done, 0, (size_t)B * sizeof(int))); HIPCHK(hipMemset(b->db.needs_reset, 0, (size_t)B * sizeof(int)));
  } else {
    for (int e = 0; e < B; e++) {
      if (!mask[e]) continue;
      HIPCHK(hipMemcpy(b->db.qpos + (size_t)e * nq, q0.data(), nq * sizeof(float), hipMemcpyHostToDevice));
      HIPCHK(hipMemset(b->db.qvel + (size_t)e * nv, 0, nv * sizeof(float)));
      HIPCHK(hipMemset(b->db.qacc_ws + (size_t)e * nv, 0, nv * sizeof(float)));
      HIPCHK(hipMemset(b->db.ctrl + (size_t)e * nu, 0, nu * sizeof(float)));
      HIPCHK(hipMemset(b->db.qfrc_applied + (size_t)e * nv, 0, nv * sizeof(float)));
      HIPCHK(hipMemset(b->db.time + e, 0, sizeof(float)));
      HIPCHK(hipMemset(b->db.cstate + (size_t)e * b->cs, 0, b->cs * sizeof(float)));
      HIPCHK(hipMemset(b->db.ep_step + e, 0, sizeof(int))); HIPCHK(hipMemset(b->db.done + e, 0, sizeof(int))); HIPCHK(hipMemset(b->db.needs_reset + e, 0, sizeof(int)));
    }
  }
  if (b->db.mprc) HIPCHK(hipMemset(b->db.mprc, 0, (size_t)B * m->npair * 12 * sizeof(float)));   // no warm start carried into a reset state (bitwise replays)
  b->gen++;
  return 0;
}

// Stream groups.  A control step is one launch whose duration is that of its slowest environment (contact-rich envs take 3-4 x the median),
// while the envs are independent of each other: with G groups, env block g steps on its own HIP stream, so block g's control step t + 1
// starts when ITS slowest env has finished step t and fills the CUs the other blocks' stragglers leave idle.  Nothing changes for the caller:
// rsim_control_step() still enqueues one step of all B envs and returns; every other entry point first makes the batch's main stream wait
// for the group streams (join), and the first control step after that makes the group streams wait for the main stream (fork).
static int join_groups(rsim_batch* b) {
  if (!b->forked) return 0;
  for (int g = 0; g < b->groups; g++) {
    HIPCHK(hipEventRecord(b->gev[g], b->gstream[g]));
    HIPCHK(hipStreamWaitEvent(b->stream, b->gev[g], 0));
  }
  b->forked = 0;
  return 0;
}
static int fork_groups(rsim_batch* b) {
  if (b->forked) return 0;
  HIPCHK(hipEventRecord(b->mev, b->stream));
  for (int g = 0; g < b->groups; g++) HIPCHK(hipStreamWaitEvent(b->gstream[g], b->mev, 0));
  b->forked = 1;
  return 0;
}
extern "C" int rsim_set_stream_groups(rsim_batch* b, int groups) {
  if (groups < 1 || groups > RSIM_MAX_GROUPS || groups > b->B) return fail("rsim_set_stream_groups: groups must be in 1..%d and <= the batch size", RSIM_MAX_GROUPS);
  HIPCHK(hipSetDevice(b->device));
  if (join_groups(b)) return 1;
  HIPCHK(hipStreamSynchronize(b->stream));
  for (int g = b->groups; g < groups; g++) {   // streams / events are created once and kept until the batch is freed
    {
      // a stream with a CU mask owns a hardware queue of its own; plain streams share the runtime's small pool of queues (GPU_MAX_HW_QUEUES,
      // 4 by default, torch's own streams included), and kernels of two groups that land on one queue run back to back instead of side by side
      hipDeviceProp_t prop;
      HIPCHK(hipGetDeviceProperties(&prop, b->device));
      const uint32_t words = (uint32_t)((prop.multiProcessorCount + 31) / 32);
      std::vector<uint32_t> mask(words, 0xFFFFFFFFu);
      if (getenv("RSIM_PLAIN_GROUP_STREAMS") || hipExtStreamCreateWithCUMask(&b->gstream[g], words, mask.data()) != hipSuccess) {
        (void)hipGetLastError();
        HIPCHK(hipStreamCreate(&b->gstream[g]));   // still correct, but groups that share a hardware queue serialise
      }
    }
    HIPCHK(hipEventCreateWithFlags(&b->gev[g], hipEventDisableTiming));
  }
  if (!b->mev) HIPCHK(hipEventCreateWithFlags(&b->mev, hipEventDisableTiming));
  if (groups > b->groups) b->groups = groups;
  if (b->cfg_w >= 0) HIPCHK(hipMemset(b->d_wcount, 0, (size_t)2 * (2 + 2 * RSIM_MAX_GROUPS) * sizeof(int)));   // another partition of the batch: no list length of the old one survives
  b->ngroups = groups;
  b->have_cost = 0;
  return 0;
}

// The controller may be re-configured on the model after the batch exists (gains, limits, even the type), but the per-env controller
// state buffer was sized when the batch was created: the kernels stride it by b->cs, so a controller that needs more state than that
// is refused instead of running past the end of the buffer.
static int sync_controller(rsim_batch* b) {
  b->dm.ctrl = b->m->ctrl;
  if (b->m->ctrl.enabled && b->m->ctrl.cs_size > b->cs)
    return fail("the controller configured on the model needs %d floats of per-env state but the batch was created with %d: call rsim_model_set_controller "
                "before rsim_batch_create (or create a new batch)", b->m->ctrl.cs_size, b->cs);
  b->dm.ctrl.cs_size = b->cs;
  return 0;
}

// (Re)build the constant blocks when a model parameter or the controller changed since they were built: one block while every env reads the
// shared float table, one per env from the moment a field has per-env values (per-episode object sizes, domain randomisation).
static int ensure_constants(rsim_batch* b) {
  if (!b->cm_dirty && !memcmp(&b->cm_ctrl, &b->dm.ctrl, sizeof(DCtrl))) return 0;
  const bool per_env = b->per_env && b->dm.fenv != 0;
  b->db.cm_stride = per_env ? (long long)b->cm_bytes : 0;
  {   // the shared block: every field from the shared float table
    DModel dm0 = b->dm; DBatch db0 = b->db;
    dm0.fenv = 0; db0.cm_env = b->d_cm; db0.cm_stride = 0;
    int e = k_prepare_launch[b->cfg](&dm0, &db0, 1, 0, b->stream);
    if (e) return fail("constant-block kernel launch failed: %s", hipGetErrorString((hipError_t)e));
  }
  if (per_env) {
    int e = k_prepare_launch[b->cfg](&b->dm, &b->db, b->B, 0, b->stream);
    if (e) return fail("constant-block kernel launch failed: %s", hipGetErrorString((hipError_t)e));
  }
  if (b->cfg_w >= 0 && !b->fused && !b->share_cm) {   // the shared block of the wide configuration (its per-env blocks are built on demand, right before a wide pass; a fused wide body reads the native blocks)
    DModel dm0 = b->dm; DBatch db0 = b->db;
    dm0.fenv = 0; db0.cm_env = b->d_cm_w; db0.cm_stride = 0;
    int e = prepare_launch_any(b->cfg_w)(&dm0, &db0, 1, 0, b->stream);
    if (e) return fail("constant-block kernel launch failed: %s", hipGetErrorString((hipError_t)e));
  }
  b->cm_ctrl = b->dm.ctrl;
  b->cm_dirty = 0;
  return 0;
}

// One wide pass of a tiered control step on `stream`: the constant blocks of the listed envs (only when envs have blocks of their own), then the step.
// pass 1 = the envs whose tier is 1 (list built by rsim_launch_tier_list), pass 2 = the redo list the native pass appended to.

// RSIM_TRACE_STEPS=N: HIP events at fixed points of the main stream in N consecutive control steps, mean intervals printed to stderr once -- where the time
// between two control-step kernels goes WITHOUT a profiler attached (rocprofv3 changes the cross-queue behaviour it is supposed to show)
struct StepTrace { int n = -1, k = 0; std::vector<hipEvent_t> ev; bool done = false; };
static StepTrace g_tr;
static const char* TR_NAMES[8] = {"step start", "order sorted", "k_step launched (fork + wide chain enqueued before it)", "k_step done", "wide pass joined", "redo pass done", "reset constants done", "reset observations done"};
static void tr_mark(rsim_batch* b, int point) {
  if (g_tr.n < 0) { const char* e = getenv("RSIM_TRACE_STEPS"); g_tr.n = e ? atoi(e) : 0; if (g_tr.n > 0) { g_tr.ev.resize((size_t)g_tr.n * 8); for (auto& x : g_tr.ev) hipEventCreate(&x); } }
  if (g_tr.n <= 0 || g_tr.done || g_tr.k >= g_tr.n) return;
  hipEventRecord(g_tr.ev[(size_t)g_tr.k * 8 + point], b->stream);
  if (point == 7 && ++g_tr.k == g_tr.n) {
    hipStreamSynchronize(b->stream);
    double acc[8] = {0}; int cnt = 0;
    for (int k = g_tr.n / 4; k + 1 < g_tr.n; k++, cnt++) {
      for (int i = 0; i < 7; i++) { float ms = 0; hipEventElapsedTime(&ms, g_tr.ev[(size_t)k * 8 + i], g_tr.ev[(size_t)k * 8 + i + 1]); acc[i] += ms; }
      float ms = 0; hipEventElapsedTime(&ms, g_tr.ev[(size_t)k * 8 + 7], g_tr.ev[(size_t)(k + 1) * 8]); acc[7] += ms;
    }
    fprintf(stderr, "[rsim trace] mean over %d control steps, us between main-stream points:\n", cnt);
    for (int i = 0; i < 8; i++) fprintf(stderr, "[rsim trace]   %-78s -> %8.1f\n", TR_NAMES[i], 1e3 * acc[i] / cnt);
    g_tr.done = true;
  }
}

static inline bool sched1_trace(int flags) { return (flags & RF_EPISODE) && (flags & RF_CTRL); }
static int wide_pass(rsim_batch* b, const float* actions, int n_sub, int flags, int pass, const int* list, const int* count, hipStream_t stream) {
  DBatch dw = b->db;
  if (!b->share_cm) { dw.cm = b->d_cm_w; dw.cm_env = (char*)b->d_cm_w + b->cm_bytes_w; dw.cm_stride = b->db.cm_stride ? (long long)b->cm_bytes_w : 0; }   // else: the native blocks, as b->db has them
  dw.order = nullptr; dw.cost = nullptr; dw.env0 = 0; dw.nenv = 0;
  dw.tier_pass = pass; dw.wlist = list; dw.wcount = count; dw.tier_con = b->lim[5]; dw.tier_efc = b->lim[6];
  const int grid = b->B < 512 ? b->B : 512;
  if (dw.cm_stride && !b->share_cm) {
    int e = prepare_launch_any(b->cfg_w)(&b->dm, &dw, grid, 2, stream);
    if (e) return fail("constant-block kernel launch failed: %s", hipGetErrorString((hipError_t)e));
  }
  int e = step_list_launch(b->cfg_w)(&b->dm, &dw, actions, n_sub, flags, grid, stream);
  if (e) return fail("wide-tier kernel launch failed: %s", hipGetErrorString((hipError_t)e));
  return 0;
}

static int launch(rsim_batch* b, const float* actions, int n_sub, int flags) {
  const int WCN = 2 + 2 * RSIM_MAX_GROUPS;   // list lengths per step parity
  HIPCHK(hipSetDevice(b->device));
  if (sync_controller(b)) return 1;
  // capacity tiers apply to the fused control step only (the debug entries keep the native capacity and count what they drop in RSIM_OVERFLOW)
  const bool tiered = b->cfg_w >= 0 && (flags & RF_EPISODE) && (flags & RF_CTRL) && !(flags & RF_DEBUG);
  b->db.tier_cur = tiered ? b->d_tier[b->tier_flip] : nullptr; b->db.tier_next = tiered ? b->d_tier[b->tier_flip ^ 1] : nullptr;
  // how close to its native capacity an env moves up BEFORE it overflows.  A flagged env is stepped by the wide pass, whose workgroups only find room once the
  // native launch has no workgroup pending (a freed 20 KB slot never fits a 32 - 116 KB workgroup): it starts ~2 ms into the step and, being one of the most
  // contact-rich envs, ends after it.  An env that overflows unflagged costs more (its step is redone after the native pass), but on the 32 x 16 build (Lift)
  // flagged-and-never-overflowing env-steps outnumber overflows by orders of magnitude (largest demand 16 contacts / 63-66 rows of 16 / 64), so that build
  // flags nothing in advance: 3.42 -> 3.31 ms per control step; Stack is flat between 0 / 0 and 2 / 6 (profiles/r04_x_tier_overhead.txt)
  // (configurations 0 and 1 flag nothing in advance: an env that does overflow is redone by the wide build in the same control step.  Lift: round 4; Stack: round 5,
  // session 14 -- with seven / eight envs per CU the wide workgroups of flagged envs wait longer for room than the rare redo costs: 6.65 -> 6.31 ms per control step)
  const bool no_advance = b->cfg == 0 || b->cfg == 1;
  b->db.tier_up_con = getenv("RSIM_TIER_UP_CON") ? atoi(getenv("RSIM_TIER_UP_CON")) : (no_advance ? 0 : 2);
  b->db.tier_up_efc = getenv("RSIM_TIER_UP_EFC") ? atoi(getenv("RSIM_TIER_UP_EFC")) : (no_advance ? 0 : 6);
  b->db.tier_pass = tiered ? 0 : -1; b->db.wlist = nullptr; b->db.wcount = nullptr; b->db.wlist2 = nullptr; b->db.wcount2 = nullptr;
  // fused tier: k_step picks the body per env and hands over in place; the thresholds at which the wide body lets an env go back are the native capacity's
  const bool fused = tiered && b->fused;
  if (fused) {
    b->db.tier_con = b->lim[5]; b->db.tier_efc = b->lim[6];
    if (const char* e = getenv("RSIM_FORCE_HANDOVER")) b->db.tier_pass = 100 + atoi(e);   // test hook: every native-tier env hands over to the wide body at this substep (tests/test_hip_edge_cases.py)
  }
  const bool grouped = (flags & RF_EPISODE) && (flags & RF_CTRL) && b->ngroups > 1;
  if (!grouped || b->cm_dirty || memcmp(&b->cm_ctrl, &b->dm.ctrl, sizeof(DCtrl))) { if (join_groups(b)) return 1; }   // main-stream work ahead
  if (ensure_constants(b)) return 1;
  if ((flags & RF_OBS) && !b->dm.task.enabled) return fail("the task (observation / reward epilogue) was configured after the batch was created");
  if ((flags & RF_CTRL) && !b->dm.ctrl.enabled) return fail("no controller configured (rsim_model_set_controller)");
  b->db.order = nullptr; b->db.cost = nullptr;
  if (grouped) {
    if (fork_groups(b)) return 1;
    const bool sched = b->schedule && b->d_order;
    for (int g = 0; g < b->ngroups; g++) {
      const int e0 = (int)((long long)b->B * g / b->ngroups), e1 = (int)((long long)b->B * (g + 1) / b->ngroups);
      DBatch db = b->db;
      db.env0 = e0; db.nenv = e1 - e0;
      if (sched) {
        if (b->have_cost) {
          int eo = rsim_launch_order(b->d_cost + e0, b->d_order + e0, e1 - e0, b->gstream[g]);
          if (eo) return fail("dispatch-order kernel launch failed: %s", hipGetErrorString((hipError_t)eo));
          db.order = b->d_order + e0;
        }
        db.cost = b->d_cost;
      }
      if (tiered && !fused) {   // this env block's lists; every pass of the block runs on the block's own stream, one after the other
        int* cnt = b->d_wcount + b->tier_flip * WCN + 2 + 2 * g;
        int el = rsim_launch_tier_list(b->db.tier_cur, b->d_wlist[0] + e0, cnt, b->d_wcount + (b->tier_flip ^ 1) * WCN + 2 + 2 * g, e0, e1 - e0, b->gstream[g]);
        if (el) return fail("tier-list kernel launch failed: %s", hipGetErrorString((hipError_t)el));
        db.wlist2 = b->d_wlist[1] + e0; db.wcount2 = cnt + 1;
      }
      int e = k_step_launch[b->cfg](&b->dm, &db, actions, n_sub, flags, b->gstream[g]);
      if (e) return fail("kernel launch failed: %s", hipGetErrorString((hipError_t)e));
      if (tiered && !fused) {
        int* cnt = b->d_wcount + b->tier_flip * WCN + 2 + 2 * g;
        if (wide_pass(b, actions, n_sub, flags, 1, b->d_wlist[0] + e0, cnt, b->gstream[g])) return 1;
        if (wide_pass(b, actions, n_sub, flags, 2, b->d_wlist[1] + e0, cnt + 1, b->gstream[g])) return 1;
      }
      if (b->db.bank && b->db.horizon > 0) {
        db.order = nullptr; db.cost = nullptr;
        if (b->db.bank_P > 0 && b->db.cm_stride) {
          e = k_prepare_launch[b->cfg](&b->dm, &db, e1 - e0, 1, b->gstream[g]);
          if (e) return fail("constant-block kernel launch failed: %s", hipGetErrorString((hipError_t)e));
        }
        if (b->dm.task.enabled) {
          // the reset-observation pass is no tier pass: an env that ended its episode on the wide tier gets its record from the native build like every other
          db.tier_cur = nullptr; db.tier_next = nullptr; db.tier_pass = -1; db.wlist = nullptr; db.wcount = nullptr; db.wlist2 = nullptr; db.wcount2 = nullptr;
          e = k_reset_obs_launch[b->cfg](&b->dm, &db, b->gstream[g]);
          if (e) return fail("reset-observation kernel launch failed: %s", hipGetErrorString((hipError_t)e));
        }
      }
    }
    if (sched) b->have_cost = 1;
    if (tiered) b->tier_flip ^= 1;
    b->gen++;
    b->derived_stale = 1;
    return 0;
  }
  const bool sched1 = (flags & RF_EPISODE) && (flags & RF_CTRL) && b->schedule && b->d_order;   // fused control steps only: forward()/step1()/step2() launches are one substep long
  const int cur = (int)(b->nstep & 1), prev = cur ^ 1;
  const bool traced = sched1_trace(flags);
  if (traced) tr_mark(b, 0);
  if (sched1) {
    if (!b->ostream) {
      HIPCHK(hipStreamCreateWithFlags(&b->ostream, hipStreamNonBlocking));
      for (int k = 0; k < 2; k++) { HIPCHK(hipEventCreateWithFlags(&b->step_done[k], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&b->ord_done[k], hipEventDisableTiming)); }
    }
    if (b->order_fresh) {
      // on the batch's stream, from the costs of the step just before this one: 12-16 us on the critical path, no cross-stream dependency.  The side-stream
      // form below sorts the costs of step t - 2; an env's duration correlates 0.55 with its duration one step earlier and 0.40-0.52 with two steps
      // earlier, and replaying the measured durations through a list scheduler puts the launch 3.4 % (Lift) / 3.7 % (Stack) shorter with the fresher
      // order (tools/sched_study.py, profiles/r04_x3_sched_study.txt)
      if (b->nstep >= 1) {
        int eo = rsim_launch_order(b->d_cost2[prev], b->d_order2[cur], b->B, b->stream);
        if (eo) return fail("dispatch-order kernel launch failed: %s", hipGetErrorString((hipError_t)eo));
        b->db.order = b->d_order2[cur];
      }
    } else if (b->ord_valid[cur]) { HIPCHK(hipStreamWaitEvent(b->stream, b->ord_done[cur], 0)); b->db.order = b->d_order2[cur]; }
    b->db.cost = b->d_cost2[cur];
  }
  if (traced) tr_mark(b, 1);
  if (tiered && !fused) {
    // beside the native pass, on a stream of its own: the envs whose tier is 1 (they were close to the native capacity, or beyond it, last step)
    int* cnt = b->d_wcount + b->tier_flip * WCN;
    if (b->tier_mode == 1) {   // everything on the batch's stream: list, wide pass, then the native pass
      int el = rsim_launch_tier_list(b->db.tier_cur, b->d_wlist[0], cnt, b->d_wcount + (b->tier_flip ^ 1) * WCN, 0, b->B, b->stream);
      if (el) return fail("tier-list kernel launch failed: %s", hipGetErrorString((hipError_t)el));
      if (wide_pass(b, actions, n_sub, flags, 1, b->d_wlist[0], cnt, b->stream)) return 1;
    } else {
      HIPCHK(hipEventRecord(b->wfork, b->stream));
      HIPCHK(hipStreamWaitEvent(b->wstream, b->wfork, 0));
      int el = rsim_launch_tier_list(b->db.tier_cur, b->d_wlist[0], cnt, b->d_wcount + (b->tier_flip ^ 1) * WCN, 0, b->B, b->wstream);
      if (el) return fail("tier-list kernel launch failed: %s", hipGetErrorString((hipError_t)el));
      if (wide_pass(b, actions, n_sub, flags, 1, b->d_wlist[0], cnt, b->wstream)) return 1;
      HIPCHK(hipEventRecord(b->wjoin, b->wstream));
    }
    b->db.wlist2 = b->d_wlist[1]; b->db.wcount2 = cnt + 1;
  }
  if (traced) tr_mark(b, 2);
  int e = k_step_launch[b->cfg](&b->dm, &b->db, actions, n_sub, flags, b->stream);
  if (e) return fail("kernel launch failed: %s", hipGetErrorString((hipError_t)e));
  if (traced) tr_mark(b, 3);
  if (tiered && !fused) {
    // after both: the envs the native pass had to hand over in mid-step (rare: most move up between steps), redone from their unchanged state
    if (b->tier_mode != 1) HIPCHK(hipStreamWaitEvent(b->stream, b->wjoin, 0));
    if (traced) tr_mark(b, 4);
    if (wide_pass(b, actions, n_sub, flags, 2, b->d_wlist[1], b->d_wcount + b->tier_flip * WCN + 1, b->stream)) return 1;
    b->tier_flip ^= 1;
  } else {
    if (fused) b->tier_flip ^= 1;
    if (traced) tr_mark(b, 4);
  }
  if (traced) tr_mark(b, 5);
  if (sched1) {
    if (!b->order_fresh) HIPCHK(hipEventRecord(b->step_done[cur], b->stream));
    if (!b->order_fresh && b->nstep >= 1) {   // beside the step just launched: sort the costs of the PREVIOUS step into the order of the NEXT one
      HIPCHK(hipStreamWaitEvent(b->ostream, b->step_done[prev], 0));
      int eo = rsim_launch_order(b->d_cost2[prev], b->d_order2[prev], b->B, b->ostream);   // order2[(t + 1) & 1] == order2[prev]
      if (eo) return fail("dispatch-order kernel launch failed: %s", hipGetErrorString((hipError_t)eo));
      HIPCHK(hipEventRecord(b->ord_done[prev], b->ostream));
      b->ord_valid[prev] = 1;
    }
    b->nstep++;
  }
  if ((flags & RF_EPISODE) && b->db.bank && b->db.horizon > 0) {
    if (b->db.bank_P > 0 && b->db.cm_stride) {
      // envs whose episode just ended were re-initialised from the reset bank, float-table patches included: rebuild their constant blocks
      e = k_prepare_launch[b->cfg](&b->dm, &b->db, b->B, 1, b->stream);
      if (e) return fail("constant-block kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    }
    if (traced) tr_mark(b, 6);
    if (b->dm.task.enabled) {
      // ... and their observation record becomes the one MujocoEnv.reset() returns (base.py:298-347): sim.forward() + observables on the reset
      // state, no reward.  The terminal record of the finished episode was moved to RSIM_TERMINAL_OBS by the control step.
      DBatch db2 = b->db;
      db2.order = nullptr; db2.cost = nullptr;
      // not a tier pass: with tier_cur set the native-pass branch of step_body would skip every env that ended its episode on the wide tier (its
      // RSIM_OBS would keep the terminal record) and could append to a redo list nobody walks any more
      db2.tier_cur = nullptr; db2.tier_next = nullptr; db2.tier_pass = -1; db2.wlist = nullptr; db2.wcount = nullptr; db2.wlist2 = nullptr; db2.wcount2 = nullptr;
      e = k_reset_obs_launch[b->cfg](&b->dm, &db2, b->stream);
      if (e) return fail("reset-observation kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    }
    if (traced) tr_mark(b, 7);
  }
  b->gen++;
  b->derived_stale = (flags & RF_DEBUG) ? 0 : 1;
  return 0;
}
// The derived arrays are written by the debug form of the kernel only (rsim_forward / step1 / step2 / step / run_controller / step2_last / observe); the fused
// control step leaves them alone.  A read of one of them after a fused step first brings them up to the CURRENT state with rsim_forward -- what
// robosuite's own sim.forward() before such a read does -- instead of handing out the values of some earlier launch.
static bool derived_field(int f) { return (f >= RSIM_XPOS && f <= RSIM_NITER) || f == RSIM_SENSORDATA; }
extern "C" int rsim_forward(rsim_batch* b);
// The refresh is a READ: it must not change what later control steps compute, nor what the drop / demand metrics say (round-4 advisor finding).  The debug
// forward therefore runs without the overflow / cap_need counters (it has the native capacity only: for an env the wide tier is stepping, the derived
// contact / force arrays hold the first NCON contacts / NEFC rows of the native configuration -- documented in include/rsim.h), without the narrow phase's
// warm-start records and broadphase list (a cold run: same contacts to the MPR tolerance) and without writing the state arrays back (RF_NOSTORE).
static int refresh_forward(rsim_batch* b) {
  DBatch keep = b->db;
  b->db.overflow = nullptr; b->db.cap_need = nullptr; b->db.mprc = nullptr; b->db.bpl = nullptr;
  const int r = launch(b, nullptr, 1, RF_POSVEL | RF_ACTSOLVE | RF_DEBUG | RF_NOSTORE);
  b->db.overflow = keep.overflow; b->db.cap_need = keep.cap_need; b->db.mprc = keep.mprc; b->db.bpl = keep.bpl;
  return r;
}
static int refresh_derived(rsim_batch* b, int field) { return (derived_field(field) && b->derived_stale) ? refresh_forward(b) : 0; }
extern "C" int rsim_forward(rsim_batch* b) { return launch(b, nullptr, 1, RF_POSVEL | RF_ACTSOLVE | RF_DEBUG); }
extern "C" int rsim_step1(rsim_batch* b) { return launch(b, nullptr, 1, RF_POSVEL | RF_DEBUG); }
extern "C" int rsim_step2(rsim_batch* b) { return launch(b, nullptr, 1, RF_POSVEL | RF_ACTSOLVE | RF_INTEGRATE | RF_DEBUG); }
extern "C" int rsim_step(rsim_batch* b) { return launch(b, nullptr, 1, RF_POSVEL | RF_ACTSOLVE | RF_INTEGRATE | RF_DEBUG); }
extern "C" int rsim_control_step(rsim_batch* b, const float* actions_dev, int n_sub) {
  if (n_sub < 1) return fail("rsim_control_step: n_sub < 1");
  if (!actions_dev) return fail("rsim_control_step: actions_dev is NULL");
  return launch(b, actions_dev, n_sub, RF_POSVEL | RF_CTRL | RF_SETGOAL | RF_ACTSOLVE | RF_INTEGRATE | (b->m->has_task ? RF_OBS : 0) | RF_EPISODE);
}
// CompositeController.run_controller() between step1 and step2 (composite_controller.py:109-116, fixed_base_robot.py:143-153): the position /
// velocity stage, then ONE evaluation of the in-kernel part controllers from the controller state as it stands -- no set_goal, no integration.
// Writes RSIM_CTRL (clipped) and the torque slots of RSIM_CSTATE: the door the parity tests pin the in-kernel control laws through.
extern "C" int rsim_run_controller(rsim_batch* b) { return launch(b, nullptr, 1, RF_POSVEL | RF_CTRL | RF_DEBUG); }
// The LAST mj_step2 of a control step whose controllers run on the host side of the boundary (robosuite_amd/controllers.py): actuation, solve and
// integration of the substep, then everything MujocoEnv.step does after its loop -- observation record, reward, success, timestep / horizon, on-device
// episode restart (base.py:501-548) -- exactly as rsim_control_step does after its last substep.  No in-kernel controller runs, so the flags that mark
// freshly restarted envs for it are cleared here; the caller learns about restarts from RSIM_DONE.
extern "C" int rsim_step2_last(rsim_batch* b) {
  if (!b->m->has_task) return fail("rsim_step2_last: no task configured");
  if (launch(b, nullptr, 1, RF_POSVEL | RF_ACTSOLVE | RF_INTEGRATE | RF_OBS | RF_EPISODE | RF_DEBUG)) return 1;
  HIPCHK(hipMemsetAsync(b->db.needs_reset, 0, (size_t)b->B * sizeof(int), b->stream));
  return 0;
}
extern "C" int rsim_observe(rsim_batch* b) {
  if (!b->m->has_task) return fail("rsim_observe: no task configured");
  return launch(b, nullptr, 1, RF_POSVEL | RF_ACTSOLVE | RF_OBS | RF_DEBUG);
}
extern "C" int rsim_set_episode(rsim_batch* b, int horizon) {
  if (horizon < 0) return fail("rsim_set_episode: horizon < 0");
  b->db.horizon = horizon;
  return 0;
}
extern "C" int rsim_param_offset(const rsim_batch* b, const char* field, int elem) {
  return param_offset_impl(b->m, field, elem);
}
extern "C" int rsim_set_reset_bank(rsim_batch* b, int n_episodes, int n_patch, const int32_t* patch_idx, const float* bank) { if (join_groups(b)) return 1;
  rsim_model* m = b->m;
  if (n_episodes < 1 || n_patch < 0) return fail("rsim_set_reset_bank: bad sizes");
  for (int p2 = 0; p2 < n_patch; p2++) {
    if (patch_idx[p2] == RSIM_PATCH_TASK_OBJECT) {
      if (!(m->has_task && m->task.task == 4 && m->task.single_object_mode == 1)) return fail("rsim_set_reset_bank: RSIM_PATCH_TASK_OBJECT needs the PickPlace task in single-object mode 1");
      continue;
    }
    if (!b->per_env) return fail("rsim_set_reset_bank: per-episode model patches need per_env_params");
    if (patch_idx[p2] < 0 || patch_idx[p2] >= (int)m->ftab.size()) return fail("rsim_set_reset_bank: patch offset out of range");
  }
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  if (rsim_bank_flush(b)) return 1;
  b->bank_poll_pending = 0;
  if (b->d_bank) { hipFree(b->d_bank); b->d_bank = nullptr; }
  if (b->d_bank_tag) { hipFree(b->d_bank_tag); b->d_bank_tag = nullptr; }
  if (b->d_patch) { hipFree(b->d_patch); b->d_patch = nullptr; }
  const size_t stride = (size_t)m->nq + n_patch, total = (size_t)b->B * n_episodes * stride;
  if (dalloc(&b->d_bank, total)) return 1;
  HIPCHK(hipMemcpy(b->d_bank, bank, total * sizeof(float), hipMemcpyHostToDevice));
  if (dalloc(&b->d_patch, (size_t)(n_patch ? n_patch : 1))) return 1;
  if (n_patch) HIPCHK(hipMemcpy(b->d_patch, patch_idx, n_patch * sizeof(int), hipMemcpyHostToDevice));
  {   // slot s of every env starts out holding episode s
    std::vector<int> tags((size_t)b->B * n_episodes);
    for (size_t i = 0; i < tags.size(); i++) tags[i] = (int)(i % (size_t)n_episodes);
    if (dalloc(&b->d_bank_tag, tags.size())) return 1;
    HIPCHK(hipMemcpy(b->d_bank_tag, tags.data(), tags.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  b->db.bank = b->d_bank; b->db.bank_tag = b->d_bank_tag; b->db.patch_idx = b->d_patch; b->db.bank_E = n_episodes; b->db.bank_P = n_patch;
  for (int p2 = 0; p2 < n_patch; p2++)
    for (int f = 0; f < FO_COUNT; f++)
      if (patch_idx[p2] >= 0 && patch_idx[p2] >= m->fo[f] && patch_idx[p2] < m->fo[f] + m->fcount[f]) b->dm.fenv |= 1ull << f;
  b->cm_dirty = 1;
  return 0;
}

extern "C" int rsim_refill_reset_bank(rsim_batch* b, int n, const int32_t* env, const int32_t* episode, const float* rows) { if (join_groups(b)) return 1;
  if (!b->d_bank) return fail("rsim_refill_reset_bank: no reset bank installed (rsim_set_reset_bank)");
  if (n < 0) return fail("rsim_refill_reset_bank: n < 0");
  if (n == 0) return 0;
  for (int i = 0; i < n; i++) if (env[i] < 0 || env[i] >= b->B || episode[i] < 0) return fail("rsim_refill_reset_bank: entry %d out of range", i);
  HIPCHK(hipSetDevice(b->device));
  const int W = b->m->nq + b->db.bank_P;
  // staging buffers live until the scatter kernel has run: allocate per call, free after a stream sync (refills are rare: once per episode and env)
  int *d_env = nullptr, *d_ep = nullptr; float* d_rows = nullptr;
  if (dalloc(&d_env, (size_t)n) || dalloc(&d_ep, (size_t)n) || dalloc(&d_rows, (size_t)n * W)) return 1;
  HIPCHK(hipMemcpyAsync(d_env, env, (size_t)n * sizeof(int), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(d_ep, episode, (size_t)n * sizeof(int), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(d_rows, rows, (size_t)n * W * sizeof(float), hipMemcpyHostToDevice, b->stream));
  int e = rsim_launch_bank_scatter(b->d_bank, b->d_bank_tag, d_env, d_ep, d_rows, n, b->db.bank_E, W, b->stream);
  HIPCHK(hipStreamSynchronize(b->stream));
  hipFree(d_env); hipFree(d_ep); hipFree(d_rows);
  if (e) return fail("bank scatter kernel launch failed: %s", hipGetErrorString((hipError_t)e));
  return 0;
}

// Asynchronous upkeep of the reset ring.  Nothing here waits for, or is waited for by, the control steps: the copies and the scatter kernel run
// on a side stream of their own, so a refill costs the stepping thread nothing and may be driven from a second host thread.  Why that is safe:
//  * RSIM_EP_INDEX only ever grows, so a copy taken while control steps are in flight is at worst a little old, and an old counter only makes
//    the host refill a slot later than it could have;
//  * a slot is overwritten only once its env has moved past the episode it held (episode <= the polled counter), so no control step reads it
//    any more, and the slot's tag is written after its row (device-scope fence in k_bank_scatter): a reset that sees the new tag sees the new row.
static int bank_side_stream(rsim_batch* b) {
  if (b->bstream) return 0;
  HIPCHK(hipStreamCreateWithFlags(&b->bstream, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&b->bev, hipEventDisableTiming));
  HIPCHK(hipHostMalloc((void**)&b->h_epidx, (size_t)b->B * sizeof(int), hipHostMallocDefault));
  return 0;
}
extern "C" int rsim_bank_poll_begin(rsim_batch* b) {
  if (!b->d_bank) return fail("rsim_bank_poll_begin: no reset bank installed (rsim_set_reset_bank)");
  HIPCHK(hipSetDevice(b->device));
  if (bank_side_stream(b)) return 1;
  if (b->bank_poll_pending) return 0;   // one poll in flight at a time
  HIPCHK(hipMemcpyAsync(b->h_epidx, b->db.ep_index, (size_t)b->B * sizeof(int), hipMemcpyDeviceToHost, b->bstream));
  HIPCHK(hipEventRecord(b->bev, b->bstream));
  b->bank_poll_pending = 1;
  return 0;
}
// returns 1 and fills ep_index[B] when the poll has completed, 0 while it is still in flight (wait != 0: block until it has), -1 on error
extern "C" int rsim_bank_poll(rsim_batch* b, int32_t* ep_index, int wait) {
  if (!b->bank_poll_pending) { fail("rsim_bank_poll: no poll in flight (rsim_bank_poll_begin)"); return -1; }
  if (hipSetDevice(b->device) != hipSuccess) { fail("rsim_bank_poll: hipSetDevice failed"); return -1; }
  if (wait) { if (hipEventSynchronize(b->bev) != hipSuccess) { fail("rsim_bank_poll: event wait failed"); return -1; } }
  else {
    hipError_t q = hipEventQuery(b->bev);
    if (q == hipErrorNotReady) return 0;
    if (q != hipSuccess) { fail("rsim_bank_poll: %s", hipGetErrorString(q)); return -1; }
  }
  memcpy(ep_index, b->h_epidx, (size_t)b->B * sizeof(int));
  b->bank_poll_pending = 0;
  return 1;
}
extern "C" int rsim_refill_reset_bank_async(rsim_batch* b, int n, const int32_t* env, const int32_t* episode, const float* rows) {
  if (!b->d_bank) return fail("rsim_refill_reset_bank_async: no reset bank installed (rsim_set_reset_bank)");
  if (n < 0) return fail("rsim_refill_reset_bank_async: n < 0");
  if (n == 0) return 0;
  for (int i = 0; i < n; i++) if (env[i] < 0 || env[i] >= b->B || episode[i] < 0) return fail("rsim_refill_reset_bank_async: entry %d out of range", i);
  HIPCHK(hipSetDevice(b->device));
  if (bank_side_stream(b)) return 1;
  const int W = b->m->nq + b->db.bank_P;
  const size_t bytes = (size_t)n * (2 * sizeof(int) + (size_t)W * sizeof(float));
  rsim_batch::Stage& st = b->bstage[b->bstage_next];
  b->bstage_next = (b->bstage_next + 1) % 4;
  if (st.busy) { HIPCHK(hipEventSynchronize(st.done)); st.busy = 0; }   // only when four refills are in flight at once
  if (st.bytes < bytes) {
    if (st.host) hipHostFree(st.host);
    if (st.dev) hipFree(st.dev);
    st.bytes = bytes + bytes / 2;
    HIPCHK(hipHostMalloc(&st.host, st.bytes, hipHostMallocDefault));
    HIPCHK(hipMalloc(&st.dev, st.bytes));
  }
  if (!st.done) HIPCHK(hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
  char* h = (char*)st.host;
  memcpy(h, env, (size_t)n * sizeof(int));
  memcpy(h + (size_t)n * sizeof(int), episode, (size_t)n * sizeof(int));
  memcpy(h + (size_t)n * 2 * sizeof(int), rows, (size_t)n * W * sizeof(float));
  HIPCHK(hipMemcpyAsync(st.dev, st.host, bytes, hipMemcpyHostToDevice, b->bstream));
  const int* d_env = (const int*)st.dev; const int* d_ep = d_env + n; const float* d_rows = (const float*)(d_ep + n);
  int e = rsim_launch_bank_scatter(b->d_bank, b->d_bank_tag, d_env, d_ep, d_rows, n, b->db.bank_E, W, b->bstream);
  if (e) return fail("bank scatter kernel launch failed: %s", hipGetErrorString((hipError_t)e));
  HIPCHK(hipEventRecord(st.done, b->bstream));
  st.busy = 1;
  return 0;
}
// all asynchronous refills issued so far have landed in the ring
extern "C" int rsim_bank_flush(rsim_batch* b) {
  if (!b->bstream) return 0;
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->bstream));
  for (auto& st : b->bstage) st.busy = 0;
  return 0;
}

extern "C" int rsim_dr_save_defaults(rsim_batch* b) { if (join_groups(b)) return 1;
  if (!b->per_env) return fail("rsim_dr_save_defaults: the batch was created without per_env_params");
  HIPCHK(hipSetDevice(b->device));
  const size_t n = b->m->ftab.size() * (size_t)b->B;
  if (!b->d_ft_base) HIPCHK(hipMalloc((void**)&b->d_ft_base, n * sizeof(float)));
  HIPCHK(hipMemcpyAsync(b->d_ft_base, b->d_ft, n * sizeof(float), hipMemcpyDeviceToDevice, b->stream));
  b->db.ft_base = b->d_ft_base;
  return 0;
}
extern "C" int rsim_randomize_dynamics(rsim_batch* b, const rsim_dr_desc* d, uint64_t seed, uint64_t step) {
  if (!b->per_env) return fail("rsim_randomize_dynamics: the batch was created without per_env_params");
  if (!b->d_ft_base) return fail("rsim_randomize_dynamics: call rsim_dr_save_defaults first");
  HIPCHK(hipSetDevice(b->device));
  DDr dd = {d->density_ratio, d->viscosity_ratio, d->position_size, d->quaternion_size, d->inertia_ratio, d->mass_ratio, d->friction_ratio, d->solref_ratio,
            d->solimp_ratio, d->frictionloss_size, d->damping_size, d->armature_size,
            d->body_mask ? d->body_mask : ~0ull, d->geom_mask ? d->geom_mask : ~0ull, d->joint_mask ? d->joint_mask : ~0ull};
  // stiffness_ratio: DynamicsModder.mod_stiffness leaves joints without a spring alone (mjmod.py:1907-1909), and models with joint springs are
  // refused by rsim_batch_create (no passive joint spring in the kernel): the draw has nothing to act on
  const unsigned long long fenv_before = b->dm.fenv;
  b->dm.fenv |= (1ull << FO_opt) | (1ull << FO_body_pos) | (1ull << FO_body_quat) | (1ull << FO_body_inertia) | (1ull << FO_body_mass) | (1ull << FO_cg_friction) |
                (1ull << FO_cg_solref) | (1ull << FO_cg_solimp) | (1ull << FO_dof_frictionloss) | (1ull << FO_dof_damping) | (1ull << FO_dof_armature);
  const unsigned long long want = (1ull << FO_opt) | (1ull << FO_body_pos) | (1ull << FO_body_quat) | (1ull << FO_body_inertia) | (1ull << FO_body_mass) | (1ull << FO_cg_friction) |
                                  (1ull << FO_cg_solref) | (1ull << FO_cg_solimp) | (1ull << FO_dof_frictionloss) | (1ull << FO_dof_damping) | (1ull << FO_dof_armature);
  if (b->ngroups > 1 && !b->cm_dirty && b->db.cm_stride && (fenv_before & want) == want && !memcmp(&b->cm_ctrl, &b->dm.ctrl, sizeof(DCtrl))) {
    // stream groups, per-env constant blocks already in place: every env block re-draws its tables and rebuilds its own constant blocks on its
    // own stream, ordered before its next control step -- the groups stay decoupled through a randomise-every-step rollout
    if (fork_groups(b)) return 1;
    for (int g = 0; g < b->ngroups; g++) {
      const int e0 = (int)((long long)b->B * g / b->ngroups), e1 = (int)((long long)b->B * (g + 1) / b->ngroups);
      DBatch db = b->db;
      db.env0 = e0; db.nenv = e1 - e0;
      int e = rsim_launch_randomize(&b->dm, &db, &dd, seed, step, b->gstream[g]);
      if (!e) e = k_prepare_launch[b->cfg](&b->dm, &db, e1 - e0, 0, b->gstream[g]);
      if (e) return fail("kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    }
    b->gen++;
    return 0;
  }
  if (join_groups(b)) return 1;
  int e = rsim_launch_randomize(&b->dm, &b->db, &dd, seed, step, b->stream);
  if (e) return fail("kernel launch failed: %s", hipGetErrorString((hipError_t)e));
  b->cm_dirty = 1;   // the next launch rebuilds the constant blocks from the re-drawn tables (same stream: ordered after this kernel)
  b->gen++;
  return 0;
}

extern "C" int rsim_ctrl_reset(rsim_batch* b, const uint8_t* mask) { if (join_groups(b)) return 1;
  HIPCHK(hipSetDevice(b->device));
  if (!b->m->ctrl.enabled) return fail("no controller configured");
  if (sync_controller(b)) return 1;
  if (ensure_constants(b)) return 1;
  const unsigned char* dmask = nullptr;
  if (mask) {
    HIPCHK(hipMemcpyAsync(b->d_mask, mask, (size_t)b->B, hipMemcpyHostToDevice, b->stream));
    dmask = b->d_mask;
  }
  int e = k_creset_launch[b->cfg](&b->dm, &b->db, dmask, b->stream);
  if (e) return fail("kernel launch failed: %s", hipGetErrorString((hipError_t)e));
  if (mask) HIPCHK(hipStreamSynchronize(b->stream));
  b->gen++;
  return 0;
}

extern "C" int rsim_profile(rsim_batch* b, int enable, unsigned long long* out, int n_out) { if (join_groups(b)) return 1;
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  if (out && b->db.prof) {
    unsigned long long tmp[RP_COUNT];
    HIPCHK(hipMemcpy(tmp, b->db.prof, sizeof(tmp), hipMemcpyDeviceToHost));
    for (int i = 0; i < n_out && i < RP_COUNT; i++) out[i] = tmp[i];
  }
  const size_t nprof = RP_COUNT + 8 * (size_t)b->B + 2 * RSIM_PAIR_MAX;  // + per candidate pair {narrow-phase visits, support calls}; before that: phase accumulators, then per-env {hw_id, xcc_id, t_start, t_end, n_mpr, n_support, n_newton, n_cand} of the last launch
  if (enable && !b->db.prof) {
    HIPCHK(hipMalloc((void**)&b->db.prof, nprof * sizeof(unsigned long long)));
  }
  if (enable) HIPCHK(hipMemset(b->db.prof, 0, nprof * sizeof(unsigned long long)));
  if (!enable && b->db.prof) { hipFree(b->db.prof); b->db.prof = nullptr; }
  return 0;
}

extern "C" int rsim_set_schedule(rsim_batch* b, int longest_first) {
  b->schedule = longest_first ? 1 : 0; b->have_cost = 0;
  if (b->ostream) { HIPCHK(hipSetDevice(b->device)); HIPCHK(hipStreamSynchronize(b->ostream)); }
  b->nstep = 0; b->ord_valid[0] = b->ord_valid[1] = 0;
  return 0;
}
extern "C" int rsim_pairlog(rsim_batch* b, unsigned long long* out) { if (join_groups(b)) return 1;
  if (!b->db.prof) return fail("rsim_pairlog: profiling is not armed (rsim_profile(b, 1, ...))");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipMemcpy(out, b->db.prof + RP_COUNT + 8 * (size_t)b->B, 2 * RSIM_PAIR_MAX * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return 0;
}
extern "C" int rsim_profile_env(rsim_batch* b, int env) { b->db.prof_env = env; return 0; }
// Capacity tier of every env for the NEXT control step (0: the native configuration steps it, 1: the wider one), host int32 [B]; all zeros for a batch without a
// tier above its configuration.  Diagnostics (tools/window_trace.py, bench.py's per-step record): an env whose entry went 0 -> 1 over a control step was handed
// over in mid-step (redone), an env at 1 is on next step's wide list.
// Everything outside a kernel's code object that decides what a control step dispatches and how its solver behaves: the compiled-in defaults and the RSIM_*
// environment overrides in force.  PMC evidence under profiles/ carries the sha of this string next to the code-object sha (round-5 advisor finding: a host-only
// change -- polish passes, tier thresholds -- left the code sha equal and the evidence was reported as current).
extern "C" const char* rsim_tuning_defaults(void) {
  static std::string s;
  char buf[1024];
  snprintf(buf, sizeof(buf), "newton_ns=%g;newton_na=%g;newton_ls=%g;newton_ng=%g;newton_wide=1;newton_exact=1;newton_refine(cfg>=3)=16;polish_tol=1;polish_gate=0;"
           "bp_reach=%g;near_thresh=%g;near_gain=%g;mpr_cone=%g;mpr_warmstart=1;mpr_portal=1;tier_up(cfg0,cfg1)=0/0;tier_up(other)=2/6;tier_mode=0;order_fresh=1;fused_tier_cfg0=%d;fused_tier_cfg1=%d;fused_tier_cfg2=%d",
           (double)RSIM_NEWTON_NS, (double)RSIM_NEWTON_NA, (double)RSIM_NEWTON_LS, (double)RSIM_NEWTON_NG, (double)RSIM_BP_REACH, (double)RSIM_NEAR_THRESH, (double)RSIM_NEAR_GAIN, (double)RSIM_MPR_CONE,
           []{ int lw[10]; return rsim_limits_w_cfg0(lw); }(), []{ int lw[10]; return rsim_limits_w_cfg1(lw); }(), []{ int lw[10]; return rsim_limits_w_cfg2(lw); }());
  s = buf;
  static const char* const envs[] = {"RSIM_NEWTON_NS", "RSIM_NEWTON_NA", "RSIM_NEWTON_LS", "RSIM_NEWTON_NG", "RSIM_NEWTON_WIDE", "RSIM_NEWTON_EXACT", "RSIM_NEWTON_REFINE",
                                     "RSIM_POLISH_TOL", "RSIM_POLISH_GATE", "RSIM_BP_REACH", "RSIM_NEAR_THRESH", "RSIM_NEAR_GAIN", "RSIM_MPR_CONE", "RSIM_NO_MPR_WARMSTART", "RSIM_NO_MPR_PORTAL_WARMSTART",
                                     "RSIM_TIER_UP_CON", "RSIM_TIER_UP_EFC", "RSIM_TIER_MODE", "RSIM_NO_TIERS", "RSIM_ORDER_FRESH", "RSIM_EULER_FULL", "RSIM_FORCE_HANDOVER"};
  for (const char* e : envs) if (const char* v = getenv(e)) { s += ";env:"; s += e; s += "="; s += v; }
  return s.c_str();
}
// {env-steps the wider capacity tier stepped, env-steps of these that were handed over (fused tier) / redone (tier kernels) in mid-step} since the batch was created
extern "C" int rsim_tier_stats(rsim_batch* b, unsigned long long* out2) {
  if (!out2) return fail("rsim_tier_stats: null destination");
  out2[0] = out2[1] = 0;
  if (b->cfg_w < 0 || !b->db.tstat) return 0;
  HIPCHK(hipSetDevice(b->device));
  if (join_groups(b)) return 1;
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipMemcpy(out2, b->db.tstat, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return 0;
}
extern "C" int rsim_tier_snapshot(rsim_batch* b, int* host_tier) {
  if (!host_tier) return fail("rsim_tier_snapshot: null destination");
  HIPCHK(hipSetDevice(b->device));
  if (b->cfg_w < 0 || !b->d_tier[0]) { memset(host_tier, 0, (size_t)b->B * sizeof(int)); return 0; }
  if (join_groups(b)) return 1;
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipMemcpy(host_tier, b->d_tier[b->tier_flip], (size_t)b->B * sizeof(int), hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int rsim_wavelog(rsim_batch* b, unsigned long long* out) { if (join_groups(b)) return 1;
  if (!b->db.prof) return fail("rsim_wavelog: profiling is not enabled");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipMemcpy(out, b->db.prof + RP_COUNT, 8 * (size_t)b->B * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return 0;
}

extern "C" void* rsim_device_ptr(rsim_batch* b, int field, size_t* count) {
  if (field < 0 || field >= RSIM_FIELD_COUNT) { fail("bad field id %d", field); return nullptr; }
  if (count) *count = b->fcount[field];
  if (refresh_derived(b, field)) return nullptr;
  return b->fptr[field];
}
extern "C" int rsim_get_array(rsim_batch* b, int field, void* dst, size_t count) { if (join_groups(b)) return 1;
  if (field < 0 || field >= RSIM_FIELD_COUNT) return fail("bad field id %d", field);
  if (count > b->fcount[field]) return fail("rsim_get_array: count %zu > field size %zu", count, b->fcount[field]);
  if (refresh_derived(b, field)) return 1;
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipMemcpy(dst, b->fptr[field], count * 4, hipMemcpyDeviceToHost));
  return 0;
}
extern "C" int rsim_set_array(rsim_batch* b, int field, const void* src, size_t count) { if (join_groups(b)) return 1;
  if (field < 0 || field >= RSIM_FIELD_COUNT) return fail("bad field id %d", field);
  if (count > b->fcount[field]) return fail("rsim_set_array: count %zu > field size %zu", count, b->fcount[field]);
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipMemcpy(b->fptr[field], src, count * 4, hipMemcpyHostToDevice));
  // positions written by the host: the narrow phase's warm-start directions belong to the states before (a replay from this state must not depend on them)
  if (field == RSIM_QPOS && b->db.mprc) HIPCHK(hipMemset(b->db.mprc, 0, (size_t)b->B * b->m->npair * 12 * sizeof(float)));
  b->gen++;
  return 0;
}

// mj_fullM (controllers/parts/controller.py:226-227: `mujoco.mj_fullM(model, mass_matrix, data.qM)`): the dense joint-space inertia of one env
extern "C" int rsim_full_M(rsim_batch* b, int env, double* M) {
  if (join_groups(b)) return 1;
  if (env < 0 || env >= b->B || !M) return fail("rsim_full_M: bad env %d / null output", env);
  if (refresh_derived(b, RSIM_QM)) return 1;
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  const int nv = b->m->nv;
  std::vector<float> h((size_t)nv * nv);
  HIPCHK(hipMemcpy(h.data(), b->db.qM + (size_t)env * nv * nv, h.size() * 4, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < h.size(); i++) M[i] = h[i];
  return 0;
}
// sim.data.contact[:ncon] of one env (binding_utils.py:1008-1035; utils/sim_utils.py:8-40 check_contact, manipulation_env.py:331-376): returns the number of
// contacts written (<= max_out), -1 on error
extern "C" int rsim_contacts(rsim_batch* b, int env, int max_out, rsim_contact* out) {
  if (join_groups(b)) return -1;
  if (env < 0 || env >= b->B || (max_out > 0 && !out)) { fail("rsim_contacts: bad env %d / null output", env); return -1; }
  if (refresh_derived(b, RSIM_CONTACT)) return -1;
  if (hipSetDevice(b->device) != hipSuccess || hipStreamSynchronize(b->stream) != hipSuccess) { fail("rsim_contacts: device error"); return -1; }
  int n = 0;
  if (hipMemcpy(&n, b->db.ncon + env, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { fail("rsim_contacts: copy failed"); return -1; }
  if (n > max_out) n = max_out;
  std::vector<float> h((size_t)(n > 0 ? n : 1) * RSIM_CON_REC);
  if (n > 0 && hipMemcpy(h.data(), b->db.contact + (size_t)env * b->lim[5] * RSIM_CON_REC, (size_t)n * RSIM_CON_REC * 4, hipMemcpyDeviceToHost) != hipSuccess) { fail("rsim_contacts: copy failed"); return -1; }
  for (int c = 0; c < n; c++) {
    const float* r = h.data() + (size_t)c * RSIM_CON_REC;
    rsim_contact& o = out[c];
    o.dist = r[0];
    for (int k = 0; k < 3; k++) o.pos[k] = r[1 + k];
    for (int k = 0; k < 9; k++) o.frame[k] = r[4 + k];
    o.geom1 = (int)r[13]; o.geom2 = (int)r[14]; o.dim = (int)r[15]; o.efc_address = (int)r[16];
    o.normal_force = r[17];
    for (int k = 0; k < 5; k++) o.friction[k] = r[18 + k];
  }
  return n;
}

static int refresh_cache(rsim_batch* b, int env) {
  if (b->derived_stale && refresh_forward(b)) return 1;   // Jacobians after a fused control step: of the current state
  if (b->cache_gen == b->gen && b->cache_env == env) return 0;
  rsim_model* m = b->m;
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  b->h_cdof.resize(m->nv * 6); b->h_rootcom.resize(m->nbody * 3); b->h_xpos.resize(m->nbody * 3); b->h_xquat.resize(m->nbody * 4);
  HIPCHK(hipMemcpy(b->h_cdof.data(), b->db.cdof + (size_t)env * m->nv * 6, m->nv * 6 * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(b->h_rootcom.data(), b->db.rootcom + (size_t)env * m->nbody * 3, m->nbody * 3 * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(b->h_xpos.data(), b->db.xpos + (size_t)env * m->nbody * 3, m->nbody * 3 * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(b->h_xquat.data(), b->db.xquat + (size_t)env * m->nbody * 4, m->nbody * 4 * 4, hipMemcpyDeviceToHost));
  b->cache_gen = b->gen; b->cache_env = env;
  return 0;
}
static void body_point(rsim_batch* b, int body, const double* local, double* out) {
  const float* q = &b->h_xquat[4 * body];
  double w = q[0], x = q[1], y = q[2], z = q[3];
  double R[9] = {w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x),
                 2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z};
  for (int i = 0; i < 3; i++) out[i] = b->h_xpos[3 * body + i] + R[3 * i] * local[0] + R[3 * i + 1] * local[1] + R[3 * i + 2] * local[2];
}
static int jac_common(rsim_batch* b, int env, int body, const double* local, double* jacp, double* jacr) {
  rsim_model* m = b->m;
  if (env < 0 || env >= b->B) return fail("env out of range");
  if (refresh_cache(b, env)) return 1;
  const int nv = m->nv;
  double p[3];
  body_point(b, body, local, p);
  if (jacp) memset(jacp, 0, sizeof(double) * 3 * nv);
  if (jacr) memset(jacr, 0, sizeof(double) * 3 * nv);
  int root = m->I("body_rootid")[body];
  double off[3];
  for (int k = 0; k < 3; k++) off[k] = p[k] - b->h_rootcom[3 * root + k];
  u64 mk = m->body_dofmask[body];
  for (int i = 0; i < nv; i++) {
    if (!((mk >> i) & 1ull)) continue;
    const float* cd = &b->h_cdof[6 * i];
    if (jacr) for (int k = 0; k < 3; k++) jacr[k * nv + i] = cd[k];
    if (jacp) {
      double t[3] = {cd[1] * off[2] - cd[2] * off[1], cd[2] * off[0] - cd[0] * off[2], cd[0] * off[1] - cd[1] * off[0]};
      for (int k = 0; k < 3; k++) jacp[k * nv + i] = cd[3 + k] + t[k];
    }
  }
  return 0;
}
extern "C" int rsim_jac_site(rsim_batch* b, int env, int site, double* jacp, double* jacr) {
  rsim_model* m = b->m;
  if (site < 0 || site >= m->nsite) return fail("site out of range");
  return jac_common(b, env, m->I("site_bodyid")[site], m->D("site_pos") + 3 * site, jacp, jacr);
}
extern "C" int rsim_jac_body(rsim_batch* b, int env, int body, double* jacp, double* jacr) {
  if (body < 0 || body >= b->m->nbody) return fail("body out of range");
  double z[3] = {0, 0, 0};
  return jac_common(b, env, body, z, jacp, jacr);
}

struct ParamMap { const char* name; int fo; int w; int geom; };
static const ParamMap* param_maps(int* n);
static int param_offset_impl(const rsim_model* m, const char* field, int elem) {
  int n = 0;
  const ParamMap* maps = param_maps(&n);
  for (int i = 0; i < n; i++)
    if (!strcmp(maps[i].name, field)) {
      if (maps[i].geom) {  // elem indexes the full geom array: (geom id, component)
        int g = elem / maps[i].w, q = elem % maps[i].w;
        if (g < 0 || g >= m->ngeom || m->geom2cg[g] < 0) return -1;
        return m->fo[maps[i].fo] + m->geom2cg[g] * maps[i].w + q;
      }
      if (elem < 0 || elem >= m->fcount[maps[i].fo]) return -1;
      return m->fo[maps[i].fo] + elem;
    }
  return -1;
}
static const ParamMap* param_maps(int* n) {
  static const ParamMap maps[] = {
      {"body_pos", FO_body_pos, 3, 0}, {"body_quat", FO_body_quat, 4, 0}, {"body_ipos", FO_body_ipos, 3, 0}, {"body_iquat", FO_body_iquat, 4, 0},
      {"body_mass", FO_body_mass, 1, 0}, {"body_inertia", FO_body_inertia, 3, 0}, {"body_invweight0", FO_body_invweight0, 2, 0},
      {"body_subtreemass", FO_body_subtreemass, 1, 0}, {"jnt_pos", FO_jnt_pos, 3, 0}, {"jnt_axis", FO_jnt_axis, 3, 0}, {"jnt_range", FO_jnt_range, 2, 0},
      {"jnt_margin", FO_jnt_margin, 1, 0}, {"jnt_solref", FO_jnt_solref, 2, 0}, {"jnt_solimp", FO_jnt_solimp, 5, 0}, {"qpos0", FO_qpos0, 1, 0},
      {"dof_armature", FO_dof_armature, 1, 0}, {"dof_damping", FO_dof_damping, 1, 0}, {"dof_frictionloss", FO_dof_frictionloss, 1, 0},
      {"dof_solref", FO_dof_solref, 2, 0}, {"dof_solimp", FO_dof_solimp, 5, 0}, {"dof_invweight0", FO_dof_invweight0, 1, 0},
      {"geom_size", FO_cg_size, 3, 1}, {"geom_pos", FO_cg_pos, 3, 1}, {"geom_quat", FO_cg_quat, 4, 1}, {"geom_friction", FO_cg_friction, 3, 1},
      {"geom_solref", FO_cg_solref, 2, 1}, {"geom_solimp", FO_cg_solimp, 5, 1}, {"geom_solmix", FO_cg_solmix, 1, 1}, {"geom_margin", FO_cg_margin, 1, 1},
      {"geom_gap", FO_cg_gap, 1, 1}, {"geom_rbound", FO_cg_rbound, 1, 1}, {"site_pos", FO_site_pos, 3, 0}, {"site_quat", FO_site_quat, 4, 0},
      {"actuator_gear", FO_act_gear, 1, 0}, {"actuator_gainprm", FO_act_gainprm, 3, 0}, {"actuator_biasprm", FO_act_biasprm, 3, 0},
      {"actuator_ctrlrange", FO_act_ctrlrange, 2, 0}, {"actuator_forcerange", FO_act_forcerange, 2, 0}, {"opt", FO_opt, 1, 0}};
  *n = (int)(sizeof(maps) / sizeof(maps[0]));
  return maps;
}

extern "C" int rsim_model_param_set(rsim_batch* b, const char* field, int env0, int nenv, const double* values, size_t cpe) { if (join_groups(b)) return 1;
  rsim_model* m = b->m;
  int nmaps = 0;
  const ParamMap* maps = param_maps(&nmaps);
  const ParamMap* mp = nullptr;
  for (int i = 0; i < nmaps; i++) if (!strcmp(maps[i].name, field)) mp = &maps[i];
  if (!mp) return fail("rsim_model_param_set: unknown or read-only field '%s'", field);
  if (env0 < 0 || nenv < 1 || env0 + nenv > b->B) return fail("rsim_model_param_set: env range out of bounds");
  if (!b->per_env && !(env0 == 0 && nenv == b->B)) return fail("rsim_model_param_set: batch was created without per-env params");
  size_t expect = mp->geom ? (size_t)m->ngeom * mp->w : (size_t)m->fcount[mp->fo];
  if (cpe != expect) return fail("rsim_model_param_set: field '%s' expects %zu values per env, got %zu", field, expect, cpe);
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  const int ncg = (int)m->cg.size();
  size_t n = m->fcount[mp->fo], fs = m->ftab.size();
  int envs = b->per_env ? nenv : 1;
  std::vector<float> tmp((size_t)envs * n);
  for (int e = 0; e < envs; e++) {
    const double* v = values + (size_t)e * cpe;
    float* t = tmp.data() + (size_t)e * n;
    if (mp->geom) {
      for (int c = 0; c < ncg; c++) for (int q = 0; q < mp->w; q++) t[(size_t)c * mp->w + q] = (float)v[(size_t)m->cg[c] * mp->w + q];
    } else for (size_t i = 0; i < n; i++) t[i] = (float)v[i];
  }
  // one strided copy: row e of `tmp` -> the field's slot inside env (env0 + e)'s float table
  const size_t base = b->per_env ? (size_t)env0 * fs : 0;
  HIPCHK(hipMemcpy2D(b->d_ft + base + m->fo[mp->fo], fs * sizeof(float), tmp.data(), n * sizeof(float), n * sizeof(float), (size_t)envs, hipMemcpyHostToDevice));
  if (b->per_env) b->dm.fenv |= 1ull << mp->fo;   // from now on the kernel reads this field from the env's own table
  else HIPCHK(hipMemcpy(b->d_ft0 + m->fo[mp->fo], tmp.data(), n * sizeof(float), hipMemcpyHostToDevice));
  b->cm_dirty = 1;
  b->gen++;
  return 0;
}

extern "C" int rsim_model_param_get(rsim_batch* b, const char* field, int env0, int nenv, double* values, size_t cpe) { if (join_groups(b)) return 1;
  rsim_model* m = b->m;
  int nmaps = 0;
  const ParamMap* maps = param_maps(&nmaps);
  const ParamMap* mp = nullptr;
  for (int i = 0; i < nmaps; i++) if (!strcmp(maps[i].name, field)) mp = &maps[i];
  if (!mp) return fail("rsim_model_param_get: unknown field '%s'", field);
  if (env0 < 0 || nenv < 1 || env0 + nenv > b->B) return fail("rsim_model_param_get: env range out of bounds");
  size_t expect = mp->geom ? (size_t)m->ngeom * mp->w : (size_t)m->fcount[mp->fo];
  if (cpe != expect) return fail("rsim_model_param_get: field '%s' has %zu values per env, got %zu", field, expect, cpe);
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  const size_t n = m->fcount[mp->fo], fs = m->ftab.size();
  std::vector<float> tmp((size_t)nenv * n);
  if (b->per_env && ((b->dm.fenv >> mp->fo) & 1ull)) HIPCHK(hipMemcpy2D(tmp.data(), n * sizeof(float), b->d_ft + (size_t)env0 * fs + m->fo[mp->fo], fs * sizeof(float), n * sizeof(float), (size_t)nenv, hipMemcpyDeviceToHost));
  else for (int e = 0; e < nenv; e++) HIPCHK(hipMemcpy(tmp.data() + (size_t)e * n, b->d_ft0 + m->fo[mp->fo], n * sizeof(float), hipMemcpyDeviceToHost));
  const int ncg = (int)m->cg.size();
  for (int e = 0; e < nenv; e++) {
    double* v = values + (size_t)e * cpe;
    const float* t = tmp.data() + (size_t)e * n;
    if (mp->geom) {
      const double* base = m->D(mp->name);   // non-colliding geoms are not on the device: report the compiled value
      for (size_t i = 0; i < cpe; i++) v[i] = base ? base[i] : 0.0;
      for (int c = 0; c < ncg; c++) for (int q = 0; q < mp->w; q++) v[(size_t)m->cg[c] * mp->w + q] = t[(size_t)c * mp->w + q];
    } else for (size_t i = 0; i < n; i++) v[i] = t[i];
  }
  return 0;
}



// ---- rollout statistics across GPUs: the one collective of the path (include/rsim.h) -----------------------------------------------------------
struct rsim_comm { ncclComm_t comm; int rank, world, device; hipStream_t stream; double* dbuf; int cap; };
static void* rccl_sym(const char* name) {
  static void* h = [] {
    void* x = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);        // PyTorch-ROCm's copy, if the process has it: one process, one RCCL
    if (!x) x = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (!x) x = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!x) x = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    return x;
  }();
  return h ? dlsym(h, name) : nullptr;
}
// (dlerror() clears the pending message when it is read: capture it once)
#define RCCL_FN(var, name, type) type var = (type)rccl_sym(name); if (!var) { const char* de_ = dlerror(); return fail("%s: librccl.so / %s not found in the process (%s)", __func__, name, de_ ? de_ : "no dlerror"); }
typedef ncclResult_t (*fn_uid)(ncclUniqueId*);
typedef ncclResult_t (*fn_init)(ncclComm_t*, int, ncclUniqueId, int);
typedef ncclResult_t (*fn_allreduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
typedef ncclResult_t (*fn_destroy)(ncclComm_t);
typedef const char* (*fn_errstr)(ncclResult_t);
static const char* rccl_err(ncclResult_t r) { fn_errstr f = (fn_errstr)rccl_sym("ncclGetErrorString"); return f ? f(r) : "?"; }

extern "C" int rsim_comm_unique_id(void* id_out, size_t bytes) {
  static_assert(sizeof(ncclUniqueId) == RSIM_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  if (!id_out || bytes < sizeof(ncclUniqueId)) return fail("rsim_comm_unique_id: need a %zu-byte buffer", sizeof(ncclUniqueId));
  RCCL_FN(f, "ncclGetUniqueId", fn_uid);
  ncclUniqueId id;
  ncclResult_t r = f(&id);
  if (r != ncclSuccess) return fail("ncclGetUniqueId: %s", rccl_err(r));
  memcpy(id_out, &id, sizeof id);
  return 0;
}

extern "C" int rsim_comm_create(const void* id, size_t bytes, int rank, int world, int device, rsim_comm** out) {
  if (!id || bytes < sizeof(ncclUniqueId) || !out || world < 1 || rank < 0 || rank >= world) return fail("rsim_comm_create: bad arguments (rank %d of %d, %zu id bytes)", rank, world, bytes);
  RCCL_FN(f, "ncclCommInitRank", fn_init);
  HIPCHK(hipSetDevice(device));
  rsim_comm* c = new rsim_comm{};
  c->rank = rank; c->world = world; c->device = device; c->cap = 64;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  ncclResult_t r = f(&c->comm, world, uid, rank);
  if (r != ncclSuccess) { delete c; return fail("ncclCommInitRank(rank %d of %d, device %d): %s", rank, world, device, rccl_err(r)); }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipMalloc(&c->dbuf, c->cap * sizeof(double)) != hipSuccess) {
    rsim_comm_free(c);
    return fail("rsim_comm_create: stream / buffer allocation failed on device %d", device);
  }
  *out = c;
  return 0;
}

extern "C" int rsim_allreduce_stats(rsim_comm* c, double* inout, int n, int op) {
  if (!c || !inout || n < 1 || n > c->cap || (op != 0 && op != 1)) return fail("rsim_allreduce_stats: bad arguments (n %d of at most %d, op %d)", n, c ? c->cap : 0, op);
  RCCL_FN(f, "ncclAllReduce", fn_allreduce);
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemcpyAsync(c->dbuf, inout, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  ncclResult_t r = f(c->dbuf, c->dbuf, (size_t)n, ncclFloat64, op == 0 ? ncclSum : ncclMax, c->comm, c->stream);
  if (r != ncclSuccess) return fail("ncclAllReduce: %s", rccl_err(r));
  HIPCHK(hipMemcpyAsync(inout, c->dbuf, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" void rsim_comm_free(rsim_comm* c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->comm) { fn_destroy f = (fn_destroy)rccl_sym("ncclCommDestroy"); if (f) f(c->comm); }
  if (c->dbuf) hipFree(c->dbuf);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}
