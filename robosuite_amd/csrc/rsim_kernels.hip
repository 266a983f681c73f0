// rsim_kernels.hip -- MI355X (gfx950) batched manipulation-sim kernels.
//
// One environment per workgroup, one 64-lane wavefront per workgroup: every per-env quantity lives in LDS for the
// whole control step, all cross-lane traffic is DPP / LDS-broadcast, and `__syncthreads()` with a 64-thread block
// lowers to a wave barrier (no s_barrier).  The 25 physics substeps of one robosuite `env.step()`
// (reference environments/base.py:494-504) and the controller evaluations between step1/step2 run inside ONE launch.
//
// Tree recursions (kinematics, CRBA, RNE) are re-formulated as lane-parallel sums over ancestor bit-masks instead of
// serial parent->child sweeps; the constraint solver is the primal Newton method MuJoCo uses by default, with one lane
// per constraint block.  Algorithm (not code) follows oracle/rsim_oracle.c, which cites the reference call sites.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsim_internal.h"

typedef unsigned long long u64;
#define SYNC() __syncthreads()
#define FMIN 1e-20f
#define PI_F 3.14159265358979f

enum { JNT_FREE = 0, JNT_BALL = 1, JNT_SLIDE = 2, JNT_HINGE = 3 };
enum { G_PLANE = 0, G_HFIELD, G_SPHERE, G_CAPSULE, G_ELLIPSOID, G_CYLINDER, G_BOX, G_MESH };
enum { C_FRICTION_DOF = 0, C_LIMIT_JOINT = 1, C_CONTACT_FRICTIONLESS = 2, C_CONTACT_ELLIPTIC = 3 };
enum { ST_SATISFIED = 0, ST_QUADRATIC = 1, ST_LINEARNEG = 2, ST_LINEARPOS = 3, ST_CONE = 4 };

// ------------------------------------------------------------------------------------------------------------
// wave-level primitives
// ------------------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
// full 64-lane sum, result uniform in all lanes (row_shr scan + row_bcast, read back from lane 63)
__device__ __forceinline__ float wave_sum(float x) {
#ifdef RSIM_NO_DPP
  x += __shfl_xor(x, 32); x += __shfl_xor(x, 16); x += __shfl_xor(x, 8); x += __shfl_xor(x, 4); x += __shfl_xor(x, 2); x += __shfl_xor(x, 1);
  return x;
#else
  x += dpp_f<0x111>(x);  // row_shr:1
  x += dpp_f<0x112>(x);  // row_shr:2
  x += dpp_f<0x114>(x);  // row_shr:4
  x += dpp_f<0x118>(x);  // row_shr:8
  x += dpp_f<0x142>(x);  // row_bcast:15
  x += dpp_f<0x143>(x);  // row_bcast:31
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
#endif
}
__device__ __forceinline__ float bcast(float x, int srclane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), srclane));
}
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ u64 lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

// optional per-phase cycle accounting (DBatch.prof != null): lane 0 accumulates s_memtime deltas and event counters
struct Prof {
  unsigned long long* p;
  long long t0;
  int lane;
  __device__ __forceinline__ void start() { if (p) t0 = clock64(); }
  __device__ __forceinline__ void mark(int id) {
    if (p) { long long t1 = clock64(); if (lane == 0) atomicAdd(p + id, (unsigned long long)(t1 - t0)); t0 = clock64(); }
  }
  __device__ __forceinline__ void count(int id, int v) { if (p && lane == 0) atomicAdd(p + id, (unsigned long long)v); }
};

// ------------------------------------------------------------------------------------------------------------
// small math (float)
// ------------------------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
__device__ __forceinline__ V3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }
__device__ __forceinline__ void st3(float* p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ float norm(V3 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ V3 normalized(V3 a, float* len = nullptr) {
  float n = norm(a);
  if (len) *len = n;
  if (n < FMIN) return v3(1, 0, 0);
  return a * (1.0f / n);
}
struct Q4 { float w, x, y, z; };
__device__ __forceinline__ Q4 ldq(const float* p) { Q4 q = {p[0], p[1], p[2], p[3]}; return q; }
__device__ __forceinline__ void stq(float* p, Q4 q) { p[0] = q.w; p[1] = q.x; p[2] = q.y; p[3] = q.z; }
__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
  Q4 r = {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
  return r;
}
__device__ __forceinline__ Q4 qnorm(Q4 q) {
  float n = sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  if (n < FMIN) { Q4 r = {1, 0, 0, 0}; return r; }
  float s = 1.0f / n;
  Q4 r = {q.w * s, q.x * s, q.y * s, q.z * s};
  return r;
}
struct M3 { float m[9]; };
__device__ __forceinline__ M3 q2m(Q4 q) {
  float w = q.w, x = q.x, y = q.y, z = q.z;
  M3 R;
  R.m[0] = w * w + x * x - y * y - z * z; R.m[1] = 2 * (x * y - w * z); R.m[2] = 2 * (x * z + w * y);
  R.m[3] = 2 * (x * y + w * z); R.m[4] = w * w - x * x + y * y - z * z; R.m[5] = 2 * (y * z - w * x);
  R.m[6] = 2 * (x * z - w * y); R.m[7] = 2 * (y * z + w * x); R.m[8] = w * w - x * x - y * y + z * z;
  return R;
}
__device__ __forceinline__ M3 ldm(const float* p) { M3 R; for (int i = 0; i < 9; i++) R.m[i] = p[i]; return R; }
__device__ __forceinline__ void stm(float* p, const M3& R) { for (int i = 0; i < 9; i++) p[i] = R.m[i]; }
__device__ __forceinline__ V3 mv(const M3& R, V3 v) { return v3(R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z, R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z, R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z); }
__device__ __forceinline__ V3 mtv(const M3& R, V3 v) { return v3(R.m[0] * v.x + R.m[3] * v.y + R.m[6] * v.z, R.m[1] * v.x + R.m[4] * v.y + R.m[7] * v.z, R.m[2] * v.x + R.m[5] * v.y + R.m[8] * v.z); }
__device__ __forceinline__ V3 col(const M3& R, int k) { return v3(R.m[k], R.m[3 + k], R.m[6 + k]); }
__device__ __forceinline__ M3 mm(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
  return r;
}
__device__ __forceinline__ M3 mtm(const M3& a, const M3& b) {  // a^T b
  M3 r;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[3 * i + j] = a.m[i] * b.m[j] + a.m[3 + i] * b.m[3 + j] + a.m[6 + i] * b.m[6 + j];
  return r;
}
__device__ __forceinline__ Q4 axisangle(V3 axis, float ang) {
  float s = sinf(0.5f * ang);
  Q4 q = {cosf(0.5f * ang), axis.x * s, axis.y * s, axis.z * s};
  return q;
}
// spatial vectors [ang; lin]
struct S6 { V3 a, l; };
__device__ __forceinline__ S6 ld6(const float* p) { S6 s = {ld3(p), ld3(p + 3)}; return s; }
__device__ __forceinline__ void st6(float* p, S6 s) { st3(p, s.a); st3(p + 3, s.l); }
__device__ __forceinline__ S6 operator+(S6 a, S6 b) { S6 r = {a.a + b.a, a.l + b.l}; return r; }
__device__ __forceinline__ S6 operator*(S6 a, float s) { S6 r = {a.a * s, a.l * s}; return r; }
__device__ __forceinline__ float dot6(S6 a, S6 b) { return dot(a.a, b.a) + dot(a.l, b.l); }
__device__ __forceinline__ S6 cross_motion(S6 v, S6 s) { S6 r = {cross(v.a, s.a), cross(v.a, s.l) + cross(v.l, s.a)}; return r; }
__device__ __forceinline__ S6 cross_force(S6 v, S6 f) { S6 r = {cross(v.a, f.a) + cross(v.l, f.l), cross(v.a, f.l)}; return r; }
__device__ __forceinline__ S6 mul_inert(const float* I, S6 v) {
  S6 r;
  r.a.x = I[0] * v.a.x + I[3] * v.a.y + I[4] * v.a.z - I[8] * v.l.y + I[7] * v.l.z;
  r.a.y = I[3] * v.a.x + I[1] * v.a.y + I[5] * v.a.z + I[8] * v.l.x - I[6] * v.l.z;
  r.a.z = I[4] * v.a.x + I[5] * v.a.y + I[2] * v.a.z - I[7] * v.l.x + I[6] * v.l.y;
  r.l.x = I[8] * v.a.y - I[7] * v.a.z + I[9] * v.l.x;
  r.l.y = I[6] * v.a.z - I[8] * v.a.x + I[9] * v.l.y;
  r.l.z = I[7] * v.a.x - I[6] * v.a.y + I[9] * v.l.z;
  return r;
}

// ------------------------------------------------------------------------------------------------------------
// per-env LDS state
// ------------------------------------------------------------------------------------------------------------
template <int NB, int NJ, int NV, int NG, int NS, int NCON, int NEFC, int NPAIR>
struct Smem {
  static constexpr int NVP = NV + 1;  // padded row stride (bank-conflict-free column access)
  static constexpr int NV_ = NV;      // register-array extent of per-dof loops (nv <= NV)
  static constexpr int CD_ = 4;       // largest contact dimension this configuration handles (condim 1, 3, 4)
  float qpos[NV + 8], qvel[NV], qacc[NV], qacc_ws[NV], ctrl[NV];
  float xpos[NB * 3], xquat[NB * 4], xmat[NB * 9], xipos[NB * 3];
  float xanchor[NJ * 3], xaxis[NJ * 3];
  float rootcom[NB * 3];
  float cinert[NB * 10], crb[NB * 10];
  float cdof[NV * 6], cdof_dot[NV * 6], fbuf[NV * 6];
  float cvel[NB * 6], cacc[NB * 6], cfrc[NB * 6], cflu[NB * 6];
  float M[NV * NVP], L[NV * NVP], H[NV * NVP], Lh[NV * NVP];
  float invdiag[NV], invdiag_h[NV];
  float qfrc_bias[NV], qfrc_passive[NV], qfrc_actuator[NV], qfrc_smooth[NV], qacc_smooth[NV], qfrc_constraint[NV];
  float va[NV], vMa[NV], vgrad[NV], vsearch[NV], vMv[NV];
  float gpos[NG * 3], gmat[NG * 9], gcen[NG * 3];
  float spos[NS * 3], smat[NS * 9];
  // contacts
  float cpos[NCON * 3], cframe[NCON * 9], cdist[NCON], cfri[NCON * 5], csolref[NCON * 2], csolimp[NCON * 5], cmu[NCON], cmargin[NCON];
  int cg1[NCON], cg2[NCON], cdim[NCON], cefc[NCON];
  // constraint rows
  float J[NEFC * NV];  // row-major, stride NV (= 16): rows are also the MFMA B operand (4 rows per instruction)
  float W[NEFC * NV];  // Hessian-weighted rows (MFMA A operand)
  float e_pos[NEFC], e_margin[NEFC], e_R[NEFC], e_D[NEFC], e_aref[NEFC], e_fl[NEFC], e_force[NEFC], e_jar[NEFC], e_jv[NEFC], e_K[NEFC], e_B[NEFC], e_imp[NEFC];
  int e_type[NEFC], e_id[NEFC], e_state[NEFC];
  int blk_start[NEFC], blk_dim[NEFC];
  int cand[NPAIR];
  float cstate[RSIM_CS_SIZE];
  float scratch[128];
  int ncon, nefc, nblk, niter;
  // model tables staged once per launch (int: shared topology; float: this env's constants)
  int tab_i[RSIM_NIT];
  float tab_f[RSIM_NFT];
};

#define IT(tab, i) (s.tab_i[m.io[tab] + (i)])
#define FP(tab, i) (s.tab_f[m.fo[tab] + (i)])

// ------------------------------------------------------------------------------------------------------------
// dense Cholesky / solve on an n x n LDS matrix with padded stride NVP, cooperative over the wave
// ------------------------------------------------------------------------------------------------------------
template <int NVP>
__device__ __forceinline__ void chol_factor(float* L, float* invdiag, const float* A, int n, int lane) {
  for (int e = lane; e < n * n; e += 64) { int i = e / n, j = e - i * n; L[i * NVP + j] = A[i * NVP + j]; }
  SYNC();
  for (int j = 0; j < n; j++) {
    float s = 0.f;
    if (lane >= j && lane < n) {
      s = L[lane * NVP + j];
      for (int k = 0; k < j; k++) s -= L[lane * NVP + k] * L[j * NVP + k];
    }
    float sj = bcast(s, j);
    float dj = sqrtf(fmaxf(sj, FMIN));
    float inv = 1.0f / dj;
    if (lane >= j && lane < n) L[lane * NVP + j] = (lane == j) ? dj : s * inv;
    if (lane == 0) invdiag[j] = inv;
    SYNC();
  }
}
// x: per-lane value (lane i < n holds b_i); returns solution component in lane i
template <int NVP>
__device__ __forceinline__ float chol_solve(const float* L, const float* invdiag, float x, int n, int lane) {
  for (int k = 0; k < n; k++) {
    float xk = bcast(x, k) * invdiag[k];
    if (lane == k) x = xk;
    else if (lane > k && lane < n) x -= L[lane * NVP + k] * xk;
  }
  for (int k = n - 1; k >= 0; k--) {
    float xk = bcast(x, k) * invdiag[k];
    if (lane == k) x = xk;
    else if (lane < k) x -= L[k * NVP + lane] * xk;
  }
  return x;
}

typedef float v4f __attribute__((ext_vector_type(4)));

// Register-resident Cholesky: lane i (< N) owns row i of the SPD matrix in a[0..N); all loops unroll so every index is a
// compile-time register and every broadcast is a v_readlane.  After the call a[k] = L[i][k] (k <= i), inv[k] = 1 / L[k][k] (uniform).
template <int N>
__device__ __forceinline__ void rchol_factor(float (&a)[N], float (&inv)[N]) {
#pragma unroll
  for (int j = 0; j < N; j++) {
    const float iv = rsqrtf(fmaxf(bcast(a[j], j), FMIN));
    inv[j] = iv;
    const float lij = a[j] * iv;
    a[j] = lij;
#pragma unroll
    for (int k = j + 1; k < N; k++) a[k] = fmaf(-lij, bcast(lij, k), a[k]);
  }
}
// x_i in lane i; a = rows of L, at[k] = L[k][i] (column i of L), inv = 1/diag.  Returns (L L^T)^-1 x, component i in lane i.
template <int N>
__device__ __forceinline__ float rchol_solve(const float (&a)[N], const float (&at)[N], const float (&inv)[N], float x, int lane) {
#pragma unroll
  for (int k = 0; k < N; k++) {
    const float xk = bcast(x, k) * inv[k];
    x = lane == k ? xk : (lane > k ? fmaf(-a[k], xk, x) : x);
  }
#pragma unroll
  for (int k = N - 1; k >= 0; k--) {
    const float xk = bcast(x, k) * inv[k];
    x = lane == k ? xk : (lane < k ? fmaf(-at[k], xk, x) : x);
  }
  return x;
}
template <int N>
__device__ __forceinline__ float sel(const float (&v)[N], int i) {
  float r = v[0];
#pragma unroll
  for (int k = 1; k < N; k++) r = i == k ? v[k] : r;
  return r;
}

// solve SPD N x N system A x = b in registers (N <= 6), evaluated uniformly by every lane
template <int N>
__device__ __forceinline__ void spd_solve_small(const float* A, const float* b, float* x) {
  float Lm[N * N];
  for (int j = 0; j < N; j++) {
    float sj = A[j * N + j];
    for (int k = 0; k < j; k++) sj -= Lm[j * N + k] * Lm[j * N + k];
    float dj = sqrtf(fmaxf(sj, FMIN));
    Lm[j * N + j] = dj;
    for (int i = j + 1; i < N; i++) {
      float t = A[i * N + j];
      for (int k = 0; k < j; k++) t -= Lm[i * N + k] * Lm[j * N + k];
      Lm[i * N + j] = t / dj;
    }
  }
  for (int i = 0; i < N; i++) { float t = b[i]; for (int k = 0; k < i; k++) t -= Lm[i * N + k] * x[k]; x[i] = t / Lm[i * N + i]; }
  for (int i = N - 1; i >= 0; i--) { float t = x[i]; for (int k = i + 1; k < N; k++) t -= Lm[k * N + i] * x[k]; x[i] = t / Lm[i * N + i]; }
}

// ------------------------------------------------------------------------------------------------------------
// the simulator (all methods are wave-cooperative: every lane of the env's wavefront calls them)
// ------------------------------------------------------------------------------------------------------------
template <class SM>
struct Sim {
  SM& s;
  const DModel& m;
  const float* fp;
  int lane;
  Prof pf;
  static constexpr int NVP = SM::NVP;
  static constexpr int NV16 = SM::NV_;
  static constexpr int CD = SM::CD_;

  __device__ Sim(SM& s_, const DModel& m_, const float* fp_, int lane_, unsigned long long* prof) : s(s_), m(m_), fp(fp_), lane(lane_) { pf.p = prof; pf.lane = lane_; pf.t0 = 0; }

  __device__ __forceinline__ u64 mask2(int tab, int i) const { return (u64)(uint32_t)IT(tab, 2 * i) | ((u64)(uint32_t)IT(tab, 2 * i + 1) << 32); }

  // ---------------------------------------------------------------- kinematics
  __device__ void kinematics() {
    const int b = lane;
    const int nb = m.nbody;
    int depth = (b < nb) ? IT(IO_body_depth, b) : -1;
    if (b == 0) {
      st3(s.xpos, v3(0, 0, 0));
      Q4 q = {1, 0, 0, 0};
      stq(s.xquat, q);
      stm(s.xmat, q2m(q));
    }
    SYNC();
    for (int lvl = 1; lvl <= m.maxdepth; lvl++) {
      if (depth == lvl) {
        int p = IT(IO_body_parentid, b), jadr = IT(IO_body_jntadr, b), jnum = IT(IO_body_jntnum, b);
        V3 pos;
        Q4 quat;
        if (jnum == 1 && IT(IO_jnt_type, jadr) == JNT_FREE) {
          int a = IT(IO_jnt_qposadr, jadr);
          pos = ld3(s.qpos + a);
          quat = qnorm(ldq(s.qpos + a + 3));
          st3(s.xanchor + 3 * jadr, pos);
          st3(s.xaxis + 3 * jadr, v3(0, 0, 1));
        } else {
          M3 Rp = ldm(s.xmat + 9 * p);
          pos = ld3(s.xpos + 3 * p) + mv(Rp, ld3(&FP(FO_body_pos, 3 * b)));
          quat = qmul(ldq(s.xquat + 4 * p), ldq(&FP(FO_body_quat, 4 * b)));
          for (int j = jadr; j < jadr + jnum; j++) {
            M3 R = q2m(quat);
            V3 jpos = ld3(&FP(FO_jnt_pos, 3 * j)), jax = ld3(&FP(FO_jnt_axis, 3 * j));
            V3 anchor = pos + mv(R, jpos), axis = mv(R, jax);
            st3(s.xanchor + 3 * j, anchor);
            st3(s.xaxis + 3 * j, axis);
            int a = IT(IO_jnt_qposadr, j), t = IT(IO_jnt_type, j);
            if (t == JNT_SLIDE) pos = pos + axis * (s.qpos[a] - FP(FO_qpos0, a));
            else {
              Q4 ql = (t == JNT_HINGE) ? axisangle(jax, s.qpos[a] - FP(FO_qpos0, a)) : qnorm(ldq(s.qpos + a));
              quat = qmul(quat, ql);
              pos = anchor - mv(q2m(quat), jpos);
            }
          }
          quat = qnorm(quat);
        }
        st3(s.xpos + 3 * b, pos);
        stq(s.xquat + 4 * b, quat);
        stm(s.xmat + 9 * b, q2m(quat));
      }
      SYNC();
    }
    if (b < nb) st3(s.xipos + 3 * b, ld3(s.xpos + 3 * b) + mv(ldm(s.xmat + 9 * b), ld3(&FP(FO_body_ipos, 3 * b))));
    // colliding geoms
    for (int g = lane; g < m.ncg; g += 64) {
      int gb = IT(IO_cg_bodyid, g);
      M3 Rb = ldm(s.xmat + 9 * gb);
      V3 gp = ld3(s.xpos + 3 * gb) + mv(Rb, ld3(&FP(FO_cg_pos, 3 * g)));
      M3 Rg = q2m(qnorm(qmul(ldq(s.xquat + 4 * gb), ldq(&FP(FO_cg_quat, 4 * g)))));
      st3(s.gpos + 3 * g, gp);
      stm(s.gmat + 9 * g, Rg);
      st3(s.gcen + 3 * g, gp + mv(Rg, ld3(&FP(FO_cg_rcenter, 3 * g))));
    }
    for (int k = lane; k < m.nsite; k += 64) {
      int sb = IT(IO_site_bodyid, k);
      st3(s.spos + 3 * k, ld3(s.xpos + 3 * sb) + mv(ldm(s.xmat + 9 * sb), ld3(&FP(FO_site_pos, 3 * k))));
      stm(s.smat + 9 * k, q2m(qnorm(qmul(ldq(s.xquat + 4 * sb), ldq(&FP(FO_site_quat, 4 * k))))));
    }
    SYNC();
  }

  // ---------------------------------------------------------------- subtree COM of every tree root, cinert, cdof
  __device__ void com_pos() {
    const int nb = m.nbody;
    if (lane < nb && IT(IO_body_isroot, lane)) {
      int r = lane;
      V3 acc = v3(0, 0, 0);
      for (int b = r; b < nb; b++)
        if (IT(IO_body_rootid, b) == r) acc = acc + ld3(s.xipos + 3 * b) * FP(FO_body_mass, b);
      float mt = FP(FO_body_subtreemass, r);
      st3(s.rootcom + 3 * r, mt < 1e-15f ? ld3(s.xipos + 3 * r) : acc * (1.0f / mt));
    }
    if (lane == 0) st3(s.rootcom, ld3(s.xipos));
    SYNC();
    if (lane < nb) {
      int b = lane;
      M3 R = q2m(qmul(ldq(s.xquat + 4 * b), ldq(&FP(FO_body_iquat, 4 * b))));
      V3 I = ld3(&FP(FO_body_inertia, 3 * b));
      float mass = FP(FO_body_mass, b);
      V3 off = ld3(s.xipos + 3 * b) - ld3(s.rootcom + 3 * IT(IO_body_rootid, b));
      float Iw[9];
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Iw[3 * i + j] = R.m[3 * i] * I.x * R.m[3 * j] + R.m[3 * i + 1] * I.y * R.m[3 * j + 1] + R.m[3 * i + 2] * I.z * R.m[3 * j + 2];
      float d2 = dot(off, off);
      float* ci = s.cinert + 10 * b;
      ci[0] = Iw[0] + mass * (d2 - off.x * off.x);
      ci[1] = Iw[4] + mass * (d2 - off.y * off.y);
      ci[2] = Iw[8] + mass * (d2 - off.z * off.z);
      ci[3] = Iw[1] - mass * off.x * off.y;
      ci[4] = Iw[2] - mass * off.x * off.z;
      ci[5] = Iw[5] - mass * off.y * off.z;
      ci[6] = mass * off.x; ci[7] = mass * off.y; ci[8] = mass * off.z; ci[9] = mass;
    }
    if (lane < m.njnt) {
      int j = lane, b = IT(IO_jnt_bodyid, j), da = IT(IO_jnt_dofadr, j), t = IT(IO_jnt_type, j);
      V3 off = ld3(s.rootcom + 3 * IT(IO_body_rootid, b)) - ld3(s.xanchor + 3 * j);
      if (t == JNT_FREE) {
        M3 R = ldm(s.xmat + 9 * b);
        for (int k = 0; k < 3; k++) {
          S6 c = {v3(0, 0, 0), v3(k == 0, k == 1, k == 2)};
          st6(s.cdof + 6 * (da + k), c);
          V3 ax = col(R, k);
          S6 cr = {ax, cross(ax, off)};
          st6(s.cdof + 6 * (da + 3 + k), cr);
        }
      } else if (t == JNT_BALL) {
        M3 R = ldm(s.xmat + 9 * b);
        for (int k = 0; k < 3; k++) { V3 ax = col(R, k); S6 cr = {ax, cross(ax, off)}; st6(s.cdof + 6 * (da + k), cr); }
      } else if (t == JNT_SLIDE) {
        S6 c = {v3(0, 0, 0), ld3(s.xaxis + 3 * j)};
        st6(s.cdof + 6 * da, c);
      } else {
        V3 ax = ld3(s.xaxis + 3 * j);
        S6 c = {ax, cross(ax, off)};
        st6(s.cdof + 6 * da, c);
      }
    }
    SYNC();
  }

  // ---------------------------------------------------------------- CRBA -> dense M, Cholesky
  __device__ void crb() {
    const int nb = m.nbody, nv = m.nv;
    for (int item = lane; item < nb * 10; item += 64) {
      int b = item / 10, k = item - b * 10;
      float acc = 0.f;
      if (IT(IO_body_dofnum, b) > 0) {
        for (int d = b; d < nb; d++)
          if ((mask2(IO_body_ancmask, d) >> b) & 1ull) acc += s.cinert[10 * d + k];
      }
      s.crb[item] = acc;
    }
    SYNC();
    if (lane < nv) st6(s.fbuf + 6 * lane, mul_inert(s.crb + 10 * IT(IO_dof_bodyid, lane), ld6(s.cdof + 6 * lane)));
    SYNC();
    for (int e = lane; e < nv * nv; e += 64) {
      int i = e / nv, j = e - i * nv;
      if (j > i) continue;
      float v = 0.f;
      if ((mask2(IO_dof_ancmask, i) >> j) & 1ull) v = dot6(ld6(s.cdof + 6 * j), ld6(s.fbuf + 6 * i));
      if (i == j) v += FP(FO_dof_armature, i);
      s.M[i * NVP + j] = v;
      s.M[j * NVP + i] = v;
    }
    SYNC();
    {
      float mr[NV16], minv[NV16];
      const int rr = lane & (NV16 - 1);
#pragma unroll
      for (int k = 0; k < NV16; k++) mr[k] = (rr < nv && k < nv) ? s.M[rr * NVP + k] : (rr == k ? 1.f : 0.f);
      rchol_factor<NV16>(mr, minv);
      if (lane < NV16) {
#pragma unroll
        for (int k = 0; k < NV16; k++) s.L[lane * NVP + k] = mr[k];
        s.invdiag[lane] = sel(minv, lane);
      }
    }
    SYNC();
  }

  // ---------------------------------------------------------------- velocity stage: cvel, cdof_dot, bias, passive
  __device__ void velocity() {
    const int nb = m.nbody, nv = m.nv;
    const float density = FP(FO_opt, 4), viscosity = FP(FO_opt, 5);
    const V3 grav = v3(FP(FO_opt, 1), FP(FO_opt, 2), FP(FO_opt, 3)), wind = v3(FP(FO_opt, 7), FP(FO_opt, 8), FP(FO_opt, 9));
    if (lane < nv) {
      int i = lane;
      S6 cv = {v3(0, 0, 0), v3(0, 0, 0)};
      u64 mk = mask2(IO_dof_cvelmask, i);
      while (mk) { int k = __ffsll((long long)mk) - 1; mk &= mk - 1; cv = cv + ld6(s.cdof + 6 * k) * s.qvel[k]; }
      S6 cd = {v3(0, 0, 0), v3(0, 0, 0)};
      if (!IT(IO_dof_zerodot, i)) cd = cross_motion(cv, ld6(s.cdof + 6 * i));
      st6(s.cdof_dot + 6 * i, cd);
    }
    if (lane < nb) {
      int b = lane;
      S6 cv = {v3(0, 0, 0), v3(0, 0, 0)};
      u64 mk = mask2(IO_body_dofmask, b);
      while (mk) { int k = __ffsll((long long)mk) - 1; mk &= mk - 1; cv = cv + ld6(s.cdof + 6 * k) * s.qvel[k]; }
      st6(s.cvel + 6 * b, cv);
    }
    SYNC();
    if (lane < nb) {
      int b = lane;
      S6 zero = {v3(0, 0, 0), v3(0, 0, 0)};
      S6 frc = zero, flu = zero;
      if (IT(IO_body_moving, b)) {
        S6 ca = {v3(0, 0, 0), -grav};
        u64 mk = mask2(IO_body_dofmask, b);
        while (mk) { int k = __ffsll((long long)mk) - 1; mk &= mk - 1; ca = ca + ld6(s.cdof_dot + 6 * k) * s.qvel[k]; }
        S6 cv = ld6(s.cvel + 6 * b);
        frc = mul_inert(s.cinert + 10 * b, ca) + cross_force(cv, mul_inert(s.cinert + 10 * b, cv));
        float mass = FP(FO_body_mass, b);
        if (mass >= 1e-15f && (density > 0.f || viscosity > 0.f)) {
          // inertia-box fluid model: force/torque at the body COM, folded into a spatial force about the tree COM
          V3 I = ld3(&FP(FO_body_inertia, 3 * b));
          M3 R = q2m(qmul(ldq(s.xquat + 4 * b), ldq(&FP(FO_body_iquat, 4 * b))));
          float bx = sqrtf(fmaxf(1e-15f, I.y + I.z - I.x) / mass * 6.0f), by = sqrtf(fmaxf(1e-15f, I.x + I.z - I.y) / mass * 6.0f),
                bz = sqrtf(fmaxf(1e-15f, I.x + I.y - I.z) / mass * 6.0f);
          V3 off = ld3(s.xipos + 3 * b) - ld3(s.rootcom + 3 * IT(IO_body_rootid, b));
          V3 gl = cv.l + cross(cv.a, off) - wind;
          V3 la = mtv(R, cv.a), ll = mtv(R, gl);
          V3 ft = v3(0, 0, 0), ff = v3(0, 0, 0);
          if (viscosity > 0.f) {
            float diam = (bx + by + bz) / 3.0f;
            ft = la * (-PI_F * diam * diam * diam * viscosity);
            ff = ll * (-3.0f * PI_F * diam * viscosity);
          }
          if (density > 0.f) {
            ff.x -= 0.5f * density * by * bz * fabsf(ll.x) * ll.x;
            ff.y -= 0.5f * density * bx * bz * fabsf(ll.y) * ll.y;
            ff.z -= 0.5f * density * bx * by * fabsf(ll.z) * ll.z;
            float bx4 = bx * bx * bx * bx, by4 = by * by * by * by, bz4 = bz * bz * bz * bz;
            ft.x -= density * bx * (by4 + bz4) * fabsf(la.x) * la.x / 64.0f;
            ft.y -= density * by * (bx4 + bz4) * fabsf(la.y) * la.y / 64.0f;
            ft.z -= density * bz * (bx4 + by4) * fabsf(la.z) * la.z / 64.0f;
          }
          V3 gt = mv(R, ft), gf = mv(R, ff);
          flu.a = gt + cross(off, gf);
          flu.l = gf;
        }
      }
      st6(s.cfrc + 6 * b, frc);
      st6(s.cflu + 6 * b, flu);
    }
    SYNC();
    if (lane < nv) {
      int i = lane, bi = IT(IO_dof_bodyid, i);
      S6 zero = {v3(0, 0, 0), v3(0, 0, 0)};
      S6 sf = zero, sl = zero;
      for (int d = bi; d < nb; d++)
        if ((mask2(IO_body_ancmask, d) >> bi) & 1ull) { sf = sf + ld6(s.cfrc + 6 * d); sl = sl + ld6(s.cflu + 6 * d); }
      S6 cd = ld6(s.cdof + 6 * i);
      s.qfrc_bias[i] = dot6(cd, sf);
      s.qfrc_passive[i] = -FP(FO_dof_damping, i) * s.qvel[i] + dot6(cd, sl);
    }
    SYNC();
  }

  // Jacobian column of world point p attached to body `b` for dof i: returns [jacr; jacp] or zero if i does not move b
  __device__ __forceinline__ S6 jac_col(int b, V3 p, int i) const {
    S6 z = {v3(0, 0, 0), v3(0, 0, 0)};
    if (!((mask2(IO_body_dofmask, b) >> i) & 1ull)) return z;
    S6 cd = ld6(s.cdof + 6 * i);
    V3 off = p - ld3(s.rootcom + 3 * IT(IO_body_rootid, b));
    S6 r = {cd.a, cd.l + cross(cd.a, off)};
    return r;
  }

  // ---------------------------------------------------------------- collision
  __device__ __forceinline__ void make_frame(V3 n, float* frame) {
    n = normalized(n);
    V3 y = (n.y < 0.5f && n.y > -0.5f) ? v3(0, 1, 0) : v3(0, 0, 1);
    y = y - n * dot(n, y);
    y = normalized(y);
    V3 z = cross(n, y);
    st3(frame, n); st3(frame + 3, y); st3(frame + 6, z);
  }

  // support point of colliding geom g along world direction dir (wave-cooperative for meshes; result uniform)
  __device__ V3 support(int g, V3 dir) {
    int t = IT(IO_cg_type, g);
    M3 R = ldm(s.gmat + 9 * g);
    V3 p = ld3(s.gpos + 3 * g), sz = ld3(&FP(FO_cg_size, 3 * g));
    V3 ld = mtv(R, dir), lp = v3(0, 0, 0);
    if (t == G_BOX) lp = v3(ld.x >= 0 ? sz.x : -sz.x, ld.y >= 0 ? sz.y : -sz.y, ld.z >= 0 ? sz.z : -sz.z);
    else if (t == G_SPHERE) { float n = norm(ld); if (n > FMIN) lp = ld * (sz.x / n); }
    else if (t == G_CYLINDER) {
      float n = sqrtf(ld.x * ld.x + ld.y * ld.y);
      if (n > FMIN) { lp.x = ld.x / n * sz.x; lp.y = ld.y / n * sz.x; }
      lp.z = ld.z >= 0 ? sz.y : -sz.y;
    } else if (t == G_CAPSULE) {
      float n = norm(ld);
      if (n > FMIN) lp = ld * (sz.x / n);
      lp.z += ld.z >= 0 ? sz.y : -sz.y;
    } else if (t == G_ELLIPSOID) {
      V3 tt = v3(ld.x * sz.x, ld.y * sz.y, ld.z * sz.z);
      float n = norm(tt);
      if (n > FMIN) lp = v3(tt.x / n * sz.x, tt.y / n * sz.y, tt.z / n * sz.z);
    } else if (t == G_MESH) {
      int adr = IT(IO_cg_meshadr, g), num = IT(IO_cg_meshnum, g);
      float bv = -3.0e38f;
      int bi = 0x7fffffff;
      for (int i = lane; i < num; i += 64) {
        const float* v = m.mesh_vert + 3 * (adr + i);
        float val = v[0] * ld.x + v[1] * ld.y + v[2] * ld.z;
        if (val > bv) { bv = val; bi = i; }
      }
      // wave arg-max, lowest index wins ties (matches the serial first-maximum scan)
      for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(bv, o);
        int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      bi = uni(bi);
      lp = ld3(m.mesh_vert + 3 * (adr + bi));
    }
    return p + mv(R, lp);
  }

  __device__ __forceinline__ void add_contact(float dist, V3 pos, V3 nrm, int g1, int g2, float margin, float gap) {
    // executed by ONE lane; caller maintains s.ncon
    int c = s.ncon;
    if (c >= (int)(sizeof(s.cdist) / sizeof(float))) return;
    s.ncon = c + 1;
    s.cdist[c] = dist;
    st3(s.cpos + 3 * c, pos);
    make_frame(nrm, s.cframe + 9 * c);
    s.cg1[c] = g1; s.cg2[c] = g2;
    s.cmargin[c] = margin - gap;
    int p1 = IT(IO_cg_priority, g1), p2 = IT(IO_cg_priority, g2);
    float fr[3];
    if (p1 != p2) {
      int gp = p1 > p2 ? g1 : g2;
      s.cdim[c] = IT(IO_cg_condim, gp);
      for (int k = 0; k < 2; k++) s.csolref[2 * c + k] = FP(FO_cg_solref, 2 * gp + k);
      for (int k = 0; k < 5; k++) s.csolimp[5 * c + k] = FP(FO_cg_solimp, 5 * gp + k);
      for (int k = 0; k < 3; k++) fr[k] = FP(FO_cg_friction, 3 * gp + k);
    } else {
      int d1 = IT(IO_cg_condim, g1), d2 = IT(IO_cg_condim, g2);
      s.cdim[c] = d1 > d2 ? d1 : d2;
      float s1 = FP(FO_cg_solmix, g1), s2 = FP(FO_cg_solmix, g2), mix;
      if (s1 >= 1e-15f && s2 >= 1e-15f) mix = s1 / (s1 + s2);
      else if (s1 < 1e-15f && s2 < 1e-15f) mix = 0.5f;
      else mix = s1 < 1e-15f ? 0.0f : 1.0f;
      float r10 = FP(FO_cg_solref, 2 * g1), r11 = FP(FO_cg_solref, 2 * g1 + 1), r20 = FP(FO_cg_solref, 2 * g2), r21 = FP(FO_cg_solref, 2 * g2 + 1);
      if (r10 > 0 && r20 > 0) { s.csolref[2 * c] = mix * r10 + (1 - mix) * r20; s.csolref[2 * c + 1] = mix * r11 + (1 - mix) * r21; }
      else { s.csolref[2 * c] = fminf(r10, r20); s.csolref[2 * c + 1] = fminf(r11, r21); }
      for (int k = 0; k < 5; k++) s.csolimp[5 * c + k] = mix * FP(FO_cg_solimp, 5 * g1 + k) + (1 - mix) * FP(FO_cg_solimp, 5 * g2 + k);
      for (int k = 0; k < 3; k++) fr[k] = fmaxf(FP(FO_cg_friction, 3 * g1 + k), FP(FO_cg_friction, 3 * g2 + k));
    }
    float* f = s.cfri + 5 * c;
    f[0] = f[1] = fr[0]; f[2] = fr[1]; f[3] = f[4] = fr[2];
  }

  // box(g1)-box(g2): SAT + face clipping / edge-edge; executed by lane 0 (serial geometry, LDS scratch polygons)
  __device__ void box_box_lane0(int g1, int g2, float margin, float gap) {
    V3 pa = ld3(s.gpos + 3 * g1), pb = ld3(s.gpos + 3 * g2);
    M3 Ra = ldm(s.gmat + 9 * g1), Rb = ldm(s.gmat + 9 * g2);
    float ha[3] = {FP(FO_cg_size, 3 * g1), FP(FO_cg_size, 3 * g1 + 1), FP(FO_cg_size, 3 * g1 + 2)};
    float hb[3] = {FP(FO_cg_size, 3 * g2), FP(FO_cg_size, 3 * g2 + 1), FP(FO_cg_size, 3 * g2 + 2)};
    V3 A[3] = {col(Ra, 0), col(Ra, 1), col(Ra, 2)}, B[3] = {col(Rb, 0), col(Rb, 1), col(Rb, 2)};
    V3 dab = pb - pa;
    float sA = -3e38f, sB = -3e38f, best_face, best_edge = -3e38f;
    int idA = 0, idB = 3, face_id, edge_id = -1;
    V3 edge_axis = v3(0, 0, 0);
    for (int i = 0; i < 6; i++) {
      V3 Lx = i < 3 ? A[i] : B[i - 3];
      float ra = 0, rb = 0;
      for (int k = 0; k < 3; k++) { ra += ha[k] * fabsf(dot(Lx, A[k])); rb += hb[k] * fabsf(dot(Lx, B[k])); }
      float sv = fabsf(dot(Lx, dab)) - ra - rb;
      if (sv > margin) return;
      if (i < 3) { if (sv > sA) { sA = sv; idA = i; } }
      else if (sv > sB) { sB = sv; idB = i; }
    }
    if (sB > sA + 1e-6f) { best_face = sB; face_id = idB; } else { best_face = sA; face_id = idA; }
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        V3 Lx = cross(A[i], B[j]);
        float n = norm(Lx);
        if (n < 1e-6f) continue;
        Lx = Lx * (1.0f / n);
        float ra = 0, rb = 0;
        for (int k = 0; k < 3; k++) { ra += ha[k] * fabsf(dot(Lx, A[k])); rb += hb[k] * fabsf(dot(Lx, B[k])); }
        float sv = fabsf(dot(Lx, dab)) - ra - rb;
        if (sv > margin) return;
        if (sv > best_edge) { best_edge = sv; edge_id = 3 * i + j; edge_axis = Lx; }
      }
    if (edge_id >= 0 && best_edge > 0.95f * best_face + 1e-5f && best_edge > best_face + 1e-5f) {
      int i = edge_id / 3, j = edge_id - 3 * i;
      V3 n = edge_axis;
      if (dot(n, dab) < 0) n = -n;
      V3 qa = pa, qb = pb;
      for (int a = 0; a < 3; a++) {
        if (a != i) qa = qa + A[a] * ((dot(n, A[a]) > 0 ? 1.f : -1.f) * ha[a]);
        if (a != j) qb = qb + B[a] * ((dot(n, B[a]) > 0 ? -1.f : 1.f) * hb[a]);
      }
      V3 r = qb - qa;
      float ab = dot(A[i], B[j]), den = 1 - ab * ab, ra_ = dot(r, A[i]), rb_ = dot(r, B[j]), sp = 0, tp = 0;
      if (den > 1e-12f) { sp = (ra_ - ab * rb_) / den; tp = (ab * ra_ - rb_) / den; }
      sp = fmaxf(-ha[i], fminf(ha[i], sp));
      tp = fmaxf(-hb[j], fminf(hb[j], tp));
      add_contact(best_edge, ((qa + A[i] * sp) + (qb + B[j] * tp)) * 0.5f, n, g1, g2, margin, gap);
      return;
    }
    bool ref_is_a = face_id < 3;
    int ax = face_id % 3;
    V3 pr = ref_is_a ? pa : pb, pi_ = ref_is_a ? pb : pa;
    const float* hr = ref_is_a ? ha : hb;
    const float* hi = ref_is_a ? hb : ha;
    const V3* Rr = ref_is_a ? A : B;
    const V3* Ri = ref_is_a ? B : A;
    V3 dri = pi_ - pr;
    float sg = dot(Rr[ax], dri) >= 0 ? 1.f : -1.f;
    V3 n = Rr[ax] * sg;
    int iax = 0;
    float bestd = -1;
    for (int k = 0; k < 3; k++) { float v = fabsf(dot(Ri[k], n)); if (v > bestd) { bestd = v; iax = k; } }
    float isg = dot(Ri[iax], n) > 0 ? -1.f : 1.f;
    int u = (iax + 1) % 3, v = (iax + 2) % 3, ru = (ax + 1) % 3, rv = (ax + 2) % 3;
    float* poly = s.scratch;        // [16][3]
    float* tmp = s.scratch + 48;    // [16][3]
    int np = 0;
    const float cs[4][2] = {{1, 1}, {-1, 1}, {-1, -1}, {1, -1}};
    for (int c = 0; c < 4; c++) {
      V3 w = pi_ + Ri[iax] * (isg * hi[iax]) + Ri[u] * (cs[c][0] * hi[u]) + Ri[v] * (cs[c][1] * hi[v]);
      V3 rel = w - pr;
      poly[3 * np] = dot(rel, Rr[ru]); poly[3 * np + 1] = dot(rel, Rr[rv]); poly[3 * np + 2] = dot(rel, n) - hr[ax];
      np++;
    }
    for (int pass = 0; pass < 4 && np > 0; pass++) {
      int axis = pass >> 1;
      float h = axis == 0 ? hr[ru] : hr[rv], sign = (pass & 1) ? -1.f : 1.f;
      int cnt = 0;
      for (int i = 0; i < np; i++) {
        const float* a = poly + 3 * i;
        const float* b = poly + 3 * ((i + 1) % np);
        float da = sign * a[axis] - h, db = sign * b[axis] - h;
        if (da <= 0) { tmp[3 * cnt] = a[0]; tmp[3 * cnt + 1] = a[1]; tmp[3 * cnt + 2] = a[2]; cnt++; }
        if ((da < 0 && db > 0) || (da > 0 && db < 0)) {
          float t = da / (da - db);
          for (int k = 0; k < 3; k++) tmp[3 * cnt + k] = a[k] + t * (b[k] - a[k]);
          cnt++;
        }
        if (cnt >= 15) break;
      }
      for (int i = 0; i < 3 * cnt; i++) poly[i] = tmp[i];
      np = cnt;
    }
    int cnt = 0;
    for (int i = 0; i < np && cnt < 8; i++) {
      float dist = poly[3 * i + 2];
      if (dist > margin) continue;
      V3 w = pr + Rr[ru] * poly[3 * i] + Rr[rv] * poly[3 * i + 1] + n * (hr[ax] + dist);
      add_contact(dist, w - n * (0.5f * dist), ref_is_a ? n : -n, g1, g2, margin, gap);
      cnt++;
    }
  }

  // closest point on triangle to the origin (barycentric)
  __device__ __forceinline__ V3 tri_closest_origin(V3 a, V3 b, V3 c, float* bary) {
    V3 ab = b - a, ac = c - a, ap = -a, bp = -b, cp = -c;
    float d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0 && d2 <= 0) { bary[0] = 1; bary[1] = 0; bary[2] = 0; }
    else {
      float d3 = dot(ab, bp), d4 = dot(ac, bp);
      if (d3 >= 0 && d4 <= d3) { bary[0] = 0; bary[1] = 1; bary[2] = 0; }
      else {
        float vc = d1 * d4 - d3 * d2;
        if (vc <= 0 && d1 >= 0 && d3 <= 0) { float v = d1 / (d1 - d3); bary[0] = 1 - v; bary[1] = v; bary[2] = 0; }
        else {
          float d5 = dot(ab, cp), d6 = dot(ac, cp);
          if (d6 >= 0 && d5 <= d6) { bary[0] = 0; bary[1] = 0; bary[2] = 1; }
          else {
            float vb = d5 * d2 - d1 * d6;
            if (vb <= 0 && d2 >= 0 && d6 <= 0) { float w = d2 / (d2 - d6); bary[0] = 1 - w; bary[1] = 0; bary[2] = w; }
            else {
              float va = d3 * d6 - d5 * d4;
              if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { float w = (d4 - d3) / ((d4 - d3) + (d5 - d6)); bary[0] = 0; bary[1] = 1 - w; bary[2] = w; }
              else { float den = 1.0f / (va + vb + vc), v = vb * den, w = vc * den; bary[0] = 1 - v - w; bary[1] = v; bary[2] = w; }
            }
          }
        }
      }
    }
    return a * bary[0] + b * bary[1] + c * bary[2];
  }

  // Minkowski Portal Refinement (uniform control flow; support() is wave-cooperative)
  __device__ void convex_convex(int g1, int g2, float margin, float gap) {
    const float tol = 1e-6f;
    V3 v0 = ld3(s.gcen + 3 * g1) - ld3(s.gcen + 3 * g2);
    if (norm(v0) < 1e-9f) v0.x = 1e-5f;
    V3 dir = normalized(-v0);
    V3 p11 = support(g1, dir), p12 = support(g2, -dir), v1 = p11 - p12;
    if (dot(v1, dir) <= 0) return;
    dir = cross(v0, v1);
    if (norm(dir) < 1e-12f) {
      V3 n = normalized(v1 - v0);
      if (lane == 0) add_contact(-dot(v1, n), (p11 + p12) * 0.5f, n, g1, g2, margin, gap);
      return;
    }
    dir = normalized(dir);
    V3 p21 = support(g1, dir), p22 = support(g2, -dir), v2 = p21 - p22;
    if (dot(v2, dir) <= 0) return;
    dir = cross(v1 - v0, v2 - v0);
    if (dot(dir, v0) > 0) {
      V3 t;
      t = v1; v1 = v2; v2 = t; t = p11; p11 = p21; p21 = t; t = p12; p12 = p22; p22 = t;
      dir = -dir;
    }
    V3 v3_, p31, p32;
    for (int it = 0;; it++) {
      if (it > 100) return;
      float len;
      dir = normalized(dir, &len);
      if (len < FMIN) return;
      p31 = support(g1, dir); p32 = support(g2, -dir); v3_ = p31 - p32;
      if (dot(v3_, dir) <= 0) return;
      if (dot(cross(v1, v3_), v0) < -1e-14f) { v2 = v3_; p21 = p31; p22 = p32; dir = cross(v1 - v0, v3_ - v0); continue; }
      if (dot(cross(v3_, v2), v0) < -1e-14f) { v1 = v3_; p11 = p31; p12 = p32; dir = cross(v3_ - v0, v2 - v0); continue; }
      break;
    }
    bool hit = false;
    for (int it = 0; it < 128; it++) {
      float len;
      dir = normalized(cross(v2 - v1, v3_ - v1), &len);
      if (len < FMIN) break;
      if (dot(dir, v1) >= 0) hit = true;
      V3 p41 = support(g1, dir), p42 = support(g2, -dir), v4 = p41 - p42;
      float dv4 = dot(v4, dir);
      if (dv4 < 0 && !hit) return;
      float delta = dv4 - dot(v3_, dir);
      if (delta <= tol || it == 127) break;
      V3 t = cross(v4, v0);
      if (dot(v1, t) > 0) {
        if (dot(v2, t) > 0) { v1 = v4; p11 = p41; p12 = p42; } else { v3_ = v4; p31 = p41; p32 = p42; }
      } else {
        if (dot(v3_, t) > 0) { v2 = v4; p21 = p41; p22 = p42; } else { v1 = v4; p11 = p41; p12 = p42; }
      }
    }
    if (!hit) return;
    float bary[3];
    V3 cp = tri_closest_origin(v1, v2, v3_, bary);
    float depth = norm(cp);
    V3 n = depth > 1e-12f ? cp * (1.0f / depth) : dir;
    V3 w1 = p11 * bary[0] + p21 * bary[1] + p31 * bary[2], w2 = p12 * bary[0] + p22 * bary[1] + p32 * bary[2];
    if (lane == 0) add_contact(-depth, (w1 + w2) * 0.5f, n, g1, g2, margin, gap);
  }

  // oriented bounding box of colliding geom g: world centre o, half extents h along the columns of gmat
  __device__ __forceinline__ void geom_obb(int g, V3& o, V3& h) const {
    const int t = IT(IO_cg_type, g);
    const V3 sz = ld3(&FP(FO_cg_size, 3 * g));
    V3 lc = v3(0, 0, 0);
    if (t == G_MESH) { lc = ld3(&FP(FO_cg_aabb, 6 * g)); h = ld3(&FP(FO_cg_aabb, 6 * g + 3)); }
    else if (t == G_SPHERE) h = v3(sz.x, sz.x, sz.x);
    else if (t == G_CAPSULE) h = v3(sz.x, sz.x, sz.x + sz.y);
    else if (t == G_CYLINDER) h = v3(sz.x, sz.x, sz.y);
    else h = sz;
    o = ld3(s.gpos + 3 * g) + mv(ldm(s.gmat + 9 * g), lc);
  }

  __device__ void collision() {
    if (lane == 0) s.ncon = 0;
    // broadphase: bounding spheres, order-preserving compaction of candidate pairs
    int ncand = 0;
    for (int base = 0; base < m.npair; base += 64) {
      int p = base + lane;
      bool pass = false;
      if (p < m.npair) {
        int g1 = IT(IO_pair_g1, p), g2 = IT(IO_pair_g2, p);
        float margin = fmaxf(FP(FO_cg_margin, g1), FP(FO_cg_margin, g2));
        V3 c2 = ld3(s.gcen + 3 * g2);
        V3 o2, h2;
        geom_obb(g2, o2, h2);
        M3 R2 = ldm(s.gmat + 9 * g2);
        if (IT(IO_cg_type, g1) == G_PLANE) {
          V3 nrm = col(ldm(s.gmat + 9 * g1), 2);
          pass = dot(c2 - ld3(s.gpos + 3 * g1), nrm) - FP(FO_cg_rbound, g2) <= margin;
          // plane vs oriented box of geom 2
          if (pass) pass = dot(o2 - ld3(s.gpos + 3 * g1), nrm) - (h2.x * fabsf(dot(nrm, col(R2, 0))) + h2.y * fabsf(dot(nrm, col(R2, 1))) + h2.z * fabsf(dot(nrm, col(R2, 2)))) <= margin;
        } else {
          V3 rel = c2 - ld3(s.gcen + 3 * g1);
          float bound = FP(FO_cg_rbound, g1) + FP(FO_cg_rbound, g2) + margin;
          pass = dot(rel, rel) <= bound * bound;
          if (pass) {
            // conservative separating-axis test on the 6 face normals of the two oriented bounding boxes
            V3 o1, h1;
            geom_obb(g1, o1, h1);
            M3 R1 = ldm(s.gmat + 9 * g1);
            M3 C = mtm(R1, R2);  // C[i][j] = A_i . B_j
            V3 t = o2 - o1, ta = mtv(R1, t), tb = mtv(R2, t);
            float sepa = fmaxf(fmaxf(fabsf(ta.x) - (h1.x + h2.x * fabsf(C.m[0]) + h2.y * fabsf(C.m[1]) + h2.z * fabsf(C.m[2])),
                                     fabsf(ta.y) - (h1.y + h2.x * fabsf(C.m[3]) + h2.y * fabsf(C.m[4]) + h2.z * fabsf(C.m[5]))),
                               fabsf(ta.z) - (h1.z + h2.x * fabsf(C.m[6]) + h2.y * fabsf(C.m[7]) + h2.z * fabsf(C.m[8])));
            float sepb = fmaxf(fmaxf(fabsf(tb.x) - (h2.x + h1.x * fabsf(C.m[0]) + h1.y * fabsf(C.m[3]) + h1.z * fabsf(C.m[6])),
                                     fabsf(tb.y) - (h2.y + h1.x * fabsf(C.m[1]) + h1.y * fabsf(C.m[4]) + h1.z * fabsf(C.m[7]))),
                               fabsf(tb.z) - (h2.z + h1.x * fabsf(C.m[2]) + h1.y * fabsf(C.m[5]) + h1.z * fabsf(C.m[8])));
            pass = fmaxf(sepa, sepb) <= margin + 1e-6f;
          }
        }
      }
      u64 mk = __ballot(pass);
      if (pass) s.cand[ncand + __popcll(mk & lanemask_lt(lane))] = p;
      ncand += __popcll(mk);
    }
    SYNC();
    pf.mark(RP_BROAD);
    pf.count(RP_N_CAND, ncand);
    for (int ci = 0; ci < ncand; ci++) {
      int p = uni(s.cand[ci]);
      int g1 = uni(IT(IO_pair_g1, p)), g2 = uni(IT(IO_pair_g2, p));
      int t1 = uni(IT(IO_cg_type, g1)), t2 = uni(IT(IO_cg_type, g2));
      float margin = fmaxf(FP(FO_cg_margin, g1), FP(FO_cg_margin, g2)), gap = fmaxf(FP(FO_cg_gap, g1), FP(FO_cg_gap, g2));
      if (t1 == G_PLANE && t2 == G_BOX) {
        M3 Rp = ldm(s.gmat + 9 * g1);
        V3 nrm = col(Rp, 2);
        float dist = 0;
        V3 wp = v3(0, 0, 0);
        bool hit = false;
        if (lane < 8) {
          V3 sz = ld3(&FP(FO_cg_size, 3 * g2));
          V3 lp = v3((lane & 1) ? sz.x : -sz.x, (lane & 2) ? sz.y : -sz.y, (lane & 4) ? sz.z : -sz.z);
          wp = ld3(s.gpos + 3 * g2) + mv(ldm(s.gmat + 9 * g2), lp);
          dist = dot(wp - ld3(s.gpos + 3 * g1), nrm);
          hit = dist <= margin;
        }
        u64 mk = __ballot(hit);
        // serialise in corner order (first four), one lane at a time
        int taken = 0;
        while (mk && taken < 4) {
          int l = __ffsll((long long)mk) - 1;
          mk &= mk - 1;
          if (lane == l) add_contact(dist, wp - nrm * (0.5f * dist), nrm, g1, g2, margin, gap);
          taken++;
          SYNC();
        }
      } else if (t1 == G_PLANE) {
        M3 Rp = ldm(s.gmat + 9 * g1);
        V3 nrm = col(Rp, 2);
        V3 sp = support(g2, -nrm);
        float dist = dot(sp - ld3(s.gpos + 3 * g1), nrm);
        if (dist <= margin && lane == 0) add_contact(dist, sp - nrm * (0.5f * dist), nrm, g1, g2, margin, gap);
      } else if (t1 == G_BOX && t2 == G_BOX) {
        if (lane == 0) box_box_lane0(g1, g2, margin, gap);
      } else {
        convex_convex(g1, g2, margin, gap);
      }
      SYNC();
    }
  }

  // ---------------------------------------------------------------- constraint rows
  __device__ __forceinline__ float impedance(const float* solimp, float x_abs) {
    float dmin = fminf(0.9999f, fmaxf(0.0001f, solimp[0])), dmax = fminf(0.9999f, fmaxf(0.0001f, solimp[1]));
    float width = fmaxf(1e-15f, solimp[2]), mid = fminf(0.9999f, fmaxf(0.0001f, solimp[3])), power = fmaxf(1.0f, solimp[4]);
    float x = x_abs / width, y;
    if (x >= 1) return dmax;
    if (x <= 0) return dmin;
    if (power == 1.0f) y = x;
    else if (x <= mid) y = powf(x, power) / powf(mid, power - 1);
    else y = 1 - powf(1 - x, power) / powf(1 - mid, power - 1);
    return dmin + y * (dmax - dmin);
  }
  __device__ __forceinline__ void row_params(int r, int type, int id, float pos, float margin, float fl, const float* solref, const float* solimp, float diag) {
    s.e_type[r] = type; s.e_id[r] = id; s.e_pos[r] = pos; s.e_margin[r] = margin; s.e_fl[r] = fl;
    float imp = impedance(solimp, fabsf(pos - margin));
    float dmax = fminf(0.9999f, fmaxf(0.0001f, solimp[1]));
    float K, Bd;
    if (solref[0] > 0) {
      float tc = fmaxf(solref[0], 2 * FP(FO_opt, 0)), dr = solref[1];
      Bd = 2 / fmaxf(1e-15f, dmax * tc);
      K = 1 / fmaxf(1e-15f, dmax * dmax * tc * tc * dr * dr);
    } else {
      K = -solref[0] / fmaxf(1e-15f, dmax * dmax);
      Bd = -solref[1] / fmaxf(1e-15f, dmax);
    }
    s.e_K[r] = K; s.e_B[r] = Bd; s.e_imp[r] = imp;
    s.e_R[r] = fmaxf(1e-15f, (1 - imp) / imp * diag);
  }

  __device__ void make_constraint() {
    const int nv = m.nv;
    constexpr int NEFC = sizeof(s.e_pos) / sizeof(float);
    int nefc = 0, nblk = 0;
    // zero J (rows >= nefc and columns >= nv feed the matrix cores as padding)
    for (int e = lane; e < NEFC * NV16; e += 64) s.J[e] = 0.f;
    SYNC();
    // (1) dof friction loss rows
    {
      bool act = lane < nv && FP(FO_dof_frictionloss, lane) > 0.f;
      u64 mk = __ballot(act);
      if (act) {
        int r = nefc + __popcll(mk & lanemask_lt(lane)), i = lane;
        s.J[r * NV16 + i] = 1.f;
        float solref[2] = {FP(FO_dof_solref, 2 * i), FP(FO_dof_solref, 2 * i + 1)}, solimp[5];
        for (int k = 0; k < 5; k++) solimp[k] = FP(FO_dof_solimp, 5 * i + k);
        row_params(r, C_FRICTION_DOF, i, 0, 0, FP(FO_dof_frictionloss, i), solref, solimp, FP(FO_dof_invweight0, i));
        s.blk_start[r] = r; s.blk_dim[r] = 1;
      }
      nefc += __popcll(mk);
    }
    // (2) joint limit rows (item = 2*joint + side, lower side first)
    for (int base = 0; base < 2 * m.njnt; base += 64) {
      int item = base + lane, j = item >> 1, side = (item & 1) ? 1 : -1;
      bool act = false;
      float dist = 0;
      if (item < 2 * m.njnt && IT(IO_jnt_limited, j)) {
        int t = IT(IO_jnt_type, j);
        if (t == JNT_HINGE || t == JNT_SLIDE) {
          float q = s.qpos[IT(IO_jnt_qposadr, j)];
          dist = side < 0 ? q - FP(FO_jnt_range, 2 * j) : FP(FO_jnt_range, 2 * j + 1) - q;
          act = dist < FP(FO_jnt_margin, j);
        }
      }
      u64 mk = __ballot(act);
      if (act) {
        int r = nefc + __popcll(mk & lanemask_lt(lane)), da = IT(IO_jnt_dofadr, j);
        s.J[r * NV16 + da] = (float)(-side);
        float solref[2] = {FP(FO_jnt_solref, 2 * j), FP(FO_jnt_solref, 2 * j + 1)}, solimp[5];
        for (int k = 0; k < 5; k++) solimp[k] = FP(FO_jnt_solimp, 5 * j + k);
        row_params(r, C_LIMIT_JOINT, j, dist, FP(FO_jnt_margin, j), 0, solref, solimp, FP(FO_dof_invweight0, da));
        s.blk_start[r] = r; s.blk_dim[r] = 1;
      }
      nefc += __popcll(mk);
    }
    nblk = nefc;
    SYNC();
    // (3) contacts
    int ncon = uni(s.ncon);
    for (int c = 0; c < ncon; c++) {
      int dim = uni(s.cdim[c]);
      bool active = s.cdist[c] < s.cmargin[c];
      if (!active || nefc + dim > NEFC) { if (lane == 0) s.cefc[c] = -1; continue; }
      int g1 = s.cg1[c], g2 = s.cg2[c], b1 = IT(IO_cg_bodyid, g1), b2 = IT(IO_cg_bodyid, g2);
      V3 pos = ld3(s.cpos + 3 * c);
      for (int e = lane; e < dim * nv; e += 64) {
        int k = e / nv, i = e - k * nv;
        V3 ax = ld3(s.cframe + 9 * c + 3 * (k < 3 ? k : k - 3));
        S6 c1 = jac_col(b1, pos, i), c2 = jac_col(b2, pos, i);
        s.J[(nefc + k) * NV16 + i] = (k < 3) ? dot(ax, c2.l - c1.l) : dot(ax, c2.a - c1.a);
      }
      if (lane < dim) {
        int k = lane;
        float tran = FP(FO_body_invweight0, 2 * b1) + FP(FO_body_invweight0, 2 * b2), rot = FP(FO_body_invweight0, 2 * b1 + 1) + FP(FO_body_invweight0, 2 * b2 + 1);
        row_params(nefc + k, dim == 1 ? C_CONTACT_FRICTIONLESS : C_CONTACT_ELLIPTIC, c, k == 0 ? s.cdist[c] : 0.f, k == 0 ? s.cmargin[c] : 0.f, 0.f,
                   s.csolref + 2 * c, s.csolimp + 5 * c, k < 3 ? tran : rot);
      }
      if (lane == 0) { s.cefc[c] = nefc; s.blk_start[nblk] = nefc; s.blk_dim[nblk] = dim; }
      SYNC();
      if (lane == 0) {
        if (dim > 1) {
          const float* f = s.cfri + 5 * c;
          float R0 = s.e_R[nefc];
          float R1 = R0 / fmaxf(1e-15f, FP(FO_opt, 6));
          s.e_R[nefc + 1] = R1;
          for (int k = 2; k < dim; k++) s.e_R[nefc + k] = R1 * f[0] * f[0] / fmaxf(1e-15f, f[k - 1] * f[k - 1]);
          s.cmu[c] = f[0] * sqrtf(R1 / R0);
        } else s.cmu[c] = 0.f;
      }
      nefc += dim;
      nblk++;
    }
    SYNC();
    for (int r = lane; r < nefc; r += 64) {
      s.e_D[r] = 1.0f / s.e_R[r];
      float v = 0;
      for (int k = 0; k < nv; k++) v += s.J[r * NV16 + k] * s.qvel[k];
      s.e_aref[r] = -s.e_B[r] * v - s.e_K[r] * s.e_imp[r] * (s.e_pos[r] - s.e_margin[r]);
    }
    if (lane == 0) { s.nefc = nefc; s.nblk = nblk; }
    SYNC();
  }

  // ---------------------------------------------------------------- actuation / smooth acceleration
  __device__ void actuation_acceleration() {
    const int nv = m.nv;
    float qs = 0.f;
    if (lane < nv) {
      int d = lane;
      float fa = 0.f;
      for (int a = 0; a < m.nu; a++) {
        int j = IT(IO_act_trnid, a);
        if (IT(IO_jnt_dofadr, j) != d) continue;
        float ctrl = s.ctrl[a], gear = FP(FO_act_gear, a);
        if (IT(IO_act_ctrllimited, a)) ctrl = fmaxf(FP(FO_act_ctrlrange, 2 * a), fminf(FP(FO_act_ctrlrange, 2 * a + 1), ctrl));
        float force = FP(FO_act_gainprm, 3 * a) * ctrl;
        if (IT(IO_act_biastype, a) == 1)
          force += FP(FO_act_biasprm, 3 * a) + FP(FO_act_biasprm, 3 * a + 1) * gear * s.qpos[IT(IO_jnt_qposadr, j)] + FP(FO_act_biasprm, 3 * a + 2) * gear * s.qvel[d];
        if (IT(IO_act_forcelimited, a)) force = fmaxf(FP(FO_act_forcerange, 2 * a), fminf(FP(FO_act_forcerange, 2 * a + 1), force));
        fa += gear * force;
      }
      s.qfrc_actuator[d] = fa;
      qs = s.qfrc_passive[d] - s.qfrc_bias[d] + fa;
      s.qfrc_smooth[d] = qs;
    }
    float as;
    {
      float lr[NV16], lt[NV16], linv[NV16];
      const int rr = lane & (NV16 - 1);
#pragma unroll
      for (int k = 0; k < NV16; k++) { lr[k] = s.L[rr * NVP + k]; lt[k] = s.L[k * NVP + rr]; linv[k] = s.invdiag[k]; }
      as = rchol_solve<NV16>(lr, lt, linv, lane < nv ? qs : 0.f, lane);
    }
    if (lane < nv) s.qacc_smooth[lane] = as;
    SYNC();
  }

  // ---------------------------------------------------------------- Newton solver (primal): lane r owns constraint row r
  // Row data (Jacobian row, D, R, aref) live in the owner lane's registers for the whole solve; the dense products
  // H = M + J^T W and J^T f run on the matrix cores (v_mfma_f32_16x16x4_f32, 4 rows per instruction); the Hessian
  // factorisation is the register-resident Cholesky above.  Algorithm = oracle solve_newton (MuJoCo's primal Newton).
  struct Row {
    float J[NV16];
    float D, R, aref, fl, mu, fr_own, Dm;
    float fj[CD - 1];
    int type, head, kk, dim;
    bool valid, ell;
  };
  __device__ __forceinline__ float row_dot(const Row& rw, float x) const {  // sum_k J[k] * x_k  (x_k lives in lane k)
    float sv = 0.f;
#pragma unroll
    for (int k = 0; k < NV16; k++) sv = fmaf(rw.J[k], bcast(x, k), sv);
    return sv;
  }
  // gather the block's friction-scaled values: out[j] = (x * fr_own) of lane head + j
  __device__ __forceinline__ void gather(const Row& rw, float x, float (&out)[CD]) const {
    float u = x * rw.fr_own;
#pragma unroll
    for (int j = 0; j < CD; j++) { float t = __shfl(u, rw.head + j); out[j] = (rw.ell && j < rw.dim) ? t : 0.f; }
  }
  // force / state / cost of this lane's row at residual jar (siblings of an elliptic block agree on the zone)
  __device__ __forceinline__ float row_update(const Row& rw, float x, float& force, int& state, float (&uj)[CD], float& T, float& g) const {
    float cost = 0.f;
    force = 0.f; state = ST_SATISFIED; T = 0.f; g = 0.f;
    gather(rw, x, uj);
    if (!rw.valid) return 0.f;
    if (rw.type == C_FRICTION_DOF) {
      if (x <= -rw.R * rw.fl) { state = ST_LINEARNEG; force = rw.fl; cost = rw.fl * (-0.5f * rw.R * rw.fl - x); }
      else if (x >= rw.R * rw.fl) { state = ST_LINEARPOS; force = -rw.fl; cost = rw.fl * (-0.5f * rw.R * rw.fl + x); }
      else { state = ST_QUADRATIC; force = -rw.D * x; cost = 0.5f * rw.D * x * x; }
    } else if (!rw.ell) {
      if (x < 0) { state = ST_QUADRATIC; force = -rw.D * x; cost = 0.5f * rw.D * x * x; }
    } else {
      float N = uj[0], T2 = 0.f;
#pragma unroll
      for (int j = 1; j < CD; j++) T2 = fmaf(uj[j], uj[j], T2);
      T = sqrtf(T2);
      const float mu = rw.mu;
      if (N >= mu * T || (T <= 0 && N >= 0)) {
      } else if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
        state = ST_QUADRATIC; force = -rw.D * x; cost = 0.5f * rw.D * x * x;
      } else {
        g = N - mu * T;
        float f0 = -rw.Dm * g * mu;
        state = ST_CONE;
        if (rw.kk == 0) { force = f0; cost = 0.5f * rw.Dm * g * g; }
        else force = -f0 / T * (x * rw.fr_own) * rw.fr_own;
      }
    }
    return cost;
  }
  // this lane's contribution to the cost and its first two derivatives along jar + alpha * jv
  __device__ __forceinline__ void row_ls(const Row& rw, float jar, float jv, const float (&g0)[CD], const float (&gv)[CD], float alpha, float& c, float& c1, float& c2) const {
    c = c1 = c2 = 0.f;
    if (!rw.valid) return;
    float x = fmaf(alpha, jv, jar), v = jv;
    if (rw.type == C_FRICTION_DOF) {
      if (x <= -rw.R * rw.fl) { c = rw.fl * (-0.5f * rw.R * rw.fl - x); c1 = -rw.fl * v; }
      else if (x >= rw.R * rw.fl) { c = rw.fl * (-0.5f * rw.R * rw.fl + x); c1 = rw.fl * v; }
      else { c = 0.5f * rw.D * x * x; c1 = rw.D * x * v; c2 = rw.D * v * v; }
    } else if (!rw.ell) {
      if (x < 0) { c = 0.5f * rw.D * x * x; c1 = rw.D * x * v; c2 = rw.D * v * v; }
    } else {
      const float mu = rw.mu;
      float N = fmaf(alpha, gv[0], g0[0]), T2 = 0.f, UV = 0.f, VV = 0.f;
#pragma unroll
      for (int j = 1; j < CD; j++) { float U = fmaf(alpha, gv[j], g0[j]); T2 = fmaf(U, U, T2); UV = fmaf(U, gv[j], UV); VV = fmaf(gv[j], gv[j], VV); }
      float T = sqrtf(T2);
      if (N >= mu * T || (T <= 0 && N >= 0)) {
      } else if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
        c = 0.5f * rw.D * x * x; c1 = rw.D * x * v; c2 = rw.D * v * v;
      } else if (rw.kk == 0) {
        float g = N - mu * T, iT = 1.0f / T;
        float g1 = gv[0] - mu * UV * iT, g2 = -mu * (VV * iT - UV * UV * iT * iT * iT);
        c = 0.5f * rw.Dm * g * g; c1 = rw.Dm * g * g1; c2 = rw.Dm * (g1 * g1 + g * g2);
      }
    }
  }
  // out_k (lane k < 16) = sum_r J[r][k] * f_r  on the matrix cores; f_r must already be in s.e_force[0..4*nch)
  __device__ __forceinline__ float jt_times_force(int nch) {
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < nch; c++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(s.J[64 * c + lane], s.e_force[4 * c + (lane >> 4)], acc, 0, 0, 0);
    if ((lane & 15) == 0) { float* o = s.scratch + 4 * (lane >> 4); o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = acc[3]; }
    SYNC();
    float r = s.scratch[lane & 15];
    SYNC();
    return r;
  }

  __device__ void solve_newton() {
    const int nv = m.nv, n = s.nefc;
    const int nch = (n + 3) >> 2;
    const float scale = 1.0f / (m.meaninertia * (nv > 1 ? nv : 1));
    const float tolerance = m.tolerance;
    // ---- per-lane row registers
    Row rw;
    rw.valid = lane < n;
    {
      const int r = rw.valid ? lane : 0;
#pragma unroll
      for (int k = 0; k < NV16; k++) rw.J[k] = rw.valid ? s.J[r * NV16 + k] : 0.f;
      rw.D = s.e_D[r]; rw.R = s.e_R[r]; rw.aref = rw.valid ? s.e_aref[r] : 0.f; rw.fl = s.e_fl[r]; rw.type = rw.valid ? s.e_type[r] : -1;
      rw.ell = rw.type == C_CONTACT_ELLIPTIC;
      const int c = rw.ell ? s.e_id[r] : 0;
      rw.head = rw.ell ? s.cefc[c] : lane; rw.kk = lane - rw.head; rw.dim = rw.ell ? s.cdim[c] : 1;
      rw.mu = s.cmu[c];
#pragma unroll
      for (int j = 0; j < CD - 1; j++) rw.fj[j] = s.cfri[5 * c + j];
      rw.fr_own = rw.kk == 0 ? rw.mu : s.cfri[5 * c + (rw.ell ? rw.kk - 1 : 0)];
      rw.Dm = s.e_D[rw.ell ? rw.head : r] / fmaxf(rw.mu * rw.mu * (1 + rw.mu * rw.mu), 1e-15f);
    }
    // M: row i in lane i (matrix-vector products) and in the MFMA accumulator layout (Hessian seed)
    float Mr[NV16];
#pragma unroll
    for (int k = 0; k < NV16; k++) Mr[k] = (lane < nv && k < nv) ? s.M[lane * NVP + k] : 0.f;
    v4f Macc;
#pragma unroll
    for (int v = 0; v < 4; v++) { int i = 4 * (lane >> 4) + v, j = lane & 15; Macc[v] = (i < nv && j < nv) ? s.M[i * NVP + j] : (i == j ? 1.f : 0.f); }
    const float a_sm = lane < nv ? s.qacc_smooth[lane] : 0.f, a_ws = lane < nv ? s.qacc_ws[lane] : 0.f, f_sm = lane < nv ? s.qfrc_smooth[lane] : 0.f;
    float force; int state; float uj[CD], T, g;
    // ---- warm start: previous acceleration unless the unconstrained one is cheaper
    float cost_sm = wave_sum(row_update(rw, row_dot(rw, a_sm) - rw.aref, force, state, uj, T, g));
    float cost_ws = wave_sum(row_update(rw, row_dot(rw, a_ws) - rw.aref, force, state, uj, T, g));
    {
      float dws = a_ws - a_sm, sv = 0.f;
#pragma unroll
      for (int k = 0; k < NV16; k++) sv = fmaf(Mr[k], bcast(dws, k), sv);
      cost_ws += wave_sum(0.5f * sv * dws);
    }
    float a = cost_ws < cost_sm ? a_ws : a_sm;
    int iter = 0;
    float jar = 0.f;
    for (;;) {
      jar = row_dot(rw, a) - rw.aref;
      float cost = wave_sum(row_update(rw, jar, force, state, uj, T, g));
      float ma = 0.f;
#pragma unroll
      for (int k = 0; k < NV16; k++) ma = fmaf(Mr[k], bcast(a, k), ma);
      const float gauss = wave_sum(0.5f * (ma - f_sm) * (a - a_sm));
      cost += gauss;
      s.e_force[lane] = force;
      SYNC();
      const float jf = jt_times_force(nch);
      float gk = lane < nv ? ma - f_sm - jf : 0.f;
      const float gn = wave_sum(gk * gk);
      if (iter >= m.iterations || scale * sqrtf(gn) < tolerance) break;
      // ---- Hessian weights W (row r): D J_r (quadratic), 0 (linear / satisfied), cone block Hc J_block
      {
        float w[NV16];
        const float dq = state == ST_QUADRATIC ? rw.D : 0.f;
#pragma unroll
        for (int k = 0; k < NV16; k++) w[k] = dq * rw.J[k];
        if (state == ST_CONE) {
          const float mu = rw.mu, iT = 1.0f / T;
          const float uo = jar * rw.fr_own;  // own friction-scaled residual U_kk
          const float grj = rw.kk == 0 ? mu : -mu * uo * rw.fr_own * iT;
#pragma unroll
          for (int k2 = 0; k2 < CD; k2++) {
            if (k2 < rw.dim) {
              const float frk = k2 == 0 ? mu : rw.fj[k2 > 0 ? k2 - 1 : 0];
              const float grk = k2 == 0 ? mu : -mu * uj[k2] * frk * iT;
              float h = grj * grk;
              if (rw.kk > 0 && k2 > 0) h += -g * mu * rw.fr_own * frk * ((rw.kk == k2 ? iT : 0.f) - uo * uj[k2] * iT * iT * iT);
              h *= rw.Dm;
              const float* Js = s.J + (rw.head + k2) * NV16;
#pragma unroll
              for (int k = 0; k < NV16; k++) w[k] = fmaf(h, Js[k], w[k]);
            }
          }
        }
#pragma unroll
        for (int k = 0; k < NV16; k++) s.W[lane * NV16 + k] = w[k];
      }
      SYNC();
      v4f acc = Macc;
      for (int c = 0; c < nch; c++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(s.W[64 * c + lane], s.J[64 * c + lane], acc, 0, 0, 0);
#pragma unroll
      for (int v = 0; v < 4; v++) s.H[(4 * (lane >> 4) + v) * NVP + (lane & 15)] = acc[v];
      SYNC();
      float sk;
      {
        float hr[NV16], hinv[NV16], ht[NV16];
        const int rr = lane & 15;
#pragma unroll
        for (int k = 0; k < NV16; k++) hr[k] = s.H[rr * NVP + k];
        rchol_factor<NV16>(hr, hinv);
        SYNC();
        if (lane < NV16) {
#pragma unroll
          for (int k = 0; k < NV16; k++) s.H[lane * NVP + k] = hr[k];
        }
        SYNC();
#pragma unroll
        for (int k = 0; k < NV16; k++) ht[k] = s.H[k * NVP + rr];
        sk = rchol_solve<NV16>(hr, ht, hinv, lane < nv ? -gk : 0.f, lane);
        if (lane >= nv) sk = 0.f;
      }
      // ---- line search along sk
      const float jv = row_dot(rw, sk);
      float mvv = 0.f;
#pragma unroll
      for (int k = 0; k < NV16; k++) mvv = fmaf(Mr[k], bcast(sk, k), mvv);
      const float q1 = wave_sum(sk * (ma - f_sm)), q2 = wave_sum(0.5f * sk * mvv), sn = sqrtf(wave_sum(sk * sk));
      if (sn < 1e-15f) break;
      float g0[CD], gvv[CD];
      gather(rw, jar, g0);
      gather(rw, jv, gvv);
      const float gtol = tolerance * 0.01f * sn / scale;
      float p0, d0, h0, p, dp, hp, lo = 0.f, hi = -1.f, alpha;
      {
        float c, c1, c2;
        row_ls(rw, jar, jv, g0, gvv, 0.f, c, c1, c2);
        p0 = gauss + wave_sum(c); d0 = q1 + wave_sum(c1); h0 = 2 * q2 + wave_sum(c2);
      }
      if (d0 >= 0 || h0 <= 0) break;
      alpha = -d0 / h0;
      // fp32 line search: stop when the directional derivative has dropped below MuJoCo's gtol, by 1e6 relative to its
      // start value (single-precision noise floor), or when the safeguarded Newton update no longer moves alpha
      const float dtol = fmaxf(gtol, 1e-6f * fabsf(d0));
      for (int ls = 0; ls < m.ls_iterations; ls++) {
        pf.count(RP_N_LS, 1);
        float c, c1, c2;
        row_ls(rw, jar, jv, g0, gvv, alpha, c, c1, c2);
        p = gauss + alpha * q1 + alpha * alpha * q2 + wave_sum(c);
        dp = q1 + 2 * alpha * q2 + wave_sum(c1);
        hp = 2 * q2 + wave_sum(c2);
        if (fabsf(dp) < dtol) break;
        if (dp < 0) lo = alpha; else hi = alpha;
        float next = hp > 0 ? alpha - dp / hp : -1.f;
        if (hi < 0) { if (next <= lo) next = 2 * alpha + 1e-12f; }
        else if (next <= lo || next >= hi) next = 0.5f * (lo + hi);
        if (fabsf(next - alpha) <= 1e-6f * fabsf(alpha)) { alpha = next; break; }
        alpha = next;
      }
      {
        float c, c1, c2;
        row_ls(rw, jar, jv, g0, gvv, alpha, c, c1, c2);
        p = gauss + alpha * q1 + alpha * alpha * q2 + wave_sum(c);
      }
      if (!(p < p0)) break;
      a = fmaf(alpha, sk, a);
      iter++;
      if (scale * (p0 - p) < tolerance) {
        jar = row_dot(rw, a) - rw.aref;
        row_update(rw, jar, force, state, uj, T, g);
        break;
      }
    }
    s.e_force[lane] = force;
    SYNC();
    const float fc = jt_times_force(nch);
    if (lane < nv) { s.qfrc_constraint[lane] = fc; s.qacc[lane] = a; }
    if (lane == 0) s.niter = iter;
    pf.count(RP_N_NEWTON, iter);
    SYNC();
  }

  __device__ void fwd_constraint() {
    if (s.nefc == 0) {
      if (lane < m.nv) { s.qacc[lane] = s.qacc_smooth[lane]; s.qfrc_constraint[lane] = 0.f; }
      if (lane == 0) s.niter = 0;
      SYNC();
      return;
    }
    solve_newton();
  }

  // ---------------------------------------------------------------- semi-implicit Euler with implicit joint damping
  __device__ void euler() {
    const int nv = m.nv;
    const float h = FP(FO_opt, 0);
    float qa;
    {
      float hr[NV16], hinv[NV16], ht[NV16];
      const int rr = lane & (NV16 - 1);
      const float hd = rr < nv ? h * FP(FO_dof_damping, rr) : 0.f;
#pragma unroll
      for (int k = 0; k < NV16; k++) hr[k] = (rr < nv && k < nv) ? s.M[rr * NVP + k] + (rr == k ? hd : 0.f) : (rr == k ? 1.f : 0.f);
      rchol_factor<NV16>(hr, hinv);
      if (lane < NV16) {
#pragma unroll
        for (int k = 0; k < NV16; k++) s.H[lane * NVP + k] = hr[k];
      }
      SYNC();
#pragma unroll
      for (int k = 0; k < NV16; k++) ht[k] = s.H[k * NVP + rr];
      qa = rchol_solve<NV16>(hr, ht, hinv, lane < nv ? s.qfrc_smooth[lane] + s.qfrc_constraint[lane] : 0.f, lane);
    }
    if (lane < nv) { s.qvel[lane] += h * qa; s.qacc_ws[lane] = s.qacc[lane]; }
    SYNC();
    if (lane < m.njnt) {
      int j = lane, pa = IT(IO_jnt_qposadr, j), da = IT(IO_jnt_dofadr, j), t = IT(IO_jnt_type, j);
      if (t == JNT_FREE || t == JNT_BALL) {
        if (t == JNT_FREE) { for (int k = 0; k < 3; k++) s.qpos[pa + k] += h * s.qvel[da + k]; pa += 3; da += 3; }
        V3 w = ld3(s.qvel + da);
        float ang = norm(w) * h;
        if (ang > 1e-15f) stq(s.qpos + pa, qnorm(qmul(ldq(s.qpos + pa), axisangle(normalized(w), ang))));
      } else s.qpos[pa] += h * s.qvel[da];
    }
    SYNC();
  }

  // ---------------------------------------------------------------- built-in controller: OSC_POSE + GRIP
  __device__ void ctrl_set_goal(const float* action) {
    const DCtrl& c = m.ctrl;
    float sc[6];
    for (int i = 0; i < 6; i++) {
      float scale = fabsf(c.out_max[i] - c.out_min[i]) / fabsf(c.in_max[i] - c.in_min[i]);
      float a = fmaxf(c.in_min[i], fminf(c.in_max[i], action[i]));
      sc[i] = (a - 0.5f * (c.in_max[i] + c.in_min[i])) * scale + 0.5f * (c.out_max[i] + c.out_min[i]);
    }
    V3 op = ld3(s.spos + 3 * c.base_site), ep = ld3(s.spos + 3 * c.eef_site);
    M3 oR = ldm(s.smat + 9 * c.base_site), eR = ldm(s.smat + 9 * c.eef_site);
    V3 gp = mtv(oR, ep - op) + v3(sc[0], sc[1], sc[2]);
    V3 d = v3(sc[3], sc[4], sc[5]);
    float ang = norm(d);
    Q4 qe = {1, 0, 0, 0};
    if (ang != 0.f) { float sn = sinf(0.5f * ang) / ang; qe.w = cosf(0.5f * ang); qe.x = d.x * sn; qe.y = d.y * sn; qe.z = d.z * sn; }
    // reference quat2mat: q *= sqrt(2/n); R = I - ... (transform_utils.py:461-487)
    float n = qe.w * qe.w + qe.x * qe.x + qe.y * qe.y + qe.z * qe.z, sq = sqrtf(2.0f / n);
    float q[4] = {qe.w * sq, qe.x * sq, qe.y * sq, qe.z * sq};
    M3 Re;
    Re.m[0] = 1.0f - q[2] * q[2] - q[3] * q[3]; Re.m[1] = q[1] * q[2] - q[3] * q[0]; Re.m[2] = q[1] * q[3] + q[2] * q[0];
    Re.m[3] = q[1] * q[2] + q[3] * q[0]; Re.m[4] = 1.0f - q[1] * q[1] - q[3] * q[3]; Re.m[5] = q[2] * q[3] - q[1] * q[0];
    Re.m[6] = q[1] * q[3] - q[2] * q[0]; Re.m[7] = q[2] * q[3] + q[1] * q[0]; Re.m[8] = 1.0f - q[1] * q[1] - q[2] * q[2];
    M3 go = mm(Re, mtm(oR, eR));
    SYNC();
    if (lane == 0) {
      st3(s.cstate + RSIM_CS_GOALPOS, gp);
      stm(s.cstate + RSIM_CS_GOALORI, go);
      if (c.ngrip > 0) {
        float a = action[6], sg = a > 0 ? 1.f : (a < 0 ? -1.f : 0.f);
        for (int i = 0; i < c.ngrip; i++) s.cstate[RSIM_CS_GRIP + i] = fmaxf(-1.f, fminf(1.f, s.cstate[RSIM_CS_GRIP + i] + c.grip_sign[i] * c.grip_speed * sg));
      }
    }
    SYNC();
  }

  // reset_goal + initial_joint capture (Controller.__init__ / OSC.reset_goal)
  __device__ void ctrl_reset() {
    const DCtrl& c = m.ctrl;
    if (lane < c.ndof) s.cstate[RSIM_CS_Q0 + lane] = s.qpos[c.qpos_idx[lane]];
    if (lane == 0) {
      st3(s.cstate + RSIM_CS_GOALPOS, ld3(s.spos + 3 * c.eef_site));
      for (int k = 0; k < 9; k++) s.cstate[RSIM_CS_GOALORI + k] = s.smat[9 * c.eef_site + k];
      for (int i = 0; i < RSIM_GRIP_MAX; i++) s.cstate[RSIM_CS_GRIP + i] = 0.f;
    }
    SYNC();
  }

  __device__ void ctrl_run() {
    const DCtrl& c = m.ctrl;
    const int nv = m.nv, n = c.ndof;
    constexpr int NA = RSIM_ARM_MAX;
    float* Jm = s.scratch;            // 6 x NA   arm Jacobian
    float* X = s.scratch + 6 * NA;    // NA x 6   Minv J^T  (stored [i*6 + r])
    float* vel = s.scratch + 12 * NA; // 12: eef vel(6), base vel(6)
    float* Ma = s.H;                  // arm mass sub-block (NVP stride), factor into s.Lh
    int eb = IT(IO_site_bodyid, c.eef_site), bb = IT(IO_site_bodyid, c.base_site);
    V3 ep = ld3(s.spos + 3 * c.eef_site), op = ld3(s.spos + 3 * c.base_site);
    if (lane < 6 * n) {
      int r = lane / n, i = lane - r * n;
      S6 jc = jac_col(eb, ep, c.dof_idx[i]);
      float v = r < 3 ? (r == 0 ? jc.l.x : (r == 1 ? jc.l.y : jc.l.z)) : (r == 3 ? jc.a.x : (r == 4 ? jc.a.y : jc.a.z));
      Jm[r * NA + i] = v;
    }
    if (lane < 12) {
      int which = lane / 6, r = lane - 6 * which, b = which ? bb : eb;
      V3 p = which ? op : ep;
      float v = 0.f;
      u64 mk = mask2(IO_body_dofmask, b);
      while (mk) {
        int k = __ffsll((long long)mk) - 1;
        mk &= mk - 1;
        S6 jc = jac_col(b, p, k);
        float jv = r < 3 ? (r == 0 ? jc.l.x : (r == 1 ? jc.l.y : jc.l.z)) : (r == 3 ? jc.a.x : (r == 4 ? jc.a.y : jc.a.z));
        v += jv * s.qvel[k];
      }
      vel[lane] = v;
    }
    for (int e = lane; e < n * n; e += 64) { int i = e / n, j = e - i * n; Ma[i * NVP + j] = s.M[c.dof_idx[i] * NVP + c.dof_idx[j]]; }
    SYNC();
    chol_factor<NVP>(s.Lh, s.invdiag_h, Ma, n, lane);
    // X = Minv J^T : lane group r (6 groups of 8 lanes) solves column r
    {
      int r = lane >> 3, i = lane & 7;
      float x = (r < 6 && i < n) ? Jm[r * NA + i] : 0.f;
      // forward / backward substitution inside 8-lane groups
      for (int k = 0; k < n; k++) {
        float xk = __shfl(x, (lane & ~7) + k) * s.invdiag_h[k];
        if (i == k) x = xk; else if (i > k && i < n) x -= s.Lh[i * NVP + k] * xk;
      }
      for (int k = n - 1; k >= 0; k--) {
        float xk = __shfl(x, (lane & ~7) + k) * s.invdiag_h[k];
        if (i == k) x = xk; else if (i < k) x -= s.Lh[k * NVP + i] * xk;
      }
      if (r < 6 && i < n) X[i * 6 + r] = x;
    }
    SYNC();
    // everything below is tiny dense algebra evaluated uniformly by all lanes
    float lfi[36];
    for (int r = 0; r < 6; r++)
      for (int q = 0; q < 6; q++) { float sv = 0; for (int k = 0; k < n; k++) sv += Jm[r * NA + k] * X[k * 6 + q]; lfi[r * 6 + q] = sv; }
    M3 oR = ldm(s.smat + 9 * c.base_site), eR = ldm(s.smat + 9 * c.eef_site);
    V3 gpos = ld3(s.cstate + RSIM_CS_GOALPOS);
    M3 gori = ldm(s.cstate + RSIM_CS_GOALORI);
    V3 perr = op + mv(oR, gpos) - ep;
    M3 dori = mm(oR, gori);
    V3 oerr = (cross(col(eR, 0), col(dori, 0)) + cross(col(eR, 1), col(dori, 1)) + cross(col(eR, 2), col(dori, 2))) * 0.5f;
    float F[3], T[3];
    float pe[3] = {perr.x, perr.y, perr.z}, oe[3] = {oerr.x, oerr.y, oerr.z};
    for (int k = 0; k < 3; k++) {
      F[k] = pe[k] * c.kp[k] - (vel[k] - vel[6 + k]) * c.kd[k];
      T[k] = oe[k] * c.kp[3 + k] - (vel[3 + k] - vel[9 + k]) * c.kd[3 + k];
    }
    float wrench[6];
    if (c.uncouple) {
      float lp[9], lo[9];
      for (int r = 0; r < 3; r++) for (int q = 0; q < 3; q++) { lp[r * 3 + q] = lfi[r * 6 + q]; lo[r * 3 + q] = lfi[(3 + r) * 6 + 3 + q]; }
      spd_solve_small<3>(lp, F, wrench);
      spd_solve_small<3>(lo, T, wrench + 3);
    } else {
      float w[6] = {F[0], F[1], F[2], T[0], T[1], T[2]};
      spd_solve_small<6>(lfi, w, wrench);
    }
    // nullspace: N^T M tmp = M tmp - J^T Lambda_full (J tmp)
    float kv = sqrtf(c.nullspace_kp) * 2, tmp[NA], jt[6], z[6];
    for (int i = 0; i < n; i++) tmp[i] = c.nullspace_kp * (s.cstate[RSIM_CS_Q0 + i] - s.qpos[c.qpos_idx[i]]) - kv * s.qvel[c.dof_idx[i]];
    for (int r = 0; r < 6; r++) { float sv = 0; for (int k = 0; k < n; k++) sv += Jm[r * NA + k] * tmp[k]; jt[r] = sv; }
    spd_solve_small<6>(lfi, jt, z);
    if (lane < n) {
      int i = lane;
      float tq = s.qfrc_bias[c.dof_idx[i]];
      for (int r = 0; r < 6; r++) tq += Jm[r * NA + i] * (wrench[r] - z[r]);
      for (int k = 0; k < n; k++) tq += Ma[i * NVP + k] * tmp[k];
      s.cstate[RSIM_CS_TAU + i] = tq;
      int a = c.act_idx[i];
      s.ctrl[a] = fmaxf(FP(FO_act_ctrlrange, 2 * a), fminf(FP(FO_act_ctrlrange, 2 * a + 1), tq));
    }
    if (lane < c.ngrip) {
      int a = c.grip_act[lane];
      float lo_ = FP(FO_act_ctrlrange, 2 * a), hi_ = FP(FO_act_ctrlrange, 2 * a + 1);
      s.ctrl[a] = fmaxf(lo_, fminf(hi_, 0.5f * (hi_ + lo_) + 0.5f * (hi_ - lo_) * s.cstate[RSIM_CS_GRIP + lane]));
    }
    SYNC();
  }
};

// ------------------------------------------------------------------------------------------------------------
// the step kernel
// ------------------------------------------------------------------------------------------------------------
template <int NB, int NJ, int NV, int NG, int NS, int NCON, int NEFC, int NPAIR>
__global__ __launch_bounds__(64) void k_step(DModel m, DBatch b, const float* __restrict__ actions, int n_sub, int flags) {
  typedef Smem<NB, NJ, NV, NG, NS, NCON, NEFC, NPAIR> SM;
  __shared__ SM s;
  const int env = blockIdx.x, lane = threadIdx.x;
  if (env >= b.B) return;
  const float* fp = m.ft + (size_t)env * m.fstride;
  Sim<SM> sim(s, m, fp, lane, b.prof);
  sim.pf.start();
  for (int i = lane; i < m.nit; i += 64) s.tab_i[i] = m.it[i];
  for (int i = lane; i < m.nft; i += 64) s.tab_f[i] = fp[i];
  // ---- load state
  for (int i = lane; i < m.nq; i += 64) s.qpos[i] = b.qpos[(size_t)env * m.nq + i];
  for (int i = lane; i < m.nv; i += 64) { s.qvel[i] = b.qvel[(size_t)env * m.nv + i]; s.qacc_ws[i] = b.qacc_ws[(size_t)env * m.nv + i]; }
  for (int i = lane; i < m.nu; i += 64) s.ctrl[i] = b.ctrl[(size_t)env * m.nu + i];
  if (lane < RSIM_CS_SIZE) s.cstate[lane] = b.cstate[(size_t)env * RSIM_CS_SIZE + lane];
  if (lane == 0) { s.ncon = 0; s.nefc = 0; s.nblk = 0; s.niter = 0; }
  SYNC();
  const float* act = actions ? actions + (size_t)env * m.ctrl.action_dim : nullptr;
  float time = b.time[env];
  sim.pf.mark(RP_LOAD);
  for (int sub = 0; sub < n_sub; sub++) {
    sim.kinematics();
    sim.pf.mark(RP_KIN);
    sim.com_pos();
    sim.pf.mark(RP_COM);
    sim.crb();
    sim.pf.mark(RP_CRB);
    sim.collision();
    sim.pf.mark(RP_NARROW);
    sim.make_constraint();
    sim.pf.mark(RP_MAKEC);
    sim.velocity();
    sim.pf.mark(RP_VEL);
    if (flags & RF_CTRL) {
      if ((flags & RF_SETGOAL) && sub == 0 && act) sim.ctrl_set_goal(act);
      sim.ctrl_run();
      sim.pf.mark(RP_CTRL);
    }
    if (flags & RF_ACTSOLVE) {
      sim.actuation_acceleration();
      sim.pf.mark(RP_ACT);
      sim.fwd_constraint();
      sim.pf.mark(RP_SOLVE);
      sim.pf.count(RP_N_CON, s.ncon);
      sim.pf.count(RP_N_EFC, s.nefc);
    }
    if (flags & RF_INTEGRATE) {
      sim.euler();
      sim.pf.mark(RP_EULER);
      time += s.tab_f[m.fo[FO_opt]];
    }
    sim.pf.count(RP_N_SUB, 1);
  }
  // ---- store state
  for (int i = lane; i < m.nq; i += 64) b.qpos[(size_t)env * m.nq + i] = s.qpos[i];
  for (int i = lane; i < m.nv; i += 64) { b.qvel[(size_t)env * m.nv + i] = s.qvel[i]; b.qacc_ws[(size_t)env * m.nv + i] = s.qacc_ws[i]; }
  for (int i = lane; i < m.nu; i += 64) b.ctrl[(size_t)env * m.nu + i] = s.ctrl[i];
  if (lane < RSIM_CS_SIZE) b.cstate[(size_t)env * RSIM_CS_SIZE + lane] = s.cstate[lane];
  if (lane == 0) b.time[env] = time;
  if (flags & RF_DEBUG) {
    const int nb = m.nbody, nv = m.nv;
    for (int i = lane; i < nb * 3; i += 64) { b.xpos[(size_t)env * nb * 3 + i] = s.xpos[i]; b.rootcom[(size_t)env * nb * 3 + i] = s.rootcom[i]; }
    for (int i = lane; i < nb * 4; i += 64) b.xquat[(size_t)env * nb * 4 + i] = s.xquat[i];
    for (int e = lane; e < nv * nv; e += 64) { int i = e / nv, j = e - i * nv; b.qM[(size_t)env * nv * nv + e] = s.M[i * SM::NVP + j]; }
    for (int i = lane; i < nv * 6; i += 64) b.cdof[(size_t)env * nv * 6 + i] = s.cdof[i];
    for (int i = lane; i < nv; i += 64) {
      size_t o = (size_t)env * nv + i;
      b.qfrc_bias[o] = s.qfrc_bias[i]; b.qfrc_passive[o] = s.qfrc_passive[i];
      if (flags & RF_ACTSOLVE) { b.qfrc_actuator[o] = s.qfrc_actuator[i]; b.qfrc_constraint[o] = s.qfrc_constraint[i]; b.qacc[o] = s.qacc[i]; }
    }
    int ncon = s.ncon;
    for (int c = lane; c < ncon; c += 64) {
      float* r = b.contact + ((size_t)env * NCON + c) * RSIM_CON_REC;
      r[0] = s.cdist[c];
      for (int k = 0; k < 3; k++) r[1 + k] = s.cpos[3 * c + k];
      for (int k = 0; k < 9; k++) r[4 + k] = s.cframe[9 * c + k];
      r[13] = (float)IT(IO_cg_geomid, s.cg1[c]); r[14] = (float)IT(IO_cg_geomid, s.cg2[c]); r[15] = (float)s.cdim[c]; r[16] = (float)s.cefc[c];
      r[17] = ((flags & RF_ACTSOLVE) && s.cefc[c] >= 0) ? s.e_force[s.cefc[c]] : 0.f;
      for (int k = 0; k < 5; k++) r[18 + k] = s.cfri[5 * c + k];
    }
    if (flags & RF_ACTSOLVE)
      for (int i = lane; i < s.nefc; i += 64) b.efc_force[(size_t)env * NEFC + i] = s.e_force[i];
    if (lane == 0) { b.ncon[env] = ncon; b.nefc[env] = s.nefc; b.niter[env] = s.niter; }
  }
}

// controller reset kernel: forward kinematics then OSC.reset_goal / initial_joint capture
template <int NB, int NJ, int NV, int NG, int NS, int NCON, int NEFC, int NPAIR>
__global__ __launch_bounds__(64) void k_ctrl_reset(DModel m, DBatch b, const unsigned char* __restrict__ mask) {
  typedef Smem<NB, NJ, NV, NG, NS, NCON, NEFC, NPAIR> SM;
  __shared__ SM s;
  const int env = blockIdx.x, lane = threadIdx.x;
  if (env >= b.B) return;
  if (mask && !mask[env]) return;
  const float* fp = m.ft + (size_t)env * m.fstride;
  Sim<SM> sim(s, m, fp, lane, nullptr);
  for (int i = lane; i < m.nit; i += 64) s.tab_i[i] = m.it[i];
  for (int i = lane; i < m.nft; i += 64) s.tab_f[i] = fp[i];
  for (int i = lane; i < m.nq; i += 64) s.qpos[i] = b.qpos[(size_t)env * m.nq + i];
  if (lane < RSIM_CS_SIZE) s.cstate[lane] = 0.f;
  SYNC();
  sim.kinematics();
  sim.ctrl_reset();
  if (lane < RSIM_CS_SIZE) b.cstate[(size_t)env * RSIM_CS_SIZE + lane] = s.cstate[lane];
}

// ------------------------------------------------------------------------------------------------------------
// standalone batched OSC torque law on explicit inputs (unit-test entry: parity against the reference's own
// OperationalSpaceController.run_controller).  One wave per sample.
// in: [B, 128] floats: ep3 eR9 ev6 op3 oR9 bv6 goal_pos3 goal_ori9 J(6x7) M(7x7) bias7 q7 qd7 q0 7 (=168?) -> packed by host
// ------------------------------------------------------------------------------------------------------------
#define OSC_IN 192
__global__ __launch_bounds__(64) void k_osc_eval(DCtrl c, const float* __restrict__ in, float* __restrict__ out, int B) {
  const int env = blockIdx.x, lane = threadIdx.x;
  if (env >= B) return;
  __shared__ float sh[OSC_IN];
  __shared__ float Lm[8 * 9], invd[8], X[8 * 6];
  for (int i = lane; i < OSC_IN; i += 64) sh[i] = in[(size_t)env * OSC_IN + i];
  SYNC();
  const int n = c.ndof;
  const float *ep = sh, *eR = sh + 3, *ev = sh + 12, *op = sh + 18, *oR = sh + 21, *bv = sh + 30, *gp = sh + 36, *go = sh + 39, *J = sh + 48,
              *M = sh + 48 + 6 * 8, *bias = M + 64, *q = bias + 8, *qd = q + 8, *q0 = qd + 8;
  // Cholesky of the arm mass block (stride 8 in, stride 9 factor)
  float* A9 = Lm;
  __shared__ float Ain[8 * 9];
  for (int e = lane; e < n * n; e += 64) { int i = e / n, j = e - i * n; Ain[i * 9 + j] = M[i * 8 + j]; }
  SYNC();
  chol_factor<9>(A9, invd, Ain, n, lane);
  {
    int r = lane >> 3, i = lane & 7;
    float x = (r < 6 && i < n) ? J[r * 8 + i] : 0.f;
    for (int k = 0; k < n; k++) { float xk = __shfl(x, (lane & ~7) + k) * invd[k]; if (i == k) x = xk; else if (i > k && i < n) x -= A9[i * 9 + k] * xk; }
    for (int k = n - 1; k >= 0; k--) { float xk = __shfl(x, (lane & ~7) + k) * invd[k]; if (i == k) x = xk; else if (i < k) x -= A9[k * 9 + i] * xk; }
    if (r < 6 && i < n) X[i * 6 + r] = x;
  }
  SYNC();
  float lfi[36];
  for (int r = 0; r < 6; r++) for (int p = 0; p < 6; p++) { float sv = 0; for (int k = 0; k < n; k++) sv += J[r * 8 + k] * X[k * 6 + p]; lfi[r * 6 + p] = sv; }
  M3 oRm = ldm(oR), eRm = ldm(eR), gom = ldm(go);
  V3 perr = ld3(op) + mv(oRm, ld3(gp)) - ld3(ep);
  M3 dori = mm(oRm, gom);
  V3 oerr = (cross(col(eRm, 0), col(dori, 0)) + cross(col(eRm, 1), col(dori, 1)) + cross(col(eRm, 2), col(dori, 2))) * 0.5f;
  float pe[3] = {perr.x, perr.y, perr.z}, oe[3] = {oerr.x, oerr.y, oerr.z}, F[3], T[3];
  for (int k = 0; k < 3; k++) { F[k] = pe[k] * c.kp[k] - (ev[k] - bv[k]) * c.kd[k]; T[k] = oe[k] * c.kp[3 + k] - (ev[3 + k] - bv[3 + k]) * c.kd[3 + k]; }
  float wrench[6];
  if (c.uncouple) {
    float lp[9], lo[9];
    for (int r = 0; r < 3; r++) for (int p = 0; p < 3; p++) { lp[r * 3 + p] = lfi[r * 6 + p]; lo[r * 3 + p] = lfi[(3 + r) * 6 + 3 + p]; }
    spd_solve_small<3>(lp, F, wrench);
    spd_solve_small<3>(lo, T, wrench + 3);
  } else {
    float w[6] = {F[0], F[1], F[2], T[0], T[1], T[2]};
    spd_solve_small<6>(lfi, w, wrench);
  }
  float kv = sqrtf(c.nullspace_kp) * 2, tmp[8], jt[6], z[6];
  for (int i = 0; i < n; i++) tmp[i] = c.nullspace_kp * (q0[i] - q[i]) - kv * qd[i];
  for (int r = 0; r < 6; r++) { float sv = 0; for (int k = 0; k < n; k++) sv += J[r * 8 + k] * tmp[k]; jt[r] = sv; }
  spd_solve_small<6>(lfi, jt, z);
  if (lane < n) {
    float tq = bias[lane];
    for (int r = 0; r < 6; r++) tq += J[r * 8 + lane] * (wrench[r] - z[r]);
    for (int k = 0; k < n; k++) tq += M[lane * 8 + k] * tmp[k];
    out[(size_t)env * 8 + lane] = tq;
  }
}

// explicit instantiations + launchers ------------------------------------------------------------------------
#define RSIM_INST(NB, NJ, NV, NG, NS, NCON, NEFC, NPAIR)                                                                          \
  template __global__ void k_step<NB, NJ, NV, NG, NS, NCON, NEFC, NPAIR>(DModel, DBatch, const float*, int, int);                 \
  template __global__ void k_ctrl_reset<NB, NJ, NV, NG, NS, NCON, NEFC, NPAIR>(DModel, DBatch, const unsigned char*);

RSIM_INST(32, 16, 16, 24, 16, 16, 64, 192)

extern "C" int rsim_launch_step_cfg0(const DModel* m, const DBatch* b, const float* actions, int n_sub, int flags, hipStream_t stream) {
  hipLaunchKernelGGL((k_step<32, 16, 16, 24, 16, 16, 64, 192>), dim3(b->B), dim3(64), 0, stream, *m, *b, actions, n_sub, flags);
  return (int)hipGetLastError();
}
extern "C" int rsim_launch_ctrl_reset_cfg0(const DModel* m, const DBatch* b, const unsigned char* mask, hipStream_t stream) {
  hipLaunchKernelGGL((k_ctrl_reset<32, 16, 16, 24, 16, 16, 64, 192>), dim3(b->B), dim3(64), 0, stream, *m, *b, mask);
  return (int)hipGetLastError();
}
extern "C" int rsim_launch_osc_eval(const DCtrl* c, const float* in, float* out, int B, hipStream_t stream) {
  hipLaunchKernelGGL(k_osc_eval, dim3(B), dim3(64), 0, stream, *c, in, out, B);
  return (int)hipGetLastError();
}
extern "C" int rsim_cfg0_limits(int* lim) {
  lim[0] = 32; lim[1] = 16; lim[2] = 16; lim[3] = 24; lim[4] = 16; lim[5] = 16; lim[6] = 64; lim[7] = 192;
  return 0;
}
