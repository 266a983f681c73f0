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
#include <type_traits>

#include "../../include/rsim.h"
#include "rsim_internal.h"

// Eight builds of this file (0-4 serve models, 5-7 are the capacity tiers above 3, 0 and 1): RSIM_CFG 0 = 32 bodies x 16 dofs (Lift/Panda; tree products as incidence-matrix MFMAs with compile-time bit
// fields, every dense nv x nv product on one 16x16 MFMA tile, register-resident Cholesky), 1 = 32 x 32 (Stack/Panda: two free cubes),
// 2 = 64 x 16 (Baxter), 3 = 64 x 48 (PickPlace / IIWA + Robotiq140), 4 = 64 x 64.  The larger builds keep the lane roles, the collision
// pipeline, the constraint rows and the Newton algorithm; beyond 32 x 16 the tree products use per-lane 64-bit incidence words
// (incidence_mfma), beyond 16 dofs the dense products run tile by tile with the factorisations on the LDS matrix.  `sm` is one file-scope
// LDS object, so each configuration is its own translation unit.
#ifndef RSIM_CFG
#define RSIM_CFG 0
#endif
#if RSIM_CFG == 0
#define RSIM_DIMS 32, 16, 16, 24, 16, 16, 64, 192
#define RSIM_SYM(x) x##_cfg0
#ifdef RSIM_FUSED_TIER
// The capacity tier above this configuration (32 contacts x 128 rows: the dimensions of configuration 6) compiled INTO this configuration's control-step kernel:
// an env that outgrows 16 contacts / 64 rows in mid-step carries on with the wide body from the substep it is in, inside the same workgroup (k_step below), instead
// of being redone by another kernel after the launch.  The wide body keeps its Jacobian and contact block in the per-env global buffer so that both bodies
// fit the same 20 KB of LDS (eight envs per CU).
#define RSIM_DIMS_W 32, 16, 16, 24, 16, 32, 128, 192
#endif
#elif RSIM_CFG == 1
#define RSIM_DIMS 32, 16, 32, 24, 16, 32, 64, 192
#define RSIM_SYM(x) x##_cfg1
#ifdef RSIM_FUSED_TIER   /* the Stack-class tier (32 contacts x 128 rows: the dimensions of configuration 7) as a second body of this configuration's kernel, as for configuration 0 */
#define RSIM_DIMS_W 32, 16, 32, 24, 16, 32, 128, 192
#endif
#elif RSIM_CFG == 2  // 64 bodies x 16 dofs (Baxter: 36 bodies, 29 colliding geoms, 17 sites, 299 candidate pairs): tree products as mask loops, dense ones on one tile
#define RSIM_DIMS 64, 16, 16, 32, 32, 32, 64, 320
#define RSIM_SYM(x) x##_cfg2
#ifdef RSIM_FUSED_TIER   /* the tier above this configuration as a second body of its kernel: the same lane roles with 128 rows (before: the 64 x 48 build's kernel served as its tier) */
#define RSIM_DIMS_W 64, 16, 16, 32, 32, 32, 128, 320
#endif
#elif RSIM_CFG == 3  // 64 bodies x 48 dofs x 128 constraint rows (PickPlace / IIWA + Robotiq140: 36 bodies, 37 dofs, 41 colliding geoms, 622 candidate
       // pairs, tendon rows; a closed Robotiq gripper alone holds ~30 rows of self-contact).  Three 16-dof tiles instead of four: the dense matrices
       // (M, H, J) shrink to 75 KB of LDS per environment = TWO environments per CU
#define RSIM_DIMS 64, 32, 48, 64, 32, 32, 128, 640
#define RSIM_SYM(x) x##_cfg3
#elif RSIM_CFG == 4  // 64 bodies x 64 dofs x 128 constraint rows: the widest configuration (one environment per CU)
#define RSIM_DIMS 64, 32, 64, 64, 32, 32, 128, 640
#define RSIM_SYM(x) x##_cfg4
#elif RSIM_CFG == 6  // capacity tier above configuration 0 (Lift class: 32 bodies x 16 dofs): 32 contacts x 128 rows, the same one-tile algebra with two rows per lane
#define RSIM_DIMS 32, 16, 16, 24, 16, 32, 128, 192
#define RSIM_SYM(x) x##_cfg6
#elif RSIM_CFG == 7  // capacity tier above configuration 1 (Stack class: 32 bodies x 32 dofs): 32 contacts x 128 rows
#define RSIM_DIMS 32, 16, 32, 24, 16, 32, 128, 192
#define RSIM_SYM(x) x##_cfg7
#else  // 64 bodies x 48 dofs with 64 contacts x 256 constraint rows (four per lane), one environment per CU: the capacity tier ABOVE configuration 3.  No
       // model is assigned to it; the PickPlace envs whose substep asks for more than 32 contacts / 128 rows (a handful of 8192 at any time: objects
       // wedged between fingers, bin walls and each other) are stepped by it for as long as they do (rsim_api.cpp launch(): capacity tiers)
#define RSIM_DIMS 64, 32, 48, 64, 32, 64, 256, 640
#define RSIM_SYM(x) x##_cfg5
#endif

#ifndef RSIM_MINWAVES
#define RSIM_MINWAVES 1  /* waves per SIMD the register allocator must leave room for (1: 512 registers, 2: 256) */
#endif

typedef unsigned long long u64;
#ifdef RSIM_SUBPROF   /* profiling build (tools/subprof.sh): sub-phase marks inside the solver and the OSC controller -> slots x0..x9 */
#define SUBMARK(id) pf.mark(id)
#else
#define SUBMARK(id)
#endif
#if defined(RSIM_SUBPROF) && RSIM_SUBPROF == 2   /* tools/subprof.sh 2: slots x7..x9 split the Hessian step (assembly | factorisation | solve) instead of the OSC controller */
#define SUBMARK_OSC(id)
#define SUBMARK_H(id) pf.mark(id)
#else
#define SUBMARK_OSC(id) SUBMARK(id)
#define SUBMARK_H(id)
#endif
#if defined(RSIM_SUBPROF) && RSIM_SUBPROF == 3   /* tools/subprof.sh 3: slots x0..x7 split the CRB and velocity stages (wide configurations) instead of the solver */
#undef SUBMARK
#define SUBMARK(id)
#undef SUBMARK_OSC
#define SUBMARK_OSC(id)
#define SUBMARK_T(id) pf.mark(id)
#else
#define SUBMARK_T(id)
#endif
#if defined(RSIM_SUBPROF) && RSIM_SUBPROF == 4   /* tools/subprof.sh 4: x0..x6 split constraint assembly, kinematics and geom frames; x7..x9 stay with the OSC controller */
#undef SUBMARK
#define SUBMARK(id)
#define SUBMARK_U(id) pf.mark(id)
#else
#define SUBMARK_U(id)
#endif
#ifdef RSIM_MPRSTAT   /* profiling build (tools/subprof.sh mpr): how MPR runs end -> event counters in slots x0..x7 (tools/tail_report.py) */
#define MPRSTAT(slot, v) pf.count(RP_X0 + (slot), (v))
#else
#define MPRSTAT(slot, v)
#endif
// One workgroup = one wavefront: LDS instructions of a wave execute in issue order, so cross-lane communication through LDS needs no
// s_waitcnt / s_barrier, only a compiler-level ordering point (wavefront-scope fences emit no instructions; __syncthreads() would
// drain the LDS queue with s_waitcnt lgkmcnt(0) at every one of the ~110 sites).
#ifdef RSIM_JGLOBAL
#define RSIM_JG_ENABLED 1
#else
#define RSIM_JG_ENABLED 0
#endif
#ifdef RSIM_JG256
#define RSIM_JG256_ENABLED 1
#else
#define RSIM_JG256_ENABLED 0
#endif
#ifdef RSIM_MGLOBAL
#define RSIM_MG_ENABLED 1
#else
#define RSIM_MG_ENABLED 0
#endif
#ifdef RSIM_CGLOBAL
#define RSIM_CG_ENABLED 1
#else
#define RSIM_CG_ENABLED 0
#endif
#ifdef RSIM_FUSED_TIER
#define RSIM_FUSED_ENABLED 1
#else
#define RSIM_FUSED_ENABLED 0
#endif
#ifndef RSIM_NOHULLPOOL
#define RSIM_NOHULLPOOL 0   /* 1: no LDS-resident hull vertices in the middle configurations either (all hulls scanned from global memory, as the 32 x 16 build does) */
#endif
#ifndef RSIM_LS_MAXSLOT
#define RSIM_LS_MAXSLOT 4   // rows per lane up to which the polish carries the fp64 line search (4: every configuration)
#endif
#define SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define FMIN 1e-20f
#define PI_F 3.14159265358979f

enum { JNT_FREE = 0, JNT_BALL = 1, JNT_SLIDE = 2, JNT_HINGE = 3 };
enum { G_PLANE = 0, G_HFIELD, G_SPHERE, G_CAPSULE, G_ELLIPSOID, G_CYLINDER, G_BOX, G_MESH };
enum { C_FRICTION_DOF = 0, C_LIMIT_JOINT = 1, C_CONTACT_FRICTIONLESS = 2, C_CONTACT_ELLIPTIC = 3,
       C_EQUALITY = 4 /* equality/tendon: bilateral, always quadratic */, C_LIMIT_TENDON = 5 /* limit on a fixed tendon's length */,
       C_FRICTION_TENDON = 6 /* friction loss along a fixed tendon: the solver sees it as C_FRICTION_DOF on the tendon's coefficient row */ };
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
__device__ __forceinline__ double wave_sum_f64(double x) {   // butterfly over the 64 lanes (the refinement pass of the wide Newton solver: a handful of calls per substep)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}
// DPP steps that leave lanes without a source unchanged (reductions whose identity is not 0)
template <int CTRL>
__device__ __forceinline__ int dpp_keep_i(int x) { return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, false); }
// Wave-wide max / min with ONE dpp instruction per step (round 6).  Written as C (update_dpp + fmaxf) each of the six steps compiled to five instructions -- a
// register copy, the two wait states a DPP read of a fresh VALU result needs, v_mov_b32_dpp, a canonicalising v_max x, x (fmaxf must quiet signalling NaNs) and the
// v_max itself -- a ~170-cycle dependent chain in front of every mesh support evaluation of the narrow phase, whose runs are what the slowest env of a launch is made
// of.  v_max_f32_dpp with the shifted lane as first source does a step in one instruction (lanes without a source keep their value: no bound_ctrl); same result bit
// for bit (max and min are exact; the operands here are never NaN).
__device__ __forceinline__ float wave_max(float x) {
#ifdef RSIM_NO_DPP_ASM
#define RSIM_MX(C) x = fmaxf(x, __builtin_bit_cast(float, dpp_keep_i<C>(__builtin_bit_cast(int, x))))
  RSIM_MX(0x111); RSIM_MX(0x112); RSIM_MX(0x114); RSIM_MX(0x118); RSIM_MX(0x142); RSIM_MX(0x143);
#undef RSIM_MX
#else
  asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0" : "+v"(x));
#endif
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}
__device__ __forceinline__ float wave_min_f(float x) { return -wave_max(-x); }
__device__ __forceinline__ int wave_min_i(int x) {
#ifdef RSIM_NO_DPP_ASM
#define RSIM_MN(C) { int t_ = dpp_keep_i<C>(x); x = t_ < x ? t_ : x; }
  RSIM_MN(0x111) RSIM_MN(0x112) RSIM_MN(0x114) RSIM_MN(0x118) RSIM_MN(0x142) RSIM_MN(0x143)
#undef RSIM_MN
#else
  asm("s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0" : "+v"(x));
#endif
  return __builtin_amdgcn_readlane(x, 63);
}
__device__ __forceinline__ float bcast(float x, int srclane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), srclane));
}
// value of lane K of the caller's own 16-lane row (DPP row_newbcast, gfx90a+): a full-rate VALU operand modifier, no SGPR round trip.
// The dense 16x16 algebra keeps row r of a matrix in lanes r, 16+r, 32+r, 48+r so every lane finds its operand in its own row.
template <int K>
__device__ __forceinline__ float rbcast(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x150 + K, 0xf, 0xf, true));
}
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ u64 lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

// optional per-phase cycle accounting (DBatch.prof != null): lane 0 accumulates s_memtime deltas and event counters
struct Prof {
  unsigned long long* p;
  unsigned long long* pairs = nullptr;   // per-pair {visits, supports} counters
  long long t0;
  int lane;
  int c_mpr = 0, c_support = 0, c_newton = 0, c_cand = 0;   // per-env event counts of this launch (wave log)
  bool acc = true;                                          // this env contributes to the shared accumulators
  __device__ __forceinline__ void start() { if (p) t0 = clock64(); }
  __device__ __forceinline__ void mark(int id) {
    if (p) { long long t1 = clock64(); if (lane == 0 && acc) atomicAdd(p + id, (unsigned long long)(t1 - t0)); t0 = clock64(); }
  }
  __device__ __forceinline__ void count(int id, int v) {
    if (p) {
      if (id == RP_N_MPR) c_mpr += v; else if (id == RP_N_SUPPORT) c_support += v; else if (id == RP_N_NEWTON) c_newton += v; else if (id == RP_N_CAND) c_cand += v;
      if (lane == 0 && acc) atomicAdd(p + id, (unsigned long long)v);
    }
  }
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

// LDS words of the per-lane constant block: 29 body fields x NB, 13 dof x NV, 11 geom x 32, 8 site x NS, 10 actuator x 16, 5 ctrl x 16, pair rows + mfbits x 64
#define RSIM_KC_WORDS(NB, NV, NGW, NS, NPAIR) (29 * (NB) + 13 * (NV) + 11 * (NGW) + 8 * (NS) + 10 * 16 + 5 * 16 + ((NPAIR) / 64 + 1) * 64)

typedef float v4f __attribute__((ext_vector_type(4)));
// explicitly global (address space 1) views of the model tables: pointers that arrive inside the by-value DModel would otherwise be
// treated as generic and compile to flat_load
typedef const float __attribute__((address_space(1)))* gcf;
typedef const int __attribute__((address_space(1)))* gci;
typedef float __attribute__((address_space(1)))* gwf;
__device__ __forceinline__ V3 ld3(gcf p) { return v3(p[0], p[1], p[2]); }
__device__ __forceinline__ void st3(gwf p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
__device__ __forceinline__ Q4 ldq(gcf p) { Q4 q = {p[0], p[1], p[2], p[3]}; return q; }

// ------------------------------------------------------------------------------------------------------------
// per-env LDS state (one wavefront = one environment; everything below lives in LDS for all substeps of a launch)
// ------------------------------------------------------------------------------------------------------------
template <int NB, int NJ, int NV, int NG, int NS, int NCON, int NEFC, int NPAIR>
struct Smem {
  static_assert((NB == 32 || NB == 64) && (NV == 16 || NV == 32 || NV == 48 || NV == 64) && (NEFC == 64 || NEFC == 128 || NEFC == 256) && NCON <= 64 && NG <= 64 && (NS == 16 || NS == 32) && NPAIR % 64 == 0 && NPAIR <= 640,
                "lane roles: body / site columns are powers of two, dofs whole 16-column tiles, one lane per constraint row, candidate pairs in rows of 64");
  static constexpr bool TREE_TILE_ = NB == 32 && NV == 16;   // tree products as 32-body x 16-dof incidence-matrix MFMAs
  static constexpr int NVP = NV + 1;  // padded row stride of the dense nv x nv matrices (conflict-free column reads)
  static constexpr int NV_ = NV;
  static constexpr int JS_ = NV + 1, CS6_ = 9, FS_ = 17;  // LDS row strides (odd => bank-conflict-free lane-per-row access)
  static constexpr int NM_ = TREE_TILE_ ? 1 : NV;       // extent of the tree bit-mask tables (mask-loop configurations only)
  static constexpr int NS_ = NS, NPT_ = NPAIR / 64;
  static constexpr int NGW_ = NG <= 32 ? 32 : 64;       // columns of the geom role in the per-lane constant block
  typedef typename std::conditional<(NV > 32), unsigned long long, unsigned>::type dmask_t;   // one bit per dof
  static constexpr int CD_ = 4;       // largest contact dimension this configuration handles (condim 1, 3, 4)
  static constexpr int NCON_ = NCON;
  static constexpr int NEFC_ = NEFC;                    // constraint-row capacity: 64 (one row per lane) or 128 (two)
  static constexpr bool HAS_LE_ = false;                // the factor of M + hD is not kept: the Euler step factors it again (LDS bytes decide the occupancy)
  // LDS-resident hull vertices.  Occupancy is LDS-bound (one wavefront per SIMD up to four environments per CU), so the two middle
  // configurations trade pool for a third environment per CU: 64 x 16 keeps 192 vertices (53.7 KB; 57.6 KB = two per CU with the full pool),
  // 32 x 32 keeps 64 and factors M + hD again at the Euler step instead of keeping it (53.6 KB instead of 63.2 KB).  The host assigns pool
  // slots for the largest pool and load_constants() drops what does not fit.
  static constexpr int HULLPOOL_ = (NV > 32 || (NB == 32 && NV == 16) || RSIM_NOHULLPOOL) ? 0 : (NV == 32 ? 64 : 176);   // (64 x 16: 176 pool vertices, not 192: 31 892 B = 25 LDS allocation granules of 1280 B = FIVE envs per CU, round 6)  32 x 16: 20 KB = eight environments per CU, hulls are scanned from global memory (L1-resident: every env of the CU scans the same vertices)
  static constexpr int NB_ = NB;
  float qpos[NV + 8], qvel[NV], qacc[NV], qacc_ws[NV], ctrl[NV];
  float xpos[NB * 3], xquat[NB * 4];
  // articulated trees (robot + free objects) and tendon / equality rows are compiled per configuration: the 32 x 16 one (the bench workload)
  // keeps four trees and no tendon code
  static constexpr int NROOT_ = NV > 32 ? RSIM_MAXDYNROOT : 4;
  static constexpr bool TENDONS_ = !(NB == 32 && NV == 16);
  float rootcom[(NROOT_ + 1) * 3];  // subtree COM per articulated tree; last slot = 0 for static trees
  float cinert[NB * 10 + 16];  // +16: the MFMA B-operand read pattern runs 6 floats past the last row
  float cdof[NV * 9];          // stride CS6 = 9 (odd: conflict-free row reads), components 6..8 stay zero (MFMA K padding)
  // phase-local storage: crb -> broadphase -> velocity -> controller read cvel -> solver W
  union {
    struct { float crbD[NV * 17], fpad[NV * 9]; } c;                                       // crb()
    struct { int cand[NPAIR]; float poly[96]; } b;                                         // collision(): candidates, box-box clip polygons
    struct { float cvel[NB * 9]; union { struct { float cvb[NV * 9], cdd[NV * 9]; }; float cacc[NB * 9]; };
             union { float cf[NB * 17]; float F[NV * 17]; }; } v;                          // velocity(): later stages overwrite dead earlier ones
    struct { float cvel[NB * 9]; float Jm[48], Li[36], vv[8], Lt[64], Ys[128]; } k;                  // ctrl_run(): cvel stays live from velocity()
    float rowst[NV == 16 ? 1 : 64 * 20];                                                 // make_constraint() (wide): per contact row the two 6-vectors (padded to 8) that multiply cdof, and the two bodies' dof masks
    float W[NV == 16 ? NEFC * (NV + 1) : NEFC * 5];                                        // solve_newton(): Hessian-weighted rows (one-tile configurations); wide: five words per row (four block coefficients, block head | dim | cone flag) from which the products form the weighted row on the fly
  } u;
  // RSIM_MGLOBAL (on top of RSIM_JGLOBAL): the mass matrix behind J in the same per-env global buffer: 49.7 -> 40.3 KB = FOUR environments per CU, one wavefront
  // per SIMD.  PickPlace @8192 + DR: 112.4 -> 89.5 ms per control step (+25.6 %, five round-robin reps, profiles/r05_a_ab_variants_pickplace.txt)
  static constexpr bool MG_ = RSIM_MG_ENABLED && RSIM_JG_ENABLED && ((NV == 48 && NEFC == 128) || (NV == 32 && (NEFC == 64 || NEFC == 128)));
  float M[MG_ ? 4 : NV * NVP];
  union { float L[NV * NVP]; float H[NV * NVP]; };  // L (factor of M) is dead once qacc_smooth exists; H is the solver / Euler work matrix
  float invdiag[NV];
  float Le[HAS_LE_ ? NV * NVP : 1], invdiag_e[NV];   // Cholesky factor of M + h*diag(damping) (implicit-damping Euler), computed alongside L
  float qfrc_bias[NV], qfrc_passive[NV], qfrc_actuator[NV], qfrc_smooth[NV], qacc_smooth[NV], qfrc_constraint[NV];
  float gpos[NG * 3], gmat[NG * 9], gcen[NG * 3];
  float spos[NS * 3], smat[NS * 9];
  // contacts
  // RSIM_CGLOBAL (32 x 32 build, on top of RSIM_JGLOBAL / RSIM_MGLOBAL): contact frames and contact material parameters -- written once per contact by the narrow phase, read
  // by the row builders -- in the same per-env global buffer behind J and M: 22.6 -> 19.8 KB = EIGHT environments per CU, two wavefronts on every SIMD (layout: CG_* below)
  static constexpr bool FUSEDW_ = RSIM_FUSED_ENABLED && NEFC == 128 && (NV == 16 || NV == 32);   // the wide body of a fused-tier build (RSIM_DIMS_W): J and the contact block in DBatch.jg (32 x 32: M as well, with the build's RSIM_MGLOBAL)
  // (64 x 16 build, round 6: J and the contact block out, M stays -- 23.1 KB = seven envs per CU at two wavefronts per SIMD, see the Makefile)
  static constexpr bool CG_ = (RSIM_CG_ENABLED && RSIM_JG_ENABLED && ((RSIM_MG_ENABLED && NV == 32 && NEFC == 64) || (NB == 64 && NV == 16 && NEFC == 64))) || FUSEDW_;
  static constexpr int CG_FRAME_ = 0, CG_FRI_ = NCON * 9, CG_SOLIMP_ = NCON * 14, CG_SOLREF_ = NCON * 19, CG_MARGIN_ = NCON * 21, CG_WORDS_ = NCON * 22;
  float cpos[NCON * 3], cframe[CG_ ? 1 : NCON * 9], cdist[NCON], cfri[CG_ ? 1 : NCON * 5], csolref[CG_ ? 1 : NCON * 2], csolimp[CG_ ? 1 : NCON * 5], cmu[NCON], cmargin[CG_ ? 1 : NCON];
  // geom | body << 8 of the two sides, contact dimension, first constraint row (-1: none): 16 / 8-bit fields (round 6: with the row descriptors below 0.5 KB of the wide
  // bodies' LDS, which is what lets the Stack-class tier share the native body's 20 KB)
  unsigned short cg1[NCON], cg2[NCON]; short cefc[NCON]; unsigned char cdim[NCON];
  // constraint rows
  // RSIM_JGLOBAL (64 x 48 build with 128 rows only): the constraint Jacobian lives in a per-env buffer in GLOBAL memory (DBatch.jg; 25 KB per env, L2-resident
  // for the resident envs of an XCD) instead of LDS: 74.8 -> 49.7 KB = three environments per CU instead of two
  // (RSIM_JG256: the 256-row tier of the same shape as well -- 116 -> 66 KB, two jumbo envs per CU instead of one)
  static constexpr bool JG_ = (RSIM_JG_ENABLED && ((NV == 48 && (NEFC == 128 || (RSIM_JG256_ENABLED && NEFC == 256))) || (NV == 32 && (NEFC == 64 || NEFC == 128)) || (NB == 64 && NV == 16 && NEFC == 64))) || (RSIM_FUSED_ENABLED && NV == 16 && NEFC == 128);
  float J[JG_ ? 4 : NEFC * (NV + 1)];  // row-major, stride JS = NV + 1 (odd: the row-owner lanes hit distinct banks); four rows = one MFMA B operand
  float e_R[NEFC], e_aref[NEFC], e_force[NEFC];   // e_force doubles as the row's velocity gain B between make_constraint's two halves
  unsigned short e_desc[NEFC];    // type | id<<4 | k<<12 (row k of its block): 16 bits
  float cstate[RSIM_CS_LDS];     // first RSIM_CS_LDS floats of the controller state (all of it for the OSC and plain joint-space types); the tail stays in global memory
  float red[NV];
  float hull[3 * HULLPOOL_ + 1];    // LDS-resident hull vertex pool (SoA x | y | z); filled once per launch
  int ncon, nefc, niter, polish;
};

// Model constants of one environment in the layout the kernel's lane roles read them: built by prepare_constants() (k_prepare, or the env's own
// wavefront after an on-device episode reset patched its float table) into GLOBAL memory -- one block shared by all envs while no env has
// overridden a float-table field, one per env otherwise -- and read with ordinary global loads.  It used to be rebuilt into LDS by every launch
// (18 KB of the 40 KB that limited the 32 x 16 configuration to four environments per CU).
template <int NB, int NJ, int NV, int NG, int NS, int NCON, int NEFC, int NPAIR>
struct Cmem {
  typedef Smem<NB, NJ, NV, NG, NS, NCON, NEFC, NPAIR> S;
  typedef typename S::dmask_t dmask_t;
  float opt[12];                 // timestep, gravity3, density, viscosity, impratio, wind3
  float arm[NV];                 // dof armature (1 on padding rows: keeps the padded matrices SPD)
  float fricR[NV], fricB[NV], fricFl[NV];  // dof friction-loss rows: regulariser, velocity gain, force limit
  float biw[NB * 2];             // body_invweight0
  dmask_t bdofs[NB];
  int broot[NB];
  alignas(16) float gst[NG * 8];   // colliding geom statics: half-extents 3, box centre (geom frame) 3, rbound, margin (read as two 16-byte loads)
  alignas(16) float gcap[NG * 8];  // bounding capsule of mesh hulls in the geom frame: p3 q3 R (R < 0: none)
  int gtype[NG], gbody[NG], gcp[NG], gmesh[NG];   // type, body, condim | priority<<8, hull vertex adr | count<<16
  int ghull[NG];                 // first slot of geom g in the LDS-resident hull pool, -1: scanned from global memory
  float gpar[NG * 12];           // contact material per geom: friction3 solref2 solimp5 solmix gap
  // wide configuration: tree incidence as bit masks (dof-ancestor set, dofs summed before dof i in the velocity recursion, body ancestors)
  dmask_t dmask_anc[S::NM_], dmask_cvel[S::NM_];
  unsigned long long bmask_anc[S::TREE_TILE_ ? 1 : NB];
  int dynroot[RSIM_MAXDYNROOT];  // root body of each articulated tree (a run-time index into the by-value DModel would put a copy of it into the private segment)
  // the same incidence relations as MFMA A operands (wide configurations, shared block only: topology, no float-table field): lane (q = lane >> 4,
  // r = lane & 15), bit t * KS + c = relation(row 16 t + r, column 4 c + q) for row tile t and k-step c --
  //   msub: dof row / body column: the body lies in the subtree of the dof's body (KS = NB / 4)
  //   mbd:  body row / dof column: the dof moves the body (KS = NV / 4)      mcv: dof row / dof column: summed before the dof in the velocity recursion
  unsigned long long msub[S::TREE_TILE_ ? 1 : 64], mbd[S::TREE_TILE_ ? 1 : 64], mcv[S::TREE_TILE_ ? 1 : 64];
  // per-lane model constants, one row per field (LaneConst below); phases fetch the handful they need instead of pinning ~80 VGPRs
  float kc[RSIM_KC_WORDS(NB, NV, S::NGW_, NS, NPAIR)];
};

// The one per-workgroup LDS object, declared at file scope so that every access is a direct LDS (ds_*) instruction with an
// immediate offset: routing it through a reference member made the compiler fall back to flat_* loads/stores.
typedef Smem<RSIM_DIMS> Smem0;
#ifdef RSIM_DIMS_W
// fused-tier build: the native and the wide body of one kernel share the workgroup's LDS (a workgroup runs one of them at a time; the hand-over in mid-step
// relies on the state arrays qpos .. ctrl at the head of both layouts having the same offsets)
typedef Smem<RSIM_DIMS_W> SmemW;
union SmemU { Smem0 n; SmemW w; };
static __shared__ SmemU smu;
template <class S> struct LdsOf;
template <> struct LdsOf<Smem0> { static __device__ __forceinline__ Smem0& get() { return smu.n; } };
template <> struct LdsOf<SmemW> { static __device__ __forceinline__ SmemW& get() { return smu.w; } };
static_assert(offsetof(Smem0, xpos) == offsetof(SmemW, xpos), "state arrays at the head of both LDS layouts");
#else
static __shared__ Smem0 sm0_;
template <class S> struct LdsOf { static __device__ __forceinline__ Smem0& get() { return sm0_; } };
#endif
// `sm` = the LDS object in the layout of the configuration the enclosing template is instantiated for (SM must name it at every use)
#define sm (LdsOf<SM>::get())
typedef Cmem<RSIM_DIMS> Cmem0;
typedef const Cmem0 __attribute__((address_space(1)))* cmr_t;   // read view of an env's constant block (global_load, never flat_load)
typedef Cmem0 __attribute__((address_space(1)))* cmw_t;         // write view (prepare_constants)

#define IT(tab, i) (((gci)m.it)[m.io[tab] + (i)])
// per-env value only for fields some env has overridden (per-episode object sizes, domain randomisation); everything else comes from the
// shared copy, which stays L2-resident instead of being streamed from HBM once per env and launch
#define FP(tab, i) (((gcf)(((fenv >> (tab)) & 1ull) ? fp : m.ft0))[m.fo[tab] + (i)])    // inside Sim: `fenv` is re-laundered at every phase()
#define FPM(tab, i) (((gcf)(((m.fenv >> (tab)) & 1ull) ? fp : m.ft0))[m.fo[tab] + (i)])

// Register-resident Cholesky: lane i owns row (i & 15) of the SPD matrix in a[0..N) (the four 16-lane rows of the wave hold identical
// copies); all loops unroll so every index is a compile-time register and every broadcast is a DPP row_newbcast operand.
// After the call a[k] = L[i][k] (k <= i), inv[k] = 1 / L[k][k] (uniform).
template <int N, int J, int K>
struct RcholUpd {
  static __device__ __forceinline__ void run(float (&a)[N], float lij) {
    if constexpr (K < N) { a[K] = fmaf(-lij, rbcast<K>(lij), a[K]); RcholUpd<N, J, K + 1>::run(a, lij); }
  }
};
template <int N, int J>
struct RcholStep {
  static __device__ __forceinline__ void run(float (&a)[N], float (&inv)[N], int row, float& own) {
    if constexpr (J < N) {
      const float iv = rsqrtf(fmaxf(rbcast<J>(a[J]), FMIN));
      inv[J] = iv;
      own = row == J ? iv : own;
      const float lij = a[J] * iv;
      a[J] = lij;
      RcholUpd<N, J, J + 1>::run(a, lij);
      RcholStep<N, J + 1>::run(a, inv, row, own);
    }
  }
};
template <int N>
__device__ __forceinline__ void rchol_factor(float (&a)[N], float (&inv)[N]) { float own = 0.f; RcholStep<N, 0>::run(a, inv, -1, own); }
// same, also returning 1 / L[row][row] of the caller's own row (for the lane that stores invdiag[row])
template <int N>
__device__ __forceinline__ float rchol_factor_own(float (&a)[N], float (&inv)[N], int row) { float own = 0.f; RcholStep<N, 0>::run(a, inv, row, own); return own; }
// forward substitution only: returns y = L^-1 x (component i in the lanes of row i); `row` = lane & 15
template <int N, int K>
struct RcholFwd {
  static __device__ __forceinline__ float run(const float (&a)[N], const float (&inv)[N], float x, int row) {
    if constexpr (K < N) {
      const float xk = rbcast<K>(x) * inv[K];
      x = row == K ? xk : (row > K ? fmaf(-a[K], xk, x) : x);
      return RcholFwd<N, K + 1>::run(a, inv, x, row);
    } else return x;
  }
};
template <int N, int K>
struct RcholBwd {
  static __device__ __forceinline__ float run(const float (&at)[N], const float (&inv)[N], float x, int row) {
    if constexpr (K >= 0) {
      const float xk = rbcast<K>(x) * inv[K];
      x = row == K ? xk : (row < K ? fmaf(-at[K], xk, x) : x);
      return RcholBwd<N, K - 1>::run(at, inv, x, row);
    } else return x;
  }
};
template <int N>
__device__ __forceinline__ float rchol_fwd(const float (&a)[N], const float (&inv)[N], float x, int lane) { return RcholFwd<N, 0>::run(a, inv, x, lane & 15); }
// x_i in the lanes of row i; a = rows of L, at[k] = L[k][i] (column i of L), inv = 1/diag.  Returns (L L^T)^-1 x.
template <int N>
__device__ __forceinline__ float rchol_solve(const float (&a)[N], const float (&at)[N], const float (&inv)[N], float x, int lane) {
  x = RcholFwd<N, 0>::run(a, inv, x, lane & 15);
  return RcholBwd<N, N - 1>::run(at, inv, x, lane & 15);
}
// ---- mask-free triangular solves.  The selects `row == K ? .. : row > K ? ..` above cost a lane predicate per step; those predicates are
// loop-invariant, so the compiler hoists all of them out of the substep loop as 64-bit SGPR masks and then spills / reloads them around
// every use (4 v_readlane + s_nop per step).  Here the factor rows are masked ONCE to their strictly-lower part (am[k] = 0 for k >= row;
// the transposed copy atm[k] = L[k][row] for k > row, else 0), the diagonal scaling is deferred to the end (lane K's x is not touched
// after step K), and every step is two instructions: a DPP multiply and an FMA.
__device__ __forceinline__ int opaque_lane(int x) { asm volatile("" : "+v"(x)); return x; }   // fresh value: keeps the masks from being hoisted
template <int N>
__device__ __forceinline__ void rchol_mask_lower(float (&a)[N], int row) {
#pragma unroll
  for (int k = 0; k < N; k++) a[k] = k < row ? a[k] : 0.f;
}
template <int N, int K>
struct RcholFwdM {
  static __device__ __forceinline__ float run(const float (&am)[N], const float (&inv)[N], float x) {
    if constexpr (K < N) { const float xk = rbcast<K>(x) * inv[K]; return RcholFwdM<N, K + 1>::run(am, inv, fmaf(-am[K], xk, x)); }
    else return x;
  }
};
template <int N, int K>
struct RcholBwdM {
  static __device__ __forceinline__ float run(const float (&atm)[N], const float (&inv)[N], float x) {
    if constexpr (K >= 0) { const float xk = rbcast<K>(x) * inv[K]; return RcholBwdM<N, K - 1>::run(atm, inv, fmaf(-atm[K], xk, x)); }
    else return x;
  }
};
// y = L^-1 x;  am = strictly-lower row of L, inv[k] = 1 / L[k][k] (uniform), own = 1 / L[row][row]
template <int N>
__device__ __forceinline__ float rchol_fwd_m(const float (&am)[N], const float (&inv)[N], float own, float x) { return RcholFwdM<N, 0>::run(am, inv, x) * own; }
// (L L^T)^-1 x
template <int N>
__device__ __forceinline__ float rchol_solve_m(const float (&am)[N], const float (&atm)[N], const float (&inv)[N], float own, float x) {
  x = RcholFwdM<N, 0>::run(am, inv, x) * own;
  return RcholBwdM<N, N - 1>::run(atm, inv, x) * own;
}
// sum_k r[k] * x_k, where x_k lives in lane k of every 16-lane row (replicated per-dof vector)
template <int N, int K>
struct RowDot {
  static __device__ __forceinline__ float run(const float (&r)[N], float x, float acc) {
    if constexpr (K < N) return RowDot<N, K + 1>::run(r, x, fmaf(r[K], rbcast<K>(x), acc));
    else return acc;
  }
};
template <int N>
__device__ __forceinline__ float dot_rows(const float (&r)[N], float x) { return RowDot<N, 0>::run(r, x, 0.f); }

template <int N>
__device__ __forceinline__ int seli(const int (&v)[N], int i) {
  int r = v[0];
#pragma unroll
  for (int k = 1; k < N; k++) r = i == k ? v[k] : r;
  return r;
}
template <int N>
__device__ __forceinline__ float sel(const float (&v)[N], int i) {
  float r = v[0];
#pragma unroll
  for (int k = 1; k < N; k++) r = i == k ? v[k] : r;
  return r;
}
// 3x3 SPD solve by cofactors (uniform small algebra)
__device__ __forceinline__ void solve3(const float* A, const float* b, float* x) {
  const float c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
  const float det = A[0] * c00 + A[1] * c01 + A[2] * c02, id = 1.0f / det;
  const float c10 = A[2] * A[7] - A[1] * A[8], c11 = A[0] * A[8] - A[2] * A[6], c12 = A[1] * A[6] - A[0] * A[7];
  const float c20 = A[1] * A[5] - A[2] * A[4], c21 = A[2] * A[3] - A[0] * A[5], c22 = A[0] * A[4] - A[1] * A[3];
  x[0] = (c00 * b[0] + c10 * b[1] + c20 * b[2]) * id;
  x[1] = (c01 * b[0] + c11 * b[1] + c21 * b[2]) * id;
  x[2] = (c02 * b[0] + c12 * b[1] + c22 * b[2]) * id;
}
// sin and cos of a bounded angle (|x| < ~1e3): Cody-Waite reduction to [-pi/4, pi/4] + Taylor kernels (abs error < 2e-7)
__device__ __forceinline__ void sincos_f(float x, float& sn, float& cs) {
  const float k = rintf(x * 0.636619772367581343f);
  float r = fmaf(-k, 1.57079625129699707031f, x);
  r = fmaf(-k, 7.54978941586159635335e-08f, r);
  const float r2 = r * r;
  const float sp = r * fmaf(r2, fmaf(r2, fmaf(r2, fmaf(r2, 2.7557319e-6f, -1.9841270e-4f), 8.3333333e-3f), -1.6666667e-1f), 1.0f);
  const float cp = fmaf(r2, fmaf(r2, fmaf(r2, fmaf(r2, fmaf(r2, -2.7557319e-7f, 2.4801587e-5f), -1.3888889e-3f), 4.1666667e-2f), -0.5f), 1.0f);
  const int q = (int)k & 3;
  const float s1 = (q & 1) ? cp : sp, c1 = (q & 1) ? sp : cp;
  sn = (q & 2) ? -s1 : s1;
  cs = ((q + 1) & 2) ? -c1 : c1;
}

// ---- orientation interpolator of OSC_POSE (LinearInterpolator in 'euler' mode, utils/traj_utils.py:129-146) ------------------------------
// start / goal are orientation-error vectors that the reference treats as Euler angles:
//   mat2euler(quat2mat(quat_slerp(mat2quat(euler2mat(start)), mat2quat(euler2mat(goal)), fraction)))   (transform_utils.py:151-201, 316-440)
__device__ __forceinline__ Q4 euler_to_quat(V3 e) {
  float si, ci, sj, cj, sk, ck;
  sincos_f(-e.z, si, ci); sincos_f(-e.y, sj, cj); sincos_f(-e.x, sk, ck);
  const float cc = ci * ck, cs = ci * sk, sc = si * ck, ss = si * sk;
  const float m00 = cj * ci, m01 = cj * si, m02 = -sj, m10 = sj * cs - sc, m11 = sj * ss + cc, m12 = cj * sk, m20 = sj * cc + ss, m21 = sj * sc - cs, m22 = cj * ck;
  // unit quaternion of the matrix with w >= 0 (the reference takes the dominant eigenvector of K and flips it to w >= 0)
  const float tr = m00 + m11 + m22;
  Q4 q;
  if (tr > 0.f) { const float s = 2.f * sqrtf(1.f + tr); q.w = 0.25f * s; q.x = (m21 - m12) / s; q.y = (m02 - m20) / s; q.z = (m10 - m01) / s; }
  else if (m00 > m11 && m00 > m22) { const float s = 2.f * sqrtf(1.f + m00 - m11 - m22); q.w = (m21 - m12) / s; q.x = 0.25f * s; q.y = (m01 + m10) / s; q.z = (m02 + m20) / s; }
  else if (m11 > m22) { const float s = 2.f * sqrtf(1.f + m11 - m00 - m22); q.w = (m02 - m20) / s; q.x = (m01 + m10) / s; q.y = 0.25f * s; q.z = (m12 + m21) / s; }
  else { const float s = 2.f * sqrtf(1.f + m22 - m00 - m11); q.w = (m10 - m01) / s; q.x = (m02 + m20) / s; q.y = (m12 + m21) / s; q.z = 0.25f * s; }
  const float inv = (q.w < 0.f ? -1.f : 1.f) * rsqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  q.w *= inv; q.x *= inv; q.y *= inv; q.z *= inv;
  return q;
}
__device__ __forceinline__ V3 euler_slerp(V3 start, V3 goal, float fraction) {
  const Q4 a = euler_to_quat(start);
  Q4 b = euler_to_quat(goal);
  float d = a.w * b.w + a.x * b.x + a.y * b.y + a.z * b.z, w0, w1;
  if (d < 0.f) { d = -d; b.w = -b.w; b.x = -b.x; b.y = -b.y; b.z = -b.z; }
  if (fraction >= 1.f) { w0 = 0.f; w1 = 1.f; }
  else if (d > 0.9995f) { w0 = 1.f - fraction; w1 = fraction; }   // fp32: sin(t angle) / sin(angle) -> t below 0.03 rad (rel. error 2e-4 angle^2)
  else {
    const float angle = acosf(d), isin = 1.f / sinf(angle);
    w0 = sinf((1.f - fraction) * angle) * isin; w1 = sinf(fraction * angle) * isin;
  }
  const float qw = a.w * w0 + b.w * w1, qx = a.x * w0 + b.x * w1, qy = a.y * w0 + b.y * w1, qz = a.z * w0 + b.z * w1;
  // quat2mat (scale 2 / |q|^2) -> mat2euler 'sxyz'
  const float k = 2.f / (qw * qw + qx * qx + qy * qy + qz * qz);
  const float m00 = 1.f - k * (qy * qy + qz * qz), m10 = k * (qx * qy + qz * qw), m20 = k * (qx * qz - qy * qw);
  const float m21 = k * (qy * qz + qx * qw), m22 = 1.f - k * (qx * qx + qy * qy);
  const float cy = sqrtf(m00 * m00 + m10 * m10);
  return v3(atan2f(m21, m22), atan2f(-m20, cy), atan2f(m10, m00));   // cy = 0 (gimbal lock) needs a pi/2 error component: out of the |e| <= 1 range
}
// rotate v by unit quaternion q
__device__ __forceinline__ V3 qrot(Q4 q, V3 v) {
  const V3 u = v3(q.x, q.y, q.z);
  const V3 t = cross(u, v) * 2.0f;
  return v + t * q.w + cross(u, t);
}
__device__ __forceinline__ float bitf(unsigned bits, int i) { return (float)((bits >> i) & 1u); }
__device__ __forceinline__ float comp6(S6 a, int r) { return r < 3 ? (r == 0 ? a.a.x : (r == 1 ? a.a.y : a.a.z)) : (r == 3 ? a.l.x : (r == 4 ? a.l.y : a.l.z)); }


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
// same, factoring the matrix already stored in L (lower triangle read, lower triangle written)
template <int NVP>
__device__ __forceinline__ void chol_inplace(float* L, float* invdiag, int n, int lane) {
  for (int j = 0; j < n; j++) {
    float s = 0.f, d0 = 0.f;
    if (lane >= j && lane < n) {
      s = d0 = L[lane * NVP + j];
      for (int k = 0; k < j; k++) s -= L[lane * NVP + k] * L[j * NVP + k];
    }
    // fp32 safeguard: a pivot that cancelled below 1e-6 of its diagonal entry is rounding noise (nearly dependent constraint rows on links of
    // 5e-5 kg m^2); flooring it there keeps the factor finite and the Newton direction a descent direction.  Inactive on well-conditioned H.
    const float sj = fmaxf(bcast(s, j), 1e-6f * bcast(d0, j));
    const float inv = rsqrtf(fmaxf(sj, FMIN));
    if (lane >= j && lane < n) L[lane * NVP + j] = (lane == j) ? sj * inv : s * inv;
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

// ---- blocked Cholesky on the LDS matrix for the configurations beyond one 16 x 16 tile (nv up to 64) ----------------------------------------
// The column-by-column factorisation above pays one LDS round trip and one dependent dot-product chain per column (37 of them for the
// PickPlace model, three factorisations per substep plus one per Newton iteration).  Here the matrix is walked in 16-column blocks:
//   (1) the diagonal block is factored in registers (the DPP Cholesky of the one-tile configurations, every index a compile-time constant),
//   (2) the rows below it are solved against that block, one row per lane (x L_bb^T = a: 16 steps on uniform factor entries),
//   (3) the trailing tiles take A_IJ -= L_Ib L_Jb^T on the matrix cores (16 x 16 x 4 f32 MFMA, four per tile),
// i.e. three block steps and ~a dozen LDS round trips for 48 dofs.  Layout and conventions of chol_inplace(): strictly lower triangle = L,
// invdiag[k] = 1 / L[k][k]; rows / columns n .. 16 ceil(n / 16) - 1 must hold identity padding (every caller's matrix does).
// One column block of the factorisation, right-looking over ALL rows from the block's first row down (lane i = row c0 + i): column J's pivot
// comes from lane J, every lane scales its entry and takes the rank-1 update of its remaining 15 - J entries against the entries of lanes
// J + 1 .. 15 (uniform v_readlane operands).  The diagonal block and the panel below it are the same 136 readlane + FMA pairs -- the separate
// register factorisation of the diagonal block (four redundant copies, DPP chain) and its LDS round trip are gone.  Same products in the same
// order as the two-stage form it replaces.
template <int NVP>
__device__ __forceinline__ void bchol_inplace(float* H, float* invdiag, int n, int lane) {
  const int nb = (n + 15) >> 4, r = lane & 15, q = lane >> 4;
  const float dorig = H[(lane < 16 * nb ? lane : 0) * NVP + (lane < 16 * nb ? lane : 0)];   // original diagonal (pivot floor)
  for (int b = 0; b < nb; b++) {
    const int c0 = 16 * b;
    const int row = c0 + lane;
    const bool act = row < 16 * nb;
    float x[16], iv[16];
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = H[(act ? row : c0) * NVP + c0 + k];
    const float dfl = 1.0e-6f * __shfl(dorig, c0 + r);   // lane J (J < 16) holds the floor of column c0 + J
#pragma unroll
    for (int J = 0; J < 16; J++) {
      const float piv = fmaxf(bcast(x[J], J), bcast(dfl, J));   // a pivot that cancelled below 1e-6 of its original diagonal entry is floored there
      const float ivj = rsqrtf(fmaxf(piv, FMIN));
      iv[J] = ivj;
      const float lij = x[J] * ivj;
      x[J] = lij;
#pragma unroll
      for (int K = J + 1; K < 16; K++) x[K] = fmaf(-lij, bcast(lij, K), x[K]);
    }
    SYNC();
    if (act) {
#pragma unroll
      for (int k = 0; k < 16; k++) H[row * NVP + c0 + k] = (lane >= 16 || k < lane) ? x[k] : 0.f;   // diagonal block: strictly lower part (diagonal and above: zeros, never read)
    }
    if (lane < 16) {
      float own = 0.f;
#pragma unroll
      for (int k = 0; k < 16; k++) own = lane == k ? iv[k] : own;
      invdiag[c0 + lane] = own;
    }
    if (b + 1 == nb) break;
    SYNC();
    // trailing tiles (I, J), b < J <= I < nb: A_IJ -= L_Ib L_Jb^T
    for (int I = b + 1; I < nb; I++)
      for (int Jb = b + 1; Jb <= I; Jb++) {
        v4f acc;
#pragma unroll
        for (int v = 0; v < 4; v++) acc[v] = H[(16 * I + 4 * q + v) * NVP + 16 * Jb + r];
#pragma unroll
        for (int c = 0; c < 4; c++)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(-H[(16 * I + r) * NVP + c0 + 4 * c + q], H[(16 * Jb + r) * NVP + c0 + 4 * c + q], acc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; v++) H[(16 * I + 4 * q + v) * NVP + 16 * Jb + r] = acc[v];
      }
    SYNC();
  }
  SYNC();
}
// x: per-lane value (lane i < n holds b_i, 0 beyond); returns (L L^T)^-1 b, component i in lane i.  Block forward / backward substitution on the
// factor bchol_inplace() left in LDS: the 16 x 16 diagonal solves run in registers (mask-free DPP steps), the off-diagonal parts are 16-term
// row / column products per lane.
template <int NVP>
__device__ __forceinline__ float bchol_solve(const float* H, const float* invdiag, float x, int n, int lane) {
  const int nb = (n + 15) >> 4, r = lane & 15;
  if (lane >= n) x = 0.f;
  // forward: L y = b
  for (int b = 0; b < nb; b++) {
    const int c0 = 16 * b;
    float lr[16], inv[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { lr[k] = H[(c0 + r) * NVP + c0 + k]; inv[k] = invdiag[c0 + k]; }
    const float yb = rchol_fwd_m<16>(lr, inv, invdiag[c0 + r], __shfl(x, c0 + r));   // component r of block b in every lane with lane & 15 == r
    if ((lane >> 4) == b) x = yb;
    if (b + 1 < nb) {   // rows below the block: x_i -= sum_k L[i][c0 + k] y_k  (y_k = lane k of the caller's own 16-lane row)
      const bool below = lane >= c0 + 16 && lane < 16 * nb;
      float row[16];
#pragma unroll
      for (int k = 0; k < 16; k++) row[k] = H[(below ? lane : c0) * NVP + c0 + k];
      const float acc = dot_rows<16>(row, yb);
      if (below) x -= acc;
    }
  }
  // backward: L^T z = y
  for (int b = nb - 1; b >= 0; b--) {
    const int c0 = 16 * b;
    float lt[16], inv[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { lt[k] = k > r ? H[(c0 + k) * NVP + c0 + r] : 0.f; inv[k] = invdiag[c0 + k]; }
    const float zb = RcholBwdM<16, 15>::run(lt, inv, __shfl(x, c0 + r)) * invdiag[c0 + r];
    if ((lane >> 4) == b) x = zb;
    if (b > 0) {        // rows above the block: x_i -= sum_k L[c0 + k][i] z_k
      const bool above = lane < c0;
      float colv[16];
#pragma unroll
      for (int k = 0; k < 16; k++) colv[k] = H[(c0 + k) * NVP + (above ? lane : 0)];
      const float acc = dot_rows<16>(colv, zb);
      if (above) x -= acc;
    }
  }
  return lane < n ? x : 0.f;
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

// support point of colliding geom g along world direction dir (wave-cooperative for meshes; result uniform).
// A real function (not inlined into its ~10 call sites); it only touches the LDS object and the hull vertex table.
// `org`: origin the result is expressed in.  MPR passes the first geom's position, so that its portal algebra runs on centimetre-sized coordinates
// (rounding ~4e-9 m) instead of metre-sized world coordinates (~1e-7 m): which facet of the Minkowski difference the origin ray leaves through is
// then decided by the geometry, not by fp32 noise (fine meshes in deep penetration ended on neighbouring facets, degrees apart, in half of the cases).
template <class SM>
__device__ __forceinline__ V3 geom_support(cmr_t cm, cmr_t cmg, int g, V3 dir, gcf mesh_vert, int lane, V3 org) {
  const int t = cm->gtype[g];
  const M3 R = ldm(sm.gmat + 9 * g);
  const V3 p = ld3(sm.gpos + 3 * g) - org, h = ld3(cmg->gst + 8 * g);
  const V3 ld = mtv(R, dir);
  V3 lp = v3(0, 0, 0);
  if (t == G_BOX) lp = v3(ld.x >= 0 ? h.x : -h.x, ld.y >= 0 ? h.y : -h.y, ld.z >= 0 ? h.z : -h.z);
  else if (t == G_SPHERE) { float n = norm(ld); if (n > FMIN) lp = ld * (h.x / n); }
  else if (t == G_CYLINDER) {
    float n = sqrtf(ld.x * ld.x + ld.y * ld.y);
    if (n > FMIN) { lp.x = ld.x / n * h.x; lp.y = ld.y / n * h.x; }
    lp.z = ld.z >= 0 ? h.z : -h.z;
  } else if (t == G_CAPSULE) {
    float n = norm(ld);
    if (n > FMIN) lp = ld * (h.x / n);
    lp.z += ld.z >= 0 ? (h.z - h.x) : -(h.z - h.x);
  } else if (t == G_ELLIPSOID) {
    V3 tt = v3(ld.x * h.x, ld.y * h.y, ld.z * h.z);
    float n = norm(tt);
    if (n > FMIN) lp = v3(tt.x / n * h.x, tt.y / n * h.y, tt.z / n * h.z);
  } else if (t == G_MESH) {
    const int adr = cm->gmesh[g] & 0xffff, num = cm->gmesh[g] >> 16;
    float bv = -3.0e38f, bx = 0.f, by = 0.f, bz = 0.f;
    int bi = 0x7fffffff;
    const int pool = SM::HULLPOOL_ > 0 ? cm->ghull[g] : -1;
    if (SM::HULLPOOL_ > 0 && pool >= 0) {
      // hull resident in LDS: lane l scans vertices l, l+64, ... (at most 256 per hull in the pool)
      const float* hv = sm.hull + pool;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int i = 64 * u + lane;
        if (64 * u < num) {
          const int ii = i < num ? i : 0;
          const float x = hv[ii], y = hv[SM::HULLPOOL_ + ii], z = hv[2 * SM::HULLPOOL_ + ii];
          const float val = x * ld.x + y * ld.y + z * ld.z;
          const bool take = i < num && val > bv; bv = take ? val : bv; bi = take ? i : bi; bx = take ? x : bx; by = take ? y : by; bz = take ? z : bz;
        }
      }
    } else {
      for (int base = 0; base < num; base += 256) {
        float vx[4], vy[4], vz[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {   // all loads of the chunk first, then the compares
          const int i = base + 64 * u + lane;
          gcf v = mesh_vert + 3 * (adr + (i < num ? i : 0));
          vx[u] = v[0]; vy[u] = v[1]; vz[u] = v[2];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int i = base + 64 * u + lane;
          const float val = vx[u] * ld.x + vy[u] * ld.y + vz[u] * ld.z;
          const bool take = i < num && val > bv; bv = take ? val : bv; bi = take ? i : bi; bx = take ? vx[u] : bx; by = take ? vy[u] : by; bz = take ? vz[u] : bz;
        }
      }
    }
    // wave arg-max, lowest index wins ties (matches the serial first-maximum scan).  One lane holds the maximum in all but degenerate
    // directions: its id comes from a ballot; exact ties fall back to the index reduction.  The winner's vertex is read with
    // v_readlane (uniform lane id), not a cross-lane LDS permute.
    const float mx = wave_max(bv);
    const u64 tie = __ballot(bv == mx);
    int win = __ffsll((long long)tie) - 1;
    if (__popcll(tie) > 1) win = wave_min_i(bv == mx ? bi : 0x7fffffff) & 63;  // vertex i is scanned by lane i % 64
    win = uni(win);
    lp = v3(bcast(bx, win), bcast(by, win), bcast(bz, win));
  }
  return p + mv(R, lp);
}

// One geom of an MPR run, loaded ONCE: type, frame, position relative to the run's origin, size, and -- for hulls of up to 256 vertices --
// the vertices themselves, four per lane in registers.  A support evaluation then touches no memory at all (it used to re-read the type, the
// hull address and the vertices in three dependent round trips, ~0.7 us of the ~1 us an MPR iteration took); MPR-heavy envs are the critical
// path of a launch (3-4x the median env), so this is launch time, not just instruction count.
struct SupGeom {
  int t, adr, num;
  M3 R;
  V3 p, h;
  float vx[4], vy[4], vz[4];
  bool regs;
};
template <class SM>
__device__ __forceinline__ SupGeom sup_load(cmr_t cm, cmr_t cmg, int g, gcf mesh_vert, int lane, V3 org) {
  SupGeom s;
  s.t = uni(cm->gtype[g]);          // scalar: the type dispatch of every support call becomes s_cbranch on an SGPR instead of exec-masked regions
  const int gm = uni(cm->gmesh[g]);
  s.adr = gm & 0xffff; s.num = gm >> 16;
  s.R = ldm(sm.gmat + 9 * g);
  s.p = ld3(sm.gpos + 3 * g) - org; s.h = ld3(cmg->gst + 8 * g);
  s.regs = s.t == G_MESH && s.num <= 256;
#pragma unroll
  for (int u = 0; u < 4; u++) {   // issued unconditionally (index clamped): the loads overlap the type / size loads above instead of waiting for them
    const int i = 64 * u + lane;
    gcf v = mesh_vert + 3 * (s.adr + ((s.t == G_MESH && i < s.num) ? i : 0));
    s.vx[u] = v[0]; s.vy[u] = v[1]; s.vz[u] = v[2];
  }
  return s;
}
__device__ __forceinline__ V3 sup_eval(const SupGeom& s, V3 dir, gcf mesh_vert, int lane) {
  const int t = s.t;
  const V3 h = s.h, ld = mtv(s.R, dir);
  V3 lp = v3(0, 0, 0);
  if (t == G_BOX) lp = v3(ld.x >= 0 ? h.x : -h.x, ld.y >= 0 ? h.y : -h.y, ld.z >= 0 ? h.z : -h.z);
  else if (t == G_SPHERE) { float n = norm(ld); if (n > FMIN) lp = ld * (h.x / n); }
  else if (t == G_CYLINDER) {
    float n = sqrtf(ld.x * ld.x + ld.y * ld.y);
    if (n > FMIN) { lp.x = ld.x / n * h.x; lp.y = ld.y / n * h.x; }
    lp.z = ld.z >= 0 ? h.z : -h.z;
  } else if (t == G_CAPSULE) {
    float n = norm(ld);
    if (n > FMIN) lp = ld * (h.x / n);
    lp.z += ld.z >= 0 ? (h.z - h.x) : -(h.z - h.x);
  } else if (t == G_ELLIPSOID) {
    V3 tt = v3(ld.x * h.x, ld.y * h.y, ld.z * h.z);
    float n = norm(tt);
    if (n > FMIN) lp = v3(tt.x / n * h.x, tt.y / n * h.y, tt.z / n * h.z);
  } else if (t == G_MESH) {
    const int adr = s.adr, num = s.num;
    float bv = -3.0e38f, bx = 0.f, by = 0.f, bz = 0.f;
    int bi = 0x7fffffff;
    if (s.regs) {
      // the lane's four vertices: all four projections first, then the first-maximum rule as selects (`if` blocks compiled to four exec-masked regions in a
      // row, each a VALU -> SALU -> VALU round trip on the critical path of the run)
      float val[4];
#pragma unroll
      for (int u = 0; u < 4; u++) val[u] = (64 * u + lane < num) ? s.vx[u] * ld.x + s.vy[u] * ld.y + s.vz[u] * ld.z : -3.0e38f;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const bool take = val[u] > bv;   // strictly greater: the lowest index keeps a tie, as the serial scan does
        bv = take ? val[u] : bv; bi = take ? 64 * u + lane : bi; bx = take ? s.vx[u] : bx; by = take ? s.vy[u] : by; bz = take ? s.vz[u] : bz;
      }
    } else {
      for (int base = 0; base < num; base += 256) {
        float vx[4], vy[4], vz[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int i = base + 64 * u + lane;
          gcf v = mesh_vert + 3 * (adr + (i < num ? i : 0));
          vx[u] = v[0]; vy[u] = v[1]; vz[u] = v[2];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int i = base + 64 * u + lane;
          const float val = vx[u] * ld.x + vy[u] * ld.y + vz[u] * ld.z;
          const bool take = i < num && val > bv; bv = take ? val : bv; bi = take ? i : bi; bx = take ? vx[u] : bx; by = take ? vy[u] : by; bz = take ? vz[u] : bz;
        }
      }
    }
    // wave arg-max, lowest index wins ties (same rule as geom_support)
    const float mx = wave_max(bv);
    const u64 tie = __ballot(bv == mx);
    int win = __ffsll((long long)tie) - 1;
    if (__popcll(tie) > 1) win = wave_min_i(bv == mx ? bi : 0x7fffffff) & 63;
    win = uni(win);
    lp = v3(bcast(bx, win), bcast(by, win), bcast(bz, win));
  }
  return s.p + mv(s.R, lp);
}

// ------------------------------------------------------------------------------------------------------------
// per-lane model constants, loaded once per launch and kept in registers for all substeps
// ------------------------------------------------------------------------------------------------------------
struct LaneConst {
  // body role (lane b < nbody)
  int part, part4, binfo;
  unsigned bdofs;
  V3 bpos; Q4 bquat; V3 jpos, jaxis; float q0;
  V3 ipos; Q4 iquat; float mass; V3 inertia;
  // dof role (lane i < nv)
  int dinfo;
  float damping;
  // joint-limit constants of the dof's joint (hinge / slide)
  float jr0, jr1, jmargin, jsr0, jsr1, jsi0, jsi1, jsi2, jsi3, jsi4, dinvw;
  // geom role (lane g < ncg)
  int ginfo;
  V3 gp; Q4 gq; V3 grc;
  // site role
  int sbody; V3 sp; Q4 sq;
  // actuator role (lane a < nu)
  int ainfo;
  float agear, again, ab0, ab1, ab2, acr0, acr1, afr0, afr1;
  // candidate pairs p = lane + 64 t
  int pair[10];
  unsigned mfbits;
  // built-in controller, lane i = arm joint i / gripper actuator i
  int cq, cd, ca, cga;
  float cgs;
};

// ------------------------------------------------------------------------------------------------------------
// the simulator (all methods are wave-cooperative: every lane of the env's wavefront calls them)
// ------------------------------------------------------------------------------------------------------------
template <class SM>
struct Sim {
  const DModel& m;
  const float* fp;
  int lane;
  // Constant blocks (read views): `cm` is the block every env shares (built from the shared float table), `ce` this env's own block (== cm while
  // no float-table field has per-env values).  A field is read from `ce` only if one of the float-table fields it is derived from has been
  // overridden (m.fenv, a uniform select per row): with per-episode object sizes or a partial domain randomisation almost every row still comes
  // from the shared block, which stays resident in the CU's L1 because all its wavefronts read the same lines.
  cmr_t cm, ce;
  cmw_t cw = nullptr;                              // write view while prepare_constants() builds a block
#define FM(x) (1ull << FO_##x)
  static constexpr u64 MK_gst = FM(cg_size) | FM(cg_aabb) | FM(cg_rbound) | FM(cg_margin), MK_gcap = FM(cg_capsule),
                       MK_gpar = FM(cg_friction) | FM(cg_solref) | FM(cg_solimp) | FM(cg_solmix) | FM(cg_gap), MK_arm = FM(dof_armature),
                       MK_fricRB = FM(dof_solref) | FM(dof_solimp) | FM(dof_invweight0) | FM(opt), MK_fricFl = FM(dof_frictionloss),
                       MK_biw = FM(body_invweight0), MK_opt = FM(opt);
  u64 fenv;
  long long ceoff;   // byte offset from the shared block to this env's block (an integer select: selecting between two POINTERS made the compiler
                     // keep Sim -- and with it a copy of DModel -- in the private segment)
  __device__ __forceinline__ cmr_t cmf(u64 mask) const {
    const long long off = (fenv & mask) ? ceoff : 0ll;
    return (cmr_t)((const char __attribute__((address_space(1)))*)cm + off);
  }
  static constexpr bool JG = SM::JG_;
  gwf Jg = nullptr;                                         // JG builds: this env's constraint Jacobian [NEFC][JS] in global memory (DBatch.jg)
  static constexpr bool MG = SM::MG_;
  gwf Mg = nullptr;                                         // MG builds: this env's mass matrix [NV][NVP] in global memory (behind J in DBatch.jg)
  __device__ __forceinline__ float Mrd(int i) const { if constexpr (MG) return Mg[i]; else return sm.M[i]; }
  __device__ __forceinline__ void Mwr(int i, float v) const { if constexpr (MG) Mg[i] = v; else sm.M[i] = v; }
  static constexpr bool CG = SM::CG_;
  // CG builds: this env's contact frames / material parameters in global memory, behind M in DBatch.jg (no member of its own: the layout of Sim, and with it the
  // register allocation of the builds that do not use it, stays what it was)
  __device__ __forceinline__ gwf Cgp() const { if constexpr (MG) return Mg + SM::NV_ * SM::NVP; else return Jg + SM::NEFC_ * SM::JS_; }   // behind M where the build keeps M there, behind J otherwise (fused wide body)
  __device__ __forceinline__ float cframe_rd(int i) const { if constexpr (CG) return Cgp()[SM::CG_FRAME_ + i]; else return sm.cframe[i]; }
  __device__ __forceinline__ float cfri_rd(int i) const { if constexpr (CG) return Cgp()[SM::CG_FRI_ + i]; else return sm.cfri[i]; }
  __device__ __forceinline__ float csolimp_rd(int i) const { if constexpr (CG) return Cgp()[SM::CG_SOLIMP_ + i]; else return sm.csolimp[i]; }
  __device__ __forceinline__ float csolref_rd(int i) const { if constexpr (CG) return Cgp()[SM::CG_SOLREF_ + i]; else return sm.csolref[i]; }
  __device__ __forceinline__ float cmargin_rd(int i) const { if constexpr (CG) return Cgp()[SM::CG_MARGIN_ + i]; else return sm.cmargin[i]; }
  __device__ __forceinline__ V3 cframe_ax(int i) const { return v3(cframe_rd(i), cframe_rd(i + 1), cframe_rd(i + 2)); }
  // the narrow phase's stores of the contact block have left the wavefront before another lane reads them (LDS build: the SYNC() sites between the phases do)
  __device__ __forceinline__ void csync() const {
    if constexpr (CG) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
  }
  __device__ __forceinline__ float Jrd(int i) const { if constexpr (JG) return Jg[i]; else return sm.J[i]; }
  __device__ __forceinline__ void Jwr(int i, float v) const { if constexpr (JG) Jg[i] = v; else sm.J[i] = v; }
  // J written by some lanes, read by others of the same wavefront: LDS needs the wavefront fence of SYNC(); global memory needs the stores to have left the
  // wavefront (workgroup scope: s_waitcnt vmcnt(0); the CU's L1 is coherent for its own waves)
  __device__ __forceinline__ void jsync() const {
    if constexpr (JG) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
    else SYNC();
  }
  float __attribute__((address_space(1)))* mprc = nullptr;  // this env's narrow-phase warm-start record [npair][MPRC] in global memory (DBatch.mprc), or null
  bool mpr_portal = true;                                   // contacts leave their portal directions in the record (flag 2)
  float __attribute__((address_space(1)))* cst = nullptr;   // this env's controller-state record in global memory (slots >= RSIM_CS_LDS are used in place)
  Prof pf;
  int ovf = 0;   // contacts / constraint rows this launch had to drop for lack of capacity (RSIM_OVERFLOW); MuJoCo's nconmax = 5000 never truncates
  float near_sep = 3.0e38f;   // smallest separation (m) along a separating direction any convex pair of the LAST substep's narrow phase ended on: how close the env is to a
                              // contact that does not exist yet (step_body: dispatch-order hint, DModel.near_thresh)
  int con_raw = 0, need_con = 0, need_efc = 0;   // contacts this substep's narrow phase found (before the capacity clamp); largest contact / row demand of any substep of this launch (RSIM_CAP_NEED)
  float opt_h, opt_density, opt_viscosity, opt_impratio;
  V3 opt_grav, opt_wind;
  static constexpr int NVP = SM::NVP;
  static constexpr int NV16 = SM::NV_;
  static constexpr int CD = SM::CD_;
  static constexpr int JS = SM::JS_, CS6 = SM::CS6_, FS = SM::FS_;
  static constexpr int SM_NB = SM::NB_;
  static constexpr bool FAST = SM::NV_ == 16;   // one-tile dense algebra: 16 x 16 MFMA products + register Cholesky
  static constexpr bool POLISH = SM::NV_ >= 48;  // the fp64 polish behind the wide solver is compiled for the configurations that run it (64 x 48 / 64 x 64 and their tiers: rsim_api.cpp sets DModel.newton_refine for those only)
  static constexpr bool TREE = SM::TREE_TILE_;  // tree products (CRBA composite inertias, RNE sums) as incidence-matrix MFMAs
  static constexpr int NPT = SM::NPT_;          // rows of 64 candidate pairs
  static constexpr int NROOT = SM::NROOT_;
  static constexpr int NSLOT = SM::NEFC_ / 64;  // constraint rows per lane (row r lives in lane r & 63, slot r >> 6)
  static constexpr int NEFCAP = SM::NEFC_;
  static constexpr bool TENDONS = SM::TENDONS_;
  static constexpr bool TWO_ARMS = SM::NB_ > 32;   // a second OSC arm part is compiled into the 64-body configurations only (bimanual robots need them anyway)
  typedef typename SM::dmask_t dmask_t;
  __device__ __forceinline__ dmask_t dmask_load(int tab, int i) const {   // dof bit mask i of an int-table entry stored as two 32-bit words
    if constexpr (sizeof(dmask_t) == 8) return mask2(tab, i); else return (dmask_t)(unsigned)IT(tab, 2 * i);
  }
  static constexpr int NT = SM::NV_ / 16;       // 16-dof tiles per dimension of the dense nv x nv products

  __device__ Sim(const DModel& m_, const float* fp_, int lane_, unsigned long long* prof, const void* cm_, const void* ce_) : m(m_), fp(fp_), lane(lane_), cm((cmr_t)cm_), ce((cmr_t)ce_), fenv(m_.fenv), ceoff((const char*)ce_ - (const char*)cm_) { pf.p = prof; pf.lane = lane_; pf.t0 = 0; }
  // Phase boundary for the register allocator: everything derived from the lane id (LDS addresses lane * stride, role predicates lane < n as
  // 64-bit masks) is loop invariant, so the compiler hoists all of it out of the substep loop and keeps it alive across every phase -- some
  // hundred registers at the pressure peaks.  A fresh (opaque) lane id per phase confines those values to the phase that uses them.
  __device__ __forceinline__ void phase() {
#ifndef RSIM_NO_PHASE_LANE
    lane = opaque_lane(lane); pf.lane = lane;
#endif
    // the same for the field-override mask: with a loop-invariant `fenv` every `bit ? env table : shared table` select of FP() -- some seventy
    // 64-bit pointers -- is computed before the substep loop and parked in spilled SGPRs (v_writelane / v_readlane) for the whole kernel
#ifdef RSIM_PHASE_FENV
    asm volatile("" : "+s"(fenv));
#endif
  }
  // global stores of this wavefront's lanes -> visible to its other lanes' loads (controller-state tail, constant block after an episode reset)
  __device__ __forceinline__ void gsync() const { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
  __device__ __forceinline__ void cst_sync() const { if (m.ctrl.cs_size > RSIM_CS_LDS) gsync(); }

  // articulated-tree slot of a root body (NROOT = static tree, COM unused / zero)
  __device__ __forceinline__ int root_slot(int rootbody) const {
    int sl = NROOT;
#pragma unroll
    for (int r = 0; r < NROOT; r++) if (r < m.ndynroot && m.dynroot[r] == rootbody) sl = r;
    return sl;
  }
  __device__ __forceinline__ u64 mask2(int tab, int i) const { return (u64)(uint32_t)IT(tab, 2 * i) | ((u64)(uint32_t)IT(tab, 2 * i + 1) << 32); }

  // ---------------------------------------------------------------- per-lane constants <-> LDS (row = field, column = lane of the role)
  // MK = float-table fields the row is derived from (0: lane-table / controller constants, always shared)
  template <bool STORE, u64 MK = 0> __device__ __forceinline__ void kio(float& x, int idx) const { if (STORE) cw->kc[idx] = x; else x = (MK ? cmf(MK) : cm)->kc[idx]; }
  template <bool STORE, u64 MK = 0> __device__ __forceinline__ void kio(int& x, int idx) const { if (STORE) cw->kc[idx] = __builtin_bit_cast(float, x); else x = __builtin_bit_cast(int, (MK ? cmf(MK) : cm)->kc[idx]); }
  template <bool STORE, u64 MK = 0> __device__ __forceinline__ void kio(unsigned& x, int idx) const { if (STORE) cw->kc[idx] = __builtin_bit_cast(float, x); else x = __builtin_bit_cast(unsigned, (MK ? cmf(MK) : cm)->kc[idx]); }
  template <bool STORE, u64 MK = 0> __device__ __forceinline__ void kio(V3& x, int idx, int stride) const { kio<STORE, MK>(x.x, idx); kio<STORE, MK>(x.y, idx + stride); kio<STORE, MK>(x.z, idx + 2 * stride); }
  template <bool STORE, u64 MK = 0> __device__ __forceinline__ void kio(Q4& x, int idx, int stride) const { kio<STORE, MK>(x.w, idx); kio<STORE, MK>(x.x, idx + stride); kio<STORE, MK>(x.y, idx + 2 * stride); kio<STORE, MK>(x.z, idx + 3 * stride); }
  // STORE: called once by load_constants (lanes outside a role's width skip it); fetch: every lane reads its (wrapped) column and the
  // compiler drops the rows a phase does not use
  // ARM: which arm part's controller columns a fetch reads (OSC types: arm a keeps its joints in columns 8a .. 8a + 7 and reads them into lanes 0 .. 7)
  template <bool STORE, int ARM = 0> __device__ __forceinline__ void kxfer(LaneConst& K) const {
    int o = 0;
    {  // body role, NB columns
      const int l = lane & (SM_NB - 1), W = SM_NB;
      if (!STORE || lane < W) {
        kio<STORE>(K.part, o + 0 * W + l); kio<STORE>(K.part4, o + 1 * W + l); kio<STORE>(K.binfo, o + 2 * W + l); kio<STORE>(K.bdofs, o + 3 * W + l);
        kio<STORE, FM(body_pos)>(K.bpos, o + 4 * W + l, W); kio<STORE, FM(body_quat)>(K.bquat, o + 7 * W + l, W); kio<STORE, FM(jnt_pos)>(K.jpos, o + 11 * W + l, W); kio<STORE, FM(jnt_axis)>(K.jaxis, o + 14 * W + l, W);
        kio<STORE, FM(qpos0)>(K.q0, o + 17 * W + l); kio<STORE, FM(body_ipos)>(K.ipos, o + 18 * W + l, W); kio<STORE, FM(body_iquat)>(K.iquat, o + 21 * W + l, W); kio<STORE, FM(body_mass)>(K.mass, o + 25 * W + l);
        kio<STORE, FM(body_inertia)>(K.inertia, o + 26 * W + l, W);
      }
      o += 29 * W;
    }
    {  // dof role, NV columns
      const int W = NV16, l = lane < W ? lane : lane - W;   // 48 dof columns in the 64 x 48 configuration
      if (!STORE || lane < W) {
        kio<STORE>(K.dinfo, o + 0 * W + l); kio<STORE, FM(dof_damping)>(K.damping, o + 1 * W + l); kio<STORE, FM(jnt_range)>(K.jr0, o + 2 * W + l); kio<STORE, FM(jnt_range)>(K.jr1, o + 3 * W + l);
        kio<STORE, FM(jnt_margin)>(K.jmargin, o + 4 * W + l); kio<STORE, FM(jnt_solref)>(K.jsr0, o + 5 * W + l); kio<STORE, FM(jnt_solref)>(K.jsr1, o + 6 * W + l); kio<STORE, FM(jnt_solimp)>(K.jsi0, o + 7 * W + l);
        kio<STORE, FM(jnt_solimp)>(K.jsi1, o + 8 * W + l); kio<STORE, FM(jnt_solimp)>(K.jsi2, o + 9 * W + l); kio<STORE, FM(jnt_solimp)>(K.jsi3, o + 10 * W + l); kio<STORE, FM(jnt_solimp)>(K.jsi4, o + 11 * W + l);
        kio<STORE, FM(dof_invweight0)>(K.dinvw, o + 12 * W + l);
      }
      o += 13 * W;
    }
    {  // geom role, 32 or 64 columns
      const int l = lane & (SM::NGW_ - 1), W = SM::NGW_;
      if (!STORE || lane < W) { kio<STORE>(K.ginfo, o + l); kio<STORE, FM(cg_pos)>(K.gp, o + 1 * W + l, W); kio<STORE, FM(cg_quat)>(K.gq, o + 4 * W + l, W); kio<STORE, FM(cg_rcenter)>(K.grc, o + 8 * W + l, W); }
      o += 11 * W;
    }
    {  // site role, NS columns
      const int l = lane & (SM::NS_ - 1), W = SM::NS_;
      if (!STORE || lane < W) { kio<STORE>(K.sbody, o + l); kio<STORE, FM(site_pos)>(K.sp, o + 1 * W + l, W); kio<STORE, FM(site_quat)>(K.sq, o + 4 * W + l, W); }
      o += 8 * W;
    }
    {  // actuator role, 16 columns
      const int l = lane & 15, W = 16;
      if (!STORE || lane < W) {
        kio<STORE>(K.ainfo, o + l); kio<STORE, FM(act_gear)>(K.agear, o + 1 * W + l); kio<STORE, FM(act_gainprm)>(K.again, o + 2 * W + l); kio<STORE, FM(act_biasprm)>(K.ab0, o + 3 * W + l); kio<STORE, FM(act_biasprm)>(K.ab1, o + 4 * W + l);
        kio<STORE, FM(act_biasprm)>(K.ab2, o + 5 * W + l); kio<STORE, FM(act_ctrlrange)>(K.acr0, o + 6 * W + l); kio<STORE, FM(act_ctrlrange)>(K.acr1, o + 7 * W + l); kio<STORE, FM(act_forcerange)>(K.afr0, o + 8 * W + l); kio<STORE, FM(act_forcerange)>(K.afr1, o + 9 * W + l);
      }
      o += 10 * W;
    }
    {  // controller role, 16 columns (joint-space parts drive up to 16 joints; the OSC types use 8 per arm)
      const int l = ARM ? (lane & 7) + 8 * ARM : (lane & 15), W = 16;
      if (!STORE || lane < W) { kio<STORE>(K.cq, o + l); kio<STORE>(K.cd, o + W + l); kio<STORE>(K.ca, o + 2 * W + l); kio<STORE>(K.cga, o + 3 * W + l); kio<STORE>(K.cgs, o + 4 * W + l); }
      o += 5 * W;
    }
#pragma unroll
    for (int t = 0; t < NPT; t++) kio<STORE>(K.pair[t], o + 64 * t + lane);
    kio<STORE>(K.mfbits, o + 64 * NPT + lane);
  }
  template <int ARM = 0> __device__ __forceinline__ LaneConst fetchK() const { LaneConst K; kxfer<false, ARM>(K); return K; }

  // ---------------------------------------------------------------- once per launch: the few uniform options, LDS padding, resident hulls
  __device__ __forceinline__ void load_opt() {
    // wave-uniform options: made scalar (SGPR) explicitly -- as vector registers they stay live over the whole launch and push other values into the private segment
    cmr_t co = cmf(MK_opt);
    auto su = [](float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); };
    opt_h = su(co->opt[0]); opt_grav = v3(su(co->opt[1]), su(co->opt[2]), su(co->opt[3])); opt_density = su(co->opt[4]); opt_viscosity = su(co->opt[5]);
    opt_impratio = su(co->opt[6]); opt_wind = v3(su(co->opt[7]), su(co->opt[8]), su(co->opt[9]));
  }
  __device__ __forceinline__ void init_lds() {
    // zero the LDS regions whose padding lanes / columns are read but never written
    for (int e = lane; e < SM_NB * 10 + 16; e += 64) sm.cinert[e] = 0.f;
    for (int e = lane; e < NV16 * CS6; e += 64) sm.cdof[e] = 0.f;
    if constexpr (!FAST) { for (int e = lane; e < NEFCAP * JS; e += 64) Jwr(e, 0.f); if constexpr (JG) jsync(); }   // rows beyond the last written 16-row tile are read (and ignored) by the lanes that own no row: keep them finite
    if (lane < (NROOT + 1) * 3) sm.rootcom[lane] = 0.f;
    if constexpr (SM::HULLPOOL_ > 0) {
      // resident hull pool: one pass per pooled mesh, lane-parallel over its vertices
      for (int g = 0; g < m.ncg; g++) {
        const int pool = cm->ghull[g];
        if (pool < 0) continue;
        const int adr = cm->gmesh[g] & 0xffff, num = cm->gmesh[g] >> 16;
        gcf mvp = (gcf)m.mesh_vert + 3 * adr;
        for (int i = lane; i < num; i += 64) { sm.hull[pool + i] = mvp[3 * i]; sm.hull[SM::HULLPOOL_ + pool + i] = mvp[3 * i + 1]; sm.hull[2 * SM::HULLPOOL_ + pool + i] = mvp[3 * i + 2]; }
      }
    }
    SYNC();
  }

  // ---------------------------------------------------------------- when the model parameters of this env change: constants -> its global block
  // (k_prepare for all envs; the env's own wavefront after an on-device episode reset patched its float table)
  __device__ __forceinline__ void prepare_constants(cmw_t out) {
    cw = out;
    LaneConst K;
    gci lt = (gci)m.lt;
    K.part = lt[LT_part * 64 + lane]; K.part4 = lt[LT_part4 * 64 + lane]; K.binfo = lt[LT_binfo * 64 + lane]; K.bdofs = (unsigned)lt[LT_bdofs * 64 + lane];
    K.dinfo = lt[LT_dinfo * 64 + lane]; K.ginfo = lt[LT_ginfo * 64 + lane]; K.sbody = lt[LT_sinfo * 64 + lane]; K.ainfo = lt[LT_ainfo * 64 + lane];
#pragma unroll
    for (int t = 0; t < NPT; t++) K.pair[t] = lt[(t < 3 ? LT_pair0 + t : (t < 5 ? LT_pair3 + (t - 3) : LT_pair5 + (t - 5))) * 64 + lane];
    K.mfbits = (unsigned)lt[LT_mfbits * 64 + lane];
    opt_h = FP(FO_opt, 0);   // row_scalars() below needs the timestep
    if (lane < 10) cw->opt[lane] = FP(FO_opt, lane);
    const int nb = m.nbody, nv = m.nv;
    {  // body role
      const int b = lane < nb ? lane : 0;
      K.bpos = ld3(&FP(FO_body_pos, 3 * b)); K.bquat = ldq(&FP(FO_body_quat, 4 * b));
      K.ipos = ld3(&FP(FO_body_ipos, 3 * b)); K.iquat = ldq(&FP(FO_body_iquat, 4 * b));
      K.mass = FP(FO_body_mass, b); K.inertia = ld3(&FP(FO_body_inertia, 3 * b));
      const int jt = K.binfo & 15;
      const int j = jt != 15 ? IT(IO_body_jntadr, b) : 0;
      K.jpos = ld3(&FP(FO_jnt_pos, 3 * j)); K.jaxis = ld3(&FP(FO_jnt_axis, 3 * j));
      K.q0 = FP(FO_qpos0, (K.binfo >> 4) & 255);
      if (lane >= nb) { K.part = 0; K.part4 = 0; K.binfo = 15; K.bdofs = 0; K.mass = 0.f; }
      if (lane < SM_NB) { cw->biw[2 * lane] = lane < nb ? FP(FO_body_invweight0, 2 * b) : 0.f; cw->biw[2 * lane + 1] = lane < nb ? FP(FO_body_invweight0, 2 * b + 1) : 0.f;
                          cw->bdofs[lane] = lane < nb ? dmask_load(IO_body_dofmask, b) : (dmask_t)0; cw->broot[lane] = root_slot((K.binfo >> 20) & 255); }
    }
    {  // dof role
      const int i = lane < nv ? lane : 0, j = (K.dinfo >> 18) & 255;
      K.damping = lane < nv ? FP(FO_dof_damping, i) : 0.f;
      K.jr0 = FP(FO_jnt_range, 2 * j); K.jr1 = FP(FO_jnt_range, 2 * j + 1); K.jmargin = FP(FO_jnt_margin, j);
      K.jsr0 = FP(FO_jnt_solref, 2 * j); K.jsr1 = FP(FO_jnt_solref, 2 * j + 1);
      K.jsi0 = FP(FO_jnt_solimp, 5 * j); K.jsi1 = FP(FO_jnt_solimp, 5 * j + 1); K.jsi2 = FP(FO_jnt_solimp, 5 * j + 2); K.jsi3 = FP(FO_jnt_solimp, 5 * j + 3); K.jsi4 = FP(FO_jnt_solimp, 5 * j + 4);
      K.dinvw = FP(FO_dof_invweight0, i);
      if (lane < NV16) {
        cw->arm[lane] = lane < nv ? FP(FO_dof_armature, i) : 1.0f;
        // friction-loss row of this dof: position term is identically 0, so R, the velocity gain and the limit are constants
        const float fl = lane < nv ? FP(FO_dof_frictionloss, i) : 0.f;
        float solref[2] = {FP(FO_dof_solref, 2 * i), FP(FO_dof_solref, 2 * i + 1)}, solimp[5];
        for (int k = 0; k < 5; k++) solimp[k] = FP(FO_dof_solimp, 5 * i + k);
        float R, Bd, Kimp;
        row_scalars(0.f, 0.f, solref, solimp, K.dinvw, R, Bd, Kimp);
        cw->fricR[lane] = R; cw->fricB[lane] = Bd; cw->fricFl[lane] = fl;
      }
      if (lane >= nv) K.dinfo = 0;
      if constexpr (!TREE) {
        if (lane < NV16) { cw->dmask_anc[lane] = lane < nv ? dmask_load(IO_dof_ancmask, i) : (dmask_t)0; cw->dmask_cvel[lane] = lane < nv ? dmask_load(IO_dof_cvelmask, i) : (dmask_t)0; }
        if (lane < SM_NB) cw->bmask_anc[lane] = lane < nb ? mask2(IO_body_ancmask, lane) : 0ull;
        if (m.fenv == 0) {   // the shared block (topology only; the env blocks never hold these)
          constexpr int KB = SM_NB / 4, NBT = SM_NB / 16, KV = NV16 / 4;
          const int r = lane & 15, q = lane >> 4;
          u64 ms = 0, mb = 0, mc = 0;
          for (int t = 0; t < NT; t++) {
            const int di = 16 * t + r;
            if (di >= nv) continue;
            const int bi = IT(IO_dof_bodyid, di);
            const u64 cv = (u64)dmask_load(IO_dof_cvelmask, di);
            for (int c = 0; c < KB; c++) { const int d = 4 * c + q; if (d >= 1 && d < nb && ((mask2(IO_body_ancmask, d) >> bi) & 1ull)) ms |= 1ull << (t * KB + c); }
            for (int c = 0; c < KV; c++) { const int j = 4 * c + q; if (j < nv && ((cv >> j) & 1ull)) mc |= 1ull << (t * KV + c); }
          }
          for (int t = 0; t < NBT; t++) {
            const int b = 16 * t + r;
            if (b >= nb) continue;
            const u64 bd = (u64)dmask_load(IO_body_dofmask, b);
            for (int c = 0; c < KV; c++) { const int j = 4 * c + q; if (j < nv && ((bd >> j) & 1ull)) mb |= 1ull << (t * KV + c); }
          }
          cw->msub[lane] = ms; cw->mbd[lane] = mb; cw->mcv[lane] = mc;
        }
      }
    }
    {  // geom role
      const int g = lane < m.ncg ? lane : 0;
      K.gp = ld3(&FP(FO_cg_pos, 3 * g)); K.gq = ldq(&FP(FO_cg_quat, 4 * g)); K.grc = ld3(&FP(FO_cg_rcenter, 3 * g));
      if (lane < m.ncg) {
        const int t = (K.ginfo >> 8) & 15;
        const V3 sz = ld3(&FP(FO_cg_size, 3 * g));
        V3 h = sz, c = v3(0, 0, 0);
        if (t == G_MESH) { c = ld3(&FP(FO_cg_aabb, 6 * g)); h = ld3(&FP(FO_cg_aabb, 6 * g + 3)); }
        else if (t == G_SPHERE) h = v3(sz.x, sz.x, sz.x);
        else if (t == G_CAPSULE) h = v3(sz.x, sz.x, sz.x + sz.y);
        else if (t == G_CYLINDER) h = v3(sz.x, sz.x, sz.y);
        float __attribute__((address_space(1)))* o = cw->gst + 8 * g;
        o[0] = h.x; o[1] = h.y; o[2] = h.z; o[3] = c.x; o[4] = c.y; o[5] = c.z; o[6] = FP(FO_cg_rbound, g); o[7] = FP(FO_cg_margin, g);
        cw->gtype[g] = t; cw->gbody[g] = K.ginfo & 255;
        for (int k = 0; k < 8; k++) cw->gcap[8 * g + k] = FP(FO_cg_capsule, 8 * g + k);
        cw->gcp[g] = IT(IO_cg_condim, g) | (IT(IO_cg_priority, g) << 8);
        cw->gmesh[g] = IT(IO_cg_meshadr, g) | (IT(IO_cg_meshnum, g) << 16);
        {   // configurations with a smaller (or no) resident pool scan the hulls that do not fit from global memory
          const int hs = lt[LT_ghull * 64 + lane];
          cw->ghull[g] = (hs >= 0 && hs + IT(IO_cg_meshnum, g) <= SM::HULLPOOL_) ? hs : -1;
        }
        float __attribute__((address_space(1)))* gp = cw->gpar + 12 * g;
        for (int k = 0; k < 3; k++) gp[k] = FP(FO_cg_friction, 3 * g + k);
        gp[3] = FP(FO_cg_solref, 2 * g); gp[4] = FP(FO_cg_solref, 2 * g + 1);
        for (int k = 0; k < 5; k++) gp[5 + k] = FP(FO_cg_solimp, 5 * g + k);
        gp[10] = FP(FO_cg_solmix, g); gp[11] = FP(FO_cg_gap, g);
      }
    }
    {  // site role
      const int k = lane < m.nsite ? lane : 0;
      K.sp = ld3(&FP(FO_site_pos, 3 * k)); K.sq = ldq(&FP(FO_site_quat, 4 * k));
    }
    {  // actuator role
      const int a = lane < m.nu ? lane : 0;
      K.agear = FP(FO_act_gear, a); K.again = FP(FO_act_gainprm, 3 * a);
      K.ab0 = FP(FO_act_biasprm, 3 * a); K.ab1 = FP(FO_act_biasprm, 3 * a + 1); K.ab2 = FP(FO_act_biasprm, 3 * a + 2);
      K.acr0 = FP(FO_act_ctrlrange, 2 * a); K.acr1 = FP(FO_act_ctrlrange, 2 * a + 1); K.afr0 = FP(FO_act_forcerange, 2 * a); K.afr1 = FP(FO_act_forcerange, 2 * a + 1);
    }
    {  // controller role (kernel-argument arrays are only ever indexed with compile-time constants: no private copy)
      const DCtrl& c = m.ctrl;
      K.cq = seli(c.qpos_idx, lane); K.cd = seli(c.dof_idx, lane); K.ca = seli(c.act_idx, lane);
      K.cga = seli(c.grip_act, lane); K.cgs = sel(c.grip_sign, lane);
    }
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < RSIM_MAXDYNROOT; r++) cw->dynroot[r] = m.dynroot[r];   // compile-time indices
    }
    kxfer<true>(K);
  }

  // ---------------------------------------------------------------- kinematics: pointer jumping over the body tree
  // every body composes its local transform with the transform of its 2^r-th ancestor in round r (log2(depth) rounds
  // instead of a serial root-to-leaf sweep); lane b = body b.  Returns this lane's world frame in registers.
  __device__ __forceinline__ void kinematics(V3& xp, Q4& xq) {
    const LaneConst K = fetchK();
    const int b = lane;
    const int jt = K.binfo & 15, qa = (K.binfo >> 4) & 255;
    V3 lp = K.bpos;
    Q4 lq = K.bquat;
    if (jt == JNT_FREE) {
      lp = ld3(sm.qpos + qa);
      lq = qnorm(ldq(sm.qpos + qa + 3));
    } else if (jt == JNT_HINGE) {
      float sn, cs;
      sincos_f(0.5f * (sm.qpos[qa] - K.q0), sn, cs);
      const Q4 ql = {cs, K.jaxis.x * sn, K.jaxis.y * sn, K.jaxis.z * sn};
      lq = qmul(K.bquat, ql);
      lp = K.bpos + qrot(K.bquat, K.jpos) - qrot(lq, K.jpos);
    } else if (jt == JNT_SLIDE) {
      lp = K.bpos + qrot(K.bquat, K.jaxis) * (sm.qpos[qa] - K.q0);
    }
    if (b == 0) { lp = v3(0, 0, 0); lq.w = 1.f; lq.x = lq.y = lq.z = 0.f; }
    SUBMARK_U(RP_X4);
    for (int r = 0; r < m.kin_rounds; r++) {
      if (b < SM_NB) { st3(sm.xpos + 3 * b, lp); stq(sm.xquat + 4 * b, lq); }
      SYNC();
      const int p = r < 4 ? (K.part >> (8 * r)) & 255 : K.part4;
      const V3 pp = ld3(sm.xpos + 3 * p);
      const Q4 pq = ldq(sm.xquat + 4 * p);
      SYNC();
      lp = pp + qrot(pq, lp);
      lq = qmul(pq, lq);
    }
    lq = qnorm(lq);
    xp = lp; xq = lq;
    if (b < SM_NB) { st3(sm.xpos + 3 * b, lp); stq(sm.xquat + 4 * b, lq); }
    SYNC();
    SUBMARK_U(RP_X5);
  }

  // colliding geom and site frames (lane g = geom g, lane k = site k)
  __device__ __forceinline__ void geom_site_frames() {
    const LaneConst K = fetchK();
    if (lane < m.ncg) {
      const int g = lane, gb = K.ginfo & 255;
      const Q4 bq = ldq(sm.xquat + 4 * gb);
      const V3 gp = ld3(sm.xpos + 3 * gb) + qrot(bq, K.gp);
      const M3 Rg = q2m(qnorm(qmul(bq, K.gq)));
      st3(sm.gpos + 3 * g, gp);
      stm(sm.gmat + 9 * g, Rg);
      st3(sm.gcen + 3 * g, gp + mv(Rg, K.grc));
    }
    if (lane < m.nsite) {
      const int k = lane, sb = K.sbody;
      const Q4 bq = ldq(sm.xquat + 4 * sb);
      st3(sm.spos + 3 * k, ld3(sm.xpos + 3 * sb) + qrot(bq, K.sp));
      stm(sm.smat + 9 * k, q2m(qnorm(qmul(bq, K.sq))));
    }
    SYNC();
  }

  // ---------------------------------------------------------------- subtree COM of the articulated trees, cinert, cdof
  __device__ __forceinline__ void com_pos(V3 xp, Q4 xq) {
    const LaneConst K = fetchK();
    const int b = lane, nb = m.nbody;
    const int root = (K.binfo >> 20) & 255, jt = K.binfo & 15, da = (K.binfo >> 12) & 255;
    const bool moving = (K.binfo >> 28) & 1;
    const M3 R = q2m(xq);
    const V3 xip = xp + mv(R, K.ipos);
    V3 com = xip;
#pragma unroll
    for (int r = 0; r < NROOT; r++) {
      if (r >= m.ndynroot) break;
      int rb;
      if constexpr (NROOT > 4) rb = cm->dynroot[r]; else rb = m.dynroot[r];
      const float w = (b < nb && root == rb) ? K.mass : 0.f;
      const float sw = wave_sum(w), sx = wave_sum(w * xip.x), sy = wave_sum(w * xip.y), sz = wave_sum(w * xip.z);
      const float iw = sw > 1e-15f ? 1.0f / sw : 0.f;
      const V3 cr = v3(sx * iw, sy * iw, sz * iw);
      if (root == rb) com = cr;
      if (lane == 0) st3(sm.rootcom + 3 * r, cr);
    }
    if (b < nb && moving) {
      const M3 Ri = q2m(qmul(xq, K.iquat));
      const V3 I = K.inertia;
      const float mass = K.mass;
      const V3 off = xip - com;
      float Iw[6];  // xx yy zz xy xz yz of R diag(I) R^T
      Iw[0] = Ri.m[0] * I.x * Ri.m[0] + Ri.m[1] * I.y * Ri.m[1] + Ri.m[2] * I.z * Ri.m[2];
      Iw[1] = Ri.m[3] * I.x * Ri.m[3] + Ri.m[4] * I.y * Ri.m[4] + Ri.m[5] * I.z * Ri.m[5];
      Iw[2] = Ri.m[6] * I.x * Ri.m[6] + Ri.m[7] * I.y * Ri.m[7] + Ri.m[8] * I.z * Ri.m[8];
      Iw[3] = Ri.m[0] * I.x * Ri.m[3] + Ri.m[1] * I.y * Ri.m[4] + Ri.m[2] * I.z * Ri.m[5];
      Iw[4] = Ri.m[0] * I.x * Ri.m[6] + Ri.m[1] * I.y * Ri.m[7] + Ri.m[2] * I.z * Ri.m[8];
      Iw[5] = Ri.m[3] * I.x * Ri.m[6] + Ri.m[4] * I.y * Ri.m[7] + Ri.m[5] * I.z * Ri.m[8];
      const float d2 = dot(off, off);
      float* ci = sm.cinert + 10 * b;
      ci[0] = Iw[0] + mass * (d2 - off.x * off.x);
      ci[1] = Iw[1] + mass * (d2 - off.y * off.y);
      ci[2] = Iw[2] + mass * (d2 - off.z * off.z);
      ci[3] = Iw[3] - mass * off.x * off.y;
      ci[4] = Iw[4] - mass * off.x * off.z;
      ci[5] = Iw[5] - mass * off.y * off.z;
      ci[6] = mass * off.x; ci[7] = mass * off.y; ci[8] = mass * off.z; ci[9] = mass;
    }
    if (b < nb && jt != 15) {
      float* cd = sm.cdof + CS6 * da;
      if (jt == JNT_FREE) {
        const V3 off = com - xp;
        for (int k = 0; k < 3; k++) {
          st3(cd + CS6 * k, v3(0, 0, 0)); st3(cd + CS6 * k + 3, v3(k == 0, k == 1, k == 2));
          const V3 ax = col(R, k);
          st3(cd + CS6 * (3 + k), ax); st3(cd + CS6 * (3 + k) + 3, cross(ax, off));
        }
      } else {
        const V3 ax = mv(R, K.jaxis);
        if (jt == JNT_SLIDE) { st3(cd, v3(0, 0, 0)); st3(cd + 3, ax); }
        else { const V3 off = com - (xp + mv(R, K.jpos)); st3(cd, ax); st3(cd + 3, cross(ax, off)); }
      }
    }
    SYNC();
  }

  // ---------------------------------------------------------------- CRBA on the matrix cores
  // composite inertia per dof  crbD = Sub x cinert           (Sub = subtree incidence, 16 x 32, 0/1 constants)
  // f_i = crbD_i * cdof_i ;  M = (m1 o F C^T) + (m2 o C F^T) + diag(armature)        (F, C = 16 x 6 stacks of f_i, cdof_i)
  // wide configurations: the same products tile by tile for any bodies x dofs (incidence_mfma below), factorisations on the LDS matrix
  // out tile t (rows 16 t .. 16 t + 15) = sum over k-steps c < kmax of Incidence(t, c) x B(c) on the matrix cores: the A operand is this lane's
  // incidence bit (Cmem::msub / mbd / mcv), the B operand of k-step c -- bop(c), row 4 c + (lane >> 4), column lane & 15 -- is read once and
  // serves every row tile.  Replaces the mask-guided lane loops of the wide configurations (one LDS round trip per body / dof and lane).
  template <int NTILE, int KS, class BOP>
  __device__ __forceinline__ void incidence_mfma(u64 mask, int kmax, BOP bop, v4f (&acc)[NTILE]) const {
    unsigned mt[NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; t++) { mt[t] = (unsigned)(mask >> (t * KS)) & ((KS < 32) ? ((1u << (KS & 31)) - 1u) : 0xFFFFFFFFu); acc[t] = v4f{0.f, 0.f, 0.f, 0.f}; }
    for (int c0 = 0; c0 < kmax; c0 += 2) {   // two k-steps per trip: both B operands in flight before the products
      const float b0 = bop(c0), b1 = bop(c0 + 1);
#pragma unroll
      for (int t = 0; t < NTILE; t++) {
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)(mt[t] & 1u), b0, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)((mt[t] >> 1) & 1u), b1, acc[t], 0, 0, 0);
        mt[t] >>= 2;
      }
    }
  }
  __device__ __forceinline__ void crb_composite_mfma() {
    const int nv = m.nv, nb = m.nbody, q = lane >> 4, r = lane & 15;
    constexpr int KB = SM_NB / 4;
    v4f acc[NT];
    const int kmax = ((nb + 7) >> 3) << 1;   // k-steps in pairs; rows nb .. NB - 1 of cinert are zero (init_lds) and their incidence bits clear
    incidence_mfma<NT, KB>(cm->msub[lane], kmax < KB ? kmax : KB, [&](int c) { return sm.cinert[(4 * c + q) * 10 + r]; }, acc);
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
      for (int v = 0; v < 4; v++) sm.u.c.crbD[(16 * t + 4 * q + v) * FS + r] = acc[t][v];
    SYNC();
    if (lane < NV16) {
      S6 f = {v3(0, 0, 0), v3(0, 0, 0)};
      if (lane < nv) f = mul_inert(sm.u.c.crbD + FS * lane, ld6(sm.cdof + CS6 * lane));
      float* o = sm.u.c.fpad + CS6 * lane;
      st3(o, f.a); st3(o + 3, f.l); o[6] = 0.f; o[7] = 0.f;
    }
    SYNC();
  }
  // M = m1 o (F C^T) + m2 o (C F^T) + diag(armature) tile by tile (F = composite-inertia forces, C = cdof, 6 components = two k-steps), masks
  // from the dof-ancestor sets; written to M and, in the same pass, to the factorisation's work matrix
  __device__ __forceinline__ void crb_mass_mfma() {
    const int nv = m.nv, q = lane >> 4, r = lane & 15;
    float fa[NT][2], ca[NT][2];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
      for (int kc = 0; kc < 2; kc++) { fa[t][kc] = sm.u.c.fpad[(16 * t + r) * CS6 + 4 * kc + q]; ca[t][kc] = sm.cdof[(16 * t + r) * CS6 + 4 * kc + q]; }
    dmask_t arow[NT][4], acol[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
      acol[t] = cm->dmask_anc[16 * t + r];
#pragma unroll
      for (int v = 0; v < 4; v++) arow[t][v] = cm->dmask_anc[16 * t + 4 * q + v];
    }
#pragma unroll
    for (int ti = 0; ti < NT; ti++)
#pragma unroll
      for (int tj = 0; tj < NT; tj++) {
        if (16 * ti >= nv || 16 * tj >= nv) {   // tile without dofs: identity padding
#pragma unroll
          for (int v = 0; v < 4; v++) { const int i = 16 * ti + 4 * q + v, j = 16 * tj + r; const float e = i == j ? cmf(MK_arm)->arm[i] : 0.f; Mwr(i * NVP + j, e); sm.L[i * NVP + j] = e; }
          continue;
        }
        v4f R1 = {0.f, 0.f, 0.f, 0.f}, R2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < 2; kc++) {
          R1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[ti][kc], ca[tj][kc], R1, 0, 0, 0);
          R2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[ti][kc], fa[tj][kc], R2, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 4; v++) {
          const int i = 16 * ti + 4 * q + v, j = 16 * tj + r;
          const bool a1 = (arow[ti][v] >> j) & 1, a2 = (acol[tj] >> i) & 1;
          float mij = a1 ? R1[v] : (a2 ? R2[v] : 0.f);
          if (i == j) mij += cmf(MK_arm)->arm[i];
          Mwr(i * NVP + j, mij); sm.L[i * NVP + j] = mij;
        }
      }
    if constexpr (MG) jsync();   // the stores of M have left the wavefront before any lane reads it back from global memory
    SYNC();
    SUBMARK_T(RP_X2);
    bchol_inplace<NVP>(sm.L, sm.invdiag, nv, lane);
    SUBMARK_T(RP_X3);
  }
  __device__ __forceinline__ void crb() {
    if constexpr (TREE) crb_composite_tile(); else crb_composite_mfma();
    SUBMARK_T(RP_X0);
    if constexpr (FAST) crb_mass_tile(); else crb_mass_mfma();
  }
  __device__ __forceinline__ void crb_composite_tile() {
    const LaneConst K = fetchK();
    const int nv = m.nv, q = lane >> 4, r = lane & 15;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 8; c++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bitf(K.mfbits, c), sm.cinert[(4 * c + q) * 10 + r], acc, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 4; v++) sm.u.c.crbD[(4 * q + v) * FS + r] = acc[v];
    SYNC();
    if (lane < NV16) {
      S6 f = {v3(0, 0, 0), v3(0, 0, 0)};
      if (lane < nv) f = mul_inert(sm.u.c.crbD + FS * lane, ld6(sm.cdof + CS6 * lane));
      float* o = sm.u.c.fpad + CS6 * lane;
      st3(o, f.a); st3(o + 3, f.l); o[6] = 0.f; o[7] = 0.f;
    }
    SYNC();
  }
  __device__ __forceinline__ void crb_mass_tile() {
    const LaneConst K = fetchK();
    const int nv = m.nv, q = lane >> 4, r = lane & 15;
    v4f R1 = {0.f, 0.f, 0.f, 0.f}, R2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < 2; kc++) {
      const float fa = sm.u.c.fpad[r * CS6 + 4 * kc + q], ca = sm.cdof[r * CS6 + 4 * kc + q];
      R1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, ca, R1, 0, 0, 0);
      R2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ca, fa, R2, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const int i = 4 * q + v;
      float mij = bitf(K.mfbits, 20 + v) * R1[v] + bitf(K.mfbits, 24 + v) * R2[v];
      if (i == r) mij += cmf(MK_arm)->arm[i];
      sm.M[i * NVP + r] = mij;
    }
    SYNC();
    {
      // factor of M for the smooth acceleration (the factor of M + hD for the integrator is made by euler(): keeping both cost 1 KB of LDS,
      // and with two wavefronts per SIMD the second dependent rsqrt / broadcast chain no longer needs this one to hide behind)
      float mr[NV16], minv[NV16];
#pragma unroll
      for (int k = 0; k < NV16; k++) mr[k] = sm.M[r * NVP + k];
      const int ro = opaque_lane(r);
      const float mown = rchol_factor_own<NV16>(mr, minv, ro);
      rchol_mask_lower<NV16>(mr, ro);   // stored strictly lower: rows and columns read back ready for rchol_solve_m
      if (lane < NV16) {
#pragma unroll
        for (int k = 0; k < NV16; k++) sm.L[lane * NVP + k] = mr[k];
        sm.invdiag[lane] = mown;
      }
    }
    SYNC();
  }

  // ---------------------------------------------------------------- velocity stage (RNE) on the matrix cores
  // cvel = BodyDof x (cdof qd) ; cdof_dot_i = cvel_before(i) x cdof_i ; cacc = BodyDof x (cdof_dot qd) - g ;
  // body wrench cf = I cacc + cvel x* I cvel (+ fluid) ; F = Sub x cf ; bias_i = cdof_i . F_i
  __device__ __forceinline__ void velocity(V3 xp, Q4 xq) {
    const LaneConst K = fetchK();
    const int nb = m.nbody, nv = m.nv, q = lane >> 4, r = lane & 15;
    float Bc[4];
    v4f a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
    if constexpr (TREE) {
#pragma unroll
    for (int c = 0; c < 4; c++) Bc[c] = r < 6 ? sm.cdof[(4 * c + q) * CS6 + r] * sm.qvel[4 * c + q] : 0.f;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(bitf(K.mfbits, 8 + c), Bc[c], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(bitf(K.mfbits, 12 + c), Bc[c], a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(bitf(K.mfbits, 16 + c), Bc[c], a2, 0, 0, 0);
    }
    if (r < 8) {
#pragma unroll
      for (int v = 0; v < 4; v++) { sm.u.v.cvel[(4 * q + v) * CS6 + r] = a0[v]; sm.u.v.cvel[(16 + 4 * q + v) * CS6 + r] = a1[v]; sm.u.v.cvb[(4 * q + v) * CS6 + r] = a2[v]; }
    }
    } else {
      // wide configuration: cvel = BodyDof x (cdof qd), cvb = Before x (cdof qd) as incidence products sharing the B operands
      constexpr int NBT = SM_NB / 16, KV = NV16 / 4;
      const int kmax = ((nv + 7) >> 3) << 1;
      auto bop = [&](int c) { const int i = 4 * c + q; return (r < 6 && i < nv) ? sm.cdof[i * CS6 + r] * sm.qvel[i] : 0.f; };
      v4f ab[NBT], av[NT];
      incidence_mfma<NBT, KV>(cm->mbd[lane], kmax < KV ? kmax : KV, bop, ab);
      incidence_mfma<NT, KV>(cm->mcv[lane], kmax < KV ? kmax : KV, bop, av);
      if (r < 8) {
#pragma unroll
        for (int t = 0; t < NBT; t++)
#pragma unroll
          for (int v = 0; v < 4; v++) sm.u.v.cvel[(16 * t + 4 * q + v) * CS6 + r] = ab[t][v];
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
          for (int v = 0; v < 4; v++) sm.u.v.cvb[(16 * t + 4 * q + v) * CS6 + r] = av[t][v];
      }
    }
    SYNC();
    SUBMARK_T(RP_X4);
    if (lane < NV16) {
      S6 cd = {v3(0, 0, 0), v3(0, 0, 0)};
      if (lane < nv && !((K.dinfo >> 8) & 1)) cd = cross_motion(ld6(sm.u.v.cvb + CS6 * lane), ld6(sm.cdof + CS6 * lane));
      float* o = sm.u.v.cdd + CS6 * lane;
      st3(o, cd.a); st3(o + 3, cd.l); o[6] = 0.f; o[7] = 0.f;
    }
    SYNC();
    if constexpr (TREE) {
#pragma unroll
    for (int c = 0; c < 4; c++) Bc[c] = r < 6 ? sm.u.v.cdd[(4 * c + q) * CS6 + r] * sm.qvel[4 * c + q] : 0.f;
    a0 = a1 = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; c++) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(bitf(K.mfbits, 8 + c), Bc[c], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(bitf(K.mfbits, 12 + c), Bc[c], a1, 0, 0, 0);
    }
    if (r < 8) {
#pragma unroll
      for (int v = 0; v < 4; v++) { sm.u.v.cacc[(4 * q + v) * CS6 + r] = a0[v]; sm.u.v.cacc[(16 + 4 * q + v) * CS6 + r] = a1[v]; }
    }
    } else {
      // cacc aliases cvb / cdd: all B operands are read (products issued) before any lane stores
      constexpr int NBT = SM_NB / 16, KV = NV16 / 4;
      const int kmax = ((nv + 7) >> 3) << 1;
      v4f ab[NBT];
      incidence_mfma<NBT, KV>(cm->mbd[lane], kmax < KV ? kmax : KV, [&](int c) { const int i = 4 * c + q; return (r < 6 && i < nv) ? sm.u.v.cdd[i * CS6 + r] * sm.qvel[i] : 0.f; }, ab);
      SYNC();
      if (r < 6) {
#pragma unroll
        for (int t = 0; t < NBT; t++)
#pragma unroll
          for (int v = 0; v < 4; v++) sm.u.v.cacc[(16 * t + 4 * q + v) * CS6 + r] = ab[t][v];
      }
    }
    SYNC();
    SUBMARK_T(RP_X5);
    if (lane < SM_NB) {
      const int b = lane;
      S6 zero = {v3(0, 0, 0), v3(0, 0, 0)};
      S6 frc = zero, flu = zero;
      if (b < nb && ((K.binfo >> 28) & 1)) {
        S6 ca = ld6(sm.u.v.cacc + CS6 * b);
        ca.l = ca.l - opt_grav;
        const S6 cv = ld6(sm.u.v.cvel + CS6 * b);
        frc = mul_inert(sm.cinert + 10 * b, ca) + cross_force(cv, mul_inert(sm.cinert + 10 * b, cv));
        const float mass = K.mass;
        if (mass >= 1e-15f && (opt_density > 0.f || opt_viscosity > 0.f)) {
          // inertia-box fluid model: force/torque at the body COM, folded into a spatial force about the tree COM
          const V3 I = K.inertia;
          const M3 R = q2m(qmul(xq, K.iquat));
          const float bx = sqrtf(fmaxf(1e-15f, I.y + I.z - I.x) / mass * 6.0f), by = sqrtf(fmaxf(1e-15f, I.x + I.z - I.y) / mass * 6.0f),
                      bz = sqrtf(fmaxf(1e-15f, I.x + I.y - I.z) / mass * 6.0f);
          const V3 off = xp + qrot(xq, K.ipos) - ld3(sm.rootcom + 3 * cm->broot[b]);
          const V3 gl = cv.l + cross(cv.a, off) - opt_wind;
          const V3 la = mtv(R, cv.a), ll = mtv(R, gl);
          V3 ft = v3(0, 0, 0), ff = v3(0, 0, 0);
          if (opt_viscosity > 0.f) {
            const float diam = (bx + by + bz) / 3.0f;
            ft = la * (-PI_F * diam * diam * diam * opt_viscosity);
            ff = ll * (-3.0f * PI_F * diam * opt_viscosity);
          }
          if (opt_density > 0.f) {
            ff.x -= 0.5f * opt_density * by * bz * fabsf(ll.x) * ll.x;
            ff.y -= 0.5f * opt_density * bx * bz * fabsf(ll.y) * ll.y;
            ff.z -= 0.5f * opt_density * bx * by * fabsf(ll.z) * ll.z;
            const float bx4 = bx * bx * bx * bx, by4 = by * by * by * by, bz4 = bz * bz * bz * bz;
            ft.x -= opt_density * bx * (by4 + bz4) * fabsf(la.x) * la.x / 64.0f;
            ft.y -= opt_density * by * (bx4 + bz4) * fabsf(la.y) * la.y / 64.0f;
            ft.z -= opt_density * bz * (bx4 + by4) * fabsf(la.z) * la.z / 64.0f;
          }
          const V3 gt = mv(R, ft), gf = mv(R, ff);
          flu.a = gt + cross(off, gf);
          flu.l = gf;
        }
      }
      float* o = sm.u.v.cf + FS * b;
      st3(o, frc.a); st3(o + 3, frc.l); st3(o + 6, flu.a); st3(o + 9, flu.l);
      o[12] = o[13] = o[14] = o[15] = 0.f;
    }
    SYNC();
    SUBMARK_T(RP_X6);
    if constexpr (TREE) {
    v4f af = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 8; c++) af = __builtin_amdgcn_mfma_f32_16x16x4f32(bitf(K.mfbits, c), sm.u.v.cf[(4 * c + q) * FS + r], af, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 4; v++) sm.u.v.F[(4 * q + v) * FS + r] = af[v];
    } else {
      // F = Sub x cf (F aliases cf: products first, then the stores)
      constexpr int KB = SM_NB / 4;
      const int kmax = ((nb + 7) >> 3) << 1;
      v4f af[NT];
      incidence_mfma<NT, KB>(cm->msub[lane], kmax < KB ? kmax : KB, [&](int c) { return sm.u.v.cf[(4 * c + q) * FS + r]; }, af);
      SYNC();
#pragma unroll
      for (int t = 0; t < NT; t++)
#pragma unroll
        for (int v = 0; v < 4; v++) sm.u.v.F[(16 * t + 4 * q + v) * FS + r] = af[t][v];
    }
    SYNC();
    SUBMARK_T(RP_X7);
    float tfrc = 0.f;
    if (TENDONS && m.ntendon) {
      // spring-damper on fixed-tendon lengths (mj_passive [3P]): force k (lo - len) / k (hi - len) outside the deadband minus damping on the
      // length rate, mapped onto the tendon's dofs by its coefficients; uniform loop over the tendons, lane = dof picks its own terms
      for (int t = 0; t < m.ntendon; t++) {
        const float k = FP(FO_tendon_stiffness, t), bd = FP(FO_tendon_damping, t);
        if (k <= 0.f && bd <= 0.f) continue;
        const int adr = IT(IO_tendon_adr, t), num = IT(IO_tendon_num, t);
        float len = 0.f, vel = 0.f, mine = 0.f;
        for (int w = 0; w < num; w++) {
          const float cf = FP(FO_wrap_prm, adr + w);
          const int dw = IT(IO_wrap_dof, adr + w);
          len = fmaf(cf, sm.qpos[IT(IO_wrap_qadr, adr + w)], len);
          vel = fmaf(cf, sm.qvel[dw], vel);
          mine += dw == lane ? cf : 0.f;
        }
        const float lo = FP(FO_tendon_lspring, 2 * t), hi = FP(FO_tendon_lspring, 2 * t + 1);
        const float frc = (len > hi ? k * (hi - len) : (len < lo ? k * (lo - len) : 0.f)) - bd * vel;
        tfrc = fmaf(mine, frc, tfrc);
      }
    }
    if (lane < nv) {
      const S6 cd = ld6(sm.cdof + CS6 * lane);
      const float* F = sm.u.v.F + FS * lane;
      sm.qfrc_bias[lane] = dot6(cd, ld6(F));
      sm.qfrc_passive[lane] = -K.damping * sm.qvel[lane] + dot6(cd, ld6(F + 6)) + tfrc;
    }
    SYNC();
  }

  // Acceleration-stage sensors of the compatibility path (mj_sensorAcc -> mj_rnePostConstraint [3P]; robots/robot.py:739-751, 795-815): the <force> and
  // <torque> sensors at the gripper's ft_frame site.  Runs after the constraint solve of the LAST substep, before the integrator, in the debug
  // build of the kernel only (k_step_dbg: B = 1 shim entries, forward()).  cvel and the velocity part of cacc are phase-local, so the velocity stage
  // is simply run again (the state has not changed); then lane = body:
  //   cacc_b += sum over the dofs that move b of cdof_i qacc_i;  f_b = cinert_b (cacc_b - g) + cvel_b x* (cinert_b cvel_b) - (contact wrenches on b),
  // all about the subtree COM of b's tree; lane = sensor sums f over the site body's subtree (ancestor masks) and expresses the force / the moment
  // about the site in the site frame.
  __device__ __forceinline__ void sensor_acc(V3 xp, Q4 xq, float* out) {
    velocity(xp, xq);
    const LaneConst K = fetchK();
    const int nb = m.nbody, nv = m.nv;
    S6 frc = {v3(0, 0, 0), v3(0, 0, 0)};
    if (lane < nb && ((K.binfo >> 28) & 1)) {
      const int b = lane;
      S6 ca = ld6(sm.u.v.cacc + CS6 * b);
      ca.l = ca.l - opt_grav;
      const u64 dm = (u64)dmask_load(IO_body_dofmask, b);
      for (int i = 0; i < nv; i++)
        if ((dm >> i) & 1ull) { const S6 cd = ld6(sm.cdof + CS6 * i); const float a = sm.qacc[i]; ca.a = ca.a + cd.a * a; ca.l = ca.l + cd.l * a; }
      const S6 cv = ld6(sm.u.v.cvel + CS6 * b);
      frc = mul_inert(sm.cinert + 10 * b, ca) + cross_force(cv, mul_inert(sm.cinert + 10 * b, cv));
      const V3 rc = ld3(sm.rootcom + 3 * cm->broot[b]);
      for (int c = 0; c < sm.ncon; c++) {
        const int ea = sm.cefc[c], dim = sm.cdim[c];
        if (ea < 0) continue;
        const int b1 = IT(IO_cg_bodyid, sm.cg1[c] & 255), b2 = IT(IO_cg_bodyid, sm.cg2[c] & 255);
        if (b1 != b && b2 != b) continue;
        V3 fw = v3(0, 0, 0), tw = v3(0, 0, 0);
        for (int k = 0; k < dim; k++) {
          const V3 ax = CG ? cframe_ax(9 * c + 3 * (k < 3 ? k : k - 3)) : ld3(sm.cframe + 9 * c + 3 * (k < 3 ? k : k - 3));
          if (k < 3) fw = fw + ax * sm.e_force[ea + k]; else tw = tw + ax * sm.e_force[ea + k];
        }
        const float sgn = (b2 == b ? 1.f : 0.f) - (b1 == b ? 1.f : 0.f);
        const V3 t = tw + cross(ld3(sm.cpos + 3 * c) - rc, fw);
        frc.a = frc.a - t * sgn; frc.l = frc.l - fw * sgn;
      }
    }
    SYNC();
    if (lane < nb) { float* o = sm.u.v.cf + FS * lane; st3(o, frc.a); st3(o + 3, frc.l); }
    SYNC();
    if (lane < m.nsensor) {
      const int type = IT(IO_sensor_type, lane), site = IT(IO_sensor_site, lane), adr = IT(IO_sensor_adr, lane), dim = IT(IO_sensor_adr, lane + 1) - adr;
      V3 r = v3(0, 0, 0);
      if ((type == 0 || type == 1) && site >= 0 && dim == 3) {
        const int b0 = IT(IO_site_bodyid, site);
        V3 wa = v3(0, 0, 0), wl = v3(0, 0, 0);
        for (int d = 1; d < nb; d++)
          if ((mask2(IO_body_ancmask, d) >> b0) & 1ull) { wa = wa + ld3(sm.u.v.cf + FS * d); wl = wl + ld3(sm.u.v.cf + FS * d + 3); }
        const M3 R = ldm(sm.smat + 9 * site);
        r = type == 0 ? mtv(R, wl) : mtv(R, wa - cross(ld3(sm.spos + 3 * site) - ld3(sm.rootcom + 3 * cm->broot[b0]), wl));
      }
      if (dim >= 1) out[adr] = r.x;
      if (dim >= 2) out[adr + 1] = r.y;
      if (dim >= 3) out[adr + 2] = r.z;
      for (int k = 3; k < dim; k++) out[adr + k] = 0.f;
    }
    SYNC();
  }

  // Jacobian column of world point p attached to body `b` for dof i: returns [jacr; jacp] or zero if i does not move b
  __device__ __forceinline__ S6 jac_col(int b, V3 p, int i) const {
    S6 z = {v3(0, 0, 0), v3(0, 0, 0)};
    if (!((cm->bdofs[b] >> i) & 1)) return z;
    S6 cd = ld6(sm.cdof + CS6 * i);
    V3 off = p - ld3(sm.rootcom + 3 * cm->broot[b]);
    S6 rr = {cd.a, cd.l + cross(cd.a, off)};
    return rr;
  }

  // ---------------------------------------------------------------- collision narrowphase
  // ---------------------------------------------------------------- collision
  __device__ __forceinline__ void make_frame(V3 n, float* frame) {
    n = normalized(n);
    V3 y = (n.y < 0.5f && n.y > -0.5f) ? v3(0, 1, 0) : v3(0, 0, 1);
    y = y - n * dot(n, y);
    y = normalized(y);
    V3 z = cross(n, y);
    st3(frame, n); st3(frame + 3, y); st3(frame + 6, z);
  }

  __device__ __forceinline__ V3 sup(const SupGeom& sg, V3 dir) { pf.count(RP_N_SUPPORT, 1); return sup_eval(sg, dir, (gcf)m.mesh_vert, lane); }
  __device__ __forceinline__ V3 support(int g, V3 dir, V3 org = {0.f, 0.f, 0.f}) { pf.count(RP_N_SUPPORT, 1); return geom_support<SM>(cm, cmf(MK_gst), g, dir, (gcf)m.mesh_vert, lane, org); }

  // contact parameters of a geom pair (MuJoCo's mixing rules: priority, solmix-weighted solref/solimp, max friction);
  // evaluated once per candidate pair, uniformly by every lane, from the per-geom table staged in LDS
  struct CPar { int dim; float solref[2], solimp[5], fr[3]; float margin_gap; int b1, b2; };
  __device__ __forceinline__ CPar contact_params(int g1, int g2, float margin, float gap) const {
    CPar cp;
    cp.margin_gap = margin - gap;
    cp.b1 = cm->gbody[g1]; cp.b2 = cm->gbody[g2];   // stored with the contact (geom | body << 8): the row builders need the body, not a second dependent global load
    gcf a = cmf(MK_gpar)->gpar + 12 * g1;
    gcf b = cmf(MK_gpar)->gpar + 12 * g2;
    const int c1 = cm->gcp[g1], c2 = cm->gcp[g2];
    const int p1 = c1 >> 8, p2 = c2 >> 8, d1 = c1 & 255, d2 = c2 & 255;
    if (p1 != p2) {
      gcf w = p1 > p2 ? a : b;
      cp.dim = p1 > p2 ? d1 : d2;
      for (int k = 0; k < 3; k++) cp.fr[k] = w[k];
      for (int k = 0; k < 2; k++) cp.solref[k] = w[3 + k];
      for (int k = 0; k < 5; k++) cp.solimp[k] = w[5 + k];
    } else {
      cp.dim = d1 > d2 ? d1 : d2;
      const float s1 = a[10], s2 = b[10];
      float mix;
      if (s1 >= 1e-15f && s2 >= 1e-15f) mix = s1 / (s1 + s2);
      else if (s1 < 1e-15f && s2 < 1e-15f) mix = 0.5f;
      else mix = s1 < 1e-15f ? 0.0f : 1.0f;
      if (a[3] > 0 && b[3] > 0) { cp.solref[0] = mix * a[3] + (1 - mix) * b[3]; cp.solref[1] = mix * a[4] + (1 - mix) * b[4]; }
      else { cp.solref[0] = fminf(a[3], b[3]); cp.solref[1] = fminf(a[4], b[4]); }
      for (int k = 0; k < 5; k++) cp.solimp[k] = mix * a[5 + k] + (1 - mix) * b[5 + k];
      for (int k = 0; k < 3; k++) cp.fr[k] = fmaxf(a[k], b[k]);
    }
    return cp;
  }
  // Lanes with has == true append one contact each, in lane order, at most `cap` of them (the rest are dropped);
  // every lane of the wave must call this (ballot + shared counter).
  __device__ __forceinline__ void emit_contacts(bool has, int cap, float dist, V3 pos, V3 nrm, int g1, int g2, const CPar& cp) {
    const u64 mk = __ballot(has);
    const int rank = __popcll(mk & lanemask_lt(lane));
    const int base = sm.ncon;
    int total = __popcll(mk);
    if (total > cap) total = cap;
    con_raw += total;
    if (base + total > SM::NCON_) { ovf += base + total - SM::NCON_; total = SM::NCON_ - base; }
    if (has && rank < total) {
      const int c = base + rank;
      sm.cdist[c] = dist;
      st3(sm.cpos + 3 * c, pos);
      if constexpr (CG) { float fr_[9]; make_frame(nrm, fr_);
#pragma unroll
        for (int k = 0; k < 9; k++) Cgp()[SM::CG_FRAME_ + 9 * c + k] = fr_[k]; }
      else make_frame(nrm, sm.cframe + 9 * c);
      sm.cg1[c] = g1 | (cp.b1 << 8); sm.cg2[c] = g2 | (cp.b2 << 8); sm.cdim[c] = cp.dim;
      if constexpr (CG) {
        Cgp()[SM::CG_MARGIN_ + c] = cp.margin_gap;
        Cgp()[SM::CG_SOLREF_ + 2 * c] = cp.solref[0]; Cgp()[SM::CG_SOLREF_ + 2 * c + 1] = cp.solref[1];
#pragma unroll
        for (int k = 0; k < 5; k++) Cgp()[SM::CG_SOLIMP_ + 5 * c + k] = cp.solimp[k];
        gwf f = Cgp() + SM::CG_FRI_ + 5 * c;
        f[0] = cp.fr[0]; f[1] = cp.fr[0]; f[2] = cp.fr[1]; f[3] = cp.fr[2]; f[4] = cp.fr[2];
      } else {
        sm.cmargin[c] = cp.margin_gap;
        sm.csolref[2 * c] = cp.solref[0]; sm.csolref[2 * c + 1] = cp.solref[1];
        for (int k = 0; k < 5; k++) sm.csolimp[5 * c + k] = cp.solimp[k];
        float* f = sm.cfri + 5 * c;
        f[0] = f[1] = cp.fr[0]; f[2] = cp.fr[1]; f[3] = f[4] = cp.fr[2];
      }
    }
    SYNC();
    if (lane == 0) sm.ncon = base + total;
    SYNC();
  }

  __device__ __forceinline__ V3 pick3(V3 a, V3 b, V3 c, int i) const { return i == 0 ? a : (i == 1 ? b : c); }
  __device__ __forceinline__ float pickf(V3 a, int i) const { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
  __device__ __forceinline__ V3 bcast3(V3 a, int src) const { return v3(__shfl(a.x, src), __shfl(a.y, src), __shfl(a.z, src)); }

  // box(g1)-box(g2): separating-axis test with one lane per axis (6 face + 9 edge), then either one edge-edge contact or a
  // lane-parallel Sutherland-Hodgman clip of the incident face against the reference face (lane i = polygon vertex i).
  // Same decisions, tie-breaks and contact order as the serial restatement in oracle/rsim_oracle.c box_box().
  __device__ __forceinline__ void box_box(int g1, int g2, float margin, const CPar& cp) {
    const V3 pa = ld3(sm.gpos + 3 * g1), pb = ld3(sm.gpos + 3 * g2);
    const M3 Ra = ldm(sm.gmat + 9 * g1), Rb = ldm(sm.gmat + 9 * g2);
    const V3 ha = ld3(cmf(MK_gst)->gst + 8 * g1), hb = ld3(cmf(MK_gst)->gst + 8 * g2);
    const V3 A0 = col(Ra, 0), A1 = col(Ra, 1), A2 = col(Ra, 2), B0 = col(Rb, 0), B1 = col(Rb, 1), B2 = col(Rb, 2);
    const V3 dab = pb - pa;
    // ---- one axis per lane
    const int al = lane < 15 ? lane : 0;
    bool valid = lane < 15;
    V3 L;
    if (al < 3) L = pick3(A0, A1, A2, al);
    else if (al < 6) L = pick3(B0, B1, B2, al - 3);
    else {
      const int i = (al - 6) / 3, j = (al - 6) - 3 * i;
      L = cross(pick3(A0, A1, A2, i), pick3(B0, B1, B2, j));
      const float nn = norm(L);
      if (nn < 1e-6f) { valid = false; L = v3(1, 0, 0); } else L = L * (1.0f / nn);
    }
    const float ra = ha.x * fabsf(dot(L, A0)) + ha.y * fabsf(dot(L, A1)) + ha.z * fabsf(dot(L, A2));
    const float rb = hb.x * fabsf(dot(L, B0)) + hb.y * fabsf(dot(L, B1)) + hb.z * fabsf(dot(L, B2));
    const float sv = fabsf(dot(L, dab)) - ra - rb;
    if (__ballot(valid && sv > margin)) return;
    const u64 vmask = __ballot(valid);
    // ---- axis selection on uniform scalars (first maximum wins, as in the serial scan)
    float sA = bcast(sv, 0), sB = bcast(sv, 3);
    int idA = 0, idB = 3;
    { float t = bcast(sv, 1); if (t > sA) { sA = t; idA = 1; } t = bcast(sv, 2); if (t > sA) { sA = t; idA = 2; }
      t = bcast(sv, 4); if (t > sB) { sB = t; idB = 4; } t = bcast(sv, 5); if (t > sB) { sB = t; idB = 5; } }
    float best_face; int face_id;
    if (sB > sA + 1e-6f) { best_face = sB; face_id = idB; } else { best_face = sA; face_id = idA; }
    float best_edge = -3e38f; int edge_id = -1;
#pragma unroll
    for (int e = 0; e < 9; e++) { const float t = bcast(sv, 6 + e); if (((vmask >> (6 + e)) & 1ull) && t > best_edge) { best_edge = t; edge_id = e; } }
    if (edge_id >= 0 && best_edge > 0.95f * best_face + 1e-5f && best_edge > best_face + 1e-5f) {
      const int i = edge_id / 3, j = edge_id - 3 * i;
      V3 n = bcast3(L, 6 + edge_id);
      if (dot(n, dab) < 0) n = -n;
      const V3 Ai = pick3(A0, A1, A2, i), Bj = pick3(B0, B1, B2, j);
      V3 qa = pa, qb = pb;
      if (i != 0) qa = qa + A0 * ((dot(n, A0) > 0 ? 1.f : -1.f) * ha.x);
      if (i != 1) qa = qa + A1 * ((dot(n, A1) > 0 ? 1.f : -1.f) * ha.y);
      if (i != 2) qa = qa + A2 * ((dot(n, A2) > 0 ? 1.f : -1.f) * ha.z);
      if (j != 0) qb = qb + B0 * ((dot(n, B0) > 0 ? -1.f : 1.f) * hb.x);
      if (j != 1) qb = qb + B1 * ((dot(n, B1) > 0 ? -1.f : 1.f) * hb.y);
      if (j != 2) qb = qb + B2 * ((dot(n, B2) > 0 ? -1.f : 1.f) * hb.z);
      const V3 r = qb - qa;
      const float ab = dot(Ai, Bj), den = 1 - ab * ab, ra_ = dot(r, Ai), rb_ = dot(r, Bj);
      float sp = 0, tp = 0;
      if (den > 1e-12f) { sp = (ra_ - ab * rb_) / den; tp = (ab * ra_ - rb_) / den; }
      const float hai = pickf(ha, i), hbj = pickf(hb, j);
      sp = fmaxf(-hai, fminf(hai, sp));
      tp = fmaxf(-hbj, fminf(hbj, tp));
      emit_contacts(lane == 0, 1, best_edge, ((qa + Ai * sp) + (qb + Bj * tp)) * 0.5f, n, g1, g2, cp);
      return;
    }
    // ---- face contact: the reference box owns the axis
    const bool ref_is_a = face_id < 3;
    const int ax = ref_is_a ? face_id : face_id - 3;
    const V3 pr = ref_is_a ? pa : pb, pi_ = ref_is_a ? pb : pa, hr = ref_is_a ? ha : hb, hi = ref_is_a ? hb : ha;
    const V3 Rr0 = ref_is_a ? A0 : B0, Rr1 = ref_is_a ? A1 : B1, Rr2 = ref_is_a ? A2 : B2;
    const V3 Ri0 = ref_is_a ? B0 : A0, Ri1 = ref_is_a ? B1 : A1, Ri2 = ref_is_a ? B2 : A2;
    const V3 Rax = pick3(Rr0, Rr1, Rr2, ax);
    const V3 n = Rax * (dot(Rax, pi_ - pr) >= 0 ? 1.f : -1.f);  // from reference toward incident
    int iax = 0;
    { float bestd = fabsf(dot(Ri0, n)), t = fabsf(dot(Ri1, n)); if (t > bestd) { bestd = t; iax = 1; } t = fabsf(dot(Ri2, n)); if (t > bestd) { bestd = t; iax = 2; } }
    const V3 Riax = pick3(Ri0, Ri1, Ri2, iax);
    const float isg = dot(Riax, n) > 0 ? -1.f : 1.f;
    const int u = iax == 2 ? 0 : iax + 1, v = u == 2 ? 0 : u + 1, ru = ax == 2 ? 0 : ax + 1, rv = ru == 2 ? 0 : ru + 1;
    const V3 Riu = pick3(Ri0, Ri1, Ri2, u), Riv = pick3(Ri0, Ri1, Ri2, v), Rru = pick3(Rr0, Rr1, Rr2, ru), Rrv = pick3(Rr0, Rr1, Rr2, rv);
    const float hrax = pickf(hr, ax), hru = pickf(hr, ru), hrv = pickf(hr, rv);
    // incident-face corners in the reference-face frame (along ru, along rv, height above the face): lane c = corner c
    float px, py, pz;
    {
      const float cx = (lane == 0 || lane == 3) ? 1.f : -1.f, cy = lane < 2 ? 1.f : -1.f;
      const V3 w = pi_ + Riax * (isg * pickf(hi, iax)) + Riu * (cx * pickf(hi, u)) + Riv * (cy * pickf(hi, v));
      const V3 rel = w - pr;
      px = dot(rel, Rru); py = dot(rel, Rrv); pz = dot(rel, n) - hrax;
    }
    int np = 4;
    float* poly = sm.u.b.poly;  // [16][3] staging for the order-preserving scatter
    for (int pass = 0; pass < 4 && np > 0; pass++) {
      const bool ax0 = pass < 2;
      const float h = ax0 ? hru : hrv, sign = (pass & 1) ? -1.f : 1.f;
      const int nxt = lane + 1 >= np ? 0 : lane + 1;
      const float bx = __shfl(px, nxt), by = __shfl(py, nxt), bz = __shfl(pz, nxt);
      const bool act = lane < np;
      const float da = sign * (ax0 ? px : py) - h, db = sign * (ax0 ? bx : by) - h;
      const bool e1 = act && da <= 0, e2 = act && ((da < 0 && db > 0) || (da > 0 && db < 0));
      const u64 m1 = __ballot(e1), m2 = __ballot(e2);
      const int off = __popcll(m1 & lanemask_lt(lane)) + __popcll(m2 & lanemask_lt(lane));
      if (e1 && off < 16) { poly[3 * off] = px; poly[3 * off + 1] = py; poly[3 * off + 2] = pz; }
      if (e2 && off + (e1 ? 1 : 0) < 16) {
        const int o = off + (e1 ? 1 : 0);
        const float t = da / (da - db);
        poly[3 * o] = px + t * (bx - px); poly[3 * o + 1] = py + t * (by - py); poly[3 * o + 2] = pz + t * (bz - pz);
      }
      np = __popcll(m1) + __popcll(m2);
      if (np > 15) np = 15;
      SYNC();
      if (lane < np) { px = poly[3 * lane]; py = poly[3 * lane + 1]; pz = poly[3 * lane + 2]; }
      SYNC();
    }
    const bool hit = lane < np && pz <= margin;
    const V3 w = pr + Rru * px + Rrv * py + n * (hrax + pz);
    emit_contacts(hit, 8, pz, w - n * (0.5f * pz), ref_is_a ? n : -n, g1, g2, cp);
  }

  // closest point on triangle to the origin (barycentric)
  __device__ __forceinline__ V3 tri_closest_origin(V3 a, V3 b, V3 c, V3& bw) {
    float bary[3];
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
    bw = v3(bary[0], bary[1], bary[2]);
    return a * bary[0] + b * bary[1] + c * bary[2];
  }

  // Minkowski Portal Refinement (uniform control flow; support() is wave-cooperative)
  // Warm start (`wd`, valid if `wh`; `wout` = the pair's record, may be null).  A run that ends without contact ends on a SEPARATING direction -- the
  // first shape's farthest point along it does not reach the second's nearest -- and a hand hovering over the table keeps such pairs alive for
  // hundreds of substeps: each of them cost ~18 support evaluations per substep to rediscover (portal refinement down to the tolerance before the
  // sign shows), and they are what makes the slowest env of a launch 3x the median (tools/tail_report.py).  The direction the previous substep's run
  // ended on is tried first: two support evaluations, and if it still separates the pair is done.  Separation along ANY direction is proof that the
  // shapes are disjoint, so the verdict is the one the full run reaches (MPR is exact for separated pairs); contacts always come from the full run.
  // A pair IN contact keeps its contact for hundreds of substeps too (a finger resting on the table), and its run is the expensive one: ~5 portal
  // discovery steps and ~9 refinement steps, 28 support evaluations, every substep, to arrive at (nearly) the portal of the substep before.  The
  // record therefore also keeps the three DIRECTIONS whose supports span the final portal of a contact (flag 2).  The next run evaluates the
  // supports along them (6 evaluations); if they still form a portal -- every support beyond the origin along its direction, the origin ray inside
  // the three side planes: the invariant the discovery loop ends on -- discovery is skipped and the refinement, started next to its fixed point,
  // ends after one or two steps.  Unlike the separating-direction test this is a different (equally valid) path to the contact: depth and normal
  // are those of the same facet of the Minkowski difference, the contact POINT is a barycentric blend over a possibly different triangle of that
  // facet (it slides by rounding-level amounts along the contact face; bounds in tests/test_hip_edge_cases.py).  RSIM_NO_MPR_PORTAL_WARMSTART=1
  // (read when the batch is created) keeps only the exact separating-direction part.
  static constexpr int MPRC = 12;   // floats per candidate pair in DBatch.mprc: (d1 | separating direction, flag) (d2, -) (d3, -)
  __device__ __forceinline__ void mpr_store(gwf wout, V3 d, float valid) const {
    if (wout && lane == 0) { typedef v4f __attribute__((address_space(1)))* gw4; *(gw4)wout = v4f{d.x, d.y, d.z, valid}; }
  }
  // The three directions live in LDS while a run is under way (lane 0 writes a slot when a portal vertex is replaced: nine registers less in the
  // phase with the highest register pressure of the kernel); LDS operations of one lane execute in program order, so no fence is needed.
  __device__ __forceinline__ float* mpr_dirs() const { return sm.u.b.poly + 48; }   // [3][3] behind the box-box clip polygons (16 x 3 floats)
  __device__ __forceinline__ void mpr_setd(int k, V3 d) const { if (lane == 0) st3(mpr_dirs() + 3 * k, d); }
  __device__ __forceinline__ void mpr_store_portal(gwf wout) const {
    if (wout && lane == 0) {
      typedef v4f __attribute__((address_space(1)))* gw4;
      const float* q = mpr_dirs();
      ((gw4)wout)[0] = v4f{q[0], q[1], q[2], mpr_portal ? 2.f : 0.f}; ((gw4)wout)[1] = v4f{q[3], q[4], q[5], 0.f}; ((gw4)wout)[2] = v4f{q[6], q[7], q[8], 0.f};
    }
  }
  __device__ __forceinline__ void convex_convex(int g1, int g2, float margin, const CPar& cp, V3 wd, int wh, gwf wout) {
    const float tol = 1e-6f;
    const int mprstat_s0 = pf.c_support; (void)mprstat_s0;
    const V3 org = ld3(sm.gpos + 3 * g1);   // all support points relative to the first geom's position (see geom_support)
    const SupGeom sg1 = sup_load<SM>(cm, cmf(MK_gst), g1, (gcf)m.mesh_vert, lane, org), sg2 = sup_load<SM>(cm, cmf(MK_gst), g2, (gcf)m.mesh_vert, lane, org);
    if (wh == 1) {
      const V3 a1 = sup(sg1, wd), a2 = sup(sg2, -wd);
      const float s_w = dot(a1 - a2, wd);
      if (s_w <= 0) { MPRSTAT(8, 1); near_sep = fminf(near_sep, -s_w); return; }
    }
    V3 v0 = (ld3(sm.gcen + 3 * g1) - org) - (ld3(sm.gcen + 3 * g2) - org);
    if (norm(v0) < 1e-9f) v0.x = 1e-5f;
    V3 dir, v1, v2, v3_, p11, p12, p21, p22, p31, p32;   // portal vertices and their witness points on the two shapes (the directions they are supports of: mpr_dirs())
    bool warm = false;
    if (wh == 2) {
      typedef const v4f __attribute__((address_space(1)))* gc4;
      const v4f r1 = ((gc4)wout)[1], r2 = ((gc4)wout)[2];
      const V3 d1 = wd, d2 = v3(r1[0], r1[1], r1[2]), d3 = v3(r2[0], r2[1], r2[2]);
      mpr_setd(0, d1); mpr_setd(1, d2); mpr_setd(2, d3);
      p11 = sup(sg1, d1); p12 = sup(sg2, -d1); v1 = p11 - p12;
      p21 = sup(sg1, d2); p22 = sup(sg2, -d2); v2 = p21 - p22;
      p31 = sup(sg1, d3); p32 = sup(sg2, -d3); v3_ = p31 - p32;
      warm = dot(v1, d1) > 0.f && dot(v2, d2) > 0.f && dot(v3_, d3) > 0.f && dot(cross(v1, v3_), v0) >= 0.f && dot(cross(v3_, v2), v0) >= 0.f && dot(cross(v2, v1), v0) >= 0.f;
      MPRSTAT(9, warm ? 1 : 0);
    }
    if (wh == 3) {
      // A contact in which one shape is SMOOTH (cylinder, capsule, sphere, ellipsoid) persists like any other -- a finger or a link resting against the mount's
      // cylinder -- and is the expensive kind: the refinement gains one bit of the tolerance per step on a curved surface (the portal has to shrink to
      // sqrt(2 R tol), ~0.6 mm on the mount), 13 - 20 support pairs per run, every substep; such runs are 60 % of the slowest env of a Lift launch
      // (profiles/r06_d_window_trace.txt), and that env IS the launch.  The converged portal itself cannot be reused (its three directions are nearly
      // parallel and their supports coincide: see the note at the end of this function), but its NORMAL can: the supports along three directions on a cone
      // of half-angle DModel.mpr_cone around it span a well-shaped triangle of the Minkowski difference (~cone x R across: millimetres, not micrometres)
      // right where the origin ray left through last time.  If it satisfies the invariant the discovery loop ends on -- every support beyond the origin
      // along its direction, the origin ray inside the three side planes -- it IS a portal and the refinement starts from it, two or three steps from its
      // tolerance; if not (the contact moved by more than the cone), the run starts cold.  A valid portal is a valid state of the algorithm however it was
      // found, and next to a smooth shape the boundary has no near-coplanar facets for the path to choose between: same DEPTH to the tolerance.
      // OFF by default (DModel.mpr_cone = 0, RSIM_MPR_CONE): measured +5 % on the Lift headline (profiles/r06_g_ab_mpr_cone.txt), but MPR's normal on a curved
      // surface is the plane normal of the LAST portal -- good to (portal size / R), 0.3 degrees on a 6 cm cylinder -- and a run that arrives from another side
      // ends on another portal: Baxter's elbow-cylinder contacts then agree with the cold-started fp64 oracle in 83 % of the sampled envs instead of 94 %
      // (tests/test_full_size_parity.py bound: 85 %).  Both are MPR answers; only the cold one is the path the oracle (and any cold-starting reference) takes.
      const V3 nn = wd, ax = fabsf(nn.x) < 0.6f ? v3(1.f, 0.f, 0.f) : v3(0.f, 1.f, 0.f);
      const V3 t1 = normalized(cross(nn, ax)), t2 = cross(nn, t1);
      const float ce = m.mpr_cone;
      const V3 d1 = normalized(nn + t1 * ce), d2 = normalized(nn + (t1 * -0.5f + t2 * 0.8660254f) * ce), d3 = normalized(nn + (t1 * -0.5f - t2 * 0.8660254f) * ce);
      p11 = sup(sg1, d1); p12 = sup(sg2, -d1); v1 = p11 - p12;
      p21 = sup(sg1, d2); p22 = sup(sg2, -d2); v2 = p21 - p22;
      p31 = sup(sg1, d3); p32 = sup(sg2, -d3); v3_ = p31 - p32;
      const float s13 = dot(cross(v1, v3_), v0), s32 = dot(cross(v3_, v2), v0), s21 = dot(cross(v2, v1), v0);
      const bool beyond = dot(v1, d1) > 0.f && dot(v2, d2) > 0.f && dot(v3_, d3) > 0.f;
      if (beyond && s13 >= 0.f && s32 >= 0.f && s21 >= 0.f) { warm = true; mpr_setd(0, d1); mpr_setd(1, d2); mpr_setd(2, d3); }
      else if (beyond && s13 <= 0.f && s32 <= 0.f && s21 <= 0.f && (s13 < 0.f || s32 < 0.f || s21 < 0.f)) {   // the other winding: second and third vertex swapped
        V3 t;
        t = v2; v2 = v3_; v3_ = t; t = p21; p21 = p31; p31 = t; t = p22; p22 = p32; p32 = t;
        warm = true; mpr_setd(0, d1); mpr_setd(1, d3); mpr_setd(2, d2);
      }
      MPRSTAT(9, warm ? 1 : 0);
    }
    if (!warm) {
    // A box or cylinder against anything: MPR's own exit test (a direction D, oriented from geom 1 to geom 2, in which the first shape's
    // farthest point does not reach the second's nearest) tried first on the primitive's most promising face / radial axis.  The table top or
    // the mount pedestal against a gripper mesh hovering over it -- most narrow-phase visits of the Lift workload -- end here after ONE hull
    // scan; from the centre-to-centre direction MPR needs ~6 portal steps (12.5 support calls) to find a separating direction of such a
    // wide, flat pair.  A pair that does touch pays one extra support pair.  Same verdict as the full run for separated pairs (MPR is exact
    // there), so the contact set is unchanged.
    {
      const int t1 = uni(sg1.t), t2 = uni(sg2.t);
      const bool prim1 = t1 == G_BOX || t1 == G_CYLINDER, prim2 = t2 == G_BOX || t2 == G_CYLINDER;
      if (prim1 || prim2) {
        const bool first = prim1 && (!prim2 || t1 == G_BOX);   // two primitives: the box's axes
        // the primitive's frame, picked field by field (a reference selected between the two structs would put both into the private segment)
        struct { M3 R; V3 p, h; } sp;
#pragma unroll
        for (int k = 0; k < 9; k++) sp.R.m[k] = first ? sg1.R.m[k] : sg2.R.m[k];
        sp.p = first ? sg1.p : sg2.p; sp.h = first ? sg1.h : sg2.h;
        const int go = first ? g2 : g1, tp = first ? t1 : t2;
        const M3 Ro = ldm(sm.gmat + 9 * go);
        V3 oo, ho;
        geom_obb(go, Ro, oo, ho);
        const V3 tl = mtv(sp.R, (oo - org) - sp.p);   // the other shape's box centre in the primitive's frame
        const M3 C = mtm(sp.R, Ro);
        const float ex = ho.x * fabsf(C.m[0]) + ho.y * fabsf(C.m[1]) + ho.z * fabsf(C.m[2]), ey = ho.x * fabsf(C.m[3]) + ho.y * fabsf(C.m[4]) + ho.z * fabsf(C.m[5]),
                    ez = ho.x * fabsf(C.m[6]) + ho.y * fabsf(C.m[7]) + ho.z * fabsf(C.m[8]);
        V3 dl = v3(0, 0, tl.z >= 0 ? 1.f : -1.f);
        float best = fabsf(tl.z) - sp.h.z - ez;
        if (tp == G_BOX) {
          const float bx = fabsf(tl.x) - sp.h.x - ex, by = fabsf(tl.y) - sp.h.y - ey;
          if (bx > best) { best = bx; dl = v3(tl.x >= 0 ? 1.f : -1.f, 0, 0); }
          if (by > best) { best = by; dl = v3(0, tl.y >= 0 ? 1.f : -1.f, 0); }
        } else {
          const float rho = sqrtf(tl.x * tl.x + tl.y * tl.y);
          if (rho > 1e-6f) {
            const V3 rl = v3(tl.x / rho, tl.y / rho, 0);
            const float br = rho - sp.h.x - (ex * fabsf(rl.x) + ey * fabsf(rl.y));
            if (br > best) { best = br; dl = rl; }
          }
        }
        V3 D = mv(sp.R, dl);
        if (!first) D = -D;
        const V3 a1 = sup(sg1, D), a2 = sup(sg2, -D);
        const float s_p = dot(a1 - a2, D);
        if (s_p <= 0) { MPRSTAT(0, 1); mpr_store(wout, D, 1.f); near_sep = fminf(near_sep, -s_p); return; }
      }
    }
    dir = normalized(-v0);
    p11 = sup(sg1, dir); p12 = sup(sg2, -dir); v1 = p11 - p12; mpr_setd(0, dir);
    if (dot(v1, dir) <= 0) { MPRSTAT(1, 1); mpr_store(wout, dir, 1.f); return; }
    dir = cross(v0, v1);
    if (norm(dir) < 1e-12f) {
      V3 n = normalized(v1 - v0);
      emit_contacts(lane == 0, 1, -dot(v1, n), org + (p11 + p12) * 0.5f, n, g1, g2, cp);
      return;
    }
    dir = normalized(dir);
    p21 = sup(sg1, dir); p22 = sup(sg2, -dir); v2 = p21 - p22; mpr_setd(1, dir);
    if (dot(v2, dir) <= 0) { MPRSTAT(2, 1); mpr_store(wout, dir, 1.f); return; }
    dir = cross(v1 - v0, v2 - v0);
    if (dot(dir, v0) > 0) {
      V3 t;
      t = v1; v1 = v2; v2 = t; t = p11; p11 = p21; p21 = t; t = p12; p12 = p22; p22 = t;
      if (lane == 0) { float* q = mpr_dirs(); const V3 a = ld3(q), b2 = ld3(q + 3); st3(q, b2); st3(q + 3, a); }
      dir = -dir;
    }
    for (int it = 0;; it++) {
      if (it > 100) return;
      float len;
      dir = normalized(dir, &len);
      if (len < FMIN) return;
      p31 = sup(sg1, dir); p32 = sup(sg2, -dir); v3_ = p31 - p32;
      if (dot(v3_, dir) <= 0) { MPRSTAT(3, 1); MPRSTAT(7, pf.c_support - mprstat_s0); mpr_store(wout, dir, 1.f); return; }
      if (dot(cross(v1, v3_), v0) < -1e-14f) { v2 = v3_; p21 = p31; p22 = p32; mpr_setd(1, dir); dir = cross(v1 - v0, v3_ - v0); continue; }
      if (dot(cross(v3_, v2), v0) < -1e-14f) { v1 = v3_; p11 = p31; p12 = p32; mpr_setd(0, dir); dir = cross(v3_ - v0, v2 - v0); continue; }
      mpr_setd(2, dir);
      break;
    }
    }   // !warm
    bool hit = false;
    for (int it = 0; it < 128; it++) {
      float len;
      dir = normalized(cross(v2 - v1, v3_ - v1), &len);
      if (len < FMIN) break;
      if (dot(dir, v1) >= 0) hit = true;
      V3 p41 = sup(sg1, dir), p42 = sup(sg2, -dir), v4 = p41 - p42;
      float dv4 = dot(v4, dir);
      if (dv4 < 0 && !hit) { MPRSTAT(4, 1); MPRSTAT(7, pf.c_support - mprstat_s0); mpr_store(wout, dir, 1.f); return; }
      float delta = dv4 - dot(v3_, dir);
      if (delta <= tol || it == 127) break;
      V3 t = cross(v4, v0);
      if (dot(v1, t) > 0) {
        if (dot(v2, t) > 0) { v1 = v4; p11 = p41; p12 = p42; mpr_setd(0, dir); } else { v3_ = v4; p31 = p41; p32 = p42; mpr_setd(2, dir); }
      } else {
        if (dot(v3_, t) > 0) { v2 = v4; p21 = p41; p22 = p42; mpr_setd(1, dir); } else { v1 = v4; p11 = p41; p12 = p42; mpr_setd(0, dir); }
      }
    }
    if (!hit) { MPRSTAT(4, 1); MPRSTAT(7, pf.c_support - mprstat_s0); mpr_store(wout, v3(0.f, 0.f, 0.f), 0.f); return; }
    MPRSTAT(5, 1); MPRSTAT(6, pf.c_support - mprstat_s0);
    V3 bw;
    V3 cpt = tri_closest_origin(v1, v2, v3_, bw);
    float depth = norm(cpt);
    V3 n = depth > 1e-12f ? cpt * (1.0f / depth) : dir;
    // fp32: the closest point of the portal to the origin is a difference of nearly equal support points (rounding ~1e-7 m on metre-sized
    // coordinates), so its DIRECTION is only good to 1e-7 / depth -- degrees for the micrometre penetrations of a resting contact.  When the
    // point is interior to the portal triangle it is the foot of the perpendicular, i.e. the direction is the plane normal, which comes from
    // centimetre-sized edges and is good to 1e-5 rad; same point, same depth in exact arithmetic (the fp64 oracle keeps the one formula).
    if (bw.x > 0.f && bw.y > 0.f && bw.z > 0.f && norm(dir) > 0.5f) {
      const float dn = dot(dir, v1);
      n = dn >= 0.f ? dir : -dir;
      depth = fabsf(dn);
    }
    // in contact: the next run starts from this portal -- for a BOX against a hull, up to 5 mm deep.  What the restart relies on is that the cached
    // directions give back a well-shaped portal on the facet the origin ray leaves through:
    //  * curved shapes (cylinder, capsule, sphere) fail it: the three directions of a converged portal are nearly parallel, their supports nearly
    //    coincide and the restarted portal is a sliver whose plane normal is rounding noise (Baxter's elbow cylinders on the torso hull lost 15 % of
    //    their normals to > 0.1 degree that way);
    //  * two fine meshes, or anything interpenetrating by a centimetre (the Robotiq's finger / knuckle links, in every pose), have several
    //    near-coplanar facets around the origin ray; which one a run ends on depends on its path, a restarted run keeps choosing its own, and the
    //    finger links -- 5e-5 kg m^2, no damping -- then follow another trajectory than the cold-started oracle (UR5e / PickPlace finger tracking went
    //    from 0.04 to 0.17 rad over 20 control steps).
    // A table, bin wall or cube face against a gripper or object hull is the case that matters (the hand held against the table is what makes the
    // slowest envs of a Lift launch) and the well-posed one: one large flat facet.
    {
      const int ta = uni(sg1.t), tb = uni(sg2.t);
      const bool smooth_a = ta == G_CYLINDER || ta == G_CAPSULE || ta == G_SPHERE || ta == G_ELLIPSOID, smooth_b = tb == G_CYLINDER || tb == G_CAPSULE || tb == G_SPHERE || tb == G_ELLIPSOID;
      if (((ta == G_BOX && tb == G_MESH) || (ta == G_MESH && tb == G_BOX)) && depth < 5e-3f) mpr_store_portal(wout);
      else if ((smooth_a || smooth_b) && m.mpr_cone > 0.f && norm(dir) > 0.5f) mpr_store(wout, dot(dir, v1) >= 0.f ? dir : -dir, 3.f);   // the final portal's outward normal (flag 3 above)
      else mpr_store(wout, v3(0.f, 0.f, 0.f), 0.f);
    }
    V3 w1 = p11 * bw.x + p21 * bw.y + p31 * bw.z, w2 = p12 * bw.x + p22 * bw.y + p32 * bw.z;
    emit_contacts(lane == 0, 1, -depth, org + (w1 + w2) * 0.5f, n, g1, g2, cp);
  }

  // oriented bounding box of colliding geom g: world centre o, half extents h along the columns of gmat
  __device__ __forceinline__ void geom_obb(int g, const M3& R, V3& o, V3& h) const {
    gcf st = cmf(MK_gst)->gst + 8 * g;
    h = ld3(st);
    o = ld3(sm.gpos + 3 * g) + mv(R, ld3(st + 3));
  }

  // squared distance between segments [p1,q1] and [p2,q2] (clamped closest-point parameters)
  __device__ __forceinline__ float segment_dist2(V3 p1, V3 q1, V3 p2, V3 q2) const {
    const V3 d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
    const float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r), c = dot(d1, r), b = dot(d1, d2);
    float sp = 0.f, tp = 0.f;
    if (a > 1e-12f && e > 1e-12f) {
      const float den = a * e - b * b;
      sp = den > 1e-12f ? fminf(1.f, fmaxf(0.f, (b * f - c * e) / den)) : 0.f;
      tp = (b * sp + f) / e;
      if (tp < 0.f) { tp = 0.f; sp = fminf(1.f, fmaxf(0.f, -c / a)); }
      else if (tp > 1.f) { tp = 1.f; sp = fminf(1.f, fmaxf(0.f, (b - c) / a)); }
    } else if (a > 1e-12f) sp = fminf(1.f, fmaxf(0.f, -c / a));
    else if (e > 1e-12f) tp = fminf(1.f, fmaxf(0.f, f / e));
    const V3 dd = (p1 + d1 * sp) - (p2 + d2 * tp);
    return dot(dd, dd);
  }

  // Active pair list of the broadphase.  Of the ~140 candidate pairs of the Lift model some 90 are never closer than decimetres (arm links against
  // the floor, the mount against the cube ...), yet every substep tested all of them: three rows of 64 lanes, each row paying its full path (13 % of a
  // contact-free substep).  When the list is (re)built, every pair whose bounding spheres (sphere and plane) are within `reach` of touching goes
  // onto it, and lane g remembers where the centre of geom g's bounding sphere was.  While no centre has moved by more than reach / 2 since, a pair
  // that is NOT on the list still has a positive gap, i.e. still fails the first test of the broadphase -- so the other substeps test ONE row,
  // lane i = i-th listed pair (in pair order: candidate order, and with it the contact order, is unchanged) and get exactly the candidates the full
  // rows would give.  No velocity bound is involved: the displacement is measured every substep (an episode reset inside the launch is just a
  // large displacement).  More than 64 near pairs: no list, full rows.
  // The list lives in global memory (DBatch.bpl, 5 x 64 words per env, L2-resident): five more registers alive across the whole substep loop cost
  // the 256-register build more in spills than the list saves.
  int task_obj = 0;                     // PickPlace single-object mode 1: this env's object (DBatch.task_object)
  int act_n = -1;                       // pairs on the list (-1: none)
  float applied = 0.f;   // this lane's dof: mjData.qfrc_applied (RSIM_QFRC_APPLIED), read by the debug form of the kernel only (step_body)
  int __attribute__((address_space(1)))* bpl = nullptr;   // [0..2][lane g]: centre of geom g's bounding sphere when the list was built; [3][lane i]: packed
                                                          // constants (geom1 | geom2 << 8 | enabled << 16) of the i-th listed pair; [4][lane i]: its index
  // one candidate pair: does it pass the broadphase now (pass), could it within `reach` (near: bounding spheres / plane distance only)
  __device__ __forceinline__ void bp_test(int pr, float reach, bool& pass, bool& near) const {
    pass = false; near = false;
    // all constants of the pair first, as eight 16-byte global loads in flight at once (they used to be LDS reads issued where needed; from
    // global memory three dependent stages per round would be three round trips)
    const int g1 = pr & 255, g2 = (pr >> 8) & 255;   // lanes without a pair read geom 0
    typedef const v4f __attribute__((address_space(1)))* gc4;
    gc4 st1 = (gc4)(cmf(MK_gst)->gst + 8 * g1), st2 = (gc4)(cmf(MK_gst)->gst + 8 * g2);
    gc4 kp1 = (gc4)(cmf(MK_gcap)->gcap + 8 * g1), kp2 = (gc4)(cmf(MK_gcap)->gcap + 8 * g2);
    const v4f s1a = st1[0], s1b = st1[1], s2a = st2[0], s2b = st2[1], k1a = kp1[0], k1b = kp1[1], k2a = kp2[0], k2b = kp2[1];
    const int t1 = cm->gtype[g1];
    if ((pr >> 16) & 1) {
      const float margin = fmaxf(s1b[3], s2b[3]);
      const V3 c2 = ld3(sm.gcen + 3 * g2);
      const M3 R2 = ldm(sm.gmat + 9 * g2);
      const V3 h2 = v3(s2a[0], s2a[1], s2a[2]), o2 = ld3(sm.gpos + 3 * g2) + mv(R2, v3(s2a[3], s2b[0], s2b[1]));
      if (t1 == G_PLANE) {
        const V3 nrm = v3(sm.gmat[9 * g1 + 2], sm.gmat[9 * g1 + 5], sm.gmat[9 * g1 + 8]);
        const V3 pp = ld3(sm.gpos + 3 * g1);
        const float gap = dot(c2 - pp, nrm) - s2b[2] - margin;
        pass = gap <= 0.f;
        near = gap <= reach;
        if (pass) pass = dot(o2 - pp, nrm) - (h2.x * fabsf(dot(nrm, col(R2, 0))) + h2.y * fabsf(dot(nrm, col(R2, 1))) + h2.z * fabsf(dot(nrm, col(R2, 2)))) <= margin;
      } else {
        const V3 rel = c2 - ld3(sm.gcen + 3 * g1);
        const float bound = s1b[2] + s2b[2] + margin, d2 = dot(rel, rel);
        pass = d2 <= bound * bound;
        near = d2 <= (bound + reach) * (bound + reach);
        if (pass) {
          const M3 R1 = ldm(sm.gmat + 9 * g1);
          const V3 h1 = v3(s1a[0], s1a[1], s1a[2]), o1 = ld3(sm.gpos + 3 * g1) + mv(R1, v3(s1a[3], s1b[0], s1b[1]));
          const M3 C = mtm(R1, R2);  // C[i][j] = A_i . B_j
          const V3 tt = o2 - o1, ta = mtv(R1, tt), tb = mtv(R2, tt);
          const float sepa = fmaxf(fmaxf(fabsf(ta.x) - (h1.x + h2.x * fabsf(C.m[0]) + h2.y * fabsf(C.m[1]) + h2.z * fabsf(C.m[2])),
                                         fabsf(ta.y) - (h1.y + h2.x * fabsf(C.m[3]) + h2.y * fabsf(C.m[4]) + h2.z * fabsf(C.m[5]))),
                                   fabsf(ta.z) - (h1.z + h2.x * fabsf(C.m[6]) + h2.y * fabsf(C.m[7]) + h2.z * fabsf(C.m[8])));
          const float sepb = fmaxf(fmaxf(fabsf(tb.x) - (h2.x + h1.x * fabsf(C.m[0]) + h1.y * fabsf(C.m[3]) + h1.z * fabsf(C.m[6])),
                                         fabsf(tb.y) - (h2.y + h1.x * fabsf(C.m[1]) + h1.y * fabsf(C.m[4]) + h1.z * fabsf(C.m[7]))),
                                   fabsf(tb.z) - (h2.z + h1.x * fabsf(C.m[2]) + h1.y * fabsf(C.m[5]) + h1.z * fabsf(C.m[8])));
          pass = fmaxf(sepa, sepb) <= margin + 1e-6f;
          // bounding capsules (mesh hulls): distance between the two axis segments against the radii
          if (pass && k1b[2] >= 0.f && k2b[2] >= 0.f) {
            const V3 gp1 = ld3(sm.gpos + 3 * g1), gp2 = ld3(sm.gpos + 3 * g2);
            const V3 p1 = gp1 + mv(R1, v3(k1a[0], k1a[1], k1a[2])), q1 = gp1 + mv(R1, v3(k1a[3], k1b[0], k1b[1])),
                     p2 = gp2 + mv(R2, v3(k2a[0], k2a[1], k2a[2])), q2 = gp2 + mv(R2, v3(k2a[3], k2b[0], k2b[1]));
            const float rch = k1b[2] + k2b[2] + margin + 1e-6f;
            pass = segment_dist2(p1, q1, p2, q2) <= rch * rch;
          }
        }
      }
    }
  }
  __device__ __forceinline__ void collision(int sub_left) {
    const LaneConst K = fetchK();
    if (lane == 0) sm.ncon = 0;
    con_raw = 0;
    near_sep = 3.0e38f;
    // broadphase: lane p tests candidate pair p (bounding spheres, then the 6 face axes of the two oriented boxes);
    // order-preserving compaction of the survivors
    int ncand = 0;
    bool valid = act_n >= 0;
    typedef const float __attribute__((address_space(1)))* gcf1;
    if (valid) {
      // a plane that is not fixed to the world can tilt without its centre moving: such a geom counts as moved
      gcf1 c0 = (gcf1)bpl;
      const V3 d = ld3(sm.gcen + 3 * (lane < m.ncg ? lane : 0)) - v3(c0[lane], c0[64 + lane], c0[128 + lane]);
      const bool moved = lane < m.ncg && (dot(d, d) > 0.25f * m.bp_reach * m.bp_reach || (cm->gtype[lane] == G_PLANE && (K.ginfo & 255) != 0));
      valid = !__ballot(moved);
    }
    if (valid) {
      // ---- one row over the listed pairs
      const int pr = bpl[192 + lane], pi = bpl[256 + lane];
      bool pass, near;
      bp_test(lane < act_n ? pr : 0, 0.f, pass, near);
      pass = pass && lane < act_n;
      const u64 mk = __ballot(pass);
      if (pass) sm.u.b.cand[__popcll(mk & lanemask_lt(lane))] = pi;
      ncand = __popcll(mk);
    } else {
      // ---- all rows; with a list to build: the near pairs, compacted in pair order
      const bool build = bpl && m.bp_reach > 0.f && sub_left > 1;
      int na = 0;
      int* alist = (int*)sm.u.b.poly;   // 64 pair indices (the clip polygons are not in use during the broadphase)
#pragma unroll
      for (int t = 0; t < NPT; t++) {
        if (64 * t >= m.npair) break;
        bool pass, near;
        bp_test(K.pair[t], m.bp_reach, pass, near);
        const u64 mk = __ballot(pass);
        if (pass) sm.u.b.cand[ncand + __popcll(mk & lanemask_lt(lane))] = 64 * t + lane;
        ncand += __popcll(mk);
        if (build) {
          const u64 nk = __ballot(near);
          const int at = na + __popcll(nk & lanemask_lt(lane));
          if (near && at < 64) alist[at] = 64 * t + lane;
          na += __popcll(nk);
        }
      }
      act_n = -1;
      if (build && na <= 64) {
        SYNC();
        const int pi = lane < na ? alist[lane] : 0;
        bpl[256 + lane] = pi;
        bpl[192 + lane] = lane < na ? (IT(IO_pair_g1, pi) | (IT(IO_pair_g2, pi) << 8) | (1 << 16)) : 0;
        const V3 c = ld3(sm.gcen + 3 * (lane < m.ncg ? lane : 0));
        float __attribute__((address_space(1)))* c0 = (float __attribute__((address_space(1)))*)bpl;
        c0[lane] = c.x; c0[64 + lane] = c.y; c0[128 + lane] = c.z;
        act_n = na;
      }
    }
    SYNC();
    pf.mark(RP_BROAD);
    pf.count(RP_N_CAND, ncand);
    // warm-start records of the candidates, one per lane, in flight while the first pairs are processed (candidates beyond 64 start cold)
    v4f wc = {0.f, 0.f, 0.f, 0.f};
    if (mprc && lane < ncand) { typedef const v4f __attribute__((address_space(1)))* gc4; wc = *(gc4)(mprc + MPRC * sm.u.b.cand[lane]); }
    for (int ci = 0; ci < ncand; ci++) {
      phase();
      int p = uni(sm.u.b.cand[ci]);
      int g1 = uni(IT(IO_pair_g1, p)), g2 = uni(IT(IO_pair_g2, p));
      int t1 = uni(cm->gtype[g1]), t2 = uni(cm->gtype[g2]);
      const float margin = fmaxf(cmf(MK_gst)->gst[8 * g1 + 7], cmf(MK_gst)->gst[8 * g2 + 7]), gap = fmaxf(cmf(MK_gpar)->gpar[12 * g1 + 11], cmf(MK_gpar)->gpar[12 * g2 + 11]);
      const CPar cp = contact_params(g1, g2, margin, gap);
      const int sup0 = pf.c_support;
      if (t1 == G_PLANE && t2 == G_BOX) {
        const V3 nrm = v3(sm.gmat[9 * g1 + 2], sm.gmat[9 * g1 + 5], sm.gmat[9 * g1 + 8]);
        float dist = 0;
        V3 wp = v3(0, 0, 0);
        bool hit = false;
        if (lane < 8) {
          const V3 sz = ld3(cmf(MK_gst)->gst + 8 * g2);
          const V3 lp = v3((lane & 1) ? sz.x : -sz.x, (lane & 2) ? sz.y : -sz.y, (lane & 4) ? sz.z : -sz.z);
          wp = ld3(sm.gpos + 3 * g2) + mv(ldm(sm.gmat + 9 * g2), lp);
          dist = dot(wp - ld3(sm.gpos + 3 * g1), nrm);
          hit = dist <= margin;
        }
        emit_contacts(hit, 4, dist, wp - nrm * (0.5f * dist), nrm, g1, g2, cp);  // first four corners in corner order
      } else if (t1 == G_PLANE) {
        const V3 nrm = v3(sm.gmat[9 * g1 + 2], sm.gmat[9 * g1 + 5], sm.gmat[9 * g1 + 8]);
        const V3 sp = support(g2, -nrm);
        const float dist = dot(sp - ld3(sm.gpos + 3 * g1), nrm);
        emit_contacts(lane == 0 && dist <= margin, 1, dist, sp - nrm * (0.5f * dist), nrm, g1, g2, cp);
      } else if (t1 == G_BOX && t2 == G_BOX) {
        pf.mark(RP_PLANE);
        box_box(g1, g2, margin, cp);
        pf.mark(RP_BOXBOX); pf.count(RP_N_BOXBOX, 1);
      } else {
        pf.mark(RP_PLANE);
        V3 wd = v3(0.f, 0.f, 0.f);
        int wh = 0;
        if (mprc && ci < 64) { wd = v3(bcast(wc[0], ci), bcast(wc[1], ci), bcast(wc[2], ci)); wh = uni((int)bcast(wc[3], ci)); if ((wh == 2 || wh == 3) && !mpr_portal) wh = 0; }
        convex_convex(g1, g2, margin, cp, wd, wh, mprc ? mprc + MPRC * p : nullptr);
        pf.mark(RP_MPR); pf.count(RP_N_MPR, 1);
      }
      if (pf.pairs && lane == 0 && pf.acc) { atomicAdd(pf.pairs + p, 1ull); atomicAdd(pf.pairs + RSIM_PAIR_MAX + p, (unsigned long long)(pf.c_support - sup0)); }
      SYNC();
    }
  }

  // ---------------------------------------------------------------- constraint rows
  __device__ __forceinline__ float impedance(const float* solimp, float x_abs) const {
    float dmin = fminf(0.9999f, fmaxf(0.0001f, solimp[0])), dmax = fminf(0.9999f, fmaxf(0.0001f, solimp[1]));
    float width = fmaxf(1e-15f, solimp[2]), mid = fminf(0.9999f, fmaxf(0.0001f, solimp[3])), power = fmaxf(1.0f, solimp[4]);
    float x = x_abs / width, y;
    if (x >= 1) return dmax;
    if (x <= 0) return dmin;
    if (power == 1.0f) y = x;
    else if (power == 2.0f) y = x <= mid ? x * x / mid : 1 - (1 - x) * (1 - x) / (1 - mid);
    else if (x <= mid) y = powf(x, power) / powf(mid, power - 1);
    else y = 1 - powf(1 - x, power) / powf(1 - mid, power - 1);
    return dmin + y * (dmax - dmin);
  }
  // regulariser R, velocity gain B and the position term K*imp*(pos-margin) of a constraint row
  __device__ __forceinline__ void row_scalars(float pos, float margin, const float* solref, const float* solimp, float diag, float& R, float& Bd, float& Kterm) const {
    const float imp = impedance(solimp, fabsf(pos - margin));
    const float dmax = fminf(0.9999f, fmaxf(0.0001f, solimp[1]));
    float Kk;
    if (solref[0] > 0) {
      const float tc = fmaxf(solref[0], 2 * opt_h), dr = solref[1];
      Bd = 2 / fmaxf(1e-15f, dmax * tc);
      Kk = 1 / fmaxf(1e-15f, dmax * dmax * tc * tc * dr * dr);
    } else {
      Kk = -solref[0] / fmaxf(1e-15f, dmax * dmax);
      Bd = -solref[1] / fmaxf(1e-15f, dmax);
    }
    Kterm = Kk * imp * (pos - margin);
    // fp32 safeguard: a contact between a static body and a link whose COM sits on its joint axis has diagApprox = 0 (e.g. Panda link0-link1
    // once domain randomisation closes their 1 mm gap); MuJoCo's floor of 1e-15 gives D = 1e15, which multiplies the ~1e-7 rounding noise of
    // a single-precision Jacobian row into a real torque.  1e-7 keeps D J_noise^2 negligible against M and changes no regular contact
    // (their R is >= 1e-5).
    R = fmaxf(1e-7f, (1 - imp) / imp * diag);
  }

  // Row list: (1) friction-loss dofs, (2) joint limits (lower side first), (3) contacts in detection order.
  // The lanes that own the source objects (dof / joint / contact) publish a row descriptor and the row scalars; then lane r
  // builds Jacobian row r in registers, writes it once to LDS (MFMA operand) and finishes aref with its own J.qvel.
  // length of fixed tendon t = sum coef_w * q_w  (uniform or per-lane t; the tables are global, at most four joints per tendon)
  __device__ __forceinline__ float tendon_length(int t) const {
    const int adr = IT(IO_tendon_adr, t), num = IT(IO_tendon_num, t);
    float len = 0.f;
    for (int w = 0; w < num; w++) len = fmaf(FP(FO_wrap_prm, adr + w), sm.qpos[IT(IO_wrap_qadr, adr + w)], len);
    return len;
  }

  __device__ __forceinline__ void make_constraint() {
    const LaneConst K = fetchK();
    const int nv = m.nv;
    int nefc = 0;
    const bool isdof = lane < nv;
    const int jt = (K.dinfo >> 26) & 15, qa = (K.dinfo >> 10) & 255;
    // (0) equality/tendon rows come first (MuJoCo row order: equality, friction loss, limits, contacts): lane e owns constraint e;
    //     residual = (length - length0) - polycoef[0] of a fixed tendon, always active
    if (TENDONS && m.neq) {
      if (lane < m.neq) {
        const int t = IT(IO_eq_tendon, lane);
        const float pos = tendon_length(t) - FP(FO_tendon_len0, t) - FP(FO_eq_data0, lane);
        const float solref[2] = {FP(FO_eq_solref, 2 * lane), FP(FO_eq_solref, 2 * lane + 1)};
        float solimp[5];
        for (int k = 0; k < 5; k++) solimp[k] = FP(FO_eq_solimp, 5 * lane + k);
        float R, Bd, Kt;
        row_scalars(pos, 0.f, solref, solimp, FP(FO_tendon_invw, t), R, Bd, Kt);
        sm.e_desc[lane] = C_EQUALITY | (t << 4);
        sm.e_R[lane] = R; sm.e_force[lane] = Bd; sm.e_aref[lane] = Kt;
      }
      nefc += m.neq;
    }
    // (1) friction loss
    {
      const bool act = isdof && cmf(MK_fricFl)->fricFl[lane] > 0.f;
      const u64 mk = __ballot(act);
      if (act) {
        const int r = nefc + __popcll(mk & lanemask_lt(lane));
        if (r < NEFCAP) { sm.e_desc[r] = C_FRICTION_DOF | (lane << 4); sm.e_R[r] = cmf(MK_fricRB)->fricR[lane]; sm.e_force[r] = cmf(MK_fricRB)->fricB[lane]; sm.e_aref[r] = 0.f; }
      }
      nefc += __popcll(mk);
      if (nefc > NEFCAP) { ovf += nefc - NEFCAP; nefc = NEFCAP; }   // rows beyond the capacity (64 or 128) are dropped, and counted
    }
    // (1b) friction loss along fixed tendons (after the dof rows, mj_instantiateFriction [3P]): lane t owns tendon t
    if (TENDONS && m.ntendon) {
      const int tl = lane < m.ntendon ? lane : 0;
      const bool act = lane < m.ntendon && FP(FO_tendon_fl, tl) > 0.f;
      const u64 mk = __ballot(act);
      if (act) {
        const int r = nefc + __popcll(mk & lanemask_lt(lane));
        const float solref[2] = {FP(FO_tendon_solref_fri, 2 * tl), FP(FO_tendon_solref_fri, 2 * tl + 1)};
        float solimp[5];
        for (int k = 0; k < 5; k++) solimp[k] = FP(FO_tendon_solimp_fri, 5 * tl + k);
        float R, Bd, Kt;
        row_scalars(0.f, 0.f, solref, solimp, FP(FO_tendon_invw, tl), R, Bd, Kt);
        if (r < NEFCAP) { sm.e_desc[r] = C_FRICTION_TENDON | (lane << 4); sm.e_R[r] = R; sm.e_force[r] = Bd; sm.e_aref[r] = 0.f; }
      }
      nefc += __popcll(mk);
      if (nefc > NEFCAP) { ovf += nefc - NEFCAP; nefc = NEFCAP; }
    }
    // (2) joint limits: dof lane i owns its hinge / slide joint; lower side before upper side
    {
      const bool lim = isdof && ((K.dinfo >> 9) & 1) && (jt == JNT_HINGE || jt == JNT_SLIDE);
      const float qv = lim ? sm.qpos[qa] : 0.f;
      const float dlo = qv - K.jr0, dhi = K.jr1 - qv;
      const bool alo = lim && dlo < K.jmargin, ahi = lim && dhi < K.jmargin;
      const u64 mlo = __ballot(alo), mhi = __ballot(ahi);
      const int before = __popcll(mlo & lanemask_lt(lane)) + __popcll(mhi & lanemask_lt(lane));
      const float solref[2] = {K.jsr0, K.jsr1}, solimp[5] = {K.jsi0, K.jsi1, K.jsi2, K.jsi3, K.jsi4};
      if (alo) {
        const int r = nefc + before;
        float R, Bd, Kt;
        row_scalars(dlo, K.jmargin, solref, solimp, K.dinvw, R, Bd, Kt);
        if (r < NEFCAP) { sm.e_desc[r] = C_LIMIT_JOINT | (lane << 4) | (0 << 12); sm.e_R[r] = R; sm.e_force[r] = Bd; sm.e_aref[r] = Kt; }
      }
      if (ahi) {
        const int r = nefc + before + (alo ? 1 : 0);
        float R, Bd, Kt;
        row_scalars(dhi, K.jmargin, solref, solimp, K.dinvw, R, Bd, Kt);
        if (r < NEFCAP) { sm.e_desc[r] = C_LIMIT_JOINT | (lane << 4) | (1 << 12); sm.e_R[r] = R; sm.e_force[r] = Bd; sm.e_aref[r] = Kt; }
      }
      nefc += __popcll(mlo) + __popcll(mhi);
      if (nefc > NEFCAP) { ovf += nefc - NEFCAP; nefc = NEFCAP; }
    }
    // (2b) limits on fixed-tendon lengths: lane t owns tendon t, lower side before upper side
    if (TENDONS && m.ntendon) {
      const int tl = lane < m.ntendon ? lane : 0;
      const bool lim = lane < m.ntendon && IT(IO_tendon_limited, tl);
      const float len = lim ? tendon_length(tl) : 0.f, margin = FP(FO_tendon_margin, tl);
      const float dlo = len - FP(FO_tendon_range, 2 * tl), dhi = FP(FO_tendon_range, 2 * tl + 1) - len;
      const bool alo = lim && dlo < margin, ahi = lim && dhi < margin;
      const u64 mlo = __ballot(alo), mhi = __ballot(ahi);
      const int before = __popcll(mlo & lanemask_lt(lane)) + __popcll(mhi & lanemask_lt(lane));
      const float solref[2] = {FP(FO_tendon_solref, 2 * tl), FP(FO_tendon_solref, 2 * tl + 1)};
      float solimp[5];
      for (int k = 0; k < 5; k++) solimp[k] = FP(FO_tendon_solimp, 5 * tl + k);
      const float invw = FP(FO_tendon_invw, tl);
      if (alo) {
        const int r = nefc + before;
        float R, Bd, Kt;
        row_scalars(dlo, margin, solref, solimp, invw, R, Bd, Kt);
        if (r < NEFCAP) { sm.e_desc[r] = C_LIMIT_TENDON | (lane << 4) | (0 << 12); sm.e_R[r] = R; sm.e_force[r] = Bd; sm.e_aref[r] = Kt; }
      }
      if (ahi) {
        const int r = nefc + before + (alo ? 1 : 0);
        float R, Bd, Kt;
        row_scalars(dhi, margin, solref, solimp, invw, R, Bd, Kt);
        if (r < NEFCAP) { sm.e_desc[r] = C_LIMIT_TENDON | (lane << 4) | (1 << 12); sm.e_R[r] = R; sm.e_force[r] = Bd; sm.e_aref[r] = Kt; }
      }
      nefc += __popcll(mlo) + __popcll(mhi);
      if (nefc > NEFCAP) { ovf += nefc - NEFCAP; nefc = NEFCAP; }
    }
    SUBMARK_U(RP_X0);
    // (3) contacts: lane c owns contact c; exclusive scan of the active dimensions gives the first row of each block
    {
      csync();   // (CG builds) first reader of the contact block after the narrow phase
      const int ncon = uni(sm.ncon);
      const bool has = lane < ncon;
      const int dim = has ? sm.cdim[lane] : 0;
      // fp32: hulls that share a face plane (UR5e base / shoulder) come out of MPR at -5e-9 m, which is rounding, not penetration; a contact
      // becomes active 0.1 um inside the margin (MuJoCo: dist < margin), so that such pairs do not add constraint rows the fp64 path lacks
      const bool active = has && sm.cdist[lane] < (CG ? cmargin_rd(lane) : sm.cmargin[lane]) - 1.0e-7f;
      int need = active ? dim : 0, incl = need;
#pragma unroll
      for (int o = 1; o < SM::NCON_; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
      int first = nefc + incl - need;
      const bool fits = active && first + dim <= NEFCAP;
      // a block that does not fit is dropped together with everything after it (rows must stay contiguous)
      const u64 bad = __ballot(active && !fits);
      const bool keep = fits && (bad == 0 || lane < (__ffsll((long long)bad) - 1));
      ovf += __popcll(__ballot(active && !keep));
      if (has) sm.cefc[lane] = keep ? first : -1;
      if (keep) {
        const int b1 = sm.cg1[lane] >> 8, b2 = sm.cg2[lane] >> 8;
        const float tran = cmf(MK_biw)->biw[2 * b1] + cmf(MK_biw)->biw[2 * b2], rot = cmf(MK_biw)->biw[2 * b1 + 1] + cmf(MK_biw)->biw[2 * b2 + 1];
        float R0, Bd, Kt;
        float sr_[CG ? 2 : 1], si_[CG ? 5 : 1], fg_[CG ? 5 : 1];   // CG builds: the contact's parameters fetched from the global block
        if constexpr (CG) {
#pragma unroll
          for (int k = 0; k < 2; k++) sr_[k] = csolref_rd(2 * lane + k);
#pragma unroll
          for (int k = 0; k < 5; k++) { si_[k] = csolimp_rd(5 * lane + k); fg_[k] = cfri_rd(5 * lane + k); }
          row_scalars(sm.cdist[lane], cmargin_rd(lane), sr_, si_, tran, R0, Bd, Kt);
        } else row_scalars(sm.cdist[lane], sm.cmargin[lane], sm.csolref + 2 * lane, sm.csolimp + 5 * lane, tran, R0, Bd, Kt);
        const int type = dim == 1 ? C_CONTACT_FRICTIONLESS : C_CONTACT_ELLIPTIC;
        const float* f = CG ? fg_ : sm.cfri + 5 * lane;
        const float R1 = R0 / fmaxf(1e-15f, opt_impratio);
        (void)rot;  // friction rows: same gains, zero position term; their regularisers follow the cone scaling of R0
#pragma unroll
        for (int k = 0; k < CD; k++) {
          if (k < dim) {
            const int r = first + k;
            sm.e_desc[r] = type | (lane << 4) | (k << 12);
            sm.e_R[r] = k == 0 ? R0 : (k == 1 ? R1 : R1 * f[0] * f[0] / fmaxf(1e-15f, f[k - 1] * f[k - 1]));
            sm.e_force[r] = Bd; sm.e_aref[r] = k == 0 ? Kt : 0.f;
          }
        }
        sm.cmu[lane] = dim > 1 ? f[0] * sqrtf(R1 / R0) : 0.f;
      }
      const u64 kept = __ballot(keep);
      const int lastc = kept ? 63 - __clzll((long long)kept) : -1;
      const int add = lastc >= 0 ? __shfl(first + dim, lastc) - nefc : 0;
      // demand of this substep, had nothing been dropped (RSIM_CAP_NEED): the rows before the contacts (those stages drop only when they alone exceed
      // the capacity), the rows of every active contact in the list, and the largest contact dimension for each contact the narrow phase could not store
      const int want = nefc + __shfl(incl, SM::NCON_ - 1) + (con_raw - ncon) * m.maxcondim;
      need_efc = want > need_efc ? want : need_efc;
      need_con = con_raw > need_con ? con_raw : need_con;
      nefc += add;
    }
    if (lane == 0) sm.nefc = nefc;
    SYNC();
    SUBMARK_U(RP_X1);
    if constexpr (!FAST) { make_rows_wide(nefc); return; }
    // ---- lane r builds row r (and row r + 64 in the 128-row configuration)
#pragma unroll
    for (int slot = 0; slot < NSLOT; slot++) {
      const int row = lane + 64 * slot;
      const bool valid = row < nefc;
      const int desc = valid ? sm.e_desc[row] : 0;
      const int type = desc & 15, id = (desc >> 4) & 255, kk = (desc >> 12) & 15;
      float Jr[NV16];
#pragma unroll
      for (int k = 0; k < NV16; k++) Jr[k] = 0.f;
      if (valid && type == C_FRICTION_DOF) {
#pragma unroll
        for (int k = 0; k < NV16; k++) Jr[k] = k == id ? 1.f : 0.f;
      } else if (valid && type == C_LIMIT_JOINT) {
        const float sg = kk ? -1.f : 1.f;  // lower limit: +dq increases the distance; upper: decreases it
#pragma unroll
        for (int k = 0; k < NV16; k++) Jr[k] = k == id ? sg : 0.f;
      } else if (TENDONS && valid && (type == C_EQUALITY || type == C_LIMIT_TENDON || type == C_FRICTION_TENDON)) {
        // row of a fixed tendon: its coefficients on the dofs of its (at most four) joints; an upper limit takes the negative row
        const float sg = (type == C_LIMIT_TENDON && kk) ? -1.f : 1.f;
        const int adr = IT(IO_tendon_adr, id), num = IT(IO_tendon_num, id);
        int wd[4];
        float wc[4];
#pragma unroll
        for (int w = 0; w < 4; w++) { wd[w] = w < num ? IT(IO_wrap_dof, adr + w) : -1; wc[w] = w < num ? sg * FP(FO_wrap_prm, adr + w) : 0.f; }
#pragma unroll
        for (int k = 0; k < NV16; k++) Jr[k] = (wd[0] == k ? wc[0] : 0.f) + (wd[1] == k ? wc[1] : 0.f) + (wd[2] == k ? wc[2] : 0.f) + (wd[3] == k ? wc[3] : 0.f);
      } else if (valid) {
        const int c = id;
        const int b1 = sm.cg1[c] >> 8, b2 = sm.cg2[c] >> 8;
        const dmask_t d1 = cm->bdofs[b1], d2 = cm->bdofs[b2];
        const V3 pos = ld3(sm.cpos + 3 * c);
        const V3 ax = CG ? cframe_ax(9 * c + 3 * (kk < 3 ? kk : kk - 3)) : ld3(sm.cframe + 9 * c + 3 * (kk < 3 ? kk : kk - 3));
        const V3 o1 = pos - ld3(sm.rootcom + 3 * cm->broot[b1]), o2 = pos - ld3(sm.rootcom + 3 * cm->broot[b2]);
        const V3 t1 = cross(o1, ax), t2 = cross(o2, ax);  // ax . (ca x o) = ca . (o x ax)
        const bool lin = kk < 3;
#pragma unroll
        for (int k = 0; k < NV16; k++) {
          const S6 cd = ld6(sm.cdof + CS6 * k);
          const float s1 = (float)((d1 >> k) & 1), s2 = (float)((d2 >> k) & 1);
          const float dl = dot(ax, cd.l), da = dot(ax, cd.a);
          const float v1 = lin ? dl + dot(t1, cd.a) : da, v2 = lin ? dl + dot(t2, cd.a) : da;
          Jr[k] = s2 * v2 - s1 * v1;
        }
      }
      float jv = 0.f;
#pragma unroll
      for (int k = 0; k < NV16; k++) { Jwr(row * JS + k, Jr[k]); jv = fmaf(Jr[k], sm.qvel[k], jv); }
      if (valid) sm.e_aref[row] = -sm.e_force[row] * jv - sm.e_aref[row];
    }
    SYNC();
    if constexpr (JG) jsync();   // the rows have left the wavefront before the solver's lanes read other lanes' rows back from global memory
  }

  // Wide configurations: the Jacobian rows on the matrix cores.  A contact row is J[r][k] = s2(r, k) (w2_r . cdof_k) - s1(r, k) (w1_r . cdof_k) with
  // w = (o x ax | ax) for a translational axis, (ax | 0) for a rotational one, and s = "dof k moves the row's body": two 6-component products per
  // 16 x 16 tile (K = 6 in two k-steps) and a mask from the two bodies' dof sets, instead of nv unrolled cdof reads per row and lane.  The sparse
  // rows (friction loss, limits, tendons) stage zeros and write their one to four entries afterwards.
  __device__ __forceinline__ void make_rows_wide(int nefc) {
    const int nv = m.nv, q = lane >> 4, r = lane & 15;
    float* st = sm.u.rowst;   // [64][20], one slot (64 rows) of the lanes at a time: w2[8] | -w1[8] | mask2 lo hi | mask1 lo hi
    float cb[NT][2];   // B operands: cdof rows of every dof tile, components 4 kc + q (6 and 7 are the zero padding of the stride-9 rows)
#pragma unroll
    for (int ct = 0; ct < NT; ct++)
#pragma unroll
      for (int kc = 0; kc < 2; kc++) cb[ct][kc] = sm.cdof[(16 * ct + r) * CS6 + 4 * kc + q];
    const int nrt = (nefc + 15) >> 4;
    // slot by slot (round 6): the staging holds 64 rows whatever the row capacity -- 5 KB instead of 10 / 20 KB in the 128 / 256-row builds, where it was the
    // largest member of the phase union
#pragma unroll
    for (int slot = 0; slot < NSLOT; slot++) {
      if (64 * slot >= nefc) continue;   // this slot of the lanes holds no row (the 128 / 256-row configurations at the usual row counts)
      if (slot > 0) SYNC();              // the products of the slot before have read the staging
      const int row = lane + 64 * slot;
      const bool valid = row < nefc;
      const int desc = valid ? sm.e_desc[row] : 0;
      const int type = desc & 15, id = (desc >> 4) & 255, kk = (desc >> 12) & 15;
      float w2[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, w1[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      u64 m1 = 0, m2 = 0;
      if (valid && (type == C_CONTACT_FRICTIONLESS || type == C_CONTACT_ELLIPTIC)) {
        const int c = id;
        const int b1 = sm.cg1[c] >> 8, b2 = sm.cg2[c] >> 8;
        m1 = (u64)cm->bdofs[b1]; m2 = (u64)cm->bdofs[b2];
        const V3 pos = ld3(sm.cpos + 3 * c);
        const V3 ax = CG ? cframe_ax(9 * c + 3 * (kk < 3 ? kk : kk - 3)) : ld3(sm.cframe + 9 * c + 3 * (kk < 3 ? kk : kk - 3));
        if (kk < 3) {
          const V3 t1 = cross(pos - ld3(sm.rootcom + 3 * cm->broot[b1]), ax), t2 = cross(pos - ld3(sm.rootcom + 3 * cm->broot[b2]), ax);   // ax . (ca x o) = ca . (o x ax)
          w1[0] = -t1.x; w1[1] = -t1.y; w1[2] = -t1.z; w1[3] = -ax.x; w1[4] = -ax.y; w1[5] = -ax.z;
          w2[0] = t2.x; w2[1] = t2.y; w2[2] = t2.z; w2[3] = ax.x; w2[4] = ax.y; w2[5] = ax.z;
        } else {
          w1[0] = -ax.x; w1[1] = -ax.y; w1[2] = -ax.z;
          w2[0] = ax.x; w2[1] = ax.y; w2[2] = ax.z;
        }
      }
      float* o = st + 20 * lane;
#pragma unroll
      for (int k = 0; k < 6; k++) { o[k] = w2[k]; o[8 + k] = w1[k]; }
      o[6] = 0.f; o[7] = 0.f; o[14] = 0.f; o[15] = 0.f;
      ((unsigned*)o)[16] = (unsigned)m2; ((unsigned*)o)[17] = (unsigned)(m2 >> 32); ((unsigned*)o)[18] = (unsigned)m1; ((unsigned*)o)[19] = (unsigned)(m1 >> 32);
      SYNC();
      const int rt1 = nrt < 4 * slot + 4 ? nrt : 4 * slot + 4;
      for (int rt = 4 * slot; rt < rt1; rt++) {
        const int lt = rt - 4 * slot;   // row tile inside the staging
        const float* o = st + 20 * (16 * lt + r);
        const float a2[2] = {o[q], o[4 + q]}, a1[2] = {o[8 + q], o[12 + q]};
        u64 k2[4], k1[4];
#pragma unroll
        for (int v = 0; v < 4; v++) {
          const unsigned* w = (const unsigned*)(st + 20 * (16 * lt + 4 * q + v)) + 16;
          k2[v] = (u64)w[0] | ((u64)w[1] << 32); k1[v] = (u64)w[2] | ((u64)w[3] << 32);
        }
#pragma unroll
        for (int ct = 0; ct < NT; ct++) {
          if (16 * ct >= nv) {
#pragma unroll
            for (int v = 0; v < 4; v++) Jwr((16 * rt + 4 * q + v) * JS + 16 * ct + r, 0.f);
            continue;
          }
          v4f P2 = {0.f, 0.f, 0.f, 0.f}, P1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kc = 0; kc < 2; kc++) {
            P2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[kc], cb[ct][kc], P2, 0, 0, 0);
            P1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[kc], cb[ct][kc], P1, 0, 0, 0);
          }
#pragma unroll
          for (int v = 0; v < 4; v++) {
            const int k = 16 * ct + r;
            Jwr((16 * rt + 4 * q + v) * JS + k, (((k2[v] >> k) & 1ull) ? P2[v] : 0.f) + (((k1[v] >> k) & 1ull) ? P1[v] : 0.f));
          }
        }
      }
    }
    jsync();
    const float qv = lane < nv ? sm.qvel[lane] : 0.f;
#pragma unroll
    for (int slot = 0; slot < NSLOT; slot++) {
      if (64 * slot >= nefc) continue;
      const int row = lane + 64 * slot;
      const bool valid = row < nefc;
      const int desc = valid ? sm.e_desc[row] : 0;
      const int type = desc & 15, id = (desc >> 4) & 255, kk = (desc >> 12) & 15;
      const int jr0 = row * JS;
      if (valid && type == C_FRICTION_DOF) Jwr(jr0 + id, 1.f);
      else if (valid && type == C_LIMIT_JOINT) Jwr(jr0 + id, kk ? -1.f : 1.f);   // lower limit: +dq increases the distance; upper: decreases it
      else if (TENDONS && valid && (type == C_EQUALITY || type == C_LIMIT_TENDON || type == C_FRICTION_TENDON)) {
        // row of a fixed tendon: its coefficients on the dofs of its (at most four) joints; an upper limit takes the negative row
        const float sg = (type == C_LIMIT_TENDON && kk) ? -1.f : 1.f;
        const int adr = IT(IO_tendon_adr, id), num = IT(IO_tendon_num, id);
        for (int w = 0; w < num && w < 4; w++) { const int at = jr0 + IT(IO_wrap_dof, adr + w); Jwr(at, Jrd(at) + sg * FP(FO_wrap_prm, adr + w)); }
      }
    }
    jsync();
#pragma unroll
    for (int slot = 0; slot < NSLOT; slot++) {
      if (64 * slot >= nefc) continue;
      const int row = lane + 64 * slot;
      const float jv = j_row_dot(row * JS, qv);
      if (row < nefc) sm.e_aref[row] = -sm.e_force[row] * jv - sm.e_aref[row];
    }
    SYNC();
  }

  // ---------------------------------------------------------------- actuation / smooth acceleration
  __device__ __forceinline__ void actuation_acceleration() {
    const LaneConst K = fetchK();
    const int nv = m.nv;
    if (lane < NV16) sm.qfrc_actuator[lane] = 0.f;
    SYNC();
    if (lane < m.nu) {
      const int d = K.ainfo & 255, qa = (K.ainfo >> 8) & 255;
      float ctrl = sm.ctrl[lane];
      if ((K.ainfo >> 18) & 1) ctrl = fmaxf(K.acr0, fminf(K.acr1, ctrl));
      float force = K.again * ctrl;
      if (((K.ainfo >> 16) & 3) == 1) force += K.ab0 + K.ab1 * K.agear * sm.qpos[qa] + K.ab2 * K.agear * sm.qvel[d];
      if ((K.ainfo >> 19) & 1) force = fmaxf(K.afr0, fminf(K.afr1, force));
      atomicAdd(&sm.qfrc_actuator[d], K.agear * force);  // one actuator per dof in every supported model: order-independent
    }
    SYNC();
    float qs = 0.f;
    if (lane < nv) {
      qs = sm.qfrc_passive[lane] - sm.qfrc_bias[lane] + sm.qfrc_actuator[lane] + applied;   // mjData.qfrc_applied: the debug form only (0 in the fused kernels)
      sm.qfrc_smooth[lane] = qs;
    }
    float as;
    if constexpr (!FAST) as = bchol_solve<NVP>(sm.L, sm.invdiag, lane < nv ? qs : 0.f, nv, lane);
    else {
      float lr[NV16], lt[NV16], linv[NV16];
      const int rr = lane & (NV16 - 1);
#pragma unroll
      for (int k = 0; k < NV16; k++) { lr[k] = sm.L[rr * NVP + k]; lt[k] = sm.L[k * NVP + rr]; linv[k] = sm.invdiag[k]; }
      as = rchol_solve_m<NV16>(lr, lt, linv, sm.invdiag[rr], lane < nv ? qs : 0.f);
    }
    if (lane < nv) sm.qacc_smooth[lane] = as;
    SYNC();
  }

  // ---------------------------------------------------------------- semi-implicit Euler with implicit joint damping
  __device__ __forceinline__ void euler() {
    const LaneConst K = fetchK();
    const int nv = m.nv;
    const float h = opt_h;
    // Implicit joint damping (MuJoCo's Euler [3P]): qacc' = (M + hD)^-1 (qfrc_smooth + qfrc_constraint).  At the solver's optimum the right-hand side
    // IS M qacc, so qacc' = qacc - (M + hD)^-1 (hD qacc) -- the form used here.  In exact arithmetic and with a converged solver the two are the
    // same number; in fp32 they are not equally good: constraint forces are f = -D (J qacc - aref) with D up to 1e6 on a residual that cancels
    // to 1e-7 relative, i.e. good to ~1 % on a stiff contact, and Jt f divided by the 1e-4 kg m^2 inertia of an object (or the 5e-5 of a
    // Robotiq link) turned that into hundreds of rad/s^2 -- the traced cause of the PickPlace envs that hit the bad-state guard.  qacc itself
    // comes out of the Newton solve filtered by H^-1 (H = M + Jt D J, large in exactly those directions) and is accurate to rounding.
    float qa;
    const float acc_own = lane < nv ? sm.qacc[lane] : 0.f;
    if constexpr (!FAST && !SM::HAS_LE_) {
      // the factor of M + h diag(damping) is not kept in this configuration: build it in the solver's (now free) work matrix
      // Only the damped trees need it: M is block diagonal over the kinematic trees and the right-hand side hD qacc is zero on a tree without
      // damping (the free objects: 24 of PickPlace's 37 dofs, 12 of Stack's 21), so the correction there is exactly zero.  m.nv_damped = dofs up
      // to the last tree that has a damped joint in the model as ingested; a randomisation that gives a later dof damping falls back to all.
      const float hd = lane < nv ? h * K.damping : 0.f;
      const int nd = __ballot(lane >= m.nv_damped && hd != 0.f) ? nv : m.nv_damped;
      SYNC();
      const int nvt = (nd + 15) & ~15;
      for (int e = lane; e < nvt * nvt; e += 64) { const int i = e / nvt, j = e - i * nvt; sm.H[i * NVP + j] = Mrd(i * NVP + j); }
      SYNC();
      if (lane < nd) sm.H[lane * NVP + lane] += hd;
      SYNC();
      qa = acc_own;
      if (nd > 0) {
        bchol_inplace<NVP>(sm.H, sm.invdiag_e, nd, lane);
        const float corr = bchol_solve<NVP>(sm.H, sm.invdiag_e, lane < nd ? hd * acc_own : 0.f, nd, lane);
        if (lane < nd) qa -= corr;
      }
    } else if constexpr (!FAST) qa = chol_solve<NVP>(sm.Le, sm.invdiag_e, lane < nv ? sm.qfrc_smooth[lane] + sm.qfrc_constraint[lane] : 0.f, nv, lane);
    else {
      // register Cholesky of M + h diag(damping) (row r in lanes r, 16 + r, ...: K is fetched per 16-lane row); the transposed rows go through
      // the solver's work matrix, which is dead by now
      float er[NV16], einv[NV16], et[NV16];
      const int rr = lane & (NV16 - 1);
      const float hd = rr < nv ? h * K.damping : 0.f;
#pragma unroll
      for (int k = 0; k < NV16; k++) er[k] = sm.M[rr * NVP + k] + (k == rr ? hd : 0.f);
      const int ro = opaque_lane(rr);
      const float eown = rchol_factor_own<NV16>(er, einv, ro);
      rchol_mask_lower<NV16>(er, ro);
      SYNC();
      if (lane < NV16) {
#pragma unroll
        for (int k = 0; k < NV16; k++) sm.H[lane * NVP + k] = er[k];
      }
      SYNC();
#pragma unroll
      for (int k = 0; k < NV16; k++) et[k] = sm.H[k * NVP + rr];
      qa = acc_own - rchol_solve_m<NV16>(er, et, einv, eown, lane < nv ? hd * acc_own : 0.f);
    }
    if (lane < nv) { sm.qvel[lane] += h * qa; sm.qacc_ws[lane] = sm.qacc[lane]; }
    SYNC();
    // positions: lane b integrates the joint of body b
    const int jt = K.binfo & 15, pa = (K.binfo >> 4) & 255, da = (K.binfo >> 12) & 255;
    if (lane < m.nbody && jt != 15) {
      if (jt == JNT_FREE) {
        for (int k = 0; k < 3; k++) sm.qpos[pa + k] += h * sm.qvel[da + k];
        const V3 w = ld3(sm.qvel + da + 3);
        const float wn = norm(w), ang = wn * h;
        if (ang > 1e-15f) {
          float sn, cs;
          sincos_f(0.5f * ang, sn, cs);
          const float k = sn / wn;
          const Q4 dq = {cs, w.x * k, w.y * k, w.z * k};
          stq(sm.qpos + pa + 3, qnorm(qmul(ldq(sm.qpos + pa + 3), dq)));
        }
      } else sm.qpos[pa] += h * sm.qvel[da];
    }
    SYNC();
  }

  // ---------------------------------------------------------------- built-in controller: OSC_POSE + GRIP
  __device__ __forceinline__ void ctrl_set_goal(const float* action) {
    const LaneConst K = fetchK();
    const DCtrl& c = m.ctrl;
    // variable-impedance action layouts (osc.py:243-253, joint_pos.py:204-214): [damping_ratio x n, kp x n, goal update] or [kp x n, goal update]
    if (c.imp_mode) {
      const int li = lane & (RSIM_JNT_MAX - 1);
      const float kmin = sel(c.kp_min, li), kmax = sel(c.kp_max, li), dmin = sel(c.dr_min, li), dmax = sel(c.dr_max, li);
      if (lane < c.nimp) {
        const float kp = fmaxf(kmin, fminf(kmax, action[(c.imp_mode == 1 ? c.nimp : 0) + lane]));
        const float dr = c.imp_mode == 1 ? fmaxf(dmin, fminf(dmax, action[lane])) : 1.f;
        cst[RSIM_CS_KP + lane] = kp;
        cst[RSIM_CS_KD + lane] = 2.f * sqrtf(kp) * dr;
      }
      action += c.nimp * (c.imp_mode == 1 ? 2 : 1);
    }
    if (c.type >= RSIM_CTRL_JOINT_POSITION) {
      // joint-space parts: lane i scales its own component (controller.py:149-168).  JOINT_POSITION: goal_qpos = joint_pos + delta
      // (joint_pos.py:200-236); JOINT_TORQUE: goal_torque = clip(scaled, torque_limits) (joint_tor.py:111-128)
      const int li = lane & (RSIM_JNT_MAX - 1);
      const float imin = sel(c.in_min, li), imax = sel(c.in_max, li), omin = sel(c.out_min, li), omax = sel(c.out_max, li);
      const float tlo = sel(c.tl_lo, li), thi = sel(c.tl_hi, li);
      if (lane < c.ndof) {
        if (c.interp_steps) cst[RSIM_CS_ISTART + lane] = sm.cstate[RSIM_CS_GOALQ + lane];   // LinearInterpolator.set_goal: start := previous goal
        const float scale = fabsf(omax - omin) / fabsf(imax - imin);
        const float a = fmaxf(imin, fminf(imax, action[lane]));
        const float sv = (a - 0.5f * (imax + imin)) * scale + 0.5f * (omax + omin);
        // JOINT_VELOCITY: goal_vel = clip(scaled, velocity_limits) (joint_vel.py:145-148) -- same clamp, limits in the same slots
        sm.cstate[RSIM_CS_GOALQ + lane] = c.type == RSIM_CTRL_JOINT_POSITION ? sm.qpos[K.cq] + sv : fmaxf(tlo, fminf(thi, sv));
      }
      if (lane < c.ngrip) {
        const float a = action[c.cdim], sg = a > 0 ? 1.f : (a < 0 ? -1.f : 0.f);
        sm.cstate[RSIM_CS_GRIP + lane] = fmaxf(-1.f, fminf(1.f, sm.cstate[RSIM_CS_GRIP + lane] + K.cgs * c.grip_speed * sg));
      }
      if (c.interp_steps && lane == 0) cst[RSIM_CS_ISTEP] = 0.f;
      SYNC();
      cst_sync();
      return;
    }
    ctrl_set_goal_osc<0>(K, action);
    // a second arm part (Baxter's default: one OSC object per arm, composite_controller.py:97-116): its slice of the action follows the first
    // arm's (and that arm's gripper entry)
    if constexpr (TWO_ARMS) { if (c.narm == 2) ctrl_set_goal_osc<1>(fetchK<1>(), action + c.cdim + (c.ngrip > 0 ? 1 : 0)); }
  }
  // OSC set_goal of arm ARM (osc.py:225-300).  The arm index is a template constant: every access to the by-value controller description
  // keeps a compile-time index (a run-time arm index would put a copy of DModel into the private segment)
  template <int ARM>
  __device__ __forceinline__ void ctrl_set_goal_osc(const LaneConst& K, const float* action) {
    const DCtrl& c = m.ctrl;
    constexpr int AO = RSIM_ARM_MAX * ARM, CO = RSIM_CS_SIZE * ARM;
    const int eef_site = ARM ? c.eef_site2 : c.eef_site, base_site = ARM ? c.base_site2 : c.base_site;
    float sc[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {   // OSC_POSITION (cdim 3): zero orientation delta, osc.py:255-263
      if (i < c.cdim) {
        // the limits are kernel arguments: without the opaque copies the 18 derived constants are computed at kernel entry and parked in the
        // private segment until this (once per launch) call -- 100 bytes of scratch per lane, 40 MB of HBM writes per launch
        float omax = c.out_max[AO + i], omin = c.out_min[AO + i], imax = c.in_max[AO + i], imin = c.in_min[AO + i];
        asm volatile("" : "+s"(omax), "+s"(omin), "+s"(imax), "+s"(imin));
        float scale = fabsf(omax - omin) / fabsf(imax - imin);
        float a = fmaxf(imin, fminf(imax, action[i]));
        sc[i] = (a - 0.5f * (imax + imin)) * scale + 0.5f * (omax + omin);
      } else sc[i] = 0.f;
    }
    V3 op = ld3(sm.spos + 3 * base_site), ep = ld3(sm.spos + 3 * eef_site);
    M3 oR = ldm(sm.smat + 9 * base_site), eR = ldm(sm.smat + 9 * eef_site);
    V3 gp = mtv(oR, ep - op) + v3(sc[0], sc[1], sc[2]);
    V3 d = v3(sc[3], sc[4], sc[5]);
    float ang = norm(d);
    Q4 qe = {1, 0, 0, 0};
    if (ang != 0.f) { float sn, cs; sincos_f(0.5f * ang, sn, cs); sn /= ang; qe.w = cs; qe.x = d.x * sn; qe.y = d.y * sn; qe.z = d.z * sn; }
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
      if (c.interp_steps) {
        st3(cst + RSIM_CS_ISTART, ld3(sm.cstate + CO + RSIM_CS_GOALPOS)); cst[RSIM_CS_ISTEP] = 0.f;
        if (c.type == RSIM_CTRL_OSC_POSE) {   // osc.py:277-283: ori_ref = current eef orientation, goal = error of the (base-frame) goal_ori against it
          st3(cst + RSIM_CS_ISTART_ORI, ld3(cst + RSIM_CS_IGOAL_ORI));
          st3(cst + RSIM_CS_IGOAL_ORI, (cross(col(eR, 0), col(go, 0)) + cross(col(eR, 1), col(go, 1)) + cross(col(eR, 2), col(go, 2))) * 0.5f);
        }
      }
      st3(sm.cstate + CO + RSIM_CS_GOALPOS, gp);
      stm(sm.cstate + CO + RSIM_CS_GOALORI, go);
    }
    if (ARM == 0 && lane < c.ngrip) {
      const float a = action[c.cdim], sg = a > 0 ? 1.f : (a < 0 ? -1.f : 0.f);
      sm.cstate[RSIM_CS_GRIP + lane] = fmaxf(-1.f, fminf(1.f, sm.cstate[RSIM_CS_GRIP + lane] + K.cgs * c.grip_speed * sg));
    }
    SYNC();
    cst_sync();
  }

  // reset_goal + initial_joint capture (Controller.__init__ / OSC.reset_goal)
  __device__ __forceinline__ void ctrl_reset() {
    const LaneConst K = fetchK();
    const DCtrl& c = m.ctrl;
    if (c.type < RSIM_CTRL_JOINT_POSITION && lane < c.ndof) sm.cstate[RSIM_CS_Q0 + lane] = sm.qpos[K.cq];
    if (TWO_ARMS && c.type < RSIM_CTRL_JOINT_POSITION && c.narm == 2) {   // second arm part: its own initial_joint / goal block
      const LaneConst K1 = fetchK<1>();
      if (lane < c.ndof2) sm.cstate[RSIM_CS_SIZE + RSIM_CS_Q0 + lane] = sm.qpos[K1.cq];
      if (lane == 0) {
        st3(sm.cstate + RSIM_CS_SIZE + RSIM_CS_GOALPOS, ld3(sm.spos + 3 * c.eef_site2));
        for (int k = 0; k < 9; k++) sm.cstate[RSIM_CS_SIZE + RSIM_CS_GOALORI + k] = sm.smat[9 * c.eef_site2 + k];
      }
    }
    if (c.imp_mode) {   // the constructor's gains stay in force until the first set_goal
      const int li = lane & (RSIM_JNT_MAX - 1);
      const float kp0 = sel(c.kp, li), kd0 = sel(c.kd, li);
      if (lane < c.nimp) { cst[RSIM_CS_KP + lane] = kp0; cst[RSIM_CS_KD + lane] = kd0; }
    }
    if (c.type >= RSIM_CTRL_JOINT_POSITION) {   // joint_pos.py:268-276 (goal_qpos = joint_pos), joint_tor.py:170-178 (goal_torque = 0)
      if (lane < c.ndof) sm.cstate[RSIM_CS_GOALQ + lane] = c.type == RSIM_CTRL_JOINT_POSITION ? sm.qpos[K.cq] : 0.f;
      if (lane < RSIM_GRIP_MAX) sm.cstate[RSIM_CS_GRIP + lane] = 0.f;
      // JOINT_VELOCITY: fresh PID state (joint_vel.py:105-110; RingBuffer starts with ptr = length - 1, size 0); the caller zeroed the block
      if (c.type == RSIM_CTRL_JOINT_VELOCITY && lane == 0) cst[RSIM_CS_JV_PTR] = 4.f;
    } else if (lane == 0) {
      st3(sm.cstate + RSIM_CS_GOALPOS, ld3(sm.spos + 3 * c.eef_site));
      for (int k = 0; k < 9; k++) sm.cstate[RSIM_CS_GOALORI + k] = sm.smat[9 * c.eef_site + k];
      for (int i = 0; i < RSIM_GRIP_MAX; i++) sm.cstate[RSIM_CS_GRIP + i] = 0.f;
    }
    SYNC();
    cst_sync();
  }

  // OperationalSpaceController.run_controller (osc.py:403-495) + SimpleGripController, tau clipped into ctrl.
  // Lambda^-1 = J Ma^-1 J^T = Y^T Y with Y = La^-1 J^T (register Cholesky of the arm block, 6 forward solves);
  // N^T Ma tmp = Ma tmp - J^T Lambda (J tmp), so no explicit inverse of Ma is ever formed.
  // arm torques -> clipped ctrl (fixed_base_robot.py:143-153) + SimpleGripController output (simple_grip.py:150-186)
  template <int ARM = 0>
  __device__ __forceinline__ void ctrl_write(const LaneConst& K, float tq) {
    const DCtrl& c = m.ctrl;
    const float alo = __shfl(K.acr0, K.ca), ahi = __shfl(K.acr1, K.ca);
    if (lane < (ARM ? c.ndof2 : c.ndof)) {
      sm.cstate[RSIM_CS_SIZE * ARM + (c.type >= RSIM_CTRL_JOINT_POSITION ? RSIM_CS_TAU_JOINT : RSIM_CS_TAU) + lane] = tq;
      sm.ctrl[K.ca] = fmaxf(alo, fminf(ahi, tq));
    }
    const float glo = __shfl(K.acr0, K.cga), ghi = __shfl(K.acr1, K.cga);
    if (ARM == 0 && lane < c.ngrip) sm.ctrl[K.cga] = fmaxf(glo, fminf(ghi, 0.5f * (ghi + glo) + 0.5f * (ghi - glo) * sm.cstate[RSIM_CS_GRIP + lane]));
    SYNC();
    cst_sync();
  }

  // JointPositionController.run_controller (joint_pos.py:238-266): tau = M_arm (kp (goal - q) - kd qd) + qfrc_bias[arm];
  // JointTorqueController.run_controller (joint_tor.py:130-167): tau = goal_torque + qfrc_bias[arm].  Lane i owns arm joint i.
  __device__ __forceinline__ void ctrl_run_joint(const LaneConst& K) {
    const DCtrl& c = m.ctrl;
    const int n = c.ndof;
    constexpr int NA = RSIM_JNT_MAX;
    const int li = lane & (NA - 1);
    const int di = lane < n ? K.cd : 0, qi = lane < n ? K.cq : 0;
    float goal = lane < n ? sm.cstate[RSIM_CS_GOALQ + lane] : 0.f;
    if (c.interp_steps) {   // LinearInterpolator.get_interpolated_goal (traj_utils.py:118-155)
      const float step = cst[RSIM_CS_ISTEP], start = lane < n ? cst[RSIM_CS_ISTART + lane] : 0.f;
      goal = start + (goal - start) / ((float)c.interp_steps - step);
      SYNC();
      if (lane == 0 && step < (float)(c.interp_steps - 1)) cst[RSIM_CS_ISTEP] = step + 1.f;
    }
    float tq = lane < n ? sm.qfrc_bias[di] : 0.f;
    if (c.type == RSIM_CTRL_JOINT_POSITION) {
      const float kpj = c.imp_mode ? cst[RSIM_CS_KP + li] : sel(c.kp, li), kdj = c.imp_mode ? cst[RSIM_CS_KD + li] : sel(c.kd, li);
      const float des = lane < n ? kpj * (goal - sm.qpos[qi]) - kdj * sm.qvel[di] : 0.f;
      const int mypart = seli(c.part_of, li);
#pragma unroll
      for (int k = 0; k < NA; k++) {   // the part's own mass-matrix block (each arm of a multi-arm robot is its own controller object)
        const float mk = (lane < n && k < n && c.part_of[k] == mypart) ? Mrd(di * NVP + c.dof_idx[k]) : 0.f;
        tq = fmaf(mk, bcast(des, k), tq);
      }
    } else if (c.type == RSIM_CTRL_JOINT_VELOCITY) {
      // joint_vel.py:166-198: err = goal - qd; derr ring (5) mean; integrator frozen while this part saturated on the previous call
      const float kp = sel(c.kp, li), alo = __shfl(K.acr0, K.ca), ahi = __shfl(K.acr1, K.ca);
      const int mypart = seli(c.part_of, li);
      const int ptr = ((int)cst[RSIM_CS_JV_PTR] + 1) % 5, size = min((int)cst[RSIM_CS_JV_SIZE] + 1, 5);
      float pid = 0.f;
      bool over = false;
      if (lane < n) {
        const float err = goal - sm.qvel[di], derr = err - sm.cstate[RSIM_CS_JV_LASTERR + lane];
        sm.cstate[RSIM_CS_JV_LASTERR + lane] = err;
        cst[RSIM_CS_JV_RING + 16 * ptr + lane] = derr;
        float summed = cst[RSIM_CS_JV_SUMMED + lane];
        if (cst[RSIM_CS_JV_SAT + mypart] == 0.f) summed += err;
        cst[RSIM_CS_JV_SUMMED + lane] = summed;
        float avg = 0.f;
        for (int k = 0; k < size; k++) avg += cst[RSIM_CS_JV_RING + 16 * k + lane];   // RingBuffer.average: mean of buf[:size]
        avg /= (float)size;
        pid = kp * err + 0.005f * kp * summed + 0.001f * kp * avg;
        const float raw = pid + tq;
        over = fmaxf(alo, fminf(ahi, raw)) != raw;
      }
      tq += pid;
      SYNC();
#pragma unroll
      for (int p2 = 0; p2 < 4; p2++) {   // saturated = any clipped torque within the part (one controller object per arm)
        const bool any = __ballot(over && mypart == p2) != 0;
        if (lane == 0) cst[RSIM_CS_JV_SAT + p2] = any ? 1.f : 0.f;
      }
      if (lane == 0) { cst[RSIM_CS_JV_PTR] = (float)ptr; cst[RSIM_CS_JV_SIZE] = (float)size; }
    } else tq += goal;
    ctrl_write(K, tq);
  }

  __device__ __forceinline__ void ctrl_run() {
    const DCtrl& c = m.ctrl;
    if (c.type >= RSIM_CTRL_JOINT_POSITION) { ctrl_run_joint(fetchK()); return; }
    ctrl_run_osc<0>();
    if constexpr (TWO_ARMS) { if (c.narm == 2) { phase(); ctrl_run_osc<1>(); } }   // one OSC object per arm (composite_controller.py:97-116), each on its own mass-matrix block
  }
  template <int ARM>
  __device__ __forceinline__ void ctrl_run_osc() {
    const LaneConst K = fetchK<ARM>();
    const DCtrl& c = m.ctrl;
    constexpr int AO = RSIM_ARM_MAX * ARM, CO = RSIM_CS_SIZE * ARM;
    const int n = ARM ? c.ndof2 : c.ndof;
    const int eef_site = ARM ? c.eef_site2 : c.eef_site, base_site = ARM ? c.base_site2 : c.base_site;
    constexpr int NA = RSIM_ARM_MAX;
    float* Li = sm.u.k.Li;             // [6][6]   Lambda^-1
    float* vv = sm.u.k.vv;             // 6: J tmp
    const int eb = __shfl(K.sbody, eef_site), bb = __shfl(K.sbody, base_site);
    const V3 ep = ld3(sm.spos + 3 * eef_site), op = ld3(sm.spos + 3 * base_site);
    // this lane's arm dof (lanes 0..n-1) and its joint-space quantities
    const int di = lane < n ? K.cd : 0, qi = lane < n ? K.cq : 0;
    const float qd_i = lane < n ? sm.qvel[di] : 0.f;
    const float tmp_i = lane < n ? c.nullspace_kp * (sm.cstate[CO + RSIM_CS_Q0 + lane] - sm.qpos[qi]) - 2.f * sqrtf(c.nullspace_kp) * qd_i : 0.f;
    // Jacobian column of the eef site for this lane's dof
    S6 jc = {v3(0, 0, 0), v3(0, 0, 0)};
    if (lane < n) jc = jac_col(eb, ep, di);
    // arm block of M: row i in lane i, Cholesky in registers
    const int row8 = opaque_lane(lane & (NA - 1));
    float mr[NA], minv[NA];
    {
      int dk[NA];
#pragma unroll
      for (int k = 0; k < NA; k++) dk[k] = k < n ? c.dof_idx[AO + k] : 0;
#pragma unroll
      for (int k = 0; k < NA; k++) { const float v = Mrd(di * NVP + dk[k]); mr[k] = (lane < n && k < n) ? v : (row8 == k ? 1.f : 0.f); }   // di = 0 on padding lanes
    }
    float matmp = 0.f;   // (Ma tmp)_i
#pragma unroll
    for (int k = 0; k < NA; k++) matmp = fmaf(mr[k], bcast(tmp_i, k), matmp);
    const float mown = rchol_factor_own<NA>(mr, minv, row8);
    rchol_mask_lower<NA>(mr, row8);
    // Y[:, r] = La^-1 J[r, :]^T  (component i in lane i)
    float Y[6];
    Y[0] = rchol_fwd_m<NA>(mr, minv, mown, jc.l.x); Y[1] = rchol_fwd_m<NA>(mr, minv, mown, jc.l.y); Y[2] = rchol_fwd_m<NA>(mr, minv, mown, jc.l.z);
    Y[3] = rchol_fwd_m<NA>(mr, minv, mown, jc.a.x); Y[4] = rchol_fwd_m<NA>(mr, minv, mown, jc.a.y); Y[5] = rchol_fwd_m<NA>(mr, minv, mown, jc.a.z);
    SUBMARK_OSC(RP_X7);
    // Lambda^-1[r][q] = sum_i Y[i][r] Y[i][q] and (J tmp)[r] = sum_i J[r][i] tmp_i in one 16 x 16 x 8 product on the matrix cores:
    // staging row i (arm joint) = [Y[i][0..5] | J[0..5][i] | tmp_i | 0 0 0]; A[a][i] = S[i][a], B[i][b] = (b < 6 ? S[i][b] : b == 6 ? tmp_i : 0)
    // => D[r][q] = Lambda^-1 (r, q < 6, exactly symmetric: same products in the same order) and D[6 + r][6] = (J tmp)[r]
    {
      float* S = sm.u.k.Ys;   // [NA][16]
      static_assert(NA == 8, "two K = 4 chunks");
      if (lane < NA) {
#pragma unroll
        for (int r = 0; r < 6; r++) { S[lane * 16 + r] = Y[r]; S[lane * 16 + 6 + r] = comp6(jc, r < 3 ? r + 3 : r - 3); }
        S[lane * 16 + 12] = tmp_i; S[lane * 16 + 13] = 0.f; S[lane * 16 + 14] = 0.f; S[lane * 16 + 15] = 0.f;
      }
      SYNC();
      const int col = lane & 15, kq = lane >> 4;
      const float a0 = S[kq * 16 + col], a1 = S[(4 + kq) * 16 + col];
      const int bi = col < 6 ? col : 12;
      float b0 = S[kq * 16 + bi], b1 = S[(4 + kq) * 16 + bi];
      if (col > 6) { b0 = 0.f; b1 = 0.f; }
      v4f acc = {0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc, 0, 0, 0);
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const int row = 4 * kq + v;
        if (col < 6 && row < 6) Li[row * 6 + col] = acc[v];
        if (col == 6 && row >= 6 && row < 12) vv[row - 6] = acc[v];
      }
    }
    SYNC();
    SUBMARK_OSC(RP_X8);
    // operational-space errors and wrench (uniform small algebra)
    const M3 oR = ldm(sm.smat + 9 * base_site), eR = ldm(sm.smat + 9 * eef_site);
    const V3 gpos = ld3(sm.cstate + CO + RSIM_CS_GOALPOS);
    const M3 gori = ldm(sm.cstate + CO + RSIM_CS_GOALORI);
    V3 perr = op + mv(oR, gpos) - ep;
    const float istep = c.interp_steps ? cst[RSIM_CS_ISTEP] : 0.f;
    if (c.interp_steps) {
      // osc.py:418-423: with an interpolator the (base-frame) goal values, linearly ramped, ARE the desired world position
      const float step = istep;
      const V3 start = ld3(cst + RSIM_CS_ISTART);
      perr = start + (gpos - start) * (1.0f / ((float)c.interp_steps - step)) - ep;
      SYNC();
      if (lane == 0 && step < (float)(c.interp_steps - 1)) cst[RSIM_CS_ISTEP] = step + 1.f;
    }
    const M3 dori = mm(oR, gori);
    V3 oerr = (cross(col(eR, 0), col(dori, 0)) + cross(col(eR, 1), col(dori, 1)) + cross(col(eR, 2), col(dori, 2))) * 0.5f;
    if (c.interp_steps && c.type == RSIM_CTRL_OSC_POSE)   // osc.py:433-437: the ramped error vector replaces the measured one
      oerr = euler_slerp(ld3(cst + RSIM_CS_ISTART_ORI), ld3(cst + RSIM_CS_IGOAL_ORI), (istep + 1.f) / (float)c.interp_steps);
    // site velocities from the body spatial velocities of the velocity stage: v = cvel.l + w x (p - com)
    const S6 ce = ld6(sm.u.v.cvel + CS6 * eb), cb = ld6(sm.u.v.cvel + CS6 * bb);
    const V3 evl = ce.l + cross(ce.a, ep - ld3(sm.rootcom + 3 * cm->broot[eb])), bvl = cb.l + cross(cb.a, op - ld3(sm.rootcom + 3 * cm->broot[bb]));
    const V3 dvl = evl - bvl, dva = ce.a - cb.a;
    float kp6[6], kd6[6];
#pragma unroll
    for (int i = 0; i < 6; i++) { kp6[i] = c.imp_mode ? cst[RSIM_CS_KP + i] : c.kp[AO + i]; kd6[i] = c.imp_mode ? cst[RSIM_CS_KD + i] : c.kd[AO + i]; }
    float F[3] = {perr.x * kp6[0] - dvl.x * kd6[0], perr.y * kp6[1] - dvl.y * kd6[1], perr.z * kp6[2] - dvl.z * kd6[2]};
    float T[3] = {oerr.x * kp6[3] - dva.x * kd6[3], oerr.y * kp6[4] - dva.y * kd6[4], oerr.z * kp6[5] - dva.z * kd6[5]};
    float wrench[6], z[6];
    // 6x6 SPD solves with Lambda^-1: register Cholesky, row r in lane r
    float lr6[NA], linv6[NA], lt6[NA];
#pragma unroll
    for (int k = 0; k < NA; k++) lr6[k] = ((lane & (NA - 1)) < 6 && k < 6) ? Li[(lane & (NA - 1)) * 6 + k] : ((lane & (NA - 1)) == k ? 1.f : 0.f);
    if (c.uncouple) {
      float lp[9], lo[9];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int q = 0; q < 3; q++) { lp[r * 3 + q] = Li[r * 6 + q]; lo[r * 3 + q] = Li[(3 + r) * 6 + 3 + q]; }
      solve3(lp, F, wrench);
      solve3(lo, T, wrench + 3);
    }
    SUBMARK_OSC(RP_X9);
    const float own6 = rchol_factor_own<NA>(lr6, linv6, row8);
    rchol_mask_lower<NA>(lr6, row8);
    SYNC();
    float* Lt = sm.u.k.Lt;        // [8][8] transpose staging
    if (lane < NA) {
#pragma unroll
      for (int k = 0; k < NA; k++) Lt[lane * NA + k] = lr6[k];
    }
    SYNC();
#pragma unroll
    for (int k = 0; k < NA; k++) lt6[k] = Lt[k * NA + (lane & (NA - 1))];
    {
      const float zl = rchol_solve_m<NA>(lr6, lt6, linv6, own6, lane < 6 ? vv[lane] : 0.f);
#pragma unroll
      for (int r = 0; r < 6; r++) z[r] = bcast(zl, r);
      if (!c.uncouple) {
        const float wl = rchol_solve_m<NA>(lr6, lt6, linv6, own6, lane < 3 ? F[lane] : (lane < 6 ? T[lane - 3] : 0.f));
#pragma unroll
        for (int r = 0; r < 6; r++) wrench[r] = bcast(wl, r);
      }
    }
    float tq = 0.f;
    if (lane < n) {
      tq = sm.qfrc_bias[di] + matmp;
#pragma unroll
      for (int r = 0; r < 6; r++) tq = fmaf(comp6(jc, r < 3 ? r + 3 : r - 3), wrench[r] - z[r], tq);
    }
    ctrl_write<ARM>(K, tq);
  }

  // ---------------------------------------------------------------- Newton solver (primal): lane r owns constraint row r
  // Row data (Jacobian row, D, R, aref) live in the owner lane's registers for the whole solve; the dense products
  // H = M + J^T W and J^T f run on the matrix cores (v_mfma_f32_16x16x4_f32, 4 rows per instruction); the Hessian
  // factorisation is the register-resident Cholesky above.  Algorithm = oracle solve_newton (MuJoCo's primal Newton).
  struct Row {
    float J[FAST ? NV16 : 1];   // one-tile configuration: the Jacobian row stays in registers; wide configurations read it from LDS / the global buffer
    float D, R, aref, fl, mu, fr_own, Dm;
    float fj[CD - 1];
    int row, type, head, kk, dim;
    bool valid, ell;
  };
  // sum_k r[k] * x_k.  One-tile configuration: x is replicated in every 16-lane row (DPP row broadcast); wide: x_k lives in lane k (readlane)
  __device__ __forceinline__ float row_dot(const Row& rw, float x) const {
    if constexpr (FAST) return dot_rows<NV16>(rw.J, x);
    else return j_row_dot(rw.row * JS, x);
  }
  // wide configurations: sum_k p[k] x_k over the 16-column tiles that hold dofs (columns nv .. 16 ceil(nv / 16) - 1 hold zeros, x_k = 0 there).
  // One wavefront per SIMD and nothing else to switch to: a read-then-use loop pays the full LDS latency per element, so the sixteen reads of a
  // tile are issued back to back and consumed afterwards.
  // row of J at element offset `base` times x (JG builds: the row comes from global memory, all its tiles in one batch of loads)
  __device__ __forceinline__ float j_row_dot(int base, float x) const {
    if constexpr (!JG) return lds_row_dot(sm.J + base, x);
    else {
      float a[NV16];
#pragma unroll
      for (int u = 0; u < NV16; u++) a[u] = Jg[base + u];
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < NV16; u++) acc = fmaf(a[u], bcast(x, u), acc);   // columns nv .. NV16 - 1 hold zeros
      return acc;
    }
  }
  __device__ __forceinline__ double row_res64(const Row& rw, float x, float xl) const {
    return j_row_res64(rw.row * JS, x, xl, rw.aref);
  }
  __device__ __forceinline__ double j_row_res64(int base, float x, float xl, float aref) const {
    if constexpr (!JG) return lds_row_res64(sm.J + base, x, xl, aref);
    else {
      float a[NV16];
#pragma unroll
      for (int u = 0; u < NV16; u++) a[u] = Jg[base + u];
      double acc = -(double)aref;
      float lo = 0.f;
#pragma unroll
      for (int u = 0; u < NV16; u++) { acc = fma((double)a[u], (double)bcast(x, u), acc); lo = fmaf(a[u], bcast(xl, u), lo); }
      return acc + (double)lo;
    }
  }
  __device__ __forceinline__ float lds_row_dot(const float* p, float x) const {
    float acc = 0.f;
    for (int k0 = 0; k0 < m.nv; k0 += 16) {
      float a[16];
#pragma unroll
      for (int u = 0; u < 16; u++) a[u] = p[k0 + u];
#pragma unroll
      for (int u = 0; u < 16; u++) acc = fmaf(a[u], bcast(x, k0 + u), acc);
    }
    return acc;
  }
  // Residual J_r . (x + xl) - aref of a constraint row with the products summed in fp64 (the refinement pass of the wide Newton solver).  J's entries and both
  // parts of the acceleration are floats, so every product is exact in fp64.
  __device__ __forceinline__ double lds_row_res64(const float* p, float x, float xl, float aref) const {
    double acc = -(double)aref;
    float lo = 0.f;
    for (int k0 = 0; k0 < m.nv; k0 += 16) {
      float a[16];
#pragma unroll
      for (int u = 0; u < 16; u++) a[u] = p[k0 + u];
#pragma unroll
      for (int u = 0; u < 16; u++) { acc = fma((double)a[u], (double)bcast(x, k0 + u), acc); lo = fmaf(a[u], bcast(xl, k0 + u), lo); }
    }
    return acc + (double)lo;
  }
  // (M x)_i for the dof of this lane
  __device__ __forceinline__ float mass_dot(const float (&Mr)[FAST ? NV16 : 1], float x) const {
    if constexpr (FAST) return dot_rows<NV16>(Mr, x);
    else {
      float acc;
      if constexpr (MG) {
        float a[NV16];
        const int base = (lane < m.nv ? lane : 0) * NVP;
#pragma unroll
        for (int u = 0; u < NV16; u++) a[u] = Mg[base + u];
        acc = 0.f;
#pragma unroll
        for (int u = 0; u < NV16; u++) acc = fmaf(a[u], bcast(x, u), acc);   // columns nv .. NV16 - 1 of a dof row hold zeros (identity padding sits on the diagonal of the padding ROWS)
      } else acc = lds_row_dot(sm.M + (lane < m.nv ? lane : 0) * NVP, x);
      return lane < m.nv ? acc : 0.f;
    }
  }
  // gather every block's friction-scaled values: out[s][j] = (x * fr_own) of row head(s) + j, which lives in lane (head + j) & 63, slot (head + j) >> 6
  __device__ __forceinline__ void gather(const Row (&rw)[NSLOT], const float (&x)[NSLOT], float (&out)[NSLOT][CD]) const {
    float u[NSLOT];
#pragma unroll
    for (int s = 0; s < NSLOT; s++) u[s] = x[s] * rw[s].fr_own;
#pragma unroll
    for (int s = 0; s < NSLOT; s++)
#pragma unroll
      for (int j = 0; j < CD; j++) {
        const int R = rw[s].head + j;
        float t = __shfl(u[0], R & 63);
#pragma unroll
        for (int s2 = 1; s2 < NSLOT; s2++) { const float t1 = __shfl(u[s2], R & 63); t = (R >> 6) == s2 ? t1 : t; }
        out[s][j] = (rw[s].ell && j < rw[s].dim) ? t : 0.f;
      }
  }
  // force / state / cost of this lane's row at residual jar (siblings of an elliptic block agree on the zone)
  __device__ __forceinline__ float row_update(const Row& rw, float x, float& force, int& state, const float (&uj)[CD], float& T, float& g) const {
    float cost = 0.f;
    force = 0.f; state = ST_SATISFIED; T = 0.f; g = 0.f;
    if (!rw.valid) return 0.f;
    if (rw.type == C_FRICTION_DOF) {
      if (x <= -rw.R * rw.fl) { state = ST_LINEARNEG; force = rw.fl; cost = rw.fl * (-0.5f * rw.R * rw.fl - x); }
      else if (x >= rw.R * rw.fl) { state = ST_LINEARPOS; force = -rw.fl; cost = rw.fl * (-0.5f * rw.R * rw.fl + x); }
      else { state = ST_QUADRATIC; force = -rw.D * x; cost = 0.5f * rw.D * x * x; }
    } else if (TENDONS && rw.type == C_EQUALITY) {
      state = ST_QUADRATIC; force = -rw.D * x; cost = 0.5f * rw.D * x * x;
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
    } else if (TENDONS && rw.type == C_EQUALITY) {
      c = 0.5f * rw.D * x * x; c1 = rw.D * x * v; c2 = rw.D * v * v;
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
  // out_k (lane k < 16) = sum_r J[r][k] * f_r  on the matrix cores; f_r must already be in sm.e_force[0..4*nch)
  // wide configurations: every dof tile of a row chunk from one batch of reads, two chunks per trip (rows nefc .. NEFC - 1 of J and e_force hold zeros)
  template <int NBT>
  __device__ __forceinline__ void jtf_wide(int nch) {
    if constexpr (NBT <= NT) {
      v4f acc[NBT];
#pragma unroll
      for (int t = 0; t < NBT; t++) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};
      const int q = lane >> 4, col = lane & 15;
      for (int c0 = 0; c0 < nch; c0 += 2) {
        float ja[2][NBT], fb[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int r = 4 * (c0 + u) + q;
          fb[u] = sm.e_force[r];
#pragma unroll
          for (int t = 0; t < NBT; t++) ja[u][t] = Jrd(r * JS + 16 * t + col);
        }
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
          for (int t = 0; t < NBT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ja[u][t], fb[u], acc[t], 0, 0, 0);
      }
      if (col == 0) {
#pragma unroll
        for (int t = 0; t < NBT; t++) { float* o = sm.red + 16 * t + 4 * q; o[0] = acc[t][0]; o[1] = acc[t][1]; o[2] = acc[t][2]; o[3] = acc[t][3]; }
      }
    }
  }
  // wide configurations: H = M + J^T W for the NBT (NBT + 1) / 2 lower tiles at once.  W is never stored: a chunk of four rows reads its
  // staged coefficients (prefetched one chunk ahead), its J rows (the B operands, and the A operands of rows outside the cone state) and -- only
  // if the chunk holds a cone-state row -- the other rows of those blocks, all in one batch, and then issues the tile products on independent
  // accumulators.  Same products in the same order as the stored-W form it replaces.
  template <int NBT>
  __device__ __forceinline__ void hess_wide(int nch) {
    if constexpr (NBT <= NT) {
      constexpr int NTL = NBT * (NBT + 1) / 2;
      const int q = lane >> 4, col = lane & 15;
      v4f acc[NTL];
      {
        int t = 0;
#pragma unroll
        for (int ti = 0; ti < NBT; ti++)
#pragma unroll
          for (int tj = 0; tj <= ti; tj++, t++)
#pragma unroll
            for (int v = 0; v < 4; v++) acc[t][v] = Mrd((16 * ti + 4 * q + v) * NVP + 16 * tj + col);
      }
      float cf[4];
      int bd;
      {
        const float* o = sm.u.W + 5 * q;
        cf[0] = o[0]; cf[1] = o[1]; cf[2] = o[2]; cf[3] = o[3]; bd = ((const int*)o)[4];
      }
      for (int c = 0; c < nch; c++) {
        const int r = 4 * c + q;
        float cfn[4];
        int bdn;
        {
          const float* o = sm.u.W + 5 * (r + 4 < NEFCAP ? r + 4 : r);   // next chunk's coefficients
          cfn[0] = o[0]; cfn[1] = o[1]; cfn[2] = o[2]; cfn[3] = o[3]; bdn = ((const int*)o)[4];
        }
        float bj[NBT], aj[NBT];
#pragma unroll
        for (int t = 0; t < NBT; t++) bj[t] = Jrd(r * JS + 16 * t + col);
        if (__ballot((bd >> 16) & 1)) {
          const int head = bd & 255, dm1 = ((bd >> 8) & 7) - 1;
          const int j0 = head * JS + col;
          const int o1 = (dm1 < 1 ? dm1 : 1) * JS, o2 = (dm1 < 2 ? dm1 : 2) * JS, o3 = (dm1 < 3 ? dm1 : 3) * JS;
          float x0[NBT], x1[NBT], x2[NBT], x3[NBT];
#pragma unroll
          for (int t = 0; t < NBT; t++) { x0[t] = Jrd(j0 + 16 * t); x1[t] = Jrd(j0 + o1 + 16 * t); x2[t] = Jrd(j0 + o2 + 16 * t); x3[t] = Jrd(j0 + o3 + 16 * t); }
#pragma unroll
          for (int t = 0; t < NBT; t++) aj[t] = fmaf(cf[3], x3[t], fmaf(cf[2], x2[t], fmaf(cf[1], x1[t], cf[0] * x0[t])));
        } else {
#pragma unroll
          for (int t = 0; t < NBT; t++) aj[t] = cf[0] * bj[t];
        }
        {
          int t = 0;
#pragma unroll
          for (int ti = 0; ti < NBT; ti++)
#pragma unroll
            for (int tj = 0; tj <= ti; tj++, t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aj[ti], bj[tj], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) cf[k] = cfn[k];
        bd = bdn;
      }
      {
        int t = 0;
#pragma unroll
        for (int ti = 0; ti < NBT; ti++)
#pragma unroll
          for (int tj = 0; tj <= ti; tj++, t++)
#pragma unroll
            for (int v = 0; v < 4; v++) sm.H[(16 * ti + 4 * q + v) * NVP + 16 * tj + col] = acc[t][v];
      }
    }
  }
  __device__ __forceinline__ float jt_times_force(int nch) {
#pragma unroll
    for (int t = 0; t < (FAST ? NT : 0); t++) {
      v4f acc = {0.f, 0.f, 0.f, 0.f};
      if constexpr (FAST) {   // four row chunks per trip (rows >= nefc of J and e_force are zero up to row 63)
        for (int c0 = 0; c0 < nch; c0 += 4) {
          float ja[4], fb[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { const int r = 4 * (c0 + u) + (lane >> 4); ja[u] = Jrd(r * JS + (lane & 15)); fb[u] = sm.e_force[r]; }
#pragma unroll
          for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ja[u], fb[u], acc, 0, 0, 0);
        }
      }
      if constexpr (FAST) if ((lane & 15) == 0) { float* o = sm.red + 16 * t + 4 * (lane >> 4); o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = acc[3]; }
    }
    if constexpr (!FAST) {
      switch ((m.nv + 15) >> 4) {   // dof tiles of this model: a compile-time count keeps the reads of a trip in one batch (no branch per tile)
        case 1: jtf_wide<1>(nch); break;
        case 2: jtf_wide<2>(nch); break;
        case 3: jtf_wide<3>(nch); break;
        default: jtf_wide<4>(nch); break;
      }
    }
    SYNC();
    float r = FAST ? sm.red[lane & 15] : (lane < NV16 ? sm.red[lane] : 0.f);
    SYNC();
    return r;
  }

  __device__ __forceinline__ void solve_newton() {
    const int nv = m.nv, n = sm.nefc;
    const int nch = (n + 3) >> 2, nvt = (nv + 15) & ~15;
    const float scale = 1.0f / (m.meaninertia * (nv > 1 ? nv : 1));
    const float tolerance = m.tolerance;
    // rows in the lanes' further slots (128 / 256-row configurations): slot s holds rows iff n > 64 s; typical states have none there, and then none of that work is done
#define SLOT_ON(s) (64 * (s) < n || (s) == 0)
    // ---- per-lane row data (NSLOT rows per lane)
    Row rw[NSLOT];
#pragma unroll
    for (int s = 0; s < NSLOT; s++) {
      Row& w_ = rw[s];
      const int row = lane + 64 * s;
      w_.row = row;
      w_.valid = row < n;
      const int r = w_.valid ? row : 0;
      if constexpr (FAST) {
#pragma unroll
        for (int k = 0; k < NV16; k++) w_.J[k] = Jrd(row * JS + k);  // rows >= n were written as zeros
      }
      const int desc = w_.valid ? sm.e_desc[r] : 0;
      w_.type = w_.valid ? (desc & 15) : -1;
      const bool tfric = TENDONS && w_.type == C_FRICTION_TENDON;   // a tendon friction row behaves as a dof friction row from here on
      if (tfric) w_.type = C_FRICTION_DOF;
      w_.R = sm.e_R[r]; w_.D = 1.0f / w_.R; w_.aref = w_.valid ? sm.e_aref[r] : 0.f; w_.fl = tfric ? FP(FO_tendon_fl, (desc >> 4) & 255) : (w_.type == C_FRICTION_DOF ? cmf(MK_fricFl)->fricFl[(desc >> 4) & 255] : 0.f);
      w_.ell = w_.type == C_CONTACT_ELLIPTIC;
      const int c = w_.ell ? (desc >> 4) & 255 : 0;
      w_.kk = w_.ell ? (desc >> 12) & 15 : 0; w_.head = row - w_.kk; w_.dim = w_.ell ? sm.cdim[c] : 1;
      w_.mu = sm.cmu[c];
#pragma unroll
      for (int j = 0; j < CD - 1; j++) w_.fj[j] = cfri_rd(5 * c + j);
      w_.fr_own = w_.kk == 0 ? w_.mu : cfri_rd(5 * c + w_.kk - 1);
      w_.Dm = (1.0f / sm.e_R[w_.ell ? w_.head : r]) / fmaxf(w_.mu * w_.mu * (1 + w_.mu * w_.mu), 1e-15f);
    }
    // M: row i in lane i (matrix-vector products) and in the MFMA accumulator layout (Hessian seed)
    // per-dof vectors (a, gradient, search direction): one-tile configuration = replicated in all four 16-lane rows and reduced over the
    // first row only; wide = component k in lane k
    const int rr = FAST ? (lane & 15) : lane;
    const bool dofl = lane < NV16;
    float Mr[FAST ? NV16 : 1];
    v4f Macc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (FAST) {
#pragma unroll
      for (int k = 0; k < NV16; k++) Mr[k] = (rr < nv && k < nv) ? sm.M[rr * NVP + k] : 0.f;
#pragma unroll
      for (int v = 0; v < 4; v++) { int i = 4 * (lane >> 4) + v, j = lane & 15; Macc[v] = (i < nv && j < nv) ? sm.M[i * NVP + j] : (i == j ? 1.f : 0.f); }
    }
    SUBMARK(RP_X0);
    const float a_sm = rr < nv ? sm.qacc_smooth[rr] : 0.f, a_ws = rr < nv ? sm.qacc_ws[rr] : 0.f, f_sm = rr < nv ? sm.qfrc_smooth[rr] : 0.f;
    float force[NSLOT], jar[NSLOT], uj[NSLOT][CD], T[NSLOT], g[NSLOT];
    int state[NSLOT];
#pragma unroll
    for (int s = 0; s < NSLOT; s++) { force[s] = 0.f; jar[s] = 0.f; T[s] = 0.f; g[s] = 0.f; state[s] = ST_SATISFIED; }
    // residuals of all rows of this lane at acceleration x, then cost / force / state of each (the cone blocks gather their siblings first)
    auto evaluate = [&](float x) -> float {
#pragma unroll
      for (int s = 0; s < NSLOT; s++) if SLOT_ON(s) jar[s] = row_dot(rw[s], x) - rw[s].aref;
      gather(rw, jar, uj);
      float c = 0.f;
#pragma unroll
      for (int s = 0; s < NSLOT; s++) if SLOT_ON(s) c += row_update(rw[s], jar[s], force[s], state[s], uj[s], T[s], g[s]);
      return c;
    };
    // Hessian weights W (row r): D J_r (quadratic), 0 (linear / satisfied), cone block Hc J_block -- from the rows' states and residuals as the last evaluation
    // left them (state, jar, uj, T, g); written to LDS for the matrix-core product.  Used by the iteration and by the fp64-evaluated passes behind it.
    auto weights = [&]() {
#pragma unroll
      for (int s = 0; s < NSLOT; s++) {
        const Row& w_ = rw[s];
        const float dq = state[s] == ST_QUADRATIC ? w_.D : 0.f;
        float hk[CD];
#pragma unroll
        for (int k2 = 0; k2 < CD; k2++) hk[k2] = 0.f;
        if (state[s] == ST_CONE) {
          const float mu = w_.mu, iT = 1.0f / T[s];
          const float uo = jar[s] * w_.fr_own;  // own friction-scaled residual U_kk
          const float grj = w_.kk == 0 ? mu : -mu * uo * w_.fr_own * iT;
#pragma unroll
          for (int k2 = 0; k2 < CD; k2++) {
            if (k2 < w_.dim) {
              const float frk = k2 == 0 ? mu : w_.fj[k2 > 0 ? k2 - 1 : 0];
              const float grk = k2 == 0 ? mu : -mu * uj[s][k2] * frk * iT;
              float h = grj * grk;
              if (w_.kk > 0 && k2 > 0) h += -g[s] * mu * w_.fr_own * frk * ((w_.kk == k2 ? iT : 0.f) - uo * uj[s][k2] * iT * iT * iT);
              hk[k2] = h * w_.Dm;
            }
          }
        }
        if constexpr (FAST) {
          float w[NV16];
#pragma unroll
          for (int k = 0; k < NV16; k++) w[k] = dq * w_.J[k];
          if (state[s] == ST_CONE) {
#pragma unroll
            for (int k2 = 0; k2 < CD; k2++) {
              if (k2 < w_.dim) {
                const int js = (w_.head + k2) * JS;
#pragma unroll
                for (int k = 0; k < NV16; k++) w[k] = fmaf(hk[k2], Jrd(js + k), w[k]);
              }
            }
          }
#pragma unroll
          for (int k = 0; k < NV16; k++) sm.u.W[w_.row * JS + k] = w[k];
        } else {
          // weighted row r = sum_k2 coef[k2] J[head + min(k2, dim - 1)]: (D, 0, 0, 0) on the row itself outside the cone state, the row's line of
          // the block Hessian on the block's rows inside it.  hess_wide() forms it while it feeds the matrix cores.
          if SLOT_ON(s) {
            float* o = sm.u.W + 5 * w_.row;
            const bool cone = state[s] == ST_CONE;
            o[0] = cone ? hk[0] : dq; o[1] = hk[1]; o[2] = hk[2]; o[3] = hk[3];
            ((int*)o)[4] = cone ? (w_.head | (w_.dim << 8) | (1 << 16)) : (w_.row | (1 << 8));
          }
        }
      }
    };
    // wide configurations: H = M + J^T W as MFMA tiles (lower triangle of tiles: the factorisation reads nothing above the diagonal blocks), factorised in place
    // on the LDS matrix (H aliases the dead factor of M); tiles beyond the model's dofs (37 of 64 in the PickPlace model) are neither formed nor read
    bool factored = false;    // sm.H holds a Cholesky factor of a Hessian of this solve (wide configurations)
    int fst[NSLOT];           // ... built on these row states
#pragma unroll
    for (int s = 0; s < NSLOT; s++) fst[s] = ST_SATISFIED;
    auto factorize = [&]() {
      if constexpr (!FAST) {
        switch ((nv + 15) >> 4) {
          case 1: hess_wide<1>(nch); break;
          case 2: hess_wide<2>(nch); break;
          case 3: hess_wide<3>(nch); break;
          default: hess_wide<4>(nch); break;
        }
        SYNC();
        SUBMARK_H(RP_X7);
        bchol_inplace<NVP>(sm.H, sm.invdiag, nv, lane);
        factored = true;
#pragma unroll
        for (int s = 0; s < NSLOT; s++) fst[s] = state[s];
        SUBMARK_H(RP_X8);
      }
    };
    // ---- warm start: previous acceleration unless the unconstrained one is cheaper
    float cost_sm = wave_sum(evaluate(a_sm));
    float cost_ws = wave_sum(evaluate(a_ws));
    {
      const float dws = a_ws - a_sm, sv = mass_dot(Mr, dws);
      cost_ws += wave_sum(dofl ? 0.5f * sv * dws : 0.f);
    }
    SUBMARK(RP_X1);
    float a = cost_ws < cost_sm ? a_ws : a_sm;
    int iter = 0;
    bool have_eval = false;   // the rows are already evaluated at `a` (cost_pre)
    int pol_exit = 0, pol_pass = 0, pol_sgb = 0;   // how the polish ended (RSIM_POLISH of the debug entries carries it: see the end of this function)
    float cost_pre = 0.f;
    for (;;) {
      float cost = have_eval ? cost_pre : wave_sum(evaluate(a));
      have_eval = false;
      int state0[NSLOT];
#pragma unroll
      for (int s = 0; s < NSLOT; s++) state0[s] = state[s];
      const float ma = mass_dot(Mr, a);
      const float gauss = wave_sum(dofl ? 0.5f * (ma - f_sm) * (a - a_sm) : 0.f);
      cost += gauss;
#pragma unroll
      for (int s = 0; s < NSLOT; s++) sm.e_force[lane + 64 * s] = force[s];
      SYNC();
      const float jf = jt_times_force(nch);
      float gk = rr < nv ? ma - f_sm - jf : 0.f;
      const float gn = wave_sum(dofl ? gk * gk : 0.f);
      SUBMARK(RP_X2);
      if (iter >= m.iterations || scale * sqrtf(gn) < tolerance) break;
      // fp32: the gradient is a difference of O(100) N m terms, good to ~1e-7 of THEIR size -- MuJoCo's 1e-8 (scaled) is below that floor for every
      // loaded arm, so the fp64 test above never fires here and the loop used to run until a step happened not to lower the (equally noisy) cost:
      // 150-230 iterations per control step where the fp64 oracle takes 25-80 on the same states (tools/newton_iters.py), half the run time of the
      // slowest envs of a launch.  A gradient whose every component is within the rounding noise of its own three terms is converged.
      if (m.newton_ng > 0.f && !__ballot(dofl && rr < nv && fabsf(gk) > m.newton_ng * (fabsf(ma) + fabsf(f_sm) + fabsf(jf)))) break;
      weights();
      SYNC();
      float sk;
      SUBMARK(RP_X3);
      if constexpr (!FAST) {
        factorize();
        sk = bchol_solve<NVP>(sm.H, sm.invdiag, lane < nv ? -gk : 0.f, nv, lane);
        if (lane >= nv) sk = 0.f;
        SUBMARK_H(RP_X9);
      } else {
        v4f acc = Macc;
        // four row chunks per trip: eight LDS reads in flight, then four MFMAs (rows >= nefc of W and J are zero up to row 63)
        for (int c0 = 0; c0 < nch; c0 += 4) {
          float wa[4], jb[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { const int o = (4 * (c0 + u) + (lane >> 4)) * JS + (lane & 15); wa[u] = sm.u.W[o]; jb[u] = Jrd(o); }
#pragma unroll
          for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[u], jb[u], acc, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 4; v++) sm.H[(4 * (lane >> 4) + v) * NVP + (lane & 15)] = acc[v];
        SYNC();
        float hr[NV16], hinv[NV16], ht[NV16];
        const int rr2 = lane & 15;
#pragma unroll
        for (int k = 0; k < NV16; k++) hr[k] = sm.H[rr2 * NVP + k];
        const int ro = opaque_lane(rr2);
        const float hown = rchol_factor_own<NV16>(hr, hinv, ro);
        rchol_mask_lower<NV16>(hr, ro);
        SYNC();
        if (lane < NV16) {
#pragma unroll
          for (int k = 0; k < NV16; k++) sm.H[lane * NVP + k] = hr[k];
        }
        SYNC();
#pragma unroll
        for (int k = 0; k < NV16; k++) ht[k] = sm.H[k * NVP + rr2];
        sk = rchol_solve_m<NV16>(hr, ht, hinv, hown, rr2 < nv ? -gk : 0.f);
        if (rr2 >= nv) sk = 0.f;
      }
      SUBMARK(RP_X4);
      // ---- line search along sk
      float jv[NSLOT];
#pragma unroll
      for (int s = 0; s < NSLOT; s++) jv[s] = SLOT_ON(s) ? row_dot(rw[s], sk) : 0.f;
      const float mvv = mass_dot(Mr, sk);
      const float q1 = wave_sum(dofl ? sk * (ma - f_sm) : 0.f), q2 = wave_sum(dofl ? 0.5f * sk * mvv : 0.f), sn = sqrtf(wave_sum(dofl ? sk * sk : 0.f));
      if (sn < 1e-15f) break;
      float g0[NSLOT][CD], gvv[NSLOT][CD];
      gather(rw, jar, g0);
      gather(rw, jv, gvv);
      const float gtol = tolerance * 0.01f * sn / scale;
      float p0, d0, h0, p, dp, hp, lo = 0.f, hi = -1.f, alpha;
      // cost and its first two derivatives along the search direction, summed over this lane's rows
      auto line = [&](float al, float& c, float& c1, float& c2) {
        c = c1 = c2 = 0.f;
#pragma unroll
        for (int s = 0; s < NSLOT; s++) if SLOT_ON(s) { float t, t1, t2; row_ls(rw[s], jar[s], jv[s], g0[s], gvv[s], al, t, t1, t2); c += t; c1 += t1; c2 += t2; }
      };
      {
        float c, c1, c2;
        line(0.f, c, c1, c2);
        p0 = gauss + wave_sum(c); d0 = q1 + wave_sum(c1); h0 = 2 * q2 + wave_sum(c2);
      }
      SUBMARK(RP_X5);
      if (d0 >= 0 || h0 <= 0) break;
      alpha = -d0 / h0;
      // fp32 line search: stop when the directional derivative has dropped below MuJoCo's gtol, by 1e6 relative to its
      // start value (single-precision noise floor), or when the safeguarded Newton update no longer moves alpha
      const float dtol = fmaxf(gtol, 1e-6f * fabsf(d0));
      // ... or moves it by less than what the step rule below calls settled: a correction of alpha that changes no component of the acceleration by
      // more than newton_ns |a_i| + newton_na is not worth an evaluation (the wide configurations: DModel.newton_wide, as the step rule).  `p` always belongs to
      // the alpha the loop ends on: the exits keep the point they evaluated, so nothing is evaluated twice.
      const float atol = ((FAST || m.newton_wide) && m.newton_ls > 0.f && m.newton_ns > 0.f)
                             ? m.newton_ls * wave_min_f((dofl && rr < nv && sk != 0.f) ? (m.newton_ns * fabsf(a) + m.newton_na) / fabsf(sk) : 3.0e38f) : 0.f;
      bool at_alpha = false;
      for (int ls = 0; ls < m.ls_iterations; ls++) {
        pf.count(RP_N_LS, 1);
        float c, c1, c2;
        line(alpha, c, c1, c2);
        p = gauss + alpha * q1 + alpha * alpha * q2 + wave_sum(c);
        dp = q1 + 2 * alpha * q2 + wave_sum(c1);
        hp = 2 * q2 + wave_sum(c2);
        at_alpha = true;
        if (fabsf(dp) < dtol) break;
        if (dp < 0) lo = alpha; else hi = alpha;
        float next = hp > 0 ? alpha - dp / hp : -1.f;
        if (hi < 0) { if (next <= lo) next = 2 * alpha + 1e-12f; }
        else if (next <= lo || next >= hi) next = 0.5f * (lo + hi);
        if (fabsf(next - alpha) <= fmaxf(1e-6f * fabsf(alpha), atol)) break;
        alpha = next;
        at_alpha = false;
      }
      if (!at_alpha) {   // the evaluation budget ran out on a fresh alpha
        float c, c1, c2;
        line(alpha, c, c1, c2);
        p = gauss + alpha * q1 + alpha * alpha * q2 + wave_sum(c);
      }
      SUBMARK(RP_X6);
      if (!(p < p0)) break;
      const float a_new = fmaf(alpha, sk, a);
      // fp32: a step that moves no component of the acceleration by more than a few units in its last place (plus an absolute floor) cannot be
      // improved on by another factorisation
      // (wide configurations: DModel.newton_wide.  An earlier form of the rule -- absolute floor 1e-6, before the line-search exit existed -- cost the
      // Robotiq's 5e-5 kg m^2 finger links a factor of ten in how closely they track the oracle and was kept off the wide models; with the
      // thresholds as they are the finger, arm and object tracking of the PickPlace / Stack / UR5e / Jaco fixtures is unchanged
      // (tools/newton_wide_sweep.py, profiles/r03_r_newton_wide_sweep.txt) and Stack gains 9 %: on by default.)
      const bool settled = (FAST || m.newton_wide == 1) && m.newton_ns > 0.f && !__ballot(dofl && rr < nv && fabsf(alpha * sk) > m.newton_ns * fabsf(a_new) + m.newton_na);
      a = a_new;
      iter++;
      if (scale * (p0 - p) < tolerance || settled) {
        evaluate(a);
        break;
      }
      // One Newton step is exact while the active set stands still.  Between two state changes the objective is a quadratic (quadratic rows, linear
      // friction-loss rows, satisfied rows; only rows ON the cone curve it), H is its exact Hessian and the line search has just found its minimiser along
      // the Newton direction: if every row sits in the state it had before the step, the new point minimises the piece it lies in, and by convexity the
      // whole objective.  Another Hessian, factorisation and line search could only confirm it (they used to: 2.5 iterations per substep for a cube at
      // rest on the table, whose rows never change state while the arm above it accelerates differently in every substep).
      if (m.newton_exact) {
        cost_pre = wave_sum(evaluate(a));
        have_eval = true;
        bool moved = false;
#pragma unroll
        for (int s = 0; s < NSLOT; s++) if SLOT_ON(s) moved |= rw[s].valid && (state[s] != state0[s] || state0[s] == ST_CONE);
        if (!__ballot(moved)) break;
      }
    }
    if constexpr (!FAST && POLISH) {
      // ---- polish of the solution with everything that DECIDES evaluated in fp64 (wide configurations).  What fp32 cannot see on these models: cost differences below
      // 1e-7 of a cost of 1e3.  The iteration above therefore ends where a step gains nothing it can measure -- characteristically ON a kink of the piecewise objective:
      // a cone block (the condim-4 contacts of the PickPlace objects) or a row about to change its state.  Round 5 traced the per-env tail of the full-size test to
      // exactly that (tests/test_full_size_parity.py, profiles/r05_g_*): in every one of the worst envs one contact sits in the cone (or satisfied) state at the kernel's
      // answer and in the quadratic state at the minimiser, the fp64 gradient left is 1e-2 .. 1 (scaled), forces are ordinary (20 - 1500 N) -- not conditioning (rounding
      // the solver's inputs to fp32 moves the fp64 solution by 1e-7), not the fp32 factor (a plain Newton step does not descend there even with an exact one), but the
      // missing line search ACROSS the kink: Newton with an exact fp64 line search walks through in 3 - 10 passes (1e-5, 1e-5, 1e-5, 3e-6, 2e-7, 7e-10), with an fp32
      // factor just as well (tools/emulate_polish.py).  So:
      //   * residuals J a - aref, row states, forces, objective and J^T f in fp64 from the float data (every product exact), the acceleration a pair of floats;
      //   * direction -H^-1 g with H = M + J^T W of the fp64-decided states, formed and factorised in fp32 on the matrix cores (re-done when the states changed);
      //   * exact line search along it in fp64 (MuJoCo's: safeguarded 1-D Newton on the derivative of the piecewise-quadratic + cone objective);
      //   * a point is kept unless it is higher than the rounding of the objective; passes end on MuJoCo's criteria (scaled gradient / improvement below `tolerance`,
      //     the latter not while rows are still flipping), or after `newton_refine` passes;
      //   * the first pass is a plain Newton step (most solves end there); the line search comes on when that step raises the objective or leaves more than a
      //     quarter of the gradient.
      const int R = m.newton_refine;
      if (R > 0 && n > 0) {
        // (not only behind a factorisation of the iteration above: its fp32 exits -- a gradient within the rounding noise of its own terms, which for a light body
        // beside a 2 kN squeeze is hundreds of rad/s^2 -- can end it before any Hessian was formed; exactly those solves need the fp64 look most.  Round 5: the
        // per-env tail of the PickPlace full-size test was made of such envs, whatever the number of passes.)
        bool need_factor = !factored;
        float a_lo = 0.f, a_keep = a, alo_keep = 0.f, force_keep[NSLOT], dk = 0.f;
        int state_keep[NSLOT];
        double err_keep = 1.0e300;
#pragma unroll
        for (int s = 0; s < NSLOT; s++) { force_keep[s] = force[s]; state_keep[s] = state[s]; }
        // (hi, lo) += d, exactly up to the pair's precision
        auto dfadd = [](float& hi, float& lo, float d) {
          const float sm_ = __fadd_rn(hi, d), bb = __fsub_rn(sm_, hi), se = __fadd_rn(__fsub_rn(hi, __fsub_rn(sm_, bb)), __fsub_rn(d, bb));   // hi + d = sm_ + se exactly
          const float l2 = __fadd_rn(lo, se);
          hi = __fadd_rn(sm_, l2);
          lo = __fsub_rn(l2, __fsub_rn(hi, sm_));
        };
        int flat = 0;
        bool use_ls = m.newton_polish_gate < 0.f;   // (RSIM_POLISH_GATE < 0: line search from the first pass on, for A/B)
        double gn_prev = 1.0e300;
        for (int it = 0;;) {
          double jr[NSLOT], u64[NSLOT], c64 = 0.0;   // c64: this lane's share of the objective at a + a_lo, in fp64 (same pieces as row_update)
#pragma unroll
          for (int s = 0; s < NSLOT; s++) { jr[s] = SLOT_ON(s) ? row_res64(rw[s], a, a_lo) : 0.0; u64[s] = jr[s] * (double)rw[s].fr_own; }
#pragma unroll
          for (int s = 0; s < NSLOT; s++) {
            const Row& w_ = rw[s];
            double uj64[CD];
#pragma unroll
            for (int j = 0; j < CD; j++) {        // the block's friction-scaled residuals (gather() in fp64)
              const int Rr = w_.head + j;
              double t = __shfl(u64[0], Rr & 63);
#pragma unroll
              for (int s2 = 1; s2 < NSLOT; s2++) { const double t1 = __shfl(u64[s2], Rr & 63); t = (Rr >> 6) == s2 ? t1 : t; }
              uj64[j] = (w_.ell && j < w_.dim) ? t : 0.0;
            }
            // force / state / cost of the row at its fp64 residual: row_update() in double
            double f = 0.0, Tn = 0.0, gg = 0.0;
            int st = ST_SATISFIED;
            if (w_.valid && SLOT_ON(s)) {
              const double x = jr[s], D = (double)w_.D;
              if (w_.type == C_FRICTION_DOF) {
                const double fl = (double)w_.fl, lim = (double)w_.R * fl;
                if (x <= -lim) { st = ST_LINEARNEG; f = fl; c64 += fl * (-0.5 * lim - x); }
                else if (x >= lim) { st = ST_LINEARPOS; f = -fl; c64 += fl * (-0.5 * lim + x); }
                else { st = ST_QUADRATIC; f = -D * x; c64 += 0.5 * D * x * x; }
              } else if (TENDONS && w_.type == C_EQUALITY) {
                st = ST_QUADRATIC; f = -D * x; c64 += 0.5 * D * x * x;
              } else if (!w_.ell) {
                if (x < 0) { st = ST_QUADRATIC; f = -D * x; c64 += 0.5 * D * x * x; }
              } else {
                const double N = uj64[0], mu = (double)w_.mu;
                double T2 = 0.0;
#pragma unroll
                for (int j = 1; j < CD; j++) T2 = fma(uj64[j], uj64[j], T2);
                Tn = sqrt(T2);
                if (N >= mu * Tn || (Tn <= 0 && N >= 0)) {
                } else if (mu * N + Tn <= 0 || (Tn <= 0 && N < 0)) {
                  st = ST_QUADRATIC; f = -D * x; c64 += 0.5 * D * x * x;
                } else {
                  gg = N - mu * Tn;
                  const double f0 = -(double)w_.Dm * gg * mu;
                  st = ST_CONE;
                  if (w_.kk == 0) { f = f0; c64 += 0.5 * (double)w_.Dm * gg * gg; }
                  else f = -f0 / Tn * u64[s] * (double)w_.fr_own;
                }
              }
            }
            // what weights() reads, should this point's states ask for another factor
            state[s] = st; jar[s] = (float)jr[s]; T[s] = (float)Tn; g[s] = (float)gg;
#pragma unroll
            for (int j = 0; j < CD; j++) uj[s][j] = (float)uj64[j];
            force[s] = (float)f;
            if SLOT_ON(s) {
              sm.u.W[w_.row] = force[s];                                  // the force as hi + lo floats for the dof lanes
              sm.u.W[NEFCAP + w_.row] = (float)(f - (double)force[s]);
            }
          }
          SYNC();
          double gk = 0.0, ma64 = 0.0;
          if (lane < nv) {
            double ma = 0.0, jf = 0.0;
            for (int j = 0; j < nv; j++) ma = fma((double)Mrd(lane * NVP + j), (double)bcast(a, j) + (double)bcast(a_lo, j), ma);
            for (int r = 0; r < n; r++) jf = fma((double)Jrd(r * JS + lane), (double)sm.u.W[r] + (double)sm.u.W[NEFCAP + r], jf);
            gk = ma - (double)f_sm - jf; ma64 = ma;
            // the objective's Gauss term half (M a - f_smooth) . (a - a_smooth)
            c64 += 0.5 * (ma - (double)f_sm) * ((double)a + (double)a_lo - (double)a_sm);
          }
          const double err = wave_sum_f64(c64);
          // The line search returns the minimiser along a descent direction of this convex objective, so the new point is not higher -- but where the kept point sits
          // ON a kink (a row or a cone block about to change its state: where the fp32 iteration above characteristically ends) the first steps gain next to nothing
          // and only flip the block; the gain comes in the passes after it (tools/emulate_polish.py on the dumped tail envs: 1e-5, 1e-5, 1e-5, 3e-6, 2e-7, 7e-10).
          // A point is therefore kept unless it is HIGHER by more than the objective's own rounding.
          if (!(err <= err_keep + 1e-12 * fabs(err_keep))) {
            a = a_keep; a_lo = alo_keep;
            if (!use_ls && NSLOT <= RSIM_LS_MAXSLOT) {   // the plain Newton step raised the objective: from the kept point again, with the line search from here on
              use_ls = true; err_keep = 1.0e300;
              SYNC();
              continue;
            }
#pragma unroll
            for (int s = 0; s < NSLOT; s++) { force[s] = force_keep[s]; state[s] = state_keep[s]; }
            pol_exit = 6; pol_pass = it;
            break;
          }
          const double gain = err_keep - err;
          bool flipped = false;       // some row sits in another state than at the kept point
#pragma unroll
          for (int s = 0; s < NSLOT; s++) if SLOT_ON(s) flipped |= rw[s].valid && state[s] != state_keep[s];
          flipped = __ballot(flipped) != 0;
          a_keep = a; alo_keep = a_lo; err_keep = err;
#pragma unroll
          for (int s = 0; s < NSLOT; s++) { force_keep[s] = force[s]; state_keep[s] = state[s]; }
          const double gn = wave_sum_f64(lane < nv ? gk * gk : 0.0);
          const double sg = (double)scale * sqrt(gn), tol64 = (double)tolerance * (double)m.newton_polish_tol;
          { const int bk = sg > 0.0 ? (int)(-log10(sg)) : 15; pol_sgb = bk < 0 ? 0 : (bk > 15 ? 15 : bk); pol_pass = it; }
          if (sg < tol64) { pol_exit = 1; break; }
          // MuJoCo's improvement criterion -- but not while the passes are still flipping rows at a kink (no gain yet, by construction)
          // (three passes in a row: at a kink two passes without gain and without a flip do occur before the block lets go -- 1e-5, 1e-5, 1e-5, 3e-6, ... in the emulation)
          if (it > 0 && !flipped && (double)scale * gain < tol64) { if (++flat >= 3) { pol_exit = 2; break; } } else flat = 0;
          if (it >= R) { pol_exit = 3; break; }
          // The first step is the plain Newton step: inside a piece of the objective it lands on the minimiser, and most solves are done after it (5000 of 8192 in
          // the PickPlace workload).  One that does not shrink the gradient to a quarter -- the point sits on a kink -- switches the line search on for the rest.
          if (it > 0 && gn > 0.0625 * gn_prev) use_ls = true;
          gn_prev = gn;
          it++;
          bool other = false;
#pragma unroll
          for (int s = 0; s < NSLOT; s++) if SLOT_ON(s) other |= rw[s].valid && (state[s] != fst[s] || (it >= 2 && state[s] == ST_CONE));
          SYNC();
          // no factor yet, one of another active set, or -- from the second pass on -- a row on the cone, whose Hessian moves with the point (a stale factor makes
          // the passes a quasi-Newton iteration: the envs that used up a 16-pass budget in profiles/r05_k_*): H from these states, fp32 on the matrix cores
          if (__ballot(other) || need_factor) {
            need_factor = false;
            weights();
            SYNC();
            factorize();
          }
          dk = bchol_solve<NVP>(sm.H, sm.invdiag, lane < nv ? -(float)gk : 0.f, nv, lane);
          if (lane >= nv) dk = 0.f;
          // ---- exact line search along dk, in fp64 (MuJoCo's primal line search: the objective along a line is piecewise quadratic plus the cone terms; a
          // safeguarded 1-D Newton iteration on its derivative).  The rows' residuals move as jr + alpha jv.
          double alpha = 1.0;
          if (use_ls && NSLOT <= RSIM_LS_MAXSLOT) {
            // (the gathered vectors of every cone block are kept: forming T^2 along the line from expanded sums cancels where the line passes the cone's axis -- the very
            // kinks this search has to cross; measured: the per-env tail came back with the expanded form, profiles/r05_i_*)
            double jv[NSLOT], v64[NSLOT], g0[NSLOT][CD], gv[NSLOT][CD];
#pragma unroll
            for (int s = 0; s < NSLOT; s++) { jv[s] = SLOT_ON(s) ? row_res64(rw[s], dk, 0.f) + (double)rw[s].aref : 0.0; v64[s] = jv[s] * (double)rw[s].fr_own; }   // J_r . dk with exact products
#pragma unroll
            for (int s = 0; s < NSLOT; s++)
#pragma unroll
              for (int j = 0; j < CD; j++) {
                const int Rr = rw[s].head + j;
                double tu = __shfl(u64[0], Rr & 63), tv = __shfl(v64[0], Rr & 63);
#pragma unroll
                for (int s2 = 1; s2 < NSLOT; s2++) { const double t1 = __shfl(u64[s2], Rr & 63), t2 = __shfl(v64[s2], Rr & 63); if ((Rr >> 6) == s2) { tu = t1; tv = t2; } }
                const bool on = rw[s].ell && j < rw[s].dim;
                g0[s][j] = on ? tu : 0.0; gv[s][j] = on ? tv : 0.0;
              }
            const double mvv = (double)mass_dot(Mr, dk);
            const double q1 = wave_sum_f64(lane < nv ? (double)dk * (ma64 - (double)f_sm) : 0.0), q2 = wave_sum_f64(lane < nv ? 0.5 * (double)dk * mvv : 0.0);
            auto line64 = [&](double al, double& d1, double& d2) {     // first and second derivative of the rows' part along the line, this lane's share
              d1 = d2 = 0.0;
#pragma unroll
              for (int s = 0; s < NSLOT; s++) {
                const Row& w_ = rw[s];
                if (!(w_.valid && SLOT_ON(s))) continue;
                const double x = fma(al, jv[s], jr[s]), v = jv[s], D = (double)w_.D;
                if (w_.type == C_FRICTION_DOF) {
                  const double fl = (double)w_.fl, lim = (double)w_.R * fl;
                  if (x <= -lim) d1 -= fl * v; else if (x >= lim) d1 += fl * v; else { d1 += D * x * v; d2 += D * v * v; }
                } else if (TENDONS && w_.type == C_EQUALITY) { d1 += D * x * v; d2 += D * v * v; }
                else if (!w_.ell) { if (x < 0) { d1 += D * x * v; d2 += D * v * v; } }
                else {
                  const double mu = (double)w_.mu, N = fma(al, gv[s][0], g0[s][0]);
                  double T2 = 0.0, UV = 0.0, VV = 0.0;
#pragma unroll
                  for (int j = 1; j < CD; j++) { const double U = fma(al, gv[s][j], g0[s][j]); T2 = fma(U, U, T2); UV = fma(U, gv[s][j], UV); VV = fma(gv[s][j], gv[s][j], VV); }
                  const double Tn = sqrt(T2);
                  if (N >= mu * Tn || (Tn <= 0 && N >= 0)) {
                  } else if (mu * N + Tn <= 0 || (Tn <= 0 && N < 0)) { d1 += D * x * v; d2 += D * v * v; }
                  else if (w_.kk == 0) {
                    const double g_ = N - mu * Tn, iT = 1.0 / Tn, g1 = gv[s][0] - mu * UV * iT, g2 = -mu * (VV * iT - UV * UV * iT * iT * iT);
                    d1 += (double)w_.Dm * g_ * g1; d2 += (double)w_.Dm * (g1 * g1 + g_ * g2);
                  }
                }
              }
            };
            double c1, c2;
            line64(0.0, c1, c2);
            const double d0 = q1 + wave_sum_f64(c1), h0 = 2.0 * q2 + wave_sum_f64(c2);
            if (!(d0 < 0.0) || !(h0 > 0.0)) {
              // not a descent direction of the true objective (an fp32 factor that is noise): the passes end here
              pol_exit = 5; break;
            }
            alpha = -d0 / h0;
            double lo = 0.0, hi = -1.0;
            bool conv = false;
            for (int ls = 0; ls < 24; ls++) {
              line64(alpha, c1, c2);
              const double dp = q1 + 2.0 * alpha * q2 + wave_sum_f64(c1), hp = 2.0 * q2 + wave_sum_f64(c2);
              if (fabs(dp) < 1e-9 * fabs(d0)) { conv = true; break; }
              if (dp < 0) lo = alpha; else hi = alpha;
              double next = hp > 0 ? alpha - dp / hp : -1.0;
              if (hi < 0) { if (next <= lo) next = 2.0 * alpha + 1e-12; }
              else if (next <= lo || next >= hi) next = 0.5 * (lo + hi);
              if (fabs(next - alpha) <= 1e-12 * fabs(alpha)) { conv = true; break; }
              alpha = next;
            }
            // out of evaluations on a fresh alpha: the largest step known to lie on the descending side (the objective is convex along the line: it is lower there)
            if (!conv && lo > 0.0) alpha = lo;
            // where the bracket has closed on a kink (the derivative jumps from negative to positive there) the step goes to its FAR side: the rows that change state
            // there then sit in their new state at the next evaluation and the next Hessian belongs to the piece the minimiser lies in; on the near side the same
            // direction would come back (the objective at the two ends differs by the bracket's width times the derivative: nothing)
            if (conv && hi > 0.0 && hi - lo <= 1e-6 * hi) alpha = hi;
          }
          dfadd(a, a_lo, (float)(alpha * (double)dk));
          SYNC();
        }
      }
    }
#pragma unroll
    for (int s = 0; s < NSLOT; s++) sm.e_force[lane + 64 * s] = force[s];
    SYNC();
    const float fc = jt_times_force(nch);
    if (lane < nv) { sm.qfrc_constraint[lane] = fc; sm.qacc[lane] = a; }
    // RSIM_POLISH of the debug entries (same encoding as include/rsim.h): 10000 x polish passes + 100000 x floor(-log10(scaled fp64 gradient at the kept point))
    // + 10^7 x how the polish ended (1 gradient below tolerance, 2 improvement below tolerance, 3 pass budget, 5 the direction was no descent direction, 6 a step
    // raised the objective; 0 not run)
    if (lane == 0) { sm.niter = iter; sm.polish = 10000 * pol_pass + 100000 * pol_sgb + 10000000 * pol_exit; }
    pf.count(RP_N_NEWTON, iter);
    SYNC();
#undef SLOT_ON
  }


  // ---------------------------------------------------------------- observation / reward epilogue (after the last substep)
  // lane i fills observation floats i, i + 64; poses are those of the last substep's kinematics (the reference samples its
  // Observables after the last sim.step2() of a control step, when site/body frames still belong to that substep's step1)
  __device__ __forceinline__ Q4 mat2quat_xyzw(const float* R) const {   // returned as {w=x, x=y, y=z, z=w} slots: see caller
    const float m00 = R[0], m01 = R[1], m02 = R[2], m10 = R[3], m11 = R[4], m12 = R[5], m20 = R[6], m21 = R[7], m22 = R[8];
    float qw, qx, qy, qz;
    const float tr = m00 + m11 + m22;
    if (tr > 0.f) { const float sq = sqrtf(tr + 1.f) * 2.f; qw = 0.25f * sq; qx = (m21 - m12) / sq; qy = (m02 - m20) / sq; qz = (m10 - m01) / sq; }
    else if (m00 > m11 && m00 > m22) { const float sq = sqrtf(1.f + m00 - m11 - m22) * 2.f; qw = (m21 - m12) / sq; qx = 0.25f * sq; qy = (m01 + m10) / sq; qz = (m02 + m20) / sq; }
    else if (m11 > m22) { const float sq = sqrtf(1.f + m11 - m00 - m22) * 2.f; qw = (m02 - m20) / sq; qx = (m01 + m10) / sq; qy = 0.25f * sq; qz = (m12 + m21) / sq; }
    else { const float sq = sqrtf(1.f + m22 - m00 - m11) * 2.f; qw = (m10 - m01) / sq; qx = (m02 + m20) / sq; qy = (m12 + m21) / sq; qz = 0.25f * sq; }
    if (qw < 0.f) { qw = -qw; qx = -qx; qy = -qy; qz = -qz; }   // transform_utils.mat2quat canonicalises w >= 0
    Q4 q = {qw, qx, qy, qz};
    return q;
  }
  // TwoArmPegInHole._compute_orientation (two_arm_peg_in_hole.py:523-560): parallel distance t, perpendicular distance d of the hole centre
  // from the peg axis, |cos| between the peg axis and the hole normal
  __device__ __forceinline__ void peg_orientation(int peg, int hole, float& tt, float& dd, float& cs) const {
    const M3 Rp = q2m(ldq(sm.xquat + 4 * peg)), Rh = q2m(ldq(sm.xquat + 4 * hole));
    const V3 pp = ld3(sm.xpos + 3 * peg), hp = ld3(sm.xpos + 3 * hole);
    V3 v = col(Rp, 2);
    v = v * (1.0f / norm(v));
    const V3 center = hp + col(Rh, 0) * 0.1f;
    const float vn = norm(v);
    tt = dot(center - pp, v) / (vn * vn);
    dd = norm(cross(v, pp - center)) / vn;
    const V3 hn = col(Rh, 2);
    cs = fabsf(dot(hn, v) / norm(hn) / vn);
  }
  // `reset_obs`: the record MujocoEnv.reset returns (rsim_observe), where the relative-pose sensors of PickPlace find an empty observation cache
  __device__ __forceinline__ void obs_reward(float* obs, float* __restrict__ reward, int* __restrict__ success, bool reset_obs) {
    const DTask& t = m.task;
    gci prog = (gci)t.obs_prog;
    float peg_t = 0.f, peg_d = 0.f, peg_c = 0.f;
    if (t.task == 3) peg_orientation(t.object_body, t.object2_body, peg_t, peg_d, peg_c);
    float* prev = sm.u.b.poly;   // PickPlace: object poses as sampled at the PREVIOUS control step (still in the record), 7 floats per object
    if (t.task == 4) {
      if (lane < 7 * t.nobj) prev[lane] = obs[seli(t.pos_slot, lane / 7) + lane % 7];   // select chain: kernel-argument arrays are never indexed dynamically
      SYNC();
    }
    for (int i = lane; i < t.nobs; i += 64) {
      const int kind = prog[3 * i], a = prog[3 * i + 1], b2 = prog[3 * i + 2];
      float v = 0.f;
      if (kind == RSIM_OBS_QPOS) v = sm.qpos[a];
      else if (kind == RSIM_OBS_COS) { float sn, cs; sincos_f(sm.qpos[a], sn, cs); v = cs; }
      else if (kind == RSIM_OBS_SIN) { float sn, cs; sincos_f(sm.qpos[a], sn, cs); v = sn; }
      else if (kind == RSIM_OBS_QVEL) v = sm.qvel[a];
      else if (kind == RSIM_OBS_QACC) v = sm.qacc[a];
      else if (kind == RSIM_OBS_SITE_POS) v = sm.spos[3 * a + b2];
      else if (kind == RSIM_OBS_BODY_POS) v = sm.xpos[3 * (a < 0 ? seli(t.obj_body, task_obj) : a) + b2];
      else if (kind == RSIM_OBS_TASK_OBJECT) v = (float)task_obj;
      else if (kind == RSIM_OBS_BODY_MINUS_SITE) v = sm.xpos[3 * a + (b2 & 3)] - sm.spos[3 * (b2 >> 2) + (b2 & 3)];
      else if (kind == RSIM_OBS_BODY_MINUS_BODY) v = sm.xpos[3 * a + (b2 & 3)] - sm.xpos[3 * (b2 >> 2) + (b2 & 3)];
      else if (kind == RSIM_OBS_REL_POS || kind == RSIM_OBS_REL_QUAT) {
        // world_pose_in_gripper @ obj_pose (manipulation_env.py:244-303): gripper = {grip site position, eef body quaternion} now, object = cached
        if (!reset_obs) {
          const float* o = prev + 7 * (a < 0 ? task_obj : a);
          const Q4 qo = {o[6], o[3], o[4], o[5]};                       // record holds xyzw
          const M3 Re = q2m(ldq(sm.xquat + 4 * t.eef_body)), Ro = q2m(qnorm(qo));
          if (kind == RSIM_OBS_REL_POS) {
            const V3 r = mtv(Re, v3(o[0], o[1], o[2]) - ld3(sm.spos + 3 * t.grip_site));
            v = b2 == 0 ? r.x : (b2 == 1 ? r.y : r.z);
          } else {
            const M3 Rr = mtm(Re, Ro);
            const Q4 q = mat2quat_xyzw(Rr.m);
            v = b2 == 0 ? q.x : (b2 == 1 ? q.y : (b2 == 2 ? q.z : q.w));
          }
        }
      }
      else if (kind == RSIM_OBS_PEG_COS) v = peg_c;
      else if (kind == RSIM_OBS_PEG_T) v = peg_t;
      else if (kind == RSIM_OBS_PEG_D) v = peg_d;
      else if (kind == RSIM_OBS_BODY_QUAT) v = sm.xquat[4 * (a < 0 ? seli(t.obj_body, task_obj) : a) + (b2 == 3 ? 0 : b2 + 1)];   // wxyz -> xyzw
      else if (kind == RSIM_OBS_SITE_QUAT) { const Q4 q = mat2quat_xyzw(sm.smat + 9 * a); v = b2 == 0 ? q.x : (b2 == 1 ? q.y : (b2 == 2 ? q.z : q.w)); }
      obs[i] = v;
    }
    if (t.task == 4) {
      // PickPlace._check_success + staged_rewards (pick_place.py:274-429, 737-762), all-objects mode.  Lane i = object i for the per-object
      // terms; the contact scan for the grasp runs over the contact lanes against the union of the active objects' geoms.
      const int i = lane < t.nobj ? lane : 0;
      const V3 op = ld3(sm.xpos + 3 * seli(t.obj_body, i)), grip = ld3(sm.spos + 3 * t.grip_site);
      const float dist = norm(grip - op);
      float bxl = t.bin2_pos[0], byl = t.bin2_pos[1];
      if (i == 0 || i == 2) bxl -= t.bin_size[0] * 0.5f;
      if (i < 2) byl -= t.bin_size[1] * 0.5f;
      const bool inside = bxl < op.x && op.x < bxl + t.bin_size[0] * 0.5f && byl < op.y && op.y < byl + t.bin_size[1] * 0.5f && t.bin2_pos[2] < op.z && op.z < t.bin2_pos[2] + 0.1f;
      const bool in_bin = lane < t.nobj && inside && (1.f - tanhf(10.f * dist)) < 0.6f;
      const bool activeo = lane < t.nobj && !in_bin;
      const u64 act_mask = __ballot(activeo), in_mask = __ballot(in_bin);
      u64 ageoms = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) if ((act_mask >> k) & 1ull) ageoms |= t.obj_geoms[k];
      bool lc = false, rc = false;
      if (lane < sm.ncon) {
        const u64 b1 = 1ull << (sm.cg1[lane] & 255), b2 = 1ull << (sm.cg2[lane] & 255);
        const bool o1 = (ageoms & b1) != 0, o2 = (ageoms & b2) != 0;
        lc = (o1 && (t.left_pad & b2)) || (o2 && (t.left_pad & b1));
        rc = (o1 && (t.right_pad & b2)) || (o2 && (t.right_pad & b1));
      }
      const bool grasp = __ballot(lc) != 0 && __ballot(rc) != 0;
      const float BIG = 3.0e38f;
      const float dmin = wave_min_f(activeo ? dist : BIG);
      const float zd = fmaxf(t.bin2_pos[2] + 0.25f - op.z, 0.f), zmin = wave_min_f(activeo ? zd : BIG);
      const float r_reach = act_mask ? (1.f - tanhf(10.f * dmin)) * 0.1f : 0.f;
      const float r_grasp = grasp ? 0.35f : 0.f;
      const float r_lift = (act_mask && grasp) ? 0.35f + (1.f - tanhf(15.f * zmin)) * 0.15f : 0.f;
      const float tx = sel(t.bin_target, 2 * i), ty = sel(t.bin_target, 2 * i + 1);
      const bool above = fabsf(op.x - tx) < t.bin_size[0] * 0.25f && fabsf(op.y - ty) < t.bin_size[1] * 0.25f;
      const float dxy = sqrtf((tx - op.x) * (tx - op.x) + (ty - op.y) * (ty - op.y));
      const float hov = (above ? 0.5f : r_lift) + (1.f - tanhf(10.f * dxy)) * 0.2f;
      const float r_hover = act_mask ? wave_max(activeo ? hov : -BIG) : 0.f;
      if (lane == 0) {
        const int nin = __popcll(in_mask);
        float r = (float)nin;
        if (t.reward_shaping) r += fmaxf(fmaxf(r_reach, r_grasp), fmaxf(r_lift, r_hover));
        if (reward) { *reward = t.single_mode ? r * t.reward_scale : r * t.reward_scale / 4.0f; *success = (t.single_mode ? nin > 0 : nin == t.nobj) ? 1 : 0; }
      }
    } else if (t.task == 3) {
      if (lane == 0) {
        const bool succ = peg_d < 0.06f && peg_t >= -0.12f && peg_t <= 0.14f && peg_c > 0.95f;
        float r = succ ? 1.f : 0.f;
        if (t.reward_shaping) {
          const float dist = norm(ld3(sm.xpos + 3 * t.object_body) - ld3(sm.xpos + 3 * t.object2_body));
          r += (1.f - tanhf(dist)) + (1.f - tanhf(peg_d)) + (1.f - tanhf(fabsf(peg_t))) + peg_c;
        } else r *= 5.f;
        if (reward) { *reward = r * t.reward_scale / 5.0f; *success = succ ? 1 : 0; }
      }
    } else if (t.task >= 1) {
      // grasp: both finger-pad geom groups touch the object (contact list of the last substep)
      bool lc = false, rc = false, oo = false;
      if (lane < sm.ncon) {
        const unsigned long long b1 = 1ull << (sm.cg1[lane] & 255), b2 = 1ull << (sm.cg2[lane] & 255);
        const bool obj1 = (t.object_geoms & b1) != 0, obj2 = (t.object_geoms & b2) != 0;
        lc = (obj1 && (t.left_pad & b2)) || (obj2 && (t.left_pad & b1));
        rc = (obj1 && (t.right_pad & b2)) || (obj2 && (t.right_pad & b1));
        oo = (obj1 && (t.object2_geoms & b2)) || (obj2 && (t.object2_geoms & b1));   // check_contact(cubeA, cubeB)
      }
      const bool grasp = __ballot(lc) != 0 && __ballot(rc) != 0;
      const bool touching = __ballot(oo) != 0;
      if (lane == 0 && t.task == 2) {
        // Stack.staged_rewards (stack.py:268-312): max(reach + grasp, lift + align, stack), x reward_scale / 2
        const V3 cA = ld3(sm.xpos + 3 * t.object_body), cB = ld3(sm.xpos + 3 * t.object2_body), grip = ld3(sm.spos + 3 * t.grip_site);
        float r_reach = (1.f - tanhf(10.f * norm(cA - grip))) * 0.25f;
        if (grasp) r_reach += 0.25f;
        const bool lifted = cA.z > t.table_height + t.lift_margin;
        float r_lift = lifted ? 1.f : 0.f;
        if (lifted) { const float dx = cA.x - cB.x, dy = cA.y - cB.y; r_lift += 0.5f * (1.f - tanhf(sqrtf(dx * dx + dy * dy))); }
        const float r_stack = (!grasp && r_lift > 0.f && touching) ? 2.f : 0.f;
        const float r = t.reward_shaping ? fmaxf(r_reach, fmaxf(r_lift, r_stack)) : r_stack;
        if (reward) { *reward = r * t.reward_scale / 2.0f; *success = r_stack > 0.f ? 1 : 0; }
      }
      if (lane == 0 && t.task == 1) {
        const V3 cube = ld3(sm.xpos + 3 * t.object_body), grip = ld3(sm.spos + 3 * t.grip_site);
        const bool succ = cube.z > t.table_height + t.lift_margin;
        float r = 0.f;
        if (succ) r = 2.25f;
        else if (t.reward_shaping) { r = 1.f - tanhf(10.f * norm(cube - grip)); if (grasp) r += 0.25f; }
        if (reward) { *reward = r * t.reward_scale / 2.25f; *success = succ ? 1 : 0; }
      }
    }
  }

  __device__ __forceinline__ void fwd_constraint() {
    if (sm.nefc == 0) {
      if (lane < m.nv) { sm.qacc[lane] = sm.qacc_smooth[lane]; sm.qfrc_constraint[lane] = 0.f; }
      if (lane == 0) sm.niter = 0;
      SYNC();
      return;
    }
    solve_newton();
  }
};

// ------------------------------------------------------------------------------------------------------------
// the step kernel
// ------------------------------------------------------------------------------------------------------------
// What an env carries from the native body of a fused-tier kernel (RSIM_DIMS_W) into the wide body when a substep asks for more contacts / rows than the native
// capacity: the substep to carry on with and the per-env values that live in registers.  The state arrays (qpos, qvel, qacc, qacc_warmstart, ctrl) stay where
// they are -- both LDS layouts keep them at the same offsets -- and nothing else of a substep is persistent before its solver / integrator ran.
struct Handover { int sub0; float time; bool fresh_ctrl, handed; int ndiverged, need_con, need_efc; float cstate; unsigned t_launch; };

// FUSED: 0 = the kernel holds this body only; 1 = native body of a fused-tier kernel (returns true, with `ho` filled, when the env has to carry on in the wide
// body); 2 = the wide body of such a kernel (ho->sub0 > 0: carries on from the native body's LDS state at that substep; 0: an env that was on the tier already)
template <int NB, int NJ, int NV, int NG, int NS, int NCON, int NEFC, int NPAIR, bool DBG, int FUSED = 0>
__device__ __forceinline__ bool step_body(const DModel& m, const DBatch& b, const float* __restrict__ actions, int n_sub, int flags, int slot, Handover* ho = nullptr) {
  typedef Smem<NB, NJ, NV, NG, NS, NCON, NEFC, NPAIR> SM;
  const int lane = threadIdx.x;
  // Capacity tiers (DBatch.tier_pass >= 0, rsim_api.cpp launch()): pass 0 = this configuration steps the envs whose tier is 0 and hands an env that
  // runs out of contact / row capacity to the redo list WITHOUT committing anything of the step; passes 1 / 2 = a wider configuration steps the
  // envs of a list (1: the envs that were close to the native capacity last step, 2: the redo list), `slot` = index into that list.
  // Fused-tier kernels: the native body is pass 0 without a redo list (it hands over in place), the wide body behaves as pass 2 on the env of its own slot.
  const int tpass = (!DBG && b.tier_cur) ? (FUSED == 2 ? 2 : FUSED == 1 ? 0 : b.tier_pass) : -1;
  const bool resume = FUSED == 2 && ho->sub0 > 0;
  if ((tpass <= 0 || FUSED) && slot >= (b.nenv ? b.nenv : b.B)) return false;
  // workgroups are dispatched in index order: handing the envs that were slowest in the previous launch to the first workgroups
  // (contact-rich envs stay contact-rich for many control steps) keeps the last wave of envs short
  const int env = uni((tpass > 0 && !FUSED) ? b.wlist[slot] : (b.order ? b.order[slot] : slot) + b.env0);   // scalar: every per-env base address below then lives in SGPRs
  // RF_RESET_ONLY: the pass that follows a control step and produces the observation MujocoEnv.reset() returns (forward + epilogue, no reward)
  // for the envs that step re-initialised from the reset bank; every other workgroup leaves at once
  if ((flags & RF_RESET_ONLY) && !b.needs_reset[env]) return false;
  if (!FUSED && tpass == 0 && b.tier_cur[env] != 0) return false;    // stepped by the wide configuration in this control step (fused kernels: k_step picks the body)
  const unsigned t_launch = resume ? ho->t_launch : (b.cost ? (unsigned)uni((int)(clock64() >> 6)) : 0u);   // 64-tick units, scalar
  const float* fp = m.ft + (size_t)env * m.fstride;
  Sim<SM> sim(m, fp, lane, b.prof, b.cm, b.cm_stride ? (const char*)b.cm_env + (size_t)env * b.cm_stride : (const char*)b.cm);
  sim.pf.acc = b.prof_env == -1 || b.prof_env == env;   // -1: every env, -2: none (undistorted wave log)
  if (b.prof) sim.pf.pairs = b.prof + RP_COUNT + 8 * (size_t)b.B;
  sim.pf.start();
  if (b.prof && lane == 0 && !resume) {
    unsigned long long* wl = b.prof + RP_COUNT + 8 * (size_t)env;
    wl[0] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
    wl[1] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    wl[2] = wall_clock64();
  }
  // ---- load state (an env handed over in mid-step finds qpos .. ctrl in LDS, as of the start of the substep it carries on with)
  if (!resume) {
    for (int i = lane; i < m.nq; i += 64) sm.qpos[i] = b.qpos[(size_t)env * m.nq + i];
    for (int i = lane; i < m.nv; i += 64) { sm.qvel[i] = b.qvel[(size_t)env * m.nv + i]; sm.qacc_ws[i] = b.qacc_ws[(size_t)env * m.nv + i]; }
    if (lane >= m.nv && lane < NV) { sm.qvel[lane] = 0.f; sm.qacc_ws[lane] = 0.f; sm.qacc[lane] = 0.f; }
    for (int i = lane; i < m.nu; i += 64) sm.ctrl[i] = b.ctrl[(size_t)env * m.nu + i];
  }
  const int cs = m.ctrl.cs_size, csl = cs < RSIM_CS_LDS ? cs : RSIM_CS_LDS;
  sim.cst = (gwf)(b.cstate + (size_t)env * cs);
  if (b.bpl) sim.bpl = (int __attribute__((address_space(1)))*)(b.bpl + (size_t)env * 320);
  if (b.mprc) { sim.mprc = (gwf)(b.mprc + (size_t)env * Sim<SM>::MPRC * m.npair); sim.mpr_portal = b.mprc_portal != 0; }
  if constexpr (Sim<SM>::JG) sim.Jg = (gwf)(b.jg + (size_t)env * b.jg_stride);   // one stride for every configuration that steps envs of this batch (the native and the wide pass run side by side)
  if constexpr (Sim<SM>::MG) sim.Mg = sim.Jg + SM::NEFC_ * SM::JS_;
  if (lane < csl) sm.cstate[lane] = resume ? ho->cstate : sim.cst[lane];
  if constexpr (DBG) if (b.qfrc_applied && lane < m.nv) sim.applied = b.qfrc_applied[(size_t)env * m.nv + lane];   // user forces of the B = 1 shim entries (GripperTester's gravity compensation)
  if (lane == 0) { sm.ncon = 0; sm.nefc = 0; sm.niter = 0; sm.polish = 0; }
  sim.load_opt();
  sim.init_lds();
  const float* act = actions ? actions + (size_t)env * m.ctrl.action_dim : nullptr;
  float time = __builtin_bit_cast(float, uni(__builtin_bit_cast(int, b.time[env])));   // wave-uniform, carried over all substeps: a scalar register
  bool fresh_ctrl = false;
  int ndiverged = 0;
  if (resume) { time = ho->time; fresh_ctrl = ho->fresh_ctrl; ndiverged = ho->ndiverged; sim.need_con = ho->need_con; sim.need_efc = ho->need_efc; }
  if (!resume && (flags & RF_CTRL) && b.needs_reset[env]) {
    // this env was re-initialised on the device when its previous episode ended: fresh controller objects (robots/robot.py:271)
    V3 xp0; Q4 xq0;
    if (lane < csl) sm.cstate[lane] = 0.f;
    for (int i = RSIM_CS_LDS + lane; i < cs; i += 64) sim.cst[i] = 0.f;
    SYNC();
    sim.cst_sync();
    sim.kinematics(xp0, xq0);
    sim.geom_site_frames();
    sim.ctrl_reset();
    fresh_ctrl = true;   // the flag is cleared when the step is committed (a step handed to the wide configuration must find it still set)
  }
  sim.pf.mark(RP_LOAD);
  int sub = resume ? ho->sub0 : 0;
  bool over = false;   // fused native body: this env carries on in the wide body from substep `sub`
  const int force_sub = (FUSED == 1 && b.tier_pass >= 100) ? b.tier_pass - 100 : -1;   // test hook (RSIM_FORCE_HANDOVER): hand over at this substep whatever the demand
  for (; sub < n_sub; sub++) {
    V3 xp; Q4 xq;
    sim.phase();
    sim.kinematics(xp, xq);
    sim.phase();
    sim.geom_site_frames();
    sim.pf.mark(RP_KIN);
    sim.phase();
    sim.com_pos(xp, xq);
    sim.pf.mark(RP_COM);
    sim.phase();
    sim.crb();
    sim.pf.mark(RP_CRB);
    sim.phase();
    sim.collision(n_sub - sub);
    sim.pf.mark(RP_NARROW);
    if (FUSED == 1 && tpass == 0 && (uni(sim.ovf) || sub == force_sub)) { over = true; break; }   // more contacts than this body holds: the wide body carries on with this substep (nothing of it is persistent yet)
    sim.phase();
    sim.make_constraint();
    sim.pf.mark(RP_MAKEC);
    if (FUSED == 1 && tpass == 0 && uni(sim.ovf)) { over = true; break; }   // ... or more constraint rows
    sim.phase();
    sim.velocity(xp, xq);
    sim.pf.mark(RP_VEL);
    if (flags & RF_CTRL) {
      sim.phase();
      if ((flags & RF_SETGOAL) && sub == 0 && act) {
        const float* act0 = act;
        asm volatile("" : "+s"(act0));   // opaque: keeps the action scaling from being precomputed at kernel entry and parked in the private segment until here
        sim.ctrl_set_goal(act0);
      }
      sim.ctrl_run();
      sim.pf.mark(RP_CTRL);
    }
    if (flags & RF_ACTSOLVE) {
      sim.phase();
      sim.actuation_acceleration();
      sim.pf.mark(RP_ACT);
      sim.phase();
      sim.fwd_constraint();
      sim.pf.mark(RP_SOLVE);
      sim.pf.count(RP_N_CON, sm.ncon);
      sim.pf.count(RP_N_EFC, sm.nefc);
      if constexpr (DBG) {
        if ((flags & RF_DEBUG) && b.sensordata && m.nsensor > 0 && sub == n_sub - 1) { sim.phase(); sim.sensor_acc(xp, xq, b.sensordata + (size_t)env * m.nsensordata); }
      }
    }
    if (flags & RF_INTEGRATE) {
      sim.phase();
      sim.euler();
      sim.pf.mark(RP_EULER);
      time += sim.opt_h;
      // MuJoCo's bad-state guard (mj_checkPos / mj_checkVel -> mj_resetData [3P]): a non-finite or absurdly large coordinate puts the env back
      // to qpos0 with zero velocity / control / time instead of letting NaNs run on.  Controller state is the caller's (robosuite objects).
      bool bad = false;
#ifdef RSIM_GUARD_LANE
      const int gl = sim.lane;   // the phase's opaque lane id (phase() above): with the kernel's own `lane` the LDS addresses below are computed before the substep loop and spilled across it
#else
      const int gl = lane;
#endif
      for (int i = gl; i < m.nq; i += 64) bad |= !(fabsf(sm.qpos[i]) < 1.0e10f);
      if (gl < m.nv) bad |= !(fabsf(sm.qvel[gl]) < 1.0e10f);
      if (__ballot(bad)) {
        SYNC();
        for (int i = gl; i < m.nq; i += 64) sm.qpos[i] = FPM(FO_qpos0, i);
        if (gl < NV) { sm.qvel[gl] = 0.f; sm.qacc_ws[gl] = 0.f; sm.qacc[gl] = 0.f; sm.ctrl[gl] = 0.f; }
        time = 0.f;
        ndiverged++;
        SYNC();
      }
    }
    sim.pf.count(RP_N_SUB, 1);
  }
  if (FUSED == 1 && over) {
    // hand-over in place: the env carries on in the wide body of this kernel from substep `sub`, on the LDS-resident state as the substeps before left it
    ho->handed = true; ho->sub0 = sub; ho->time = time; ho->fresh_ctrl = fresh_ctrl; ho->ndiverged = ndiverged; ho->need_con = sim.need_con; ho->need_efc = sim.need_efc;
    ho->cstate = lane < csl ? sm.cstate[lane] : 0.f; ho->t_launch = t_launch;
    SYNC();   // (a hand-over in the first substep, sub0 == 0, is simply a wide step from the stored state)
    return true;
  }
  if (!FUSED && tpass == 0 && sim.ovf) {
    // a substep asked for more contacts / rows than this configuration holds: nothing of this control step is committed (state, controller
    // state, episode counters, observation record are as they were); the env goes on the redo list and the wide configuration steps it from the
    // same state later in this same rsim_control_step.  The caches this pass touched (warm-start records, broadphase list) validate themselves.
    if (lane == 0) { b.wlist2[atomicAdd(b.wcount2, 1)] = env; b.tier_next[env] = 1; }
    if (b.cost && lane == 0) b.cost[env] = (unsigned)(clock64() >> 6) - t_launch;
    return false;
  }
  if (fresh_ctrl && lane == 0) b.needs_reset[env] = 0;
  if (ndiverged && lane == 0) b.diverged[env] += ndiverged;
  if (tpass > 0 && b.tstat && lane == 0) { atomicAdd(b.tstat, 1ull); if (FUSED == 2 ? ho->handed : tpass == 2) atomicAdd(b.tstat + 1, 1ull); }   // a step the wider tier committed (rsim_tier_stats)
  if (tpass >= 0 && lane == 0) {
    // next step's tier: up when a substep came within a few contacts / rows of the native capacity (so that most hand-overs happen between steps,
    // without a redo), down again once the demand has dropped well below it
    const bool up = tpass == 0 ? (sim.need_con > SM::NCON_ - b.tier_up_con || sim.need_efc > SM::NEFC_ - b.tier_up_efc)
                               : !(sim.need_con <= b.tier_con - b.tier_up_con - 2 && sim.need_efc <= b.tier_efc - b.tier_up_efc - 8);
    b.tier_next[env] = up ? 1 : 0;
  }
  if ((flags & RF_OBS) && m.task.enabled) {
    if (m.task.single_mode == 1) sim.task_obj = b.task_object[env] & 3;
    sim.obs_reward(b.obs + (size_t)env * m.task.nobs, (flags & RF_RESET_ONLY) ? nullptr : b.reward + env, b.success + env, !(flags & RF_INTEGRATE));   // a record taken without advancing time is the one reset() returns (rsim_observe, k_reset_obs); rsim_step2_last integrates and runs no in-kernel controller
  }
  if (flags & RF_EPISODE) {
    // MujocoEnv.step: timestep += 1; done = timestep >= horizon (base.py:508, 532-548); optional on-device reset from the bank
    int st = b.ep_step[env] + 1;
    const bool done = b.horizon > 0 && st >= b.horizon;
    if (done && b.bank) {
      // episodes are numbered for ever; the bank is a ring of bank_E slots per env that the host keeps refilling (rsim_refill_reset_bank), so no
      // reset is ever replayed.  A slot whose tag is not the episode about to start was not refilled in time: counted (RSIM_BANK_STALE), still used.
      const int ep = b.ep_index[env] + 1, slot = ep % b.bank_E;
      const float* src = b.bank + ((size_t)env * b.bank_E + slot) * (m.nq + b.bank_P);
      if (lane == 0 && b.bank_tag && b.bank_tag[(size_t)env * b.bank_E + slot] != ep) b.bank_stale[env] += 1;
      // gym auto-reset convention: the record of the step that ended the episode moves to RSIM_TERMINAL_OBS; RSIM_OBS receives the reset
      // observation from the reset-only pass that follows this launch (same lanes wrote these floats in obs_reward above)
      if (b.term_obs && m.task.enabled) for (int i = lane; i < m.task.nobs; i += 64) b.term_obs[(size_t)env * m.task.nobs + i] = b.obs[(size_t)env * m.task.nobs + i];
      SYNC();
      for (int i = lane; i < m.nq; i += 64) sm.qpos[i] = src[i];
      if (lane < NV) { sm.qvel[lane] = 0.f; sm.qacc_ws[lane] = 0.f; sm.ctrl[lane] = 0.f; }
      for (int p2 = lane; p2 < b.bank_P; p2 += 64) {
        const int pi = b.patch_idx[p2];
        if (pi < 0) { b.task_object[env] = (int)src[m.nq + p2]; continue; }   // RSIM_PATCH_TASK_OBJECT: the new episode's object (PickPlace single-object mode 1)
        b.ft_rw[(size_t)env * m.fstride + pi] = src[m.nq + p2];
        if (b.ft_base) b.ft_base[(size_t)env * m.fstride + pi] = src[m.nq + p2];
      }
      time = 0.f;
      st = 0;
      if (sim.mprc) for (int p2 = lane; p2 < m.npair; p2 += 64) sim.mprc[Sim<SM>::MPRC * p2 + 3] = 0.f;   // the new episode's narrow phase starts cold, as after a host reset
      if (lane == 0) { b.ep_index[env] = ep; b.needs_reset[env] = 1; }
      SYNC();
      // the patched float-table entries change this env's constant block: the host follows this launch with k_prepare over the envs whose
      // needs_reset flag is set (inlining prepare_constants() here makes the compiler keep a private copy of DModel)
    }
    if (lane == 0) { b.done[env] = done ? 1 : 0; b.ep_step[env] = st; }
  }
  // ---- store state (the debug form under RF_NOSTORE -- the refresh of the derived arrays when the host reads one after a fused step -- leaves it alone)
  if (!DBG || !(flags & RF_NOSTORE)) {
    for (int i = lane; i < m.nq; i += 64) b.qpos[(size_t)env * m.nq + i] = sm.qpos[i];
    for (int i = lane; i < m.nv; i += 64) { b.qvel[(size_t)env * m.nv + i] = sm.qvel[i]; b.qacc_ws[(size_t)env * m.nv + i] = sm.qacc_ws[i]; }
    for (int i = lane; i < m.nu; i += 64) b.ctrl[(size_t)env * m.nu + i] = sm.ctrl[i];
    if (lane < csl) sim.cst[lane] = sm.cstate[lane];
    if (lane == 0) b.time[env] = time;
  }
  if (b.overflow && lane == 0 && sim.ovf) b.overflow[env] += sim.ovf;   // with capacity tiers: drops of the WIDE configuration only (pass 0 left above)
  if (b.cap_need && lane == 0) { int* cn = b.cap_need + 2 * (size_t)env; if (sim.need_con > cn[0]) cn[0] = sim.need_con; if (sim.need_efc > cn[1]) cn[1] = sim.need_efc; }
  if (b.cost && lane == 0) {
    // dispatch-order key of the next step: this step's duration -- and half as much again for an env one of whose convex pairs ended within DModel.near_thresh of
    // touching: a contact that STARTS next step (full MPR runs, several times the Newton iterations) is what a duration cannot see coming, and such an env in the
    // last round of a launch is what ends it (profiles/r06_t_ab_peg_five_per_cu.txt).  Ordering only: no result depends on it.
    unsigned cst = (unsigned)(clock64() >> 6) - t_launch;
    if (sim.near_sep < m.near_thresh) cst += (unsigned)((float)cst * m.near_gain);
    b.cost[env] = cst;
  }
  if (b.prof && lane == 0) {
    unsigned long long* wl = b.prof + RP_COUNT + 8 * (size_t)env;
    wl[3] = wall_clock64(); wl[4] = sim.pf.c_mpr; wl[5] = sim.pf.c_support; wl[6] = sim.pf.c_newton; wl[7] = sim.pf.c_cand;
  }
  if constexpr (DBG) if (flags & RF_DEBUG) {
    const int nb = m.nbody, nv = m.nv;
    for (int i = lane; i < nb * 3; i += 64) { b.xpos[(size_t)env * nb * 3 + i] = sm.xpos[i]; b.rootcom[(size_t)env * nb * 3 + i] = sm.rootcom[3 * sim.cm->broot[i / 3] + i % 3]; }
    for (int i = lane; i < nb * 4; i += 64) b.xquat[(size_t)env * nb * 4 + i] = sm.xquat[i];
    for (int e = lane; e < nv * nv; e += 64) { int i = e / nv, j = e - i * nv; b.qM[(size_t)env * nv * nv + e] = sim.Mrd(i * SM::NVP + j); }
    for (int e = lane; e < nv * 6; e += 64) b.cdof[(size_t)env * nv * 6 + e] = sm.cdof[(e / 6) * 9 + e % 6];
    for (int i = lane; i < nv; i += 64) {
      size_t o = (size_t)env * nv + i;
      b.qfrc_bias[o] = sm.qfrc_bias[i]; b.qfrc_passive[o] = sm.qfrc_passive[i];
      if (flags & RF_ACTSOLVE) { b.qfrc_actuator[o] = sm.qfrc_actuator[i]; b.qfrc_constraint[o] = sm.qfrc_constraint[i]; b.qacc[o] = sm.qacc[i]; }
    }
    int ncon = sm.ncon;
    sim.csync();
    for (int c = lane; c < ncon; c += 64) {
      float* r = b.contact + ((size_t)env * NCON + c) * RSIM_CON_REC;
      r[0] = sm.cdist[c];
      for (int k = 0; k < 3; k++) r[1 + k] = sm.cpos[3 * c + k];
      for (int k = 0; k < 9; k++) r[4 + k] = sim.cframe_rd(9 * c + k);
      r[13] = (float)IT(IO_cg_geomid, sm.cg1[c] & 255); r[14] = (float)IT(IO_cg_geomid, sm.cg2[c] & 255); r[15] = (float)sm.cdim[c]; r[16] = (float)sm.cefc[c];
      r[17] = ((flags & RF_ACTSOLVE) && sm.cefc[c] >= 0) ? sm.e_force[sm.cefc[c]] : 0.f;
      for (int k = 0; k < 5; k++) r[18 + k] = sim.cfri_rd(5 * c + k);
    }
    if (flags & RF_ACTSOLVE)
      for (int i = lane; i < sm.nefc; i += 64) b.efc_force[(size_t)env * NEFC + i] = sm.e_force[i];
    if (lane == 0) { b.ncon[env] = ncon; b.nefc[env] = sm.nefc; b.niter[env] = sm.niter; if (b.polish) b.polish[env] = sm.polish; }
  }
  return false;
}

#ifdef RSIM_WAVES_PER_EU   /* experiment builds: state the occupancy target to the register allocator directly (launch_bounds' second argument tops out at two for 64-thread blocks) */
#define RSIM_KSTEP_ATTR __attribute__((amdgpu_waves_per_eu(RSIM_WAVES_PER_EU, RSIM_WAVES_PER_EU)))
#else
#define RSIM_KSTEP_ATTR
#endif
template <int NB, int NJ, int NV, int NG, int NS, int NCON, int NEFC, int NPAIR>
__global__ RSIM_KSTEP_ATTR __launch_bounds__(64, RSIM_MINWAVES) void k_step(DModel m, DBatch b, const float* __restrict__ actions, int n_sub, int flags) {
#ifdef RSIM_DIMS_W
  // fused-tier kernel: the workgroup steps its env with the native body; an env that is on the wide tier already (DBatch.tier_cur) or outgrows the native
  // capacity in mid-step (step_body returns true) is stepped / carried on by the wide body in this same workgroup.  No list, no second launch: the envs that
  // need the tier are the slowest of a control step, and a kernel of their own could not start before the native launch had drained.
  // Both bodies read the kernel arguments straight from the kernarg segment.  Left to itself the compiler copies the by-value DModel / DBatch into the private
  // segment once TWO bodies use them (2.4 KB of scratch per lane, every per-lane table lookup a scratch load: the copy is only elided while the uses of the
  // argument stay below an analysis limit, which one body does and two do not).
  typedef const char __attribute__((address_space(4)))* ka_t;
  ka_t ka = (ka_t)__builtin_amdgcn_kernarg_segment_ptr();
  constexpr size_t b_off = (sizeof(DModel) + alignof(DBatch) - 1) & ~(alignof(DBatch) - 1);
  const DModel& mk = *(const DModel*)(const DModel __attribute__((address_space(4)))*)ka;
  const DBatch& bk = *(const DBatch*)(const DBatch __attribute__((address_space(4)))*)(ka + b_off);
#define m mk
#define b bk
  const int slot = (int)blockIdx.x;
  Handover ho;
  ho.sub0 = 0; ho.time = 0.f; ho.fresh_ctrl = false; ho.handed = false; ho.ndiverged = 0; ho.need_con = 0; ho.need_efc = 0; ho.cstate = 0.f; ho.t_launch = 0u;
  bool wide = false;
#ifdef RSIM_FUSED_TEST_WIDE_ONLY
  step_body<RSIM_DIMS_W, false, 2>(m, b, actions, n_sub, flags, slot, &ho); return;
#endif
  if (b.tier_cur) {
    if (slot >= (b.nenv ? b.nenv : b.B)) return;
    wide = b.tier_cur[uni((b.order ? b.order[slot] : slot) + b.env0)] != 0;
  }
  if (!wide) {
    const int over = uni(step_body<NB, NJ, NV, NG, NS, NCON, NEFC, NPAIR, false, 1>(m, b, actions, n_sub, flags, slot, &ho) ? 1 : 0);
    if (!over) return;
#ifdef RSIM_FUSED_TEST_NATIVE_ONLY
    return;
#endif
    // wave-uniform by construction; said so explicitly (the compiler sees them leave a branch on a vector condition)
    ho.sub0 = uni(ho.sub0); ho.time = __builtin_bit_cast(float, uni(__builtin_bit_cast(int, ho.time))); ho.fresh_ctrl = uni(ho.fresh_ctrl ? 1 : 0) != 0; ho.handed = true;
    ho.ndiverged = uni(ho.ndiverged); ho.need_con = uni(ho.need_con); ho.need_efc = uni(ho.need_efc); ho.t_launch = (unsigned)uni((int)ho.t_launch);
  }
  step_body<RSIM_DIMS_W, false, 2>(m, b, actions, n_sub, flags, slot, &ho);
#undef m
#undef b
#else
  step_body<NB, NJ, NV, NG, NS, NCON, NEFC, NPAIR, false>(m, b, actions, n_sub, flags, (int)blockIdx.x);
#endif
}
// The same control step as the upper capacity tier of a batch whose model belongs to a narrower configuration: a fixed, small grid walks the list
// of envs this pass steps (DBatch.wlist / wcount, filled on the device), so a control step in which no env needs the tier costs one empty launch.
template <int NB, int NJ, int NV, int NG, int NS, int NCON, int NEFC, int NPAIR>
__global__ __launch_bounds__(64, 1) void k_step_list(DModel m, DBatch b, const float* __restrict__ actions, int n_sub, int flags) {
  const int n = *b.wcount;
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    step_body<NB, NJ, NV, NG, NS, NCON, NEFC, NPAIR, false>(m, b, actions, n_sub, flags, i);
    __syncthreads();
  }
}
// The compatibility / debug form (RF_DEBUG: the B = 1 shim entries, forward() of a batch for the parity tests): the same body plus the MuJoCo-shaped
// derived arrays and the acceleration-stage sensors.  Its own kernel, so that none of this costs the control step a register.
template <int NB, int NJ, int NV, int NG, int NS, int NCON, int NEFC, int NPAIR>
__global__ __launch_bounds__(64, 1) void k_step_dbg(DModel m, DBatch b, const float* __restrict__ actions, int n_sub, int flags) {
  step_body<NB, NJ, NV, NG, NS, NCON, NEFC, NPAIR, true>(m, b, actions, n_sub, flags, (int)blockIdx.x);
}
// The reset-observation pass that follows a control step (forward + observables for the envs it re-initialised, no reward): the same body under
// its own kernel name, so that per-kernel profiles of k_step hold control steps only (and the constant flags strip controller / integrator code)
template <int NB, int NJ, int NV, int NG, int NS, int NCON, int NEFC, int NPAIR>
__global__ __launch_bounds__(64, 1) void k_reset_obs(DModel m, DBatch b) {
  step_body<NB, NJ, NV, NG, NS, NCON, NEFC, NPAIR, false>(m, b, nullptr, 1, RF_POSVEL | RF_ACTSOLVE | RF_OBS | RF_RESET_ONLY, (int)blockIdx.x);
}

// controller reset kernel: forward kinematics then OSC.reset_goal / initial_joint capture
template <int NB, int NJ, int NV, int NG, int NS, int NCON, int NEFC, int NPAIR>
__global__ __launch_bounds__(64) void k_ctrl_reset(DModel m, DBatch b, const unsigned char* __restrict__ mask) {
  typedef Smem<NB, NJ, NV, NG, NS, NCON, NEFC, NPAIR> SM;
  const int env = blockIdx.x, lane = threadIdx.x;
  if (env >= b.B) return;
  if (mask && !mask[env]) return;
  const float* fp = m.ft + (size_t)env * m.fstride;
  Sim<SM> sim(m, fp, lane, nullptr, b.cm, b.cm_stride ? (const char*)b.cm_env + (size_t)env * b.cm_stride : (const char*)b.cm);
  for (int i = lane; i < m.nq; i += 64) sm.qpos[i] = b.qpos[(size_t)env * m.nq + i];
  const int cs = m.ctrl.cs_size, csl = cs < RSIM_CS_LDS ? cs : RSIM_CS_LDS;
  sim.cst = (gwf)(b.cstate + (size_t)env * cs);
  if constexpr (Sim<SM>::JG) sim.Jg = (gwf)(b.jg + (size_t)env * b.jg_stride);   // one stride for every configuration that steps envs of this batch (the native and the wide pass run side by side)
  if constexpr (Sim<SM>::MG) sim.Mg = sim.Jg + SM::NEFC_ * SM::JS_;
  if (lane < csl) sm.cstate[lane] = 0.f;
  for (int i = RSIM_CS_LDS + lane; i < cs; i += 64) sim.cst[i] = 0.f;
  sim.cst_sync();
  sim.load_opt();
  sim.init_lds();
  V3 xp; Q4 xq;
  sim.kinematics(xp, xq);
  sim.geom_site_frames();
  sim.ctrl_reset();
  if (lane < csl) sim.cst[lane] = sm.cstate[lane];
}

// constant blocks (Cmem) of envs [0, grid): built from the float / int / lane tables whenever a model parameter changed (model ingest,
// rsim_model_param_set, domain randomisation, a new controller); one workgroup per block.  reset_only: just the envs that the preceding
// control step re-initialised from the reset bank (their float tables were patched).
template <int NB, int NJ, int NV, int NG, int NS, int NCON, int NEFC, int NPAIR>
__global__ __launch_bounds__(64) void k_prepare(DModel m, DBatch b, int reset_only) {
  typedef Smem<NB, NJ, NV, NG, NS, NCON, NEFC, NPAIR> SM;
  const int lane = threadIdx.x;
  if (reset_only == 2) {   // the envs of a capacity-tier list (their blocks of the WIDE configuration are built on demand, right before that configuration steps them)
    const int n = *b.wcount;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
      const int env = b.wlist[i];
      char* const cmb = (char*)b.cm_env + (size_t)env * b.cm_stride;
      Sim<SM> sim(m, m.ft + (size_t)env * m.fstride, lane, nullptr, cmb, cmb);
      sim.prepare_constants((cmw_t)cmb);
    }
    return;
  }
  const int env = blockIdx.x + b.env0;
  if (reset_only && !b.needs_reset[env]) return;
  const float* fp = m.ft + (size_t)env * m.fstride;
  // the host passes the blocks to build as b.cm_env / b.cm_stride (the shared block: one workgroup, m.fenv = 0, stride 0)
  char* const cmb = (char*)b.cm_env + (size_t)env * b.cm_stride;
  Sim<SM> sim(m, fp, lane, nullptr, cmb, cmb);
  sim.prepare_constants((cmw_t)cmb);
}


#if RSIM_CFG == 0
__device__ __forceinline__ float dr_uniform(unsigned long long seed, unsigned long long step, unsigned env, unsigned item) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (step + 1) + ((unsigned long long)env << 32 | item);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;   // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
  return (float)(z >> 40) * (2.0f / 16777216.0f) - 1.0f;   // U(-1, 1), 24-bit
}
__global__ __launch_bounds__(64) void k_randomize(DModel m, DBatch b, DDr d, unsigned long long seed, unsigned long long step) {
  const int env = blockIdx.x + b.env0, lane = threadIdx.x;
  if (env >= b.B) return;
  const float* base = b.ft_base + (size_t)env * m.fstride;
  float* out = b.ft_rw + (size_t)env * m.fstride;
  gci it = (gci)m.it;
  unsigned item = 0;
  auto ratio = [&](int off, float mag, float lo, float hi, unsigned id) {
    if (mag > 0.f) out[off] = fminf(hi, fmaxf(lo, base[off] * (1.0f + mag * dr_uniform(seed, step, env, id))));
  };
  auto size_ = [&](int off, float mag, float lo, unsigned id) {
    if (mag > 0.f) out[off] = fmaxf(lo, base[off] + mag * dr_uniform(seed, step, env, id));
  };
  const float INF = 3.0e38f;
  if (lane == 0) { ratio(m.fo[FO_opt] + 4, d.density, 0.f, INF, 1); ratio(m.fo[FO_opt] + 5, d.viscosity, 0.f, INF, 2); }
  item = 16;
  for (int bd = 1 + lane; bd < m.nbody; bd += 64) {
    if (!((d.body_mask >> bd) & 1ull)) continue;       // body_names subset (domain_randomization_wrapper.py:55)
    const unsigned id = item + 16u * bd;
    for (int k = 0; k < 3; k++) size_(m.fo[FO_body_pos] + 3 * bd + k, d.pos, -INF, id + k);
    if (d.quat > 0.f) {
      float q[4], n = 0.f;
      for (int k = 0; k < 4; k++) { q[k] = base[m.fo[FO_body_quat] + 4 * bd + k] + d.quat * dr_uniform(seed, step, env, id + 3 + k); n += q[k] * q[k]; }
      n = rsqrtf(fmaxf(n, 1e-20f));
      for (int k = 0; k < 4; k++) out[m.fo[FO_body_quat] + 4 * bd + k] = q[k] * n;
    }
    for (int k = 0; k < 3; k++) ratio(m.fo[FO_body_inertia] + 3 * bd + k, d.inertia, 0.f, INF, id + 7 + k);
    ratio(m.fo[FO_body_mass] + bd, d.mass, 0.f, INF, id + 10);
  }
  item += 16u * 64;
  for (int g = lane; g < m.ncg; g += 64) {
    if (!((d.geom_mask >> g) & 1ull)) continue;        // geom_names subset (:64)
    const unsigned id = item + 16u * g;
    for (int k = 0; k < 3; k++) ratio(m.fo[FO_cg_friction] + 3 * g + k, d.friction, 0.f, INF, id + k);
    for (int k = 0; k < 2; k++) ratio(m.fo[FO_cg_solref] + 2 * g + k, d.solref, 0.f, 1.0f, id + 3 + k);
    for (int k = 0; k < 5; k++) ratio(m.fo[FO_cg_solimp] + 5 * g + k, d.solimp, 0.f, INF, id + 5 + k);
  }
  item += 16u * 64;
  for (int i = lane; i < m.nv; i += 64) {
    const int j = it[m.io[IO_dof_jntid] + i];
    if (it[m.io[IO_jnt_type] + j] == JNT_FREE) continue;   // mjmod.py:1927: free joints keep their values
    if (!((d.joint_mask >> j) & 1ull)) continue;           // joint_names subset (:71)
    const unsigned id = item + 4u * i;
    size_(m.fo[FO_dof_frictionloss] + i, d.frictionloss, 0.f, id);
    size_(m.fo[FO_dof_damping] + i, d.damping, 0.f, id + 1);
    size_(m.fo[FO_dof_armature] + i, d.armature, 0.f, id + 2);
  }
}
extern "C" int rsim_launch_randomize(const DModel* m, const DBatch* b, const DDr* d, unsigned long long seed, unsigned long long step, hipStream_t stream) {
  hipLaunchKernelGGL(k_randomize, dim3(b->nenv ? b->nenv : b->B), dim3(64), 0, stream, *m, *b, *d, seed, step);
  return (int)hipGetLastError();
}

// order[] := env indices by decreasing cost[]: one-workgroup counting sort into 256 cost bins (exact order inside a bin is irrelevant for
// longest-job-first dispatch; results never depend on the dispatch order)
__global__ __launch_bounds__(1024) void k_order(const unsigned* __restrict__ cost, int* __restrict__ order, int B) {
  __shared__ unsigned lo_hi[2];
  __shared__ int hist[256], base[256];
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int k = tid; k < 256; k += nt) hist[k] = 0;
  if (tid == 0) { lo_hi[0] = 0xFFFFFFFFu; lo_hi[1] = 0u; }
  __syncthreads();
  unsigned lo = 0xFFFFFFFFu, hi = 0u;
  for (int i = tid; i < B; i += nt) { const unsigned c = cost[i]; lo = c < lo ? c : lo; hi = c > hi ? c : hi; }
  atomicMin(&lo_hi[0], lo); atomicMax(&lo_hi[1], hi);
  __syncthreads();
  lo = lo_hi[0]; hi = lo_hi[1];
  const float scale = 255.0f / (float)(hi - lo + 1u);
  for (int i = tid; i < B; i += nt) atomicAdd(&hist[(int)((float)(hi - cost[i]) * scale)], 1);   // bin 0 = most expensive
  __syncthreads();
  if (tid == 0) { int acc = 0; for (int k = 0; k < 256; k++) { base[k] = acc; acc += hist[k]; } }
  __syncthreads();
  for (int i = tid; i < B; i += nt) order[atomicAdd(&base[(int)((float)(hi - cost[i]) * scale)], 1)] = i;
}
// capacity tiers: the envs [env0, env0 + n) whose tier is 1, compacted (any order: results never depend on it) -- the list the wide configuration walks
// `zero_next`: the two list lengths of the NEXT control step (the lengths are double-buffered by step parity, so that no launch of a step has to wait for a memset)
__global__ __launch_bounds__(256) void k_tier_list(const int* __restrict__ tier, int* __restrict__ list, int* __restrict__ count, int* __restrict__ zero_next, int env0, int n) {
  if (blockIdx.x == 0 && threadIdx.x < 2) zero_next[threadIdx.x] = 0;
  for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < n; i += blockDim.x * gridDim.x)
    if (tier[env0 + i] == 1) list[atomicAdd(count, 1)] = env0 + i;
}
extern "C" int rsim_launch_tier_list(const int* tier, int* list, int* count, int* zero_next, int env0, int n, hipStream_t stream) {
  hipLaunchKernelGGL(k_tier_list, dim3((n + 255) / 256 < 64 ? (n + 255) / 256 : 64), dim3(256), 0, stream, tier, list, count, zero_next, env0, n);
  return (int)hipGetLastError();
}
// refill of the reset-bank ring: row i of `rows` -> slot (episode[i] % E) of env[i], and the slot's tag := episode[i]
__global__ __launch_bounds__(64) void k_bank_scatter(float* bank, int* tag, const int* env, const int* episode, const float* rows, int n, int E, int W) {
  const int i = blockIdx.x;
  if (i >= n) return;
  const int e = env[i], slot = episode[i] % E;
  float* dst = bank + ((size_t)e * E + slot) * W;
  for (int k = threadIdx.x; k < W; k += 64) dst[k] = rows[(size_t)i * W + k];
  // the tag after the row, device-wide: a control step running beside an asynchronous refill that reads the new tag also reads the new row
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) tag[(size_t)e * E + slot] = episode[i];
}
extern "C" int rsim_launch_bank_scatter(float* bank, int* tag, const int* env, const int* episode, const float* rows, int n, int E, int W, hipStream_t stream) {
  hipLaunchKernelGGL(k_bank_scatter, dim3(n), dim3(64), 0, stream, bank, tag, env, episode, rows, n, E, W);
  return (int)hipGetLastError();
}
extern "C" int rsim_launch_order(const unsigned* cost, int* order, int B, hipStream_t stream) {
  // one wavefront for an env block of a stream group: its 26 registers fit next to the resident k_step waves of the other blocks, whereas a
  // 1024-thread workgroup waits (0.24 ms measured) until a whole CU has room for sixteen more wavefronts
  hipLaunchKernelGGL(k_order, dim3(1), dim3(B <= 1024 ? 64 : 1024), 0, stream, cost, order, B);
  return (int)hipGetLastError();
}
#endif  // RSIM_CFG == 0

#if RSIM_CFG == 0 || RSIM_CFG == 1 || RSIM_CFG == 2
// limits of the wide body of a fused-tier build (layout of rsim_limits); returns 0 when this build has none
extern "C" int RSIM_SYM(rsim_limits_w)(int* lim) {
#ifdef RSIM_DIMS_W
  const int dims[8] = {RSIM_DIMS_W};
  for (int i = 0; i < 8; i++) lim[i] = dims[i];
  lim[8] = SmemW::NROOT_; lim[9] = (SmemW::TENDONS_ ? 1 : 0) | (SmemW::NB_ > 32 ? 2 : 0) | (SmemW::JG_ ? 4 : 0) | (SmemW::MG_ ? 8 : 0) | (SmemW::CG_ ? 16 : 0) | 32;
  static_assert(sizeof(Cmem<RSIM_DIMS_W>) == sizeof(Cmem0), "the wide body reads the native configuration's constant blocks");
  return 1;
#else
  (void)lim; return 0;
#endif
}
#endif
// explicit instantiations + launchers (one set per configuration build) ----------------------------------------------------------
template __global__ void k_step<RSIM_DIMS>(DModel, DBatch, const float*, int, int);
template __global__ void k_step_dbg<RSIM_DIMS>(DModel, DBatch, const float*, int, int);
#if RSIM_CFG == 3 || RSIM_CFG >= 5   // the configurations that serve as the upper capacity tier of narrower ones (rsim_api.cpp pick_wide)
template __global__ void k_step_list<RSIM_DIMS>(DModel, DBatch, const float*, int, int);
extern "C" int RSIM_SYM(rsim_launch_step_list)(const DModel* m, const DBatch* b, const float* actions, int n_sub, int flags, int grid, hipStream_t stream) {
  hipLaunchKernelGGL((k_step_list<RSIM_DIMS>), dim3(grid), dim3(64), 0, stream, *m, *b, actions, n_sub, flags);
  return (int)hipGetLastError();
}
#endif
extern "C" int RSIM_SYM(rsim_launch_step)(const DModel* m, const DBatch* b, const float* actions, int n_sub, int flags, hipStream_t stream) {
  if (flags & RF_DEBUG) hipLaunchKernelGGL((k_step_dbg<RSIM_DIMS>), dim3(b->nenv ? b->nenv : b->B), dim3(64), 0, stream, *m, *b, actions, n_sub, flags);
  else hipLaunchKernelGGL((k_step<RSIM_DIMS>), dim3(b->nenv ? b->nenv : b->B), dim3(64), 0, stream, *m, *b, actions, n_sub, flags);
  return (int)hipGetLastError();
}
template __global__ void k_ctrl_reset<RSIM_DIMS>(DModel, DBatch, const unsigned char*);
template __global__ void k_prepare<RSIM_DIMS>(DModel, DBatch, int);
template __global__ void k_reset_obs<RSIM_DIMS>(DModel, DBatch);

extern "C" int RSIM_SYM(rsim_launch_reset_obs)(const DModel* m, const DBatch* b, hipStream_t stream) {
  hipLaunchKernelGGL((k_reset_obs<RSIM_DIMS>), dim3(b->nenv ? b->nenv : b->B), dim3(64), 0, stream, *m, *b);
  return (int)hipGetLastError();
}
extern "C" int RSIM_SYM(rsim_launch_ctrl_reset)(const DModel* m, const DBatch* b, const unsigned char* mask, hipStream_t stream) {
  hipLaunchKernelGGL((k_ctrl_reset<RSIM_DIMS>), dim3(b->B), dim3(64), 0, stream, *m, *b, mask);
  return (int)hipGetLastError();
}
extern "C" int RSIM_SYM(rsim_launch_prepare)(const DModel* m, const DBatch* b, int nblocks, int reset_only, hipStream_t stream) {
  hipLaunchKernelGGL((k_prepare<RSIM_DIMS>), dim3(nblocks), dim3(64), 0, stream, *m, *b, reset_only);
  return (int)hipGetLastError();
}
extern "C" int RSIM_SYM(rsim_cmem_bytes)(void) { return (int)((sizeof(Cmem0) + 255) & ~(size_t)255); }
// {nbody, njnt, nv, ncgeom, nsite, ncon, nefc, npair, articulated trees, tendon / equality rows (0 / 1)}
extern "C" int RSIM_SYM(rsim_limits)(int* lim) {
  const int dims[8] = {RSIM_DIMS};
  for (int i = 0; i < 8; i++) lim[i] = dims[i];
  lim[8] = Smem0::NROOT_; lim[9] = (Smem0::TENDONS_ ? 1 : 0) | (Smem0::NB_ > 32 ? 2 : 0) | (Smem0::JG_ ? 4 : 0) | (Smem0::MG_ ? 8 : 0) | (Smem0::CG_ ? 16 : 0) | (RSIM_FUSED_ENABLED ? 32 : 0);   // bit 5: the capacity tier above this configuration is compiled into its control-step kernel (rsim_limits_w)   // bit 2: the constraint Jacobian lives in DBatch.jg (NEFC * (NV + 1) floats per env)   // bit 1: two OSC arm parts
  return 0;
}
