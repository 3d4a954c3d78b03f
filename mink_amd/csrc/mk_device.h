// Device-side building blocks shared by the mink_amd kernels: small fixed-size
// vector / quaternion / SE3 helpers and the problem descriptors that the host
// flattens out of mk_model_desc / mk_task_desc / mk_limit_desc.
//
// Everything is float64 (float32 in the MK_F32 build of the fused kernels), one IK problem per lane; descriptors are wave-uniform, so
// descriptor reads become scalar loads.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// Math helpers are host+device so that the same source can be exercised lane by lane
// in a host-compiled unit-test build (tests/hostsim); the library itself only ever
// runs them inside kernels.
#define MK_DEV __host__ __device__ __forceinline__

constexpr int MK_MAX_CHAIN = 12;   // dofs with dense coupling in the fused path
constexpr int MK_MAX_NV = 64;      // tangent dimension supported by the fused path
constexpr int MK_MAX_NQ = 72;
constexpr int MK_MAX_POSTURE = 4;  // posture-like tasks per problem (fused path)
constexpr int MK_MAX_TERMS = 4;    // box-limit terms per dof
constexpr double MK_LIE_EPS = 1e-10;  // mink/lie/utils.py:4-8, float64
constexpr double MK_MINVAL = 1e-15;   // MuJoCo mjMINVAL
constexpr double MK_PI = 3.14159265358979323846;

enum { MK_TERM_CFG = 0, MK_TERM_VEL = 1 };
enum { MK_DOF_SCALAR = 0, MK_DOF_FREE_LIN = 1, MK_DOF_FREE_ROT = 2, MK_DOF_BALL = 3 };

// One 1-dof joint of the kinematic chain from the world to the task frame in canonical
// form: the joint moves about / along the z axis through the origin of its own frame
// (any axis and anchor are brought to this form on the host by a change of the fixed
// frames either side of the joint, like Denavit-Hartenberg frames), its qpos0 offset is
// folded into the fixed transform (pq, pp) that leads from the previous joint frame to
// this one.  (hinge, slide) is (1, 0) for a hinge, (0, 1) for a slide and (0, 0) for a
// padding slot, which then contributes a zero Jacobian column and no motion.
struct MkSegC {
  double pq[4];
  double pp[3];
  double hinge;
  double slide;
  double pad;
};

struct MkBoundTerm {
  int32_t kind;   // MK_TERM_CFG: a = lower, b = upper, gain; MK_TERM_VEL: a = vmax
  int32_t pad;
  double a, b, gain;
};

struct MkDof {
  int32_t qadr;
  int32_t kind;   // MK_DOF_*
  int32_t slot;   // index in the dense block, or -1
  int32_t nterm;
  MkBoundTerm term[MK_MAX_TERMS];
};

// Constants of one scalar dof, packed into one 64-byte record (a single scalar load):
// the sum of the posture-like tasks on it and its box limits.
struct MkDofC {
  double A, B, W;             // c_i = B * q_i - A,  H_ii += W   (sums over posture tasks)
  double lower, upper, gain;  // ConfigurationLimit (-inf / +inf / 1 when absent)
  double vmax;                // VelocityLimit (+inf when absent)
  int32_t qadr, dof;          // dof < 0: padding slot
};

struct MkPosture {
  double gain, lm;
  double cost[MK_MAX_NV];     // per dof (free-joint dofs forced to 0)
  double target[MK_MAX_NV];   // target qpos of the dof's (scalar) joint
};

struct MkFusedDesc {
  int32_t nq, nv, nc, nposture;
  int32_t nfree, target_off;
  int32_t needs_diag;   // some dof gets its Hessian diagonal from damping + mu alone (no task cost, zero Jacobian column)
  int32_t pad1;
  int32_t free_qadr[8];        // qpos address of each free joint (7 values)
  MkSegC segc[MK_MAX_CHAIN];
  int32_t seg_qadr[MK_MAX_CHAIN];     // qpos address of each chain slot (0 for padding slots)
  int32_t seg_dof[MK_MAX_CHAIN];      // dof of each chain slot (-1 for padding slots)
  double post_pos[3], post_quat[4];   // last joint frame -> task frame
  double cost[6], gain, lm;           // frame task
  MkPosture posture[MK_MAX_POSTURE];   // only read when a posture task has lm_damping > 0
  MkDof dof[MK_MAX_NV];
  int32_t nd, any_posture_lm;
  int32_t free_dofadr[8];
  MkDofC sdof[MK_MAX_CHAIN];           // chain dofs, by slot
  MkDofC ddof[MK_MAX_NV];              // scalar dofs outside the chain, increasing dof order
};

// ---- extension of the fused path: a second frame term and a collision prefilter ----
// A RelativeFrameTask on the same task frame whose root frame lies on the chain (or in the
// world) shares the chain pass: T_root,frame is the pose carried by the backward pass at
// the root's slot, and its Jacobian is the frame Jacobian with the columns of the root's
// ancestors set to zero.  CollisionAvoidanceLimit pairs are only screened here: a pair that
// a conservative bound puts beyond its detection distance contributes no row
// (mink/limits/collision_avoidance_limit.py:192-199).  A problem with a pair in range goes to
// the lane-per-problem kernel that builds the pairs' rows and solves the QP with them, and from
// there to the general kernel only when that one cannot finish it (more rows in range than it
// keeps, a pair without a closed form -- ellipsoid, cylinder, mesh --, no convergence of its dual
// iteration).
constexpr int MK_LANE_ROWS = 4;    // general rows per problem in the lane-per-problem solve
constexpr int MK_ROWS_MAX_NV = 48; // dofs of a model whose general rows the lane-per-problem solve holds
constexpr int MK_MAX_CBODY = 10;    // bodies carrying collision geoms
constexpr int MK_MAX_CSEG = 4;     // 1-dof joints between the chain and such a body
constexpr int MK_MAX_CGEOM = 24;
constexpr int MK_MAX_CPAIR = 40;
enum { MK_CB_WORLD = 0, MK_CB_TIP = 1, MK_CB_FREE = 2 };   // TIP: carried by a joint frame of the chain
constexpr int32_t MK_STATUS_PENDING = 1 << 30;   // in a status entry: the general kernel still has to solve it
// In an entry of the iterated entry's `iters` array between the two launches of a loop with an
// exit test: the lane-per-problem kernel left this problem's loop after the count in the low bits;
// the cooperative kernel takes it on from there.
constexpr int32_t MK_ITER_CONT_BIT = 1 << 29;

struct MkCollSeg {        // like MkSegC: fixed transform, then motion about / along z
  double pq[4], pp[3];
  double hinge, slide;
  int32_t qadr, dof;
};
struct MkCollBody {
  int32_t kind, nseg, free_qadr, free_dofadr;
  int32_t slot, pad0;     // MK_CB_TIP: the chain slot whose joint frame carries the body (nc - 1: the last one)
  int32_t g0, ng;         // its geoms: cgeom[g0 .. g0 + ng)
  MkCollSeg seg[MK_MAX_CSEG];
  double tq[4], tp[3];    // fixed transform behind the last joint (WORLD: the world pose)
  double reach;           // how far the body frame's origin can get from the frame that carries the body
                          // (the joints between them over their ranges); < 0: not bounded
};
struct MkCollGeom {
  int32_t body, type;     // index into cbody, MuJoCo geom type
  double lq[4], lp[3];    // pose in the body frame
  double size[3], rbound; // rbound: radius of a bounding sphere about bc
  double bc[3];           // centre of the bounding sphere in the geom frame (meshes; 0 otherwise)
  int32_t oslot, pad;     // planes and boxes: index among the oriented geoms (-1 otherwise)
  // coarse bound of the collision screen: every point of the geom lies within `reach` of its
  // anchor -- the origin of the frame behind the chain's last joint (bodies carried by it), the
  // free joint's position, or, for a geom fixed in the world, the sphere centre wc itself.
  // reach < 0: no such bound (a body on an inner link of the chain, an unlimited slide joint, a
  // plane that moves).  wn: the normal of a plane fixed in the world.
  double reach, wc[3], wn[3];
};
struct MkCollPair {
  int32_t g1, g2;         // indices into cgeom, ordered by geom type
  double d_detect;
  double d_min, gain, relax;   // row bound: gain (dist - d_min) / dt + relax, or relax at dist <= d_min
};
struct MkFusedExt {
  int32_t has_rel, rel_slot, rel_target_off, ncbody;
  int32_t ncgeom, ncpair, norient, pad1;   // norient: planes and boxes among the geoms
  double rel_cost[6], rel_gain, rel_lm;
  double rel_xq[4], rel_xp[3], pad2;   // root frame in the frame the pass carries at rel_slot
  MkCollBody cbody[MK_MAX_CBODY];
  MkCollGeom cgeom[MK_MAX_CGEOM];
  MkCollPair cpair[MK_MAX_CPAIR];
};

// ------------------------------------------------------------- scalar type ----
// The arithmetic type of the kernels.  The library's kernels compute in float64; a translation
// unit compiled with MK_F32 gets the same source in float32 (mk_fused32.hip): `real` is then a
// thin wrapper of float whose only purpose is that the float64 literals and descriptor fields of
// the shared source convert to it instead of promoting the arithmetic to float64.  Arrays in
// global memory (q, targets, v) stay float64 in both builds.
#if defined(MK_F32)
// (a namespace of its own: V3, Q4 ... have another layout than in the float64 translation units)
namespace mk_f32 {
struct real {
  float v;
  real() = default;
  MK_DEV constexpr real(double d) : v((float)d) {}
  MK_DEV constexpr real(float f) : v(f) {}
  MK_DEV constexpr real(int i) : v((float)i) {}
  MK_DEV explicit constexpr operator double() const { return (double)v; }
  MK_DEV explicit constexpr operator int() const { return (int)v; }
};
MK_DEV constexpr real operator+(real a, real b) { return real(a.v + b.v); }
MK_DEV constexpr real operator-(real a, real b) { return real(a.v - b.v); }
MK_DEV constexpr real operator*(real a, real b) { return real(a.v * b.v); }
MK_DEV constexpr real operator/(real a, real b) { return real(a.v / b.v); }
MK_DEV constexpr real operator-(real a) { return real(-a.v); }
MK_DEV constexpr real operator+(real a) { return a; }
MK_DEV real& operator+=(real& a, real b) { a.v += b.v; return a; }
MK_DEV real& operator-=(real& a, real b) { a.v -= b.v; return a; }
MK_DEV real& operator*=(real& a, real b) { a.v *= b.v; return a; }
MK_DEV constexpr bool operator<(real a, real b) { return a.v < b.v; }
MK_DEV constexpr bool operator>(real a, real b) { return a.v > b.v; }
MK_DEV constexpr bool operator<=(real a, real b) { return a.v <= b.v; }
MK_DEV constexpr bool operator>=(real a, real b) { return a.v >= b.v; }
MK_DEV constexpr bool operator==(real a, real b) { return a.v == b.v; }
MK_DEV constexpr bool operator!=(real a, real b) { return a.v != b.v; }
MK_DEV real fabs(real a) { return real(__builtin_fabsf(a.v)); }
MK_DEV real fmax(real a, real b) { return real(__builtin_fmaxf(a.v, b.v)); }
MK_DEV real fmin(real a, real b) { return real(__builtin_fminf(a.v, b.v)); }
MK_DEV real fma(real a, real b, real c) { return real(__builtin_fmaf(a.v, b.v, c.v)); }
MK_DEV real sqrt(real a) { return real(__builtin_sqrtf(a.v)); }
MK_DEV real rint(real a) { return real(__builtin_rintf(a.v)); }
MK_DEV real atan2(real a, real b) { return real(atan2f(a.v, b.v)); }
MK_DEV real asin(real a) { return real(asinf(a.v)); }
#define MK_RAW(a) ((a).v)
constexpr bool MK_IS_F32 = true;
constexpr float MK_TINY = 1e-30f;   // floor under arguments of rsqrt / rcp that may vanish
#else
typedef double real;
#define MK_RAW(a) (a)
constexpr bool MK_IS_F32 = false;
constexpr double MK_TINY = 1e-300;
#endif

// ------------------------------------------------------------------ math ----
// Reciprocal / reciprocal square root for normal-range positive-scale arguments:
// hardware seed (v_rcp_f64 / v_rsq_f64) plus two Newton steps, ~1 ulp, without the
// scaling and fix-up sequence of a full IEEE division (the operands here are norms,
// Cholesky pivots, dt and Hessian diagonals, never denormal or infinite).
#if defined(MK_F32)
// float32: the hardware reciprocal and reciprocal square root are good to 1 ulp as they are.
MK_DEV real mk_rcp(real x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return real(__builtin_amdgcn_rcpf(x.v));
#else
  return real(1.0f / x.v);
#endif
}
MK_DEV real mk_rsqrt(real x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return real(__builtin_amdgcn_rsqf(x.v));
#else
  return real(1.0f / __builtin_sqrtf(x.v));
#endif
}
// sin and cos for |x| < 2^15: two-constant reduction by pi/2, degree-9 / degree-8 kernels on
// [-pi/4, pi/4] (the float32 minimax coefficients of the usual sinf / cosf kernels); about 1 ulp
// for the joint angles and rotation-vector norms that reach it, NaN beyond the range.
MK_DEV void mk_sincos(real xr, real* sn, real* cs) {
  const float x = xr.v;
  const float k = __builtin_rintf(x * 0.636619772367581343f);
  float r = __builtin_fmaf(-k, 1.57079637050628662109375f, x);
  r = __builtin_fmaf(-k, -4.37113900018624283e-8f, r);
  const float z = r * r;
  float ps = __builtin_fmaf(z, 2.7183114939898219064e-6f, -1.98393348360966317347e-4f);
  ps = __builtin_fmaf(z, ps, 8.3333293858894631756e-3f);
  ps = __builtin_fmaf(z, ps, -1.66666666416265235595e-1f);
  const float sr = __builtin_fmaf(r * z, ps, r);
  float pc = __builtin_fmaf(z, 2.43904487962774090654e-5f, -1.38867637746099294692e-3f);
  pc = __builtin_fmaf(z, pc, 4.16666233237390631894e-2f);
  pc = __builtin_fmaf(z, pc, -4.99999997251031003120e-1f);
  const float cr = __builtin_fmaf(z, pc, 1.0f);
  const bool ok = __builtin_fabsf(x) < 32768.0f;
  const int q = ok ? (int)k : 0;
  float s = (q & 1) ? cr : sr;
  float c = (q & 1) ? sr : cr;
  s = (q & 2) ? -s : s;
  c = ((q + 1) & 2) ? -c : c;
  *sn = real(ok ? s : NAN);
  *cs = real(ok ? c : NAN);
}
#else
MK_DEV real mk_rcp(real x) {
#if defined(__HIP_DEVICE_COMPILE__)
  real r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
#else
  return 1.0 / x;
#endif
}
MK_DEV real mk_rsqrt(real x) {
#if defined(__HIP_DEVICE_COMPILE__)
  real y = __builtin_amdgcn_rsq(x);
  const real hx = 0.5 * x;
  y = fma(fma(-hx * y, y, 0.5), y, y);
  y = fma(fma(-hx * y, y, 0.5), y, y);
  return y;
#else
  return 1.0 / sqrt(x);
#endif
}
// sin and cos of one argument for |x| < 2^30 (joint angles, rotation-vector norms):
// two-constant Cody-Waite reduction by pi/2 with fused multiply-adds, then the classic
// degree-13 / degree-14 minimax kernels on [-pi/4, pi/4]; < 1 ulp.  Larger or non-finite
// arguments give NaN, which the kernels report as non-finite input.
MK_DEV void mk_sincos(real x, real* sn, real* cs) {
  const real k = rint(x * 0.63661977236758134308);
  real r = fma(-k, 1.57079632679489655800e+00, x);
  r = fma(-k, 6.12323399573676603587e-17, r);
  const real z = r * r;
  real ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(z, ps, 2.75573137070700676789e-06);
  ps = fma(z, ps, -1.98412698298579493134e-04);
  ps = fma(z, ps, 8.33333333332248946124e-03);
  ps = fma(z, ps, -1.66666666666666324348e-01);
  const real sr = fma(r * z, ps, r);
  real pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(z, pc, -2.75573143513906633035e-07);
  pc = fma(z, pc, 2.48015872894767294178e-05);
  pc = fma(z, pc, -1.38888888888741095749e-03);
  pc = fma(z, pc, 4.16666666666666019037e-02);
  const real cr = fma(z * z, pc, fma(-0.5, z, 1.0));
  const bool ok = fabs(x) < 1073741824.0;
  const int q = ok ? (int)k : 0;
  real s = (q & 1) ? cr : sr;
  real c = (q & 1) ? sr : cr;
  s = (q & 2) ? -s : s;
  c = ((q + 1) & 2) ? -c : c;
  *sn = ok ? s : NAN;
  *cs = ok ? c : NAN;
}
#endif

// Keeps the instruction scheduler from moving anything across this point.  The kernels
// are long straight-line blocks full of scalar loads of per-slot constants; unfenced, the
// scheduler hoists all of them to the top and spills scalar registers by the hundred.
// With a fence per slot and the next slot's constants requested one slot ahead, the loads
// are in flight during the arithmetic of the current slot.
#if defined(__HIP_DEVICE_COMPILE__)
#define MK_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define MK_SCHED_FENCE() ((void)0)
#endif

// Orders the arithmetic that produces / consumes the given values relative to
// MK_SCHED_FENCE (a fence alone only orders memory operations while the block is first
// linearised; arithmetic would still drift away from the loads it was meant to overlap).
#if defined(__HIP_DEVICE_COMPILE__)
#define MK_PIN4(a, b, c, d) asm volatile("" : "+v"(MK_RAW(a)), "+v"(MK_RAW(b)), "+v"(MK_RAW(c)), "+v"(MK_RAW(d)))
#define MK_PIN3(a, b, c) asm volatile("" : "+v"(MK_RAW(a)), "+v"(MK_RAW(b)), "+v"(MK_RAW(c)))
#else
#define MK_PIN4(a, b, c, d) ((void)0)
#define MK_PIN3(a, b, c) ((void)0)
#endif

// A wave-uniform zero the optimiser cannot see through.  Added to the descriptor pointer
// once per iteration of the solve loop: otherwise every per-slot constant is hoisted out
// of the loop as loop-invariant, hundreds of scalar registers are spilled to vector lanes
// and read back with v_readlane inside the loop.
MK_DEV int mk_opaque_zero() {
  int z = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+s"(z));
#endif
  return z;
}

// True when no lane of the wave still has `pending` set.
MK_DEV bool mk_wave_none(bool pending) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __ballot(pending) == 0ull;
#else
  return !pending;
#endif
}

struct V3 { real x, y, z; };
struct Q4 { real w, x, y, z; };
struct M3 { real m[9]; };  // row-major

MK_DEV V3 v3(real x, real y, real z) { return V3{x, y, z}; }
MK_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
MK_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
MK_DEV V3 operator*(real s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
MK_DEV V3 neg(V3 a) { return V3{-a.x, -a.y, -a.z}; }
MK_DEV real dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
MK_DEV V3 cross(V3 a, V3 b) {
  return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// Hamilton product (mju_mulQuat).
MK_DEV Q4 qmul(Q4 a, Q4 b) {
  return Q4{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
            a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
            a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
MK_DEV Q4 qconj(Q4 a) { return Q4{a.w, -a.x, -a.y, -a.z}; }
// Rotation of v by the unit quaternion q: v + 2 u x (u x v + w v).
MK_DEV V3 qrot(Q4 q, V3 v) {
  const V3 u = V3{q.x, q.y, q.z};
  const V3 c = cross(u, v) + q.w * v;
  const V3 d = cross(u, c);
  return V3{fma(2.0, d.x, v.x), fma(2.0, d.y, v.y), fma(2.0, d.z, v.z)};
}

// mju_normalize4: a vanishing quaternion maps to identity.
MK_DEV Q4 qnormalize(Q4 q) {
  const real n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
  if (n2 < MK_MINVAL * MK_MINVAL) return Q4{1.0, 0.0, 0.0, 0.0};
  const real r = mk_rsqrt(n2);
  return Q4{q.w * r, q.x * r, q.y * r, q.z * r};
}

// mju_quat2Mat.
MK_DEV M3 q2m(Q4 q) {
  real q00 = q.w * q.w, q11 = q.x * q.x, q22 = q.y * q.y, q33 = q.z * q.z;
  real q01 = q.w * q.x, q02 = q.w * q.y, q03 = q.w * q.z;
  real q12 = q.x * q.y, q13 = q.x * q.z, q23 = q.y * q.z;
  M3 r;
  r.m[0] = q00 + q11 - q22 - q33; r.m[1] = 2 * (q12 - q03); r.m[2] = 2 * (q13 + q02);
  r.m[3] = 2 * (q12 + q03); r.m[4] = q00 - q11 + q22 - q33; r.m[5] = 2 * (q23 - q01);
  r.m[6] = 2 * (q13 - q02); r.m[7] = 2 * (q23 + q01); r.m[8] = q00 - q11 - q22 + q33;
  return r;
}
MK_DEV V3 mul(const M3& r, V3 v) {
  return V3{r.m[0] * v.x + r.m[1] * v.y + r.m[2] * v.z,
            r.m[3] * v.x + r.m[4] * v.y + r.m[5] * v.z,
            r.m[6] * v.x + r.m[7] * v.y + r.m[8] * v.z};
}
MK_DEV V3 mulT(const M3& r, V3 v) {
  return V3{r.m[0] * v.x + r.m[3] * v.y + r.m[6] * v.z,
            r.m[1] * v.x + r.m[4] * v.y + r.m[7] * v.z,
            r.m[2] * v.x + r.m[5] * v.y + r.m[8] * v.z};
}
MK_DEV M3 mmul(const M3& a, const M3& b) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      r.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
  return r;
}
// I + a*[w]x + b*[w]x^2
MK_DEV M3 rodrigues_like(V3 w, real a, real b) {
  real xx = w.x * w.x, yy = w.y * w.y, zz = w.z * w.z;
  real xy = w.x * w.y, xz = w.x * w.z, yz = w.y * w.z;
  M3 r;
  r.m[0] = 1.0 - b * (yy + zz); r.m[1] = -a * w.z + b * xy; r.m[2] = a * w.y + b * xz;
  r.m[3] = a * w.z + b * xy; r.m[4] = 1.0 - b * (xx + zz); r.m[5] = -a * w.x + b * yz;
  r.m[6] = -a * w.y + b * xz; r.m[7] = a * w.x + b * yz; r.m[8] = 1.0 - b * (xx + yy);
  return r;
}
MK_DEV M3 skew(V3 w) {
  M3 r;
  r.m[0] = 0; r.m[1] = -w.z; r.m[2] = w.y;
  r.m[3] = w.z; r.m[4] = 0; r.m[5] = -w.x;
  r.m[6] = -w.y; r.m[7] = w.x; r.m[8] = 0;
  return r;
}
MK_DEV M3 mtrans(const M3& a) {
  M3 r;
  r.m[0] = a.m[0]; r.m[1] = a.m[3]; r.m[2] = a.m[6];
  r.m[3] = a.m[1]; r.m[4] = a.m[4]; r.m[5] = a.m[7];
  r.m[6] = a.m[2]; r.m[7] = a.m[5]; r.m[8] = a.m[8];
  return r;
}

// SO3.log of a unit quaternion (mink/lie/so3.py:176-191), safe for w < 0.  The three
// cases of the reference (small angle, |w| ~ 0, general) are all evaluated and selected,
// so that the code stays straight-line.
MK_DEV V3 so3_log(Q4 q) {
  const real n2 = q.x * q.x + q.y * q.y + q.z * q.z;
  const real iw = mk_rcp(q.w);
  const real f_small = 2.0 * iw - 2.0 / 3.0 * n2 * (iw * iw * iw);
  const real n2s = fmax(n2, MK_TINY);
  const real rn = mk_rsqrt(n2s), n = n2s * rn;
  const real f_pi = (q.w > 0.0 ? MK_PI : -MK_PI) * rn;
  const real f_gen = 2.0 * atan2(q.w < 0 ? -n : n, fabs(q.w)) * rn;
  const real f = (n2 < MK_LIE_EPS) ? f_small : ((fabs(q.w) < MK_LIE_EPS) ? f_pi : f_gen);
  return V3{f * q.x, f * q.y, f * q.z};
}

// SO3.exp of a rotation vector (mink/lie/so3.py:158-173): unit quaternion, with the
// reference's series below its small-angle switch.
MK_DEV Q4 so3_exp(V3 w) {
  const real t2 = dot(w, w);
  const bool small = t2 < MK_LIE_EPS;
  const real th = sqrt(small ? 1.0 : t2);
  real sn, cs;
  mk_sincos(0.5 * th, &sn, &cs);
  const real re = small ? 1.0 - t2 / 8.0 + t2 * t2 / 384.0 : cs;
  const real im = small ? 0.5 - t2 / 48.0 + t2 * t2 / 3840.0 : sn / th;
  return Q4{re, im * w.x, im * w.y, im * w.z};
}

// Roll, pitch, yaw of a unit quaternion (mink/lie/so3.py:116-133), i.e. the extrinsic
// x-y-z Euler angles scipy calls as_euler("xyz").
MK_DEV V3 quat_rpy(Q4 q) {
  return V3{atan2(2.0 * (q.w * q.x + q.y * q.z), 1.0 - 2.0 * (q.x * q.x + q.y * q.y)),
            asin(2.0 * (q.w * q.y - q.z * q.x)),
            atan2(2.0 * (q.w * q.z + q.x * q.y), 1.0 - 2.0 * (q.y * q.y + q.z * q.z))};
}

struct Twist { V3 v, w; };

// (1 - (theta/2) cot(theta/2)) / theta^2 as its series in t2 = theta^2 (t2 < 1; next term
// 1e-10 t2^5).  The float32 build uses it where the closed forms of SE3.log and SO3.ljacinv --
// which are this same function -- lose their leading digits to cancellation.
MK_DEV real mk_half_cot_series(real t2) {
  real p = fma(t2, 1.0 / 47900160.0, 1.0 / 1209600.0);
  p = fma(t2, p, 1.0 / 30240.0);
  p = fma(t2, p, 1.0 / 720.0);
  return fma(t2, p, 1.0 / 12.0);
}

// SE3.log (mink/lie/se3.py:159-185) of the transform (q, t).
MK_DEV Twist se3_log(Q4 q, V3 t) {
  const V3 w = so3_log(q);
  const real t2 = dot(w, w);
  const real t2s = fmax(t2, MK_TINY);
  const real rth = mk_rsqrt(t2s), th = t2s * rth;
  real s, c;
  mk_sincos(0.5 * th, &s, &c);
  const real coef_gen = (1.0 - th * c * mk_rcp(2.0 * s)) * (rth * rth);
  real coef = (t2 < MK_LIE_EPS) ? 1.0 / 12.0 : coef_gen;
  if (MK_IS_F32 && t2 < 1.0) coef = mk_half_cot_series(t2);   // the closed form cancels in float32
  M3 Vinv = rodrigues_like(w, -0.5, coef);
  return Twist{mul(Vinv, t), w};
}

// The Q block of the SE3 Jacobians (mink/lie/se3.py:222-249) for theta^2 = |w|^2 above the
// small-angle switch, given rth = 1/theta, th = theta, s = sin(theta), c = cos(theta).
//
// Q is the reference's sum of products of the skew matrices V = [v]x, W = [w]x,
//   Q = V/2 + b (WV + VW + WVW) - c (VWW - (VWW)' - 3 WVW) + d (WVWW + WWVW),
// written out with [a]x[b]x = b a' - (a.b) I, s = v.w, n = v x w:
//   WV + VW = v w' + w v' - 2 s I,   WVW = -s W,   VWW - (VWW)' = [n x w]x - 2 s W,
//   WVWW + WWVW = -2 s (w w' - |w|^2 I),
// i.e. a skew part [v/2 - (b + c) s w - c (n x w)]x plus a symmetric part.
MK_DEV M3 se3_Q_closed(Twist xi, real t2, real rth, real th, real s, real c) {
  const real it2 = rth * rth;
  const real ith = rth;
  real Bc = (th - s) * it2 * ith;
  real Cc = (1.0 - 0.5 * t2 - c) * it2 * it2;
  real Dc = (2.0 * th - 3.0 * s + th * c) * (0.5 * it2 * it2 * ith);
  if (MK_IS_F32 && t2 < 1.0) {
    // float32: the numerators above are differences of O(1) terms that leave theta^3 .. theta^5;
    // their series in t2 instead (next terms below 1e-8)
    Bc = fma(t2, fma(t2, fma(t2, -1.0 / 362880.0, 1.0 / 5040.0), -1.0 / 120.0), 1.0 / 6.0);
    Cc = fma(t2, fma(t2, fma(t2, 1.0 / 3628800.0, -1.0 / 40320.0), 1.0 / 720.0), -1.0 / 24.0);
    Dc = 0.5 * fma(t2, fma(t2, fma(t2, -1.0 / 4989600.0, 1.0 / 60480.0), -1.0 / 1260.0), 1.0 / 60.0);
  }
  const V3 v = xi.v, w = xi.w;
  const real sv = dot(v, w);
  const V3 n = cross(v, w);
  const V3 k = 0.5 * v - ((Bc + Cc) * sv) * w - Cc * cross(n, w);   // skew part [k]x
  const real d2 = -2.0 * sv * Dc;
  const real di = -2.0 * sv * Bc - d2 * t2;                         // multiple of I
  M3 Q;
  Q.m[0] = di + 2.0 * Bc * v.x * w.x + d2 * w.x * w.x;
  Q.m[4] = di + 2.0 * Bc * v.y * w.y + d2 * w.y * w.y;
  Q.m[8] = di + 2.0 * Bc * v.z * w.z + d2 * w.z * w.z;
  const real sxy = Bc * (v.x * w.y + v.y * w.x) + d2 * w.x * w.y;
  const real sxz = Bc * (v.x * w.z + v.z * w.x) + d2 * w.x * w.z;
  const real syz = Bc * (v.y * w.z + v.z * w.y) + d2 * w.y * w.z;
  Q.m[1] = sxy - k.z; Q.m[3] = sxy + k.z;
  Q.m[2] = sxz + k.y; Q.m[6] = sxz - k.y;
  Q.m[5] = syz - k.x; Q.m[7] = syz + k.x;
  return Q;
}

// Blocks of SE3.ljacinv(xi) = [[Ji, B], [0, Ji]] with B = -Ji Q Ji
// (mink/lie/se3.py:210-249, mink/lie/so3.py:214-226); Ji = I, B = 0 when the reference
// returns exactly I6 (theta^2 below the switch).
MK_DEV void se3_ljacinv_blocks(Twist xi, M3& Ji, M3& B) {
  const real t2 = dot(xi.w, xi.w);
  // (float32: no switch to the identity -- the series below are good down to theta = 0, and the
  // reference's float32 switch, theta^2 < 1e-5, would cost three digits of the Jacobian)
  if (!MK_IS_F32 && t2 < MK_LIE_EPS) {
#pragma unroll
    for (int i = 0; i < 9; ++i) { Ji.m[i] = (i % 4 == 0) ? 1.0 : 0.0; B.m[i] = 0.0; }
    return;
  }
  const real t2s = MK_IS_F32 ? fmax(t2, MK_TINY) : t2;
  const real rth = mk_rsqrt(t2s), th = t2s * rth;
  real s, c;
  mk_sincos(th, &s, &c);
  // SO3.ljacinv switches on theta (not theta^2); theta >= 1e-5 here, so closed form.
  const real it2 = rth * rth;
  real A = it2 * (1.0 - (th * s * mk_rcp(2.0 * (1.0 - c))));
  if (MK_IS_F32 && t2 < 1.0) A = mk_half_cot_series(t2);
  Ji = rodrigues_like(xi.w, -0.5, A);
  const M3 Q = se3_Q_closed(xi, t2, rth, th, s, c);
  M3 JQ = mmul(Ji, Q);
  B = mmul(JQ, Ji);
#pragma unroll
  for (int i = 0; i < 9; ++i) B.m[i] = -B.m[i];
}

// Frame-task error e = log(T_wb^-1 T_wt) (mink/tasks/frame_task.py:95-122) of the frame at
// (Q, t, R = mat(Q)) for the target (Qt, pt).
MK_DEV Twist frame_error(Q4 Q, V3 t, const M3& R, Q4 Qt, V3 pt) { return se3_log(qmul(qconj(Q), Qt), mulT(R, pt - t)); }

// LinearWaypointReach.step (envs/robot_utils.py:131-160) for the current pose (Qc, tc) and
// the goal (Qg, pg): the commanded pose (Qt, pt) -- a step of at most pos_step along the
// straight line to the goal position (the goal itself once within pos_thr), half-way along
// the geodesic to the goal orientation -- and whether the goal counts as reached (position
// within pos_thr and the norm of the x-y-z Euler angles of goal * current^-1 below rot_thr).
MK_DEV bool waypoint_step(Q4 Qc, V3 tc, Q4 Qg, V3 pg, real pos_thr, real pos_step, real rot_thr, Q4& Qt,
                          V3& pt) {
  const V3 delta = pg - tc;
  const real dist = sqrt(dot(delta, delta));
  const bool pos_reached = dist < pos_thr;
  const real step = fmin(dist, pos_step);
  const V3 moved = tc + (step / fmax(dist, MK_TINY)) * delta;
  pt = pos_reached ? pg : moved;
  const V3 half = 0.5 * so3_log(qmul(qconj(Qc), Qg));
  Qt = qmul(Qc, so3_exp(half));
  const V3 eul = quat_rpy(qmul(Qg, qconj(Qc)));
  return pos_reached && sqrt(dot(eul, eul)) < rot_thr;
}

#if defined(MK_F32)
}  // namespace mk_f32
using namespace mk_f32;
#endif
