// Cooperative differential-IK kernels: one IK problem per 16-lane row of a wave, four
// problems per wave.  The small-batch / low-latency form of the fused path (mk_fused.hip):
// there one lane runs a whole solve_ik as ~6500 sequential instructions, which is the right
// shape when the batch fills every lane of the machine and the wrong one for the callers'
// real batches (one to a few thousand problems, interactive_scripts/teleop_mouse.py:99-125,
// envs/mj_env.py:253-282) and for the last stragglers of a loop with an exit test.  Here the
// 16 lanes of a row share one problem:
//   lane k owns chain joint k: its sin / cos and local transform are formed in parallel and a
//     log-step suffix scan over the row composes the chain (mj_kinematics); the lane then
//     holds the pose of the task frame in its joint frame, i.e. column k of the body-frame
//     Jacobian (mj_jacSite + mink/configuration.py:143-153),
//   the frame-task error and the SE3 Jacobian blocks are evaluated redundantly by every lane
//     (mink/tasks/frame_task.py:95-146),
//   lane k owns row k of H and entry k of c, lo, hi (mink/tasks/task.py:125-138,
//     mink/solve_ik.py:13-22, mink/limits/configuration_limit.py:98-124, velocity_limit.py:96-101),
//   the box QP is the active-set method of mk_fused.hip's box_qp with the factorisation done
//     as a Gauss-Jordan elimination across the row (pivot rows travel by row broadcasts),
//   the dofs outside the chain are clamps, one or more per lane.
// Cross-lane traffic is DPP row operations only (row_newbcast, row_shl) -- no LDS, no
// barriers -- so the four problems of a wave never interact, and a problem's arithmetic does
// not depend on the row or wave it runs in.
//
// Every function here is device code; tests/hostsim builds the same source for the host with
// the row operations emulated by 16 cooperating fibers (MK_COOP_HOSTSIM).
#ifndef MK_COOP_FUNCS_INCLUDED
#define MK_COOP_FUNCS_INCLUDED
#include "mk_device.h"

#include <type_traits>

#if defined(MK_COOP_HOSTSIM)
#define MK_CO __host__ __device__ inline
// host emulation of the row operations (tests/hostsim/coop_sim.h)
int co_sim_lane();
const double* co_sim_exchange(double v);
#else
#define MK_CO __device__ __forceinline__
#endif

namespace mkcoop {

constexpr int kRow = 16;   // lanes per problem (one DPP row)

// ---- compile-time loops (DPP controls are instruction immediates) ----
template <int I, int N, class F>
MK_CO void co_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    co_for<I + 1, N>(f);
  }
}

// ---- row operations ----
#if defined(__HIP_DEVICE_COMPILE__)
MK_CO int co_lane() { return threadIdx.x & (kRow - 1); }
// DPP move of a double whose source lane always exists (no fill value, so no copy of one)
template <int CTRL>
MK_CO double co_dpp(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// value of lane J of the row
template <int J>
MK_CO double co_bcast(double v) { return co_dpp<0x150 + J>(v); }             // row_newbcast:J
// value of lane l + D of the row; beyond the row 0 (ONE == false) or 1 (ONE == true)
template <int D, bool ONE>
MK_CO double co_up(double v) {                                                // row_shl:D
  // lanes without a source lane take zero bits (bound_ctrl), but for the high word of 1.0
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x100 + D, 0xf, 0xf, true);
  const int hi = ONE ? __builtin_amdgcn_update_dpp(0x3ff00000, __double2hiint(v), 0x100 + D, 0xf, 0xf, false)
                     : __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x100 + D, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// value of lane l ^ M of the row (M a power of two)
template <int M>
MK_CO double co_xor(double v) {
  if constexpr (M == 1) return co_dpp<0xB1>(v);          // quad_perm [1,0,3,2]
  else if constexpr (M == 2) return co_dpp<0x4E>(v);     // quad_perm [2,3,0,1]
  else return __shfl_xor(v, M, kRow);
}
template <int M>
MK_CO int co_xor_i(int v) { return __shfl_xor(v, M, kRow); }
// value of lane `src` (the same for every lane of the row) of the row
MK_CO double co_pick(double v, int src) { return __shfl(v, src, kRow); }
// the row's 16 predicate bits
MK_CO uint32_t co_ballot(bool p) {
  const uint64_t m = __ballot(p);
  return (uint32_t)(m >> (threadIdx.x & 48)) & 0xffffu;
}
#elif defined(MK_COOP_HOSTSIM)
MK_CO int co_lane() { return co_sim_lane(); }
template <int J>
MK_CO double co_bcast(double v) { return co_sim_exchange(v)[J]; }
template <int D, bool ONE>
MK_CO double co_up(double v) {
  const int l = co_sim_lane();
  const double* x = co_sim_exchange(v);
  return l + D < kRow ? x[l + D] : (ONE ? 1.0 : 0.0);
}
template <int M>
MK_CO double co_xor(double v) { const int l = co_sim_lane(); return co_sim_exchange(v)[l ^ M]; }
template <int M>
MK_CO int co_xor_i(int v) { const int l = co_sim_lane(); return (int)co_sim_exchange((double)v)[l ^ M]; }
MK_CO double co_pick(double v, int src) { return co_sim_exchange(v)[src & (kRow - 1)]; }
MK_CO uint32_t co_ballot(bool p) {
  const double* x = co_sim_exchange(p ? 1.0 : 0.0);
  uint32_t m = 0u;
  for (int i = 0; i < kRow; ++i) m |= (x[i] != 0.0 ? 1u : 0u) << i;
  return m;
}
#else
// host pass of the library build: declarations only (device code is never called on the host)
__device__ int co_lane();
template <int J> __device__ double co_bcast(double v);
template <int D, bool ONE> __device__ double co_up(double v);
template <int M> __device__ double co_xor(double v);
template <int M> __device__ int co_xor_i(int v);
__device__ double co_pick(double v, int src);
__device__ uint32_t co_ballot(bool p);
#endif

// ---- compound row operations: the broadcast fused into the multiply-add ----
// gfx950 has 64-bit DPP forms of v_fmac_f64 and v_mov_b64 for row_newbcast, so "multiply my
// value by lane J's and accumulate" is ONE instruction where a broadcast through 32-bit DPP
// moves costs three.  The compiler has no builtin for them; they are issued from inline
// assembly in blocks that start with `s_nop 1`: a DPP instruction must not read a register
// that a vector instruction wrote within the previous two issue slots, and the compiler pads
// nothing inside or in front of an asm statement.  Within a block no instruction reads through
// DPP what an earlier one of the block wrote, except where a nop is written between them.
// Read-write operands are early-clobber so that no plain input shares their register.
// MK_COOP_PLAIN_DPP builds the same operations from co_bcast (A/B timing, and the host emulation).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MK_COOP_PLAIN_DPP)
#define MK_CO_ASM 1
#define MK_CO_DPPT " row_mask:0xf bank_mask:0xf\n\t"
#define MK_CO_ELIM(n) "v_fmac_f64_dpp %[a" #n "], %[a" #n "], %[f] row_newbcast:%[j]" MK_CO_DPPT
#define MK_CO_DOT(n) "v_fmac_f64_dpp %[y], %[x], %[h" #n "] row_newbcast:%[k" #n "]" MK_CO_DPPT
#define MK_CO_GRAM(n) "v_fmac_f64_dpp %[h], %[w" #n "], %[w" #n "] row_newbcast:%[k]" MK_CO_DPPT
#define MK_CO_MOV(n) "v_mov_b64_dpp %[o" #n "], %[i" #n "] row_newbcast:%[j]" MK_CO_DPPT
#define MK_CO_NEXT "v_mov_b64_dpp %[nx], %[a0] row_newbcast:%[jn]" MK_CO_DPPT
#define MK_CO_CHAIN(n) "s_nop 1\n\tv_fmac_f64_dpp %[r], %[r], %[f" #n "] row_newbcast:%[k" #n "]" MK_CO_DPPT
#endif

// a[k] += (a[k] of lane J) * f for k in [K0, K1): the elimination of one pivot on this lane's row.
template <int J, int K0, int K1, int M>
MK_CO void co_elim(double (&a)[M], double f) {
  if constexpr (K0 < K1) {
#if defined(MK_CO_ASM)
    constexpr int N = K1 - K0 < 4 ? K1 - K0 : 4;
    if constexpr (N == 4)
      asm volatile("s_nop 1\n\t" MK_CO_ELIM(0) MK_CO_ELIM(1) MK_CO_ELIM(2) MK_CO_ELIM(3)
                   : [a0] "+&v"(a[K0]), [a1] "+&v"(a[K0 + 1]), [a2] "+&v"(a[K0 + 2]), [a3] "+&v"(a[K0 + 3])
                   : [f] "v"(f), [j] "n"(J));
    else if constexpr (N == 3)
      asm volatile("s_nop 1\n\t" MK_CO_ELIM(0) MK_CO_ELIM(1) MK_CO_ELIM(2)
                   : [a0] "+&v"(a[K0]), [a1] "+&v"(a[K0 + 1]), [a2] "+&v"(a[K0 + 2])
                   : [f] "v"(f), [j] "n"(J));
    else if constexpr (N == 2)
      asm volatile("s_nop 1\n\t" MK_CO_ELIM(0) MK_CO_ELIM(1)
                   : [a0] "+&v"(a[K0]), [a1] "+&v"(a[K0 + 1])
                   : [f] "v"(f), [j] "n"(J));
    else
      asm volatile("s_nop 1\n\t" MK_CO_ELIM(0) : [a0] "+&v"(a[K0]) : [f] "v"(f), [j] "n"(J));
    co_elim<J, K0 + N, K1>(a, f);
#else
    co_for<K0, K1>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      a[k] = fma(co_bcast<J>(a[k]), f, a[k]);
    });
#endif
  }
}

// One pivot of the elimination as a single block: a[K0 + i] += (a[K0 + i] of lane J) * f for i in
// [0, N), then `next` = a[K0] of lane J + 1, the next pivot (a[K0] is written first, so the
// broadcast that reads it comes at least two instructions later, or after a nop).
template <int J, int K0, int N, int M>
MK_CO void co_pivot(double (&a)[M], double f, double& next) {
#if defined(MK_CO_ASM)
  static_assert(N >= 1 && N <= 12, "row length");
  if constexpr (N == 1)
    asm volatile("s_nop 1\n\t" MK_CO_ELIM(0) "s_nop 1\n\t" MK_CO_NEXT
                 : [a0] "+&v"(a[K0 + 0]), [nx] "=&v"(next)
                 : [f] "v"(f), [j] "n"(J), [jn] "n"(J + 1));
  if constexpr (N == 2)
    asm volatile("s_nop 1\n\t" MK_CO_ELIM(0) MK_CO_ELIM(1) "s_nop 1\n\t" MK_CO_NEXT
                 : [a0] "+&v"(a[K0 + 0]), [a1] "+&v"(a[K0 + 1]), [nx] "=&v"(next)
                 : [f] "v"(f), [j] "n"(J), [jn] "n"(J + 1));
  if constexpr (N == 3)
    asm volatile("s_nop 1\n\t" MK_CO_ELIM(0) MK_CO_ELIM(1) MK_CO_ELIM(2) MK_CO_NEXT
                 : [a0] "+&v"(a[K0 + 0]), [a1] "+&v"(a[K0 + 1]), [a2] "+&v"(a[K0 + 2]), [nx] "=&v"(next)
                 : [f] "v"(f), [j] "n"(J), [jn] "n"(J + 1));
  if constexpr (N == 4)
    asm volatile("s_nop 1\n\t" MK_CO_ELIM(0) MK_CO_ELIM(1) MK_CO_ELIM(2) MK_CO_ELIM(3) MK_CO_NEXT
                 : [a0] "+&v"(a[K0 + 0]), [a1] "+&v"(a[K0 + 1]), [a2] "+&v"(a[K0 + 2]), [a3] "+&v"(a[K0 + 3]), [nx] "=&v"(next)
                 : [f] "v"(f), [j] "n"(J), [jn] "n"(J + 1));
  if constexpr (N == 5)
    asm volatile("s_nop 1\n\t" MK_CO_ELIM(0) MK_CO_ELIM(1) MK_CO_ELIM(2) MK_CO_ELIM(3) MK_CO_ELIM(4) MK_CO_NEXT
                 : [a0] "+&v"(a[K0 + 0]), [a1] "+&v"(a[K0 + 1]), [a2] "+&v"(a[K0 + 2]), [a3] "+&v"(a[K0 + 3]), [a4] "+&v"(a[K0 + 4]), [nx] "=&v"(next)
                 : [f] "v"(f), [j] "n"(J), [jn] "n"(J + 1));
  if constexpr (N == 6)
    asm volatile("s_nop 1\n\t" MK_CO_ELIM(0) MK_CO_ELIM(1) MK_CO_ELIM(2) MK_CO_ELIM(3) MK_CO_ELIM(4) MK_CO_ELIM(5) MK_CO_NEXT
                 : [a0] "+&v"(a[K0 + 0]), [a1] "+&v"(a[K0 + 1]), [a2] "+&v"(a[K0 + 2]), [a3] "+&v"(a[K0 + 3]), [a4] "+&v"(a[K0 + 4]), [a5] "+&v"(a[K0 + 5]), [nx] "=&v"(next)
                 : [f] "v"(f), [j] "n"(J), [jn] "n"(J + 1));
  if constexpr (N == 7)
    asm volatile("s_nop 1\n\t" MK_CO_ELIM(0) MK_CO_ELIM(1) MK_CO_ELIM(2) MK_CO_ELIM(3) MK_CO_ELIM(4) MK_CO_ELIM(5) MK_CO_ELIM(6) MK_CO_NEXT
                 : [a0] "+&v"(a[K0 + 0]), [a1] "+&v"(a[K0 + 1]), [a2] "+&v"(a[K0 + 2]), [a3] "+&v"(a[K0 + 3]), [a4] "+&v"(a[K0 + 4]), [a5] "+&v"(a[K0 + 5]), [a6] "+&v"(a[K0 + 6]), [nx] "=&v"(next)
                 : [f] "v"(f), [j] "n"(J), [jn] "n"(J + 1));
  if constexpr (N == 8)
    asm volatile("s_nop 1\n\t" MK_CO_ELIM(0) MK_CO_ELIM(1) MK_CO_ELIM(2) MK_CO_ELIM(3) MK_CO_ELIM(4) MK_CO_ELIM(5) MK_CO_ELIM(6) MK_CO_ELIM(7) MK_CO_NEXT
                 : [a0] "+&v"(a[K0 + 0]), [a1] "+&v"(a[K0 + 1]), [a2] "+&v"(a[K0 + 2]), [a3] "+&v"(a[K0 + 3]), [a4] "+&v"(a[K0 + 4]), [a5] "+&v"(a[K0 + 5]), [a6] "+&v"(a[K0 + 6]), [a7] "+&v"(a[K0 + 7]), [nx] "=&v"(next)
                 : [f] "v"(f), [j] "n"(J), [jn] "n"(J + 1));
  if constexpr (N == 9)
    asm volatile("s_nop 1\n\t" MK_CO_ELIM(0) MK_CO_ELIM(1) MK_CO_ELIM(2) MK_CO_ELIM(3) MK_CO_ELIM(4) MK_CO_ELIM(5) MK_CO_ELIM(6) MK_CO_ELIM(7) MK_CO_ELIM(8) MK_CO_NEXT
                 : [a0] "+&v"(a[K0 + 0]), [a1] "+&v"(a[K0 + 1]), [a2] "+&v"(a[K0 + 2]), [a3] "+&v"(a[K0 + 3]), [a4] "+&v"(a[K0 + 4]), [a5] "+&v"(a[K0 + 5]), [a6] "+&v"(a[K0 + 6]), [a7] "+&v"(a[K0 + 7]), [a8] "+&v"(a[K0 + 8]), [nx] "=&v"(next)
                 : [f] "v"(f), [j] "n"(J), [jn] "n"(J + 1));
  if constexpr (N == 10)
    asm volatile("s_nop 1\n\t" MK_CO_ELIM(0) MK_CO_ELIM(1) MK_CO_ELIM(2) MK_CO_ELIM(3) MK_CO_ELIM(4) MK_CO_ELIM(5) MK_CO_ELIM(6) MK_CO_ELIM(7) MK_CO_ELIM(8) MK_CO_ELIM(9) MK_CO_NEXT
                 : [a0] "+&v"(a[K0 + 0]), [a1] "+&v"(a[K0 + 1]), [a2] "+&v"(a[K0 + 2]), [a3] "+&v"(a[K0 + 3]), [a4] "+&v"(a[K0 + 4]), [a5] "+&v"(a[K0 + 5]), [a6] "+&v"(a[K0 + 6]), [a7] "+&v"(a[K0 + 7]), [a8] "+&v"(a[K0 + 8]), [a9] "+&v"(a[K0 + 9]), [nx] "=&v"(next)
                 : [f] "v"(f), [j] "n"(J), [jn] "n"(J + 1));
  if constexpr (N == 11)
    asm volatile("s_nop 1\n\t" MK_CO_ELIM(0) MK_CO_ELIM(1) MK_CO_ELIM(2) MK_CO_ELIM(3) MK_CO_ELIM(4) MK_CO_ELIM(5) MK_CO_ELIM(6) MK_CO_ELIM(7) MK_CO_ELIM(8) MK_CO_ELIM(9) MK_CO_ELIM(10) MK_CO_NEXT
                 : [a0] "+&v"(a[K0 + 0]), [a1] "+&v"(a[K0 + 1]), [a2] "+&v"(a[K0 + 2]), [a3] "+&v"(a[K0 + 3]), [a4] "+&v"(a[K0 + 4]), [a5] "+&v"(a[K0 + 5]), [a6] "+&v"(a[K0 + 6]), [a7] "+&v"(a[K0 + 7]), [a8] "+&v"(a[K0 + 8]), [a9] "+&v"(a[K0 + 9]), [a10] "+&v"(a[K0 + 10]), [nx] "=&v"(next)
                 : [f] "v"(f), [j] "n"(J), [jn] "n"(J + 1));
  if constexpr (N == 12)
    asm volatile("s_nop 1\n\t" MK_CO_ELIM(0) MK_CO_ELIM(1) MK_CO_ELIM(2) MK_CO_ELIM(3) MK_CO_ELIM(4) MK_CO_ELIM(5) MK_CO_ELIM(6) MK_CO_ELIM(7) MK_CO_ELIM(8) MK_CO_ELIM(9) MK_CO_ELIM(10) MK_CO_ELIM(11) MK_CO_NEXT
                 : [a0] "+&v"(a[K0 + 0]), [a1] "+&v"(a[K0 + 1]), [a2] "+&v"(a[K0 + 2]), [a3] "+&v"(a[K0 + 3]), [a4] "+&v"(a[K0 + 4]), [a5] "+&v"(a[K0 + 5]), [a6] "+&v"(a[K0 + 6]), [a7] "+&v"(a[K0 + 7]), [a8] "+&v"(a[K0 + 8]), [a9] "+&v"(a[K0 + 9]), [a10] "+&v"(a[K0 + 10]), [a11] "+&v"(a[K0 + 11]), [nx] "=&v"(next)
                 : [f] "v"(f), [j] "n"(J), [jn] "n"(J + 1));
#else
  co_for<K0, K0 + N>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    a[k] = fma(co_bcast<J>(a[k]), f, a[k]);
  });
  next = co_bcast<(J + 1) % kRow>(a[K0]);
#endif
}

// y += sum over k in [0, N) of (x of lane k) * h[k]: this lane's entry of a matrix-vector product
// with the vector spread over the row.
template <int N>
MK_CO void co_dot(double& y, double x, const double (&h)[N]) {
#if defined(MK_CO_ASM)
  static_assert(N >= 1 && N <= 12, "row length");
  if constexpr (N == 1)
    asm volatile("s_nop 1\n\t" MK_CO_DOT(0)
                 : [y] "+&v"(y)
                 : [x] "v"(x), [h0] "v"(h[0]), [k0] "n"(0));
  if constexpr (N == 2)
    asm volatile("s_nop 1\n\t" MK_CO_DOT(0) MK_CO_DOT(1)
                 : [y] "+&v"(y)
                 : [x] "v"(x), [h0] "v"(h[0]), [h1] "v"(h[1]), [k0] "n"(0), [k1] "n"(1));
  if constexpr (N == 3)
    asm volatile("s_nop 1\n\t" MK_CO_DOT(0) MK_CO_DOT(1) MK_CO_DOT(2)
                 : [y] "+&v"(y)
                 : [x] "v"(x), [h0] "v"(h[0]), [h1] "v"(h[1]), [h2] "v"(h[2]), [k0] "n"(0), [k1] "n"(1), [k2] "n"(2));
  if constexpr (N == 4)
    asm volatile("s_nop 1\n\t" MK_CO_DOT(0) MK_CO_DOT(1) MK_CO_DOT(2) MK_CO_DOT(3)
                 : [y] "+&v"(y)
                 : [x] "v"(x), [h0] "v"(h[0]), [h1] "v"(h[1]), [h2] "v"(h[2]), [h3] "v"(h[3]), [k0] "n"(0), [k1] "n"(1), [k2] "n"(2), [k3] "n"(3));
  if constexpr (N == 5)
    asm volatile("s_nop 1\n\t" MK_CO_DOT(0) MK_CO_DOT(1) MK_CO_DOT(2) MK_CO_DOT(3) MK_CO_DOT(4)
                 : [y] "+&v"(y)
                 : [x] "v"(x), [h0] "v"(h[0]), [h1] "v"(h[1]), [h2] "v"(h[2]), [h3] "v"(h[3]), [h4] "v"(h[4]), [k0] "n"(0), [k1] "n"(1), [k2] "n"(2), [k3] "n"(3), [k4] "n"(4));
  if constexpr (N == 6)
    asm volatile("s_nop 1\n\t" MK_CO_DOT(0) MK_CO_DOT(1) MK_CO_DOT(2) MK_CO_DOT(3) MK_CO_DOT(4) MK_CO_DOT(5)
                 : [y] "+&v"(y)
                 : [x] "v"(x), [h0] "v"(h[0]), [h1] "v"(h[1]), [h2] "v"(h[2]), [h3] "v"(h[3]), [h4] "v"(h[4]), [h5] "v"(h[5]), [k0] "n"(0), [k1] "n"(1), [k2] "n"(2), [k3] "n"(3), [k4] "n"(4), [k5] "n"(5));
  if constexpr (N == 7)
    asm volatile("s_nop 1\n\t" MK_CO_DOT(0) MK_CO_DOT(1) MK_CO_DOT(2) MK_CO_DOT(3) MK_CO_DOT(4) MK_CO_DOT(5) MK_CO_DOT(6)
                 : [y] "+&v"(y)
                 : [x] "v"(x), [h0] "v"(h[0]), [h1] "v"(h[1]), [h2] "v"(h[2]), [h3] "v"(h[3]), [h4] "v"(h[4]), [h5] "v"(h[5]), [h6] "v"(h[6]), [k0] "n"(0), [k1] "n"(1), [k2] "n"(2), [k3] "n"(3), [k4] "n"(4), [k5] "n"(5), [k6] "n"(6));
  if constexpr (N == 8)
    asm volatile("s_nop 1\n\t" MK_CO_DOT(0) MK_CO_DOT(1) MK_CO_DOT(2) MK_CO_DOT(3) MK_CO_DOT(4) MK_CO_DOT(5) MK_CO_DOT(6) MK_CO_DOT(7)
                 : [y] "+&v"(y)
                 : [x] "v"(x), [h0] "v"(h[0]), [h1] "v"(h[1]), [h2] "v"(h[2]), [h3] "v"(h[3]), [h4] "v"(h[4]), [h5] "v"(h[5]), [h6] "v"(h[6]), [h7] "v"(h[7]), [k0] "n"(0), [k1] "n"(1), [k2] "n"(2), [k3] "n"(3), [k4] "n"(4), [k5] "n"(5), [k6] "n"(6), [k7] "n"(7));
  if constexpr (N == 9)
    asm volatile("s_nop 1\n\t" MK_CO_DOT(0) MK_CO_DOT(1) MK_CO_DOT(2) MK_CO_DOT(3) MK_CO_DOT(4) MK_CO_DOT(5) MK_CO_DOT(6) MK_CO_DOT(7) MK_CO_DOT(8)
                 : [y] "+&v"(y)
                 : [x] "v"(x), [h0] "v"(h[0]), [h1] "v"(h[1]), [h2] "v"(h[2]), [h3] "v"(h[3]), [h4] "v"(h[4]), [h5] "v"(h[5]), [h6] "v"(h[6]), [h7] "v"(h[7]), [h8] "v"(h[8]), [k0] "n"(0), [k1] "n"(1), [k2] "n"(2), [k3] "n"(3), [k4] "n"(4), [k5] "n"(5), [k6] "n"(6), [k7] "n"(7), [k8] "n"(8));
  if constexpr (N == 10)
    asm volatile("s_nop 1\n\t" MK_CO_DOT(0) MK_CO_DOT(1) MK_CO_DOT(2) MK_CO_DOT(3) MK_CO_DOT(4) MK_CO_DOT(5) MK_CO_DOT(6) MK_CO_DOT(7) MK_CO_DOT(8) MK_CO_DOT(9)
                 : [y] "+&v"(y)
                 : [x] "v"(x), [h0] "v"(h[0]), [h1] "v"(h[1]), [h2] "v"(h[2]), [h3] "v"(h[3]), [h4] "v"(h[4]), [h5] "v"(h[5]), [h6] "v"(h[6]), [h7] "v"(h[7]), [h8] "v"(h[8]), [h9] "v"(h[9]), [k0] "n"(0), [k1] "n"(1), [k2] "n"(2), [k3] "n"(3), [k4] "n"(4), [k5] "n"(5), [k6] "n"(6), [k7] "n"(7), [k8] "n"(8), [k9] "n"(9));
  if constexpr (N == 11)
    asm volatile("s_nop 1\n\t" MK_CO_DOT(0) MK_CO_DOT(1) MK_CO_DOT(2) MK_CO_DOT(3) MK_CO_DOT(4) MK_CO_DOT(5) MK_CO_DOT(6) MK_CO_DOT(7) MK_CO_DOT(8) MK_CO_DOT(9) MK_CO_DOT(10)
                 : [y] "+&v"(y)
                 : [x] "v"(x), [h0] "v"(h[0]), [h1] "v"(h[1]), [h2] "v"(h[2]), [h3] "v"(h[3]), [h4] "v"(h[4]), [h5] "v"(h[5]), [h6] "v"(h[6]), [h7] "v"(h[7]), [h8] "v"(h[8]), [h9] "v"(h[9]), [h10] "v"(h[10]), [k0] "n"(0), [k1] "n"(1), [k2] "n"(2), [k3] "n"(3), [k4] "n"(4), [k5] "n"(5), [k6] "n"(6), [k7] "n"(7), [k8] "n"(8), [k9] "n"(9), [k10] "n"(10));
  if constexpr (N == 12)
    asm volatile("s_nop 1\n\t" MK_CO_DOT(0) MK_CO_DOT(1) MK_CO_DOT(2) MK_CO_DOT(3) MK_CO_DOT(4) MK_CO_DOT(5) MK_CO_DOT(6) MK_CO_DOT(7) MK_CO_DOT(8) MK_CO_DOT(9) MK_CO_DOT(10) MK_CO_DOT(11)
                 : [y] "+&v"(y)
                 : [x] "v"(x), [h0] "v"(h[0]), [h1] "v"(h[1]), [h2] "v"(h[2]), [h3] "v"(h[3]), [h4] "v"(h[4]), [h5] "v"(h[5]), [h6] "v"(h[6]), [h7] "v"(h[7]), [h8] "v"(h[8]), [h9] "v"(h[9]), [h10] "v"(h[10]), [h11] "v"(h[11]), [k0] "n"(0), [k1] "n"(1), [k2] "n"(2), [k3] "n"(3), [k4] "n"(4), [k5] "n"(5), [k6] "n"(6), [k7] "n"(7), [k8] "n"(8), [k9] "n"(9), [k10] "n"(10), [k11] "n"(11));
#else
  co_for<0, N>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    y = fma(co_bcast<k>(x), h[k], y);
  });
#endif
}

// h += sum over r of (w[r] of lane K) * w[r]: entry K of this lane's row of w'w.
template <int K>
MK_CO void co_gram(double& h, const double (&w)[6]) {
#if defined(MK_CO_ASM)
  asm volatile("s_nop 1\n\t" MK_CO_GRAM(0) MK_CO_GRAM(1) MK_CO_GRAM(2) MK_CO_GRAM(3) MK_CO_GRAM(4) MK_CO_GRAM(5)
               : [h] "+&v"(h)
               : [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), [w4] "v"(w[4]), [w5] "v"(w[5]), [k] "n"(K));
#else
#pragma unroll
  for (int r = 0; r < 6; ++r) h = fma(co_bcast<K>(w[r]), w[r], h);
#endif
}

// r <- r + (r of lane k) * f[k] for k = K0 .. K1 - 1, one after the other (each step reads what
// the previous one wrote): the eliminations of a factorisation repeated on a right-hand side.
template <int K0, int K1, int M>
MK_CO void co_replay(double& r, const double (&f)[M]) {
  if constexpr (K0 < K1) {
#if defined(MK_CO_ASM)
    constexpr int N = K1 - K0 < 4 ? K1 - K0 : 4;
    if constexpr (N == 4)
      asm volatile(MK_CO_CHAIN(0) MK_CO_CHAIN(1) MK_CO_CHAIN(2) MK_CO_CHAIN(3)
                   : [r] "+&v"(r)
                   : [f0] "v"(f[K0]), [f1] "v"(f[K0 + 1]), [f2] "v"(f[K0 + 2]), [f3] "v"(f[K0 + 3]), [k0] "n"(K0),
                     [k1] "n"(K0 + 1), [k2] "n"(K0 + 2), [k3] "n"(K0 + 3));
    else if constexpr (N == 3)
      asm volatile(MK_CO_CHAIN(0) MK_CO_CHAIN(1) MK_CO_CHAIN(2)
                   : [r] "+&v"(r)
                   : [f0] "v"(f[K0]), [f1] "v"(f[K0 + 1]), [f2] "v"(f[K0 + 2]), [k0] "n"(K0), [k1] "n"(K0 + 1),
                     [k2] "n"(K0 + 2));
    else if constexpr (N == 2)
      asm volatile(MK_CO_CHAIN(0) MK_CO_CHAIN(1)
                   : [r] "+&v"(r)
                   : [f0] "v"(f[K0]), [f1] "v"(f[K0 + 1]), [k0] "n"(K0), [k1] "n"(K0 + 1));
    else
      asm volatile(MK_CO_CHAIN(0) : [r] "+&v"(r) : [f0] "v"(f[K0]), [k0] "n"(K0));
    co_replay<K0 + N, K1>(r, f);
#else
    co_for<K0, K1>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      r = fma(co_bcast<k>(r), f[k], r);
    });
#endif
  }
}

// value of lane J of the row, one 64-bit move
template <int J>
MK_CO double co_bcast64(double v) {
#if defined(MK_CO_ASM)
  double o;
  asm volatile("s_nop 1\n\t" MK_CO_MOV(0) : [o0] "=v"(o) : [i0] "v"(v), [j] "n"(J));
  return o;
#else
  return co_bcast<J>(v);
#endif
}

struct CoPose {
  Q4 q;
  V3 t;
};
MK_CO CoPose co_compose(const CoPose& a, const CoPose& b) { return CoPose{qmul(a.q, b.q), a.t + qrot(a.q, b.t)}; }
template <int D>
MK_CO CoPose co_up_pose(const CoPose& p) {   // identity beyond the row
  return CoPose{Q4{co_up<D, true>(p.q.w), co_up<D, false>(p.q.x), co_up<D, false>(p.q.y), co_up<D, false>(p.q.z)},
                V3{co_up<D, false>(p.t.x), co_up<D, false>(p.t.y), co_up<D, false>(p.t.z)}};
}
template <int J>
MK_CO CoPose co_bcast_pose(const CoPose& p) {
#if defined(MK_CO_ASM)
  CoPose o;
  asm volatile("s_nop 1\n\t" MK_CO_MOV(0) MK_CO_MOV(1) MK_CO_MOV(2) MK_CO_MOV(3) MK_CO_MOV(4) MK_CO_MOV(5) MK_CO_MOV(6)
               : [o0] "=&v"(o.q.w), [o1] "=&v"(o.q.x), [o2] "=&v"(o.q.y), [o3] "=&v"(o.q.z), [o4] "=&v"(o.t.x),
                 [o5] "=&v"(o.t.y), [o6] "=&v"(o.t.z)
               : [i0] "v"(p.q.w), [i1] "v"(p.q.x), [i2] "v"(p.q.y), [i3] "v"(p.q.z), [i4] "v"(p.t.x), [i5] "v"(p.t.y),
                 [i6] "v"(p.t.z), [j] "n"(J));
  return o;
#else
  return CoPose{Q4{co_bcast<J>(p.q.w), co_bcast<J>(p.q.x), co_bcast<J>(p.q.y), co_bcast<J>(p.q.z)},
                V3{co_bcast<J>(p.t.x), co_bcast<J>(p.t.y), co_bcast<J>(p.t.z)}};
#endif
}

// Arg-max over the row: the largest value and the lowest lane that holds it, in every lane.
MK_CO void co_argmax(double v, double& best, int& who) {
  int i = co_lane();
  co_for<0, 4>([&](auto sc) {
    constexpr int M = 1 << decltype(sc)::value;
    const double ov = co_xor<M>(v);
    const int oi = co_xor_i<M>(i);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  });
  best = v;
  who = i;
}

// ---- per-lane constants of a problem set ----
// Lane l < NC carries chain joint l in canonical form (MkSegC), lane NC the fixed offset from
// the last joint frame to the task frame, lanes beyond it the identity.
template <int NC>
struct CoLane {
  double pq[4], pp[3], hinge, slide;
  MkDofC sd;              // QP constants of the chain slot (lanes >= NC: an inert slot)
  int qadr, dof;          // the slot's coordinate and dof (dof < 0: padding slot or lane beyond the chain)
  double onehot[NC];      // 1 in entry l
  bool valid;             // l < NC
  // the same in every lane (read once per launch, not once per iteration)
  double cost[6], gain, lm;   // frame task
  int nd, needs_diag, nfree;
};

template <int NC>
MK_CO void co_load_lane(const MkFusedDesc* __restrict__ D, CoLane<NC>& C) {
  const int l = co_lane();
  C.valid = l < NC;
  const int s = l < NC ? l : 0;
  const MkSegC S = D->segc[s];
  const MkDofC E = D->sdof[s];
  const bool post = l == NC;
#pragma unroll
  for (int i = 0; i < 4; ++i) C.pq[i] = C.valid ? S.pq[i] : (post ? D->post_quat[i] : (i == 0 ? 1.0 : 0.0));
#pragma unroll
  for (int i = 0; i < 3; ++i) C.pp[i] = C.valid ? S.pp[i] : (post ? D->post_pos[i] : 0.0);
  C.hinge = C.valid ? S.hinge : 0.0;
  C.slide = C.valid ? S.slide : 0.0;
  C.sd = E;
  if (!C.valid) {
    C.sd.A = 0.0; C.sd.B = 0.0; C.sd.W = 1.0;
    C.sd.lower = -INFINITY; C.sd.upper = INFINITY; C.sd.gain = 1.0; C.sd.vmax = INFINITY;
    C.sd.qadr = 0; C.sd.dof = -1;
  }
  C.qadr = C.valid ? D->seg_qadr[s] : 0;
  C.dof = C.valid ? D->seg_dof[s] : -1;
#pragma unroll
  for (int k = 0; k < NC; ++k) C.onehot[k] = (l == k) ? 1.0 : 0.0;
#pragma unroll
  for (int r = 0; r < 6; ++r) C.cost[r] = D->cost[r];
  C.gain = D->gain;
  C.lm = D->lm;
  C.nd = D->nd;
  C.needs_diag = D->needs_diag;
  C.nfree = D->nfree;
}

// Chain kinematics across the row (mj_kinematics + mj_jacSite + the rotation into the body
// frame, mink/configuration.py:143-153): this lane's Jacobian column Jc (rows linear; angular)
// and the world pose W of the task frame, in every lane.
//
// Element l of the row is the transform across joint l, T_l = (pq_l rotz(theta_l), pp_l +
// slide displacement) -- what one slot of the backward pass of mk_fused.hip applies --, element
// NC the offset to the task frame.  The inclusive suffix products S_l = T_l ... T_NC come from
// a log-step scan; S_(l+1) is the pose of the task frame in the frame that joint l moves, from
// which the column follows as in the lane-per-problem pass; S_0 is the world pose.
template <int NC>
MK_CO void co_chain(const CoLane<NC>& C, double qc, double (&Jc)[6], CoPose& W) {
  double sn, cs;
  mk_sincos(0.5 * (C.hinge * qc), &sn, &cs);
  CoPose S;
  S.q = Q4{C.pq[0] * cs - C.pq[3] * sn, C.pq[1] * cs + C.pq[2] * sn, C.pq[2] * cs - C.pq[1] * sn,
           C.pq[3] * cs + C.pq[0] * sn};
  // slide displacement d along the joint's z axis: d * (third column of the rotation)
  const double d = C.slide * qc;
  S.t = V3{fma(d, 2.0 * (S.q.x * S.q.z + S.q.w * S.q.y), C.pp[0]), fma(d, 2.0 * (S.q.y * S.q.z - S.q.w * S.q.x), C.pp[1]),
           fma(d, 1.0 - 2.0 * (S.q.x * S.q.x + S.q.y * S.q.y), C.pp[2])};
  S = co_compose(S, co_up_pose<1>(S));
  S = co_compose(S, co_up_pose<2>(S));
  S = co_compose(S, co_up_pose<4>(S));
  if constexpr (NC + 1 > 8) S = co_compose(S, co_up_pose<8>(S));
  const CoPose P = co_up_pose<1>(S);
  W = co_bcast_pose<0>(S);
  const M3 R = q2m(P.q);
  const double h = C.hinge, sl = C.slide;
  const double hx = h * P.t.x, hy = h * P.t.y;
  Jc[0] = fma(hx, R.m[3], fma(-hy, R.m[0], sl * R.m[6]));
  Jc[1] = fma(hx, R.m[4], fma(-hy, R.m[1], sl * R.m[7]));
  Jc[2] = fma(hx, R.m[5], fma(-hy, R.m[2], sl * R.m[8]));
  Jc[3] = h * R.m[6];
  Jc[4] = h * R.m[7];
  Jc[5] = h * R.m[8];
}

// Frame-task error e = log(T_wb^-1 T_wt) (mink/tasks/frame_task.py:95-122, mink/lie/se3.py:159-185)
// of the frame at (Q, t) for the target (Qt, pt), together with Ji and Q of
// SE3.ljacinv(e) = [[Ji, -Ji Q Ji], [0, Ji]] (mink/lie/se3.py:210-249, mink/lie/so3.py:214-226);
// Ji = I, Q = 0 where the reference returns exactly I6.  A lane applies the blocks to its own
// column as Ji (bl - Q (Ji ba)), which saves forming -Ji Q Ji.  Every lane evaluates the same
// scalars, with one exception: SE3.log needs sin / cos of theta / 2 and the Jacobian those of
// theta, and the two are one evaluation -- even lanes take theta / 2, odd lanes theta, and lanes
// 0 and 1 hand the results to the row.
MK_CO Twist co_frame_error_jlog(Q4 Q, V3 t, Q4 Qt, V3 pt, M3& Ji, M3& Qm) {
  const M3 R = q2m(Q);
  const V3 w = so3_log(qmul(qconj(Q), Qt));
  const V3 te = mulT(R, pt - t);
  const double t2 = dot(w, w);
  const double t2s = fmax(t2, 1e-300);
  const double rth = mk_rsqrt(t2s), th = t2s * rth;
  double sn, cs;
  mk_sincos((co_lane() & 1) ? th : 0.5 * th, &sn, &cs);
  const double sh = co_bcast64<0>(sn), ch = co_bcast64<0>(cs);
  const double coef_gen = (1.0 - th * ch * mk_rcp(2.0 * sh)) * (rth * rth);
  const double coef = (t2 < MK_LIE_EPS) ? 1.0 / 12.0 : coef_gen;
  const M3 Vinv = rodrigues_like(w, -0.5, coef);
  const Twist e = Twist{mul(Vinv, te), w};
  if (t2 < MK_LIE_EPS) {
#pragma unroll
    for (int i = 0; i < 9; ++i) { Ji.m[i] = (i % 4 == 0) ? 1.0 : 0.0; Qm.m[i] = 0.0; }
    return e;
  }
  const double s = co_bcast64<1>(sn), c = co_bcast64<1>(cs);
  // SO3.ljacinv switches on theta (not theta^2); theta >= 1e-5 here, so closed form.
  const double A = (rth * rth) * (1.0 - (th * s * mk_rcp(2.0 * (1.0 - c))));
  Ji = rodrigues_like(w, -0.5, A);
  Qm = se3_Q_closed(e, t2, rth, th, s, c);
  return e;
}

// Gauss-Jordan elimination, without pivoting, of H with the rows and columns of the active
// variables replaced by identity (H symmetric positive definite, so every pivot is positive):
// lane i holds row i.  With pivot j every other row adds nf_ij = -A_ij / A_jj times the pivot
// row; the multipliers and the reciprocal of the lane's final diagonal are kept, so that a
// right-hand side is solved for by repeating the eliminations on it (co_apply; entries of active
// variables must be zero there).  An active variable's row is e_i with a zero right-hand side:
// as a pivot it changes nothing but its own (dead) column, and no pivot changes it.  The pivot
// lane's own multiplier is fma(-A_jj, 1 / A_jj, 1), the rounding residual of the reciprocal
// (a relative change of its row by 1e-16) instead of a lane test.
// co_factor_solve eliminates a right-hand side along with the matrix and returns the solution.
template <int NC>
struct CoFactor {
  double nf[NC];
  double invd;
};

template <int NC>
MK_CO double co_factor_solve(const double (&H)[NC], const double (&onehot)[NC], uint32_t act, bool valid, double rhs,
                             CoFactor<NC>& F) {
  const bool a = ((act >> co_lane()) & 1u) != 0u;
  double A[NC + 1];   // the lane's row, and its right-hand side as one more column
#pragma unroll
  for (int k = 0; k < NC; ++k) A[k] = a ? onehot[k] : H[k];
  A[NC] = a ? 0.0 : rhs;
  double piv = co_bcast64<0>(A[0]);
  co_for<0, NC - 1>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const double nf = fma(-A[j], mk_rcp(piv), onehot[j]);
    F.nf[j] = nf;
    co_pivot<j, j + 1, NC - j>(A, nf, piv);
  });
  {
    const double nf = fma(-A[NC - 1], mk_rcp(piv), onehot[NC - 1]);
    F.nf[NC - 1] = nf;
    co_elim<NC - 1, NC, NC + 1>(A, nf);
  }
  double d = 0.0;
#pragma unroll
  for (int k = 0; k < NC; ++k) d = fma(onehot[k], A[k], d);
  F.invd = valid ? mk_rcp(d) : 0.0;
  return A[NC] * F.invd;
}

template <int NC>
MK_CO double co_apply(const CoFactor<NC>& F, double rhs) {
  co_replay<0, NC>(rhs, F.nf);
  return rhs * F.invd;
}

// (H x)_l for the row's vector x.
template <int NC>
MK_CO double co_symv(const double (&H)[NC], double x) {
  double y = 0.0;
  co_dot<NC>(y, x, H);
  return y;
}

// min 1/2 x'Hx + c'x  s.t. lo <= x <= hi over the row: the active-set method of box_qp
// (mk_fused.hip) -- phase 1 guesses the active set, phase 2 is the Goldfarb-Idnani dual method
// from the S-pair phase 1 ends in -- with one variable per lane.  The active set and its sides
// are 16-bit masks, the same in every lane of the row.  hdiag = H_ll.
// Returns 0 ok / 1 iteration limit / 2 infeasible.
template <int NC>
MK_CO int co_box_qp(const double (&H)[NC], double hdiag, const double (&onehot)[NC], double c, double lo, double hi,
                    bool valid, double& x, uint32_t& act_io, uint32_t& upper_io, bool warm) {
  constexpr double kTol = 1e-14;
  const int l = co_lane();
  const uint32_t bit = 1u << l;
  const int kGuessRounds = warm ? 3 : 6;
  uint32_t act, upper = upper_io;
  const bool bad = co_ballot(valid && !(lo <= hi)) != 0u;
  if (warm) {
    const double b = (upper & bit) ? hi : lo;
    act = co_ballot(valid && (act_io & bit) != 0u && fabs(b) < INFINITY);
  } else {
    // cold start: the bounds that the diagonal estimate -c_i / H_ii violates
    const double xi = -c * mk_rcp(hdiag);
    const uint32_t ov = co_ballot(valid && xi > hi), un = co_ballot(valid && xi < lo);
    act = ov | un;
    upper = (upper | ov) & ~un;
  }
  CoFactor<NC> F;
  x = 0.0;
  // ---- phase 1 ----
  bool settled = bad;
  for (int round = 0; round < 6 + NC + 2; ++round) {
    if (!settled) {
      const bool a = (act & bit) != 0u;
      const double xa = a ? ((upper & bit) ? hi : lo) : 0.0;
      double rhs = -c;
      co_dot<NC>(rhs, -xa, H);
      const double xf = co_factor_solve<NC>(H, onehot, act, valid, rhs, F);
      x = a ? xa : xf;
      double g = c;
      co_dot<NC>(g, x, H);
      const double ui = (upper & bit) ? -g : g;
      const uint32_t drop = co_ballot(valid && a && ui < 0.0);
      const bool over = !a && x - hi > kTol, under = !a && lo - x > kTol;
      uint32_t add = co_ballot(valid && (over || under));
      const uint32_t add_up = co_ballot(valid && over);
      if (round >= kGuessRounds) add = 0u;
      if ((drop | add) == 0u) {
        settled = true;
      } else {
        act = (act & ~drop) | add;
        upper = (upper & ~add) | (add_up & add);
      }
    }
    if (mk_wave_none(!settled)) break;
  }
  // ---- phase 2 (F is valid for `act` here) ----
  int status = bad ? 2 : 0;
  bool done = bad;
  int p = -1;
  double sgn = 0.0, bound = 0.0, up_new = 0.0, u = 0.0;
  bool have_u = false;
  for (int it = 0; it < 6 * NC + 8; ++it) {
    if (!done && p < 0) {
      const bool fr = valid && (act & bit) == 0u;
      const double vh = x - hi, vl = lo - x;
      const double viol = fr ? fmax(vh, vl) : -INFINITY;
      double worst = 0.0;
      int who = -1;
      // (the arg-max only when some bound is violated at all: rarely, after phase 1)
      if (co_ballot(viol > kTol) != 0u) co_argmax(viol, worst, who);
      if (worst > kTol) {
        p = who;
        const bool up = co_pick(vh >= vl ? 1.0 : 0.0, p) != 0.0;
        sgn = up ? 1.0 : -1.0;
        bound = co_pick(up ? hi : lo, p);
      } else {
        done = true;
      }
      up_new = 0.0;
    }
    if (mk_wave_none(!done)) break;
    if (!have_u) {   // wave-uniform
      have_u = true;
      double g = c;
      co_dot<NC>(g, x, H);
      const double ui = (upper & bit) ? -g : g;
      u = (valid && (act & bit) != 0u && ui > 0.0) ? ui : 0.0;
    }
    if (!done) {
      const bool a = (act & bit) != 0u;
      double z = co_apply<NC>(F, (l == p) ? -sgn : 0.0);
      z = (a || !valid) ? 0.0 : z;
      double r = co_symv<NC>(H, z);
      r = (upper & bit) ? r : -r;
      // blocking active bound: the smallest u_k / r_k over the active k with r_k > 0 (lowest k on ties)
      double t1;
      int kb;
      co_argmax((valid && a && r > 0.0) ? -(u * mk_rcp(r)) : -INFINITY, t1, kb);
      t1 = -t1;
      if (!(t1 < INFINITY)) kb = -1;
      const double xp = co_pick(x, p), zp = co_pick(z, p);
      const double t2 = (bound - xp) * mk_rcp(zp);
      const bool full = t2 <= t1;
      const double t = full ? t2 : t1;
      up_new += t;
      x += t * z;
      if (a) u = fmax(u - t * r, 0.0);
      if (full && l == p) { x = bound; u = up_new; }
      if (!full && l == kb) u = 0.0;
      if (full) {
        act |= 1u << p;
        if (sgn > 0.0) upper |= 1u << p; else upper &= ~(1u << p);
        p = -1;
      } else if (kb >= 0) {
        act &= ~(1u << kb);
      }
      (void)co_factor_solve<NC>(H, onehot, act, valid, 0.0, F);
    }
  }
  if (!done) status = 1;
  act_io = act;
  upper_io = upper;
  return status;
}

// The state of one problem that a row keeps across iterations of the callers' loop.
template <int NC>
struct CoProblem {
  double qc;              // this lane's chain coordinate
  double qd;              // this lane's first dof outside the chain (d = l), when l < nd
  MkDofC dd;              // and its constants
  bool has_dd;
  bool input_bad;
  uint32_t act, upper;    // warm start of the next solve
};

// One solve_ik for the row's problem from its chain kinematics (mink/solve_ik.py:68-105): QP
// assembly (H row, c, bounds), the box QP on the chain dofs and the clamps of the other dofs.
// Jc is this lane's Jacobian column, e the frame-task error, (Ji, Qm) its Jacobian blocks.  Returns the status; dq of the chain dof in x, of the lane's first other dof
// in xd; dofs d = l + 16 r (r >= 1) outside the chain are handled through `rest`, called once
// per such dof as rest(dof constants, dq).
template <int NC, class RestQ, class RestOut>
MK_CO int co_solve(const MkFusedDesc* __restrict__ D, const CoLane<NC>& C, CoProblem<NC>& P, double (&Jc)[6], Twist e,
                   const M3& Ji, const M3& Qm, double dt, double damping, bool warm, double& x, double& xd,
                   RestQ&& rest_q, RestOut&& rest_out) {
  const double (&cost)[6] = C.cost;
  // Jw = diag(cost) (-jlog) J_b for this lane's column (frame_task.py:146, task.py:125-127)
  double Jw[6];
  {
    const V3 bl = v3(Jc[0], Jc[1], Jc[2]), ba = v3(Jc[3], Jc[4], Jc[5]);
    const V3 ta = mul(Ji, ba);
    const V3 tl = mul(Ji, bl - mul(Qm, ta));
    Jw[0] = -cost[0] * tl.x; Jw[1] = -cost[1] * tl.y; Jw[2] = -cost[2] * tl.z;
    Jw[3] = -cost[3] * ta.x; Jw[4] = -cost[4] * ta.y; Jw[5] = -cost[5] * ta.z;
  }
  const double g = C.gain;
  const double ew[6] = {cost[0] * (-g * e.v.x), cost[1] * (-g * e.v.y), cost[2] * (-g * e.v.z),
                        cost[3] * (-g * e.w.x), cost[4] * (-g * e.w.y), cost[5] * (-g * e.w.z)};
  const double diag = damping + C.lm * (ew[0] * ew[0] + ew[1] * ew[1] + ew[2] * ew[2] + ew[3] * ew[3] +
                                         ew[4] * ew[4] + ew[5] * ew[5]);
  // row l of H = Jw'Jw + (diag + W) I
  double H[NC];
  co_for<0, NC>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    H[k] = C.onehot[k] * (diag + C.sd.W);
    co_gram<k>(H[k], Jw);
  });
  double hdiag = diag + C.sd.W, s = 0.0;
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    hdiag = fma(Jw[r], Jw[r], hdiag);
    s = fma(ew[r], Jw[r], s);
  }
  const double c = C.sd.B * P.qc - C.sd.A - s;
  const double vb = dt * C.sd.vmax;
  const double hi = fmin(C.sd.gain * (C.sd.upper - P.qc), vb);
  const double lo = fmax(-(C.sd.gain * (P.qc - C.sd.lower)), -vb);
  int status = co_box_qp<NC>(H, hdiag, C.onehot, c, lo, hi, C.valid, x, P.act, P.upper, warm);
  // non-finite chain q or target (the error, hence every entry of c, is NaN), a non-finite
  // coordinate outside the chain, or damping = 0 with a vanishing Levenberg-Marquardt term on a
  // dof without task cost (singular P, which the reference's QP backend rejects,
  // mink/solve_ik.py:101-103)
  const bool nonfinite = co_ballot(C.valid && !(fabs(diag + c) < INFINITY)) != 0u;
  if (P.input_bad || nonfinite || (C.needs_diag && !(diag > 0.0))) {
    status = 3;
    x = 0.0;
  }
  // ---- dofs outside the chain: diagonal Hessian, box limits -> clamp ----
  const int nd = C.nd;
  bool infeasible = false;
  auto clamp_dof = [&](const MkDofC& dc, double qi) -> double {
    const double vbd = dt * dc.vmax;
    const double h = fmin(dc.gain * (dc.upper - qi), vbd);
    const double lw = fmax(-(dc.gain * (qi - dc.lower)), -vbd);
    infeasible |= !(lw <= h);
    return fmin(fmax(-(dc.B * qi - dc.A) * mk_rcp(diag + dc.W), lw), h);
  };
  xd = 0.0;
  if (P.has_dd) xd = status == 3 ? 0.0 : clamp_dof(P.dd, P.qd);
  for (int d = co_lane() + kRow; d < nd; d += kRow) {
    const MkDofC dc = D->ddof[d];
    const double qi = rest_q(dc);
    rest_out(dc, qi, status == 3 ? 0.0 : clamp_dof(dc, qi));
  }
  if (status != 3 && co_ballot(infeasible) != 0u) status = 2;
  return status;
}

}  // namespace mkcoop

namespace {
using namespace mkcoop;

// Integration of the free joints' quaternions over a step with zero velocity: mj_integratePos
// renormalises them (the fused path gives free joints zero velocity: zero Jacobian columns and
// zeroed posture rows).
MK_CO void co_free_renorm(const MkFusedDesc* __restrict__ D, double* qrow) {
  const int l = co_lane();
  if (l < D->nfree) {
    double* qq = qrow + D->free_qadr[l] + 3;
    const Q4 nq = qnormalize(Q4{qq[0], qq[1], qq[2], qq[3]});
    qq[0] = nq.w; qq[1] = nq.x; qq[2] = nq.y; qq[3] = nq.z;
  }
}

// True when a coordinate outside the chain (scalar dofs, free-joint poses) is not finite.
MK_CO bool co_rest_nonfinite(const MkFusedDesc* __restrict__ D, const double* qrow) {
  const int l = co_lane();
  double acc = 0.0;
  for (int d = l; d < D->nd; d += kRow) acc += 0.0 * qrow[D->ddof[d].qadr];
  for (int f = 0; f < D->nfree; ++f)
    if (l < 7) acc += 0.0 * qrow[D->free_qadr[f] + l];
  return co_ballot(!(acc == 0.0)) != 0u;
}

template <int NC>
MK_CO void co_load_problem(const MkFusedDesc* __restrict__ D, const CoLane<NC>& C, const double* qrow, CoProblem<NC>& P) {
  const int l = co_lane();
  const double v = qrow[C.qadr];
  P.qc = C.dof >= 0 ? v : 0.0;
  P.has_dd = l < D->nd;
  P.dd = D->ddof[P.has_dd ? l : 0];
  P.qd = P.has_dd ? qrow[P.dd.qadr] : 0.0;
  P.input_bad = co_rest_nonfinite(D, qrow);
  P.act = 0u;
  P.upper = 0u;
}

// mk_solve_ik_batch on the cooperative form: one solve_ik for problem g (one row of lanes).
template <int NC>
MK_CO void coop_step_body(const MkFusedDesc* __restrict__ D, const double* __restrict__ q,
                          const double* __restrict__ targets, int target_stride, double dt, double damping, int B,
                          double* __restrict__ v_out, int32_t* __restrict__ status_out, long g) {
  const int l = co_lane();
  const bool live = g < B;
  const long pb = live ? g : B - 1;   // a row without a problem repeats the last one and stores nothing
  const int nq = D->nq, nv = D->nv;
  const double inv_dt = 1.0 / dt;
  CoLane<NC> C;
  co_load_lane<NC>(D, C);
  const double* qrow = q + pb * nq;
  CoProblem<NC> P;
  co_load_problem<NC>(D, C, qrow, P);
  const double* tg = targets + pb * target_stride + D->target_off;
  const Q4 Qt = Q4{tg[0], tg[1], tg[2], tg[3]};
  const V3 pt = v3(tg[4], tg[5], tg[6]);
  double Jc[6];
  CoPose W;
  co_chain<NC>(C, P.qc, Jc, W);
  M3 Ji, Qm;
  const Twist e = co_frame_error_jlog(W.q, W.t, Qt, pt, Ji, Qm);
  double x, xd;
  double* vrow = v_out + pb * nv;
  const int status = co_solve<NC>(
      D, C, P, Jc, e, Ji, Qm, dt, damping, false, x, xd, [&](const MkDofC& dc) { return qrow[dc.qadr]; },
      [&](const MkDofC& dc, double, double dq) { if (live) vrow[dc.dof] = dq * inv_dt; });
  if (live) {
    if (C.dof >= 0) vrow[C.dof] = x * inv_dt;
    if (P.has_dd) vrow[P.dd.dof] = xd * inv_dt;
    for (int f = 0; f < D->nfree; ++f)
      if (l < 6) vrow[D->free_dofadr[f] + l] = 0.0;
    if (l == 0) status_out[pb] = status;
  }
}

// mk_solve_ik_iterate_batch on the cooperative form: the callers' loop
// (interactive_scripts/teleop_mouse.py:105-121) -- up to max_iters x (solve_ik ->
// integrate_inplace -> frame-task error -> exit test) -- for problem pb, with the
// configuration in registers.  A row with live == false computes and stores nothing new.
// it0 / status0: the loop resumes after it0 solves already executed (by the lane-per-problem
// kernel, see coop_resume_kernel) with their worst status.
template <int NC>
MK_CO void coop_iterate_body(const MkFusedDesc* __restrict__ D, double* __restrict__ q,
                             const double* __restrict__ targets, int target_stride, double dt, double damping,
                             int max_iters, double pos_tol, double ori_tol, double* __restrict__ v_out,
                             int32_t* __restrict__ iters_out, double* __restrict__ err_out,
                             int32_t* __restrict__ status_out, long pb, bool live, int it0, int status0) {
  const int l = co_lane();
  const int nq = D->nq, nv = D->nv;
  const double inv_dt = 1.0 / dt;
  CoLane<NC> C;
  co_load_lane<NC>(D, C);
  double* qrow = q + pb * nq;
  CoProblem<NC> P;
  co_load_problem<NC>(D, C, qrow, P);
  const double* tg = targets + pb * target_stride + D->target_off;
  const Q4 Qt = Q4{tg[0], tg[1], tg[2], tg[3]};
  const V3 pt = v3(tg[4], tg[5], tg[6]);
  double* vrow = (live && v_out) ? v_out + pb * nv : nullptr;
  Twist e = Twist{v3(0, 0, 0), v3(0, 0, 0)};
  int status = status0, iters = it0;
  bool done = false;
  double vx = 0.0, vd = 0.0;
  const double pos_tol2 = pos_tol >= 0.0 ? pos_tol * pos_tol : -1.0, ori_tol2 = ori_tol >= 0.0 ? ori_tol * ori_tol : -1.0;
  for (int it = it0; it <= max_iters; ++it) {
    if (!done) {
      double Jc[6];
      CoPose W;
      co_chain<NC>(C, P.qc, Jc, W);
      M3 Ji, Qm;
      e = co_frame_error_jlog(W.q, W.t, Qt, pt, Ji, Qm);
      // exit test of the callers' loop, on the configuration reached by the previous solve
      // (|e| <= tol as |e|^2 <= tol^2; a negative tolerance never passes)
      const bool conv = it > 0 && dot(e.v, e.v) <= pos_tol2 && dot(e.w, e.w) <= ori_tol2;
      if (conv || it == max_iters) {
        done = true;
      } else {
        double x, xd;
        const int st = co_solve<NC>(
            D, C, P, Jc, e, Ji, Qm, dt, damping, it > it0, x, xd, [&](const MkDofC& dc) { return qrow[dc.qadr]; },
            [&](const MkDofC& dc, double qi, double dq) {
              const double vi = dq * inv_dt;
              if (live) qrow[dc.qadr] = qi + dt * vi;
              if (vrow) vrow[dc.dof] = vi;
            });
        status = st > status ? st : status;
        iters = it + 1;
        vx = x * inv_dt;
        vd = xd * inv_dt;
        P.qc = P.qc + dt * vx;
        P.qd = P.qd + dt * vd;
      }
    }
    if (mk_wave_none(!done)) break;
  }
  if (live) {
    const bool solved = iters > it0;   // otherwise q and v are as they came
    if (solved) {
      if (C.dof >= 0) qrow[C.qadr] = P.qc;
      if (P.has_dd) qrow[P.dd.qadr] = P.qd;
      co_free_renorm(D, qrow);
    }
    if (vrow && (solved || it0 == 0)) {
      if (C.dof >= 0) vrow[C.dof] = vx;
      if (solved) {
        if (P.has_dd) vrow[P.dd.dof] = vd;
        for (int f = 0; f < D->nfree; ++f)
          if (l < 6) vrow[D->free_dofadr[f] + l] = 0.0;
      }
    }
    if (l == 0) {
      status_out[pb] = status;
      if (iters_out) iters_out[pb] = iters;
      if (err_out) {
        double* eo = err_out + pb * 6;
        eo[0] = e.v.x; eo[1] = e.v.y; eo[2] = e.v.z; eo[3] = e.w.x; eo[4] = e.w.y; eo[5] = e.w.z;
      }
    }
  }
}

// mk_move_to_batch on the cooperative form: waypoint reaching (envs/mj_env.py:193-276 with
// envs/robot_utils.py:113-160) for problem pb.  Per outer step: the pose of the task frame, the
// waypoint generator (bounded linear step, half-way slerp, reached test), then the callers' IK
// loop towards the commanded pose; the problem stops after the step in which the generator
// reported the goal reached, or after max_steps.  As in fused_move_to_kernel (mk_fused.hip) the
// two nested loops are one flat sequence of ticks, one IK iteration each: a tick evaluates the
// chain once, closes the outer step if its IK loop has ended on this pose, opens the next one on
// the same pose, and solves.  steps0 / solves0 / status0: the rollout resumes at an outer-step
// boundary after that many steps and solves (by the lane-per-problem kernel, see
// coop_move_to_resume_kernel) with their worst status.
template <int NC>
MK_CO void coop_move_to_body(const MkFusedDesc* __restrict__ D, double* __restrict__ q, const double* __restrict__ goals,
                             double dt, double damping, int max_steps, int max_iters, double pos_tol, double ori_tol,
                             double wp_pos_thr, double wp_pos_step, double wp_rot_thr, int32_t* __restrict__ steps_out,
                             int32_t* __restrict__ reached_out, int32_t* __restrict__ solves_out,
                             int32_t* __restrict__ status_out, long pb, bool live, int steps0, int solves0, int status0) {
  const int nq = D->nq;
  const double inv_dt = 1.0 / dt;
  CoLane<NC> C;
  co_load_lane<NC>(D, C);
  double* qrow = q + pb * nq;
  CoProblem<NC> P;
  co_load_problem<NC>(D, C, qrow, P);
  const double* gp = goals + pb * 7;
  const Q4 Qg = Q4{gp[0], gp[1], gp[2], gp[3]};
  const V3 pg = v3(gp[4], gp[5], gp[6]);
  Q4 Qt = Q4{1, 0, 0, 0};
  V3 pt = v3(0, 0, 0);
  int status = status0, steps = steps0, solves = solves0, it = 0;
  bool alive = max_steps > steps0, now = false, reached = false, in_step = false;
  const double pos_tol2 = pos_tol >= 0.0 ? pos_tol * pos_tol : -1.0, ori_tol2 = ori_tol >= 0.0 ? ori_tol * ori_tol : -1.0;
  for (;;) {
    if (alive) {
      double Jc[6];
      CoPose W;
      co_chain<NC>(C, P.qc, Jc, W);
      M3 Ji, Qm;
      Twist e = Twist{v3(0, 0, 0), v3(0, 0, 0)};
      // does the IK loop of the current outer step end on this pose?
      bool begin = !in_step;
      if (in_step) {
        e = co_frame_error_jlog(W.q, W.t, Qt, pt, Ji, Qm);
        const bool conv = it > 0 && dot(e.v, e.v) <= pos_tol2 && dot(e.w, e.w) <= ori_tol2;
        if (conv || it == max_iters) {
          ++steps;
          if (now) {
            reached = true;
            alive = false;
          } else if (steps == max_steps) {
            alive = false;
          } else {
            begin = true;
          }
        }
      }
      if (begin) {
        // the next outer step opens on the same pose: waypoint generator, then the error towards
        // the new command
        now = waypoint_step(W.q, W.t, Qg, pg, wp_pos_thr, wp_pos_step, wp_rot_thr, Qt, pt);
        it = 0;
        in_step = true;
        if (max_iters == 0) {
          // no IK iteration per step: the pose never changes, so every step repeats this one
          if (now) { ++steps; reached = true; } else { steps = max_steps; }
          alive = false;
        }
        e = co_frame_error_jlog(W.q, W.t, Qt, pt, Ji, Qm);
      }
      if (alive) {
        double x, xd;
        const int st = co_solve<NC>(
            D, C, P, Jc, e, Ji, Qm, dt, damping, it > 0 || steps > steps0, x, xd,
            [&](const MkDofC& dc) { return qrow[dc.qadr]; },
            [&](const MkDofC& dc, double qi, double dq) { if (live) qrow[dc.qadr] = qi + dt * (dq * inv_dt); });
        status = st > status ? st : status;
        ++it;
        ++solves;
        P.qc = P.qc + dt * (x * inv_dt);
        P.qd = P.qd + dt * (xd * inv_dt);
      }
    }
    if (mk_wave_none(alive)) break;
  }
  if (live) {
    if (solves > solves0) {
      if (C.dof >= 0) qrow[C.qadr] = P.qc;
      if (P.has_dd) qrow[P.dd.qadr] = P.qd;
      co_free_renorm(D, qrow);
    }
    if (co_lane() == 0) {
      status_out[pb] = status;
      if (steps_out) steps_out[pb] = steps;
      if (reached_out) reached_out[pb] = reached ? 1 : 0;
      if (solves_out) solves_out[pb] = solves;
    }
  }
}

}  // namespace
#endif  // MK_COOP_FUNCS_INCLUDED

// The device functions above are also compiled into mk_fused.hip (whose loop kernel finishes
// the problems it hands off with coop_iterate_body), which includes this file with
// MK_COOP_NO_KERNELS defined; the kernels and launchers below belong to this file's own object.
#if !defined(MK_COOP_NO_KERNELS) && !defined(MK_COOP_KERNELS_INCLUDED)
#define MK_COOP_KERNELS_INCLUDED
namespace {
template <int NC>
__global__ __launch_bounds__(64) void coop_step_kernel(const MkFusedDesc* __restrict__ D, const double* __restrict__ q,
                                                       const double* __restrict__ targets, int target_stride, double dt,
                                                       double damping, int B, double* __restrict__ v_out,
                                                       int32_t* __restrict__ status_out) {
  coop_step_body<NC>(D, q, targets, target_stride, dt, damping, B, v_out, status_out,
                     (long)blockIdx.x * 4 + (threadIdx.x >> 4));
}

template <int NC>
__global__ __launch_bounds__(64) void coop_iterate_kernel(const MkFusedDesc* __restrict__ D, double* __restrict__ q,
                                                          const double* __restrict__ targets, int target_stride,
                                                          double dt, double damping, int B, int max_iters, double pos_tol,
                                                          double ori_tol, double* __restrict__ v_out,
                                                          int32_t* __restrict__ iters_out, double* __restrict__ err_out,
                                                          int32_t* __restrict__ status_out) {
  const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 4);
  const bool live = g < B;   // a row without a problem repeats the last one and stores nothing
  coop_iterate_body<NC>(D, q, targets, target_stride, dt, damping, max_iters, pos_tol, ori_tol, v_out, iters_out, err_out,
                        status_out, live ? g : B - 1, live, 0, 0);
}

template <int NC>
__global__ __launch_bounds__(64) void coop_move_to_kernel(const MkFusedDesc* __restrict__ D, double* __restrict__ q,
                                                          const double* __restrict__ goals, double dt, double damping,
                                                          int B, int max_steps, int max_iters, double pos_tol,
                                                          double ori_tol, double wp_pos_thr, double wp_pos_step,
                                                          double wp_rot_thr, int32_t* __restrict__ steps_out,
                                                          int32_t* __restrict__ reached_out,
                                                          int32_t* __restrict__ solves_out,
                                                          int32_t* __restrict__ status_out) {
  const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 4);
  const bool live = g < B;
  coop_move_to_body<NC>(D, q, goals, dt, damping, max_steps, max_iters, pos_tol, ori_tol, wp_pos_thr, wp_pos_step,
                        wp_rot_thr, steps_out, reached_out, solves_out, status_out, live ? g : B - 1, live, 0, 0, 0);
}

// The rollouts that fused_move_to_kernel (mk_fused.hip) left at an outer-step boundary, marked
// with MK_ITER_CONT_BIT in steps_out: found and shared out like the problems of coop_resume_kernel.
template <int NC>
__global__ __launch_bounds__(64) void coop_move_to_resume_kernel(
    const MkFusedDesc* __restrict__ D, double* __restrict__ q, const double* __restrict__ goals, double dt, double damping,
    int B, int max_steps, int max_iters, double pos_tol, double ori_tol, double wp_pos_thr, double wp_pos_step,
    double wp_rot_thr, int32_t* __restrict__ steps_out, int32_t* __restrict__ reached_out,
    int32_t* __restrict__ solves_out, int32_t* __restrict__ status_out) {
  const int row = threadIdx.x >> 4;
  const int nchunks = (B + 63) / 64;
  for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const long idx = (long)ch * 64 + threadIdx.x;
    const int32_t mark = idx < B ? steps_out[idx] : 0;
    uint64_t mask = __ballot((mark & MK_ITER_CONT_BIT) != 0);
    while (mask != 0ull) {   // wave-uniform
      uint64_t m = mask;
      for (int k = 0; k < row; ++k) m &= m - 1ull;
      if (m != 0ull) {
        const long pb = (long)ch * 64 + (__ffsll((unsigned long long)m) - 1);
        coop_move_to_body<NC>(D, q, goals, dt, damping, max_steps, max_iters, pos_tol, ori_tol, wp_pos_thr, wp_pos_step,
                              wp_rot_thr, steps_out, reached_out, solves_out, status_out, pb, true,
                              steps_out[pb] & ~MK_ITER_CONT_BIT, solves_out[pb], status_out[pb]);
      }
      for (int k = 0; k < 4; ++k) mask &= mask - 1ull;
    }
  }
}

// Second phase of the callers' loop on batches that fill the machine's lanes once: the
// lane-per-problem kernel (fused_iterate_kernel, mk_fused.hip) stops a wave when only a few of
// its 64 problems are still in the loop and marks them in `iters` (MK_ITER_CONT_BIT next to the
// solves executed so far); those stragglers -- a handful per wave, but they would hold the launch
// for their remaining sequential solves at 64 problems per wave -- finish here at 16 lanes per
// problem.  Wave w takes the 64-entry chunks w, w + gridDim.x, ... of the batch: one load of the
// marks, a ballot, and the marked problems of the chunk four at a time (one per row).
template <int NC>
__global__ __launch_bounds__(64) void coop_resume_kernel(const MkFusedDesc* __restrict__ D, double* __restrict__ q,
                                                         const double* __restrict__ targets, int target_stride,
                                                         double dt, double damping, int B, int max_iters, double pos_tol,
                                                         double ori_tol, double* __restrict__ v_out,
                                                         int32_t* __restrict__ iters_out, double* __restrict__ err_out,
                                                         int32_t* __restrict__ status_out) {
  const int row = threadIdx.x >> 4;
  const int nchunks = (B + 63) / 64;
  for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const long idx = (long)ch * 64 + threadIdx.x;
    const int32_t mark = idx < B ? iters_out[idx] : 0;
    uint64_t mask = __ballot((mark & MK_ITER_CONT_BIT) != 0);
    while (mask != 0ull) {   // wave-uniform
      // row r takes the r-th marked problem still waiting, if there is one
      uint64_t m = mask;
      for (int k = 0; k < row; ++k) m &= m - 1ull;
      const bool has = m != 0ull;
      if (has) {
        const long pb = (long)ch * 64 + (__ffsll((unsigned long long)m) - 1);
        const int32_t it0 = iters_out[pb] & ~MK_ITER_CONT_BIT;
        coop_iterate_body<NC>(D, q, targets, target_stride, dt, damping, max_iters, pos_tol, ori_tol, v_out, iters_out,
                              err_out, status_out, pb, true, it0, status_out[pb]);
      }
      for (int k = 0; k < 4; ++k) mask &= mask - 1ull;
    }
  }
}

}  // namespace

// ---- host-side launchers (called from mk_api.hip) ----
hipError_t mk_coop_launch_step(int nc, const MkFusedDesc* d_desc, const double* q, const double* targets,
                               int target_stride, double dt, double damping, int B, double* v, int32_t* status,
                               hipStream_t stream) {
  const int blocks = (B + 3) / 4;
  switch (nc) {
#define MK_CASE(N)                                                                                                      \
  case N:                                                                                                               \
    hipLaunchKernelGGL(coop_step_kernel<N>, dim3(blocks), dim3(64), 0, stream, d_desc, q, targets, target_stride, dt,   \
                       damping, B, v, status);                                                                          \
    break;
    MK_CASE(4) MK_CASE(6) MK_CASE(7) MK_CASE(8) MK_CASE(10) MK_CASE(12)
#undef MK_CASE
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t mk_coop_launch_iterate(int nc, const MkFusedDesc* d_desc, double* q, const double* targets,
                                  int target_stride, double dt, double damping, int B, int max_iters, double pos_tol,
                                  double ori_tol, double* v, int32_t* iters, double* err, int32_t* status,
                                  hipStream_t stream) {
  const int blocks = (B + 3) / 4;
  switch (nc) {
#define MK_CASE(N)                                                                                                      \
  case N:                                                                                                               \
    hipLaunchKernelGGL(coop_iterate_kernel<N>, dim3(blocks), dim3(64), 0, stream, d_desc, q, targets, target_stride,    \
                       dt, damping, B, max_iters, pos_tol, ori_tol, v, iters, err, status);                             \
    break;
    MK_CASE(4) MK_CASE(6) MK_CASE(7) MK_CASE(8) MK_CASE(10) MK_CASE(12)
#undef MK_CASE
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t mk_coop_launch_move_to(int nc, const MkFusedDesc* d_desc, double* q, const double* goals, double dt,
                                  double damping, int B, int max_steps, int max_iters, double pos_tol, double ori_tol,
                                  double wp_pos_thr, double wp_pos_step, double wp_rot_thr, int32_t* steps,
                                  int32_t* reached, int32_t* solves, int32_t* status, hipStream_t stream) {
  const int blocks = (B + 3) / 4;
  switch (nc) {
#define MK_CASE(N)                                                                                                      \
  case N:                                                                                                               \
    hipLaunchKernelGGL(coop_move_to_kernel<N>, dim3(blocks), dim3(64), 0, stream, d_desc, q, goals, dt, damping, B,     \
                       max_steps, max_iters, pos_tol, ori_tol, wp_pos_thr, wp_pos_step, wp_rot_thr, steps, reached,     \
                       solves, status);                                                                                 \
    break;
    MK_CASE(4) MK_CASE(6) MK_CASE(7) MK_CASE(8) MK_CASE(10) MK_CASE(12)
#undef MK_CASE
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// The rollouts that a lane-per-problem launch marked with MK_ITER_CONT_BIT in `steps`.
hipError_t mk_coop_launch_move_to_resume(int nc, const MkFusedDesc* d_desc, double* q, const double* goals, double dt,
                                         double damping, int B, int max_steps, int max_iters, double pos_tol,
                                         double ori_tol, double wp_pos_thr, double wp_pos_step, double wp_rot_thr,
                                         int32_t* steps, int32_t* reached, int32_t* solves, int32_t* status,
                                         int max_blocks, hipStream_t stream) {
  const int nchunks = (B + 63) / 64;
  const int blocks = nchunks < max_blocks ? nchunks : max_blocks;
  switch (nc) {
#define MK_CASE(N)                                                                                                      \
  case N:                                                                                                               \
    hipLaunchKernelGGL(coop_move_to_resume_kernel<N>, dim3(blocks), dim3(64), 0, stream, d_desc, q, goals, dt, damping, \
                       B, max_steps, max_iters, pos_tol, ori_tol, wp_pos_thr, wp_pos_step, wp_rot_thr, steps, reached,  \
                       solves, status);                                                                                 \
    break;
    MK_CASE(4) MK_CASE(6) MK_CASE(7) MK_CASE(8) MK_CASE(10) MK_CASE(12)
#undef MK_CASE
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// The problems that a lane-per-problem launch marked with MK_ITER_CONT_BIT in `iters`.
hipError_t mk_coop_launch_resume(int nc, const MkFusedDesc* d_desc, double* q, const double* targets, int target_stride,
                                 double dt, double damping, int B, int max_iters, double pos_tol, double ori_tol,
                                 double* v, int32_t* iters, double* err, int32_t* status, int max_blocks,
                                 hipStream_t stream) {
  const int nchunks = (B + 63) / 64;
  const int blocks = nchunks < max_blocks ? nchunks : max_blocks;
  switch (nc) {
#define MK_CASE(N)                                                                                                      \
  case N:                                                                                                               \
    hipLaunchKernelGGL(coop_resume_kernel<N>, dim3(blocks), dim3(64), 0, stream, d_desc, q, targets, target_stride,     \
                       dt, damping, B, max_iters, pos_tol, ori_tol, v, iters, err, status);                             \
    break;
    MK_CASE(4) MK_CASE(6) MK_CASE(7) MK_CASE(8) MK_CASE(10) MK_CASE(12)
#undef MK_CASE
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

#endif  // MK_COOP_NO_KERNELS
