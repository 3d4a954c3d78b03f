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
  return CoPose{Q4{co_bcast<J>(p.q.w), co_bcast<J>(p.q.x), co_bcast<J>(p.q.y), co_bcast<J>(p.q.z)},
                V3{co_bcast<J>(p.t.x), co_bcast<J>(p.t.y), co_bcast<J>(p.t.z)}};
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

// Ji and Q of SE3.ljacinv(xi) = [[Ji, -Ji Q Ji], [0, Ji]] (mink/lie/se3.py:210-249,
// mink/lie/so3.py:214-226); Ji = I, Q = 0 where the reference returns exactly I6.  A lane
// applies the blocks to its own column as Ji (bl - Q (Ji ba)), which saves forming -Ji Q Ji.
MK_CO void co_ljacinv_parts(Twist xi, M3& Ji, M3& Qm) {
  const double t2 = dot(xi.w, xi.w);
  if (t2 < MK_LIE_EPS) {
#pragma unroll
    for (int i = 0; i < 9; ++i) { Ji.m[i] = (i % 4 == 0) ? 1.0 : 0.0; Qm.m[i] = 0.0; }
    return;
  }
  const double rth = mk_rsqrt(t2), th = t2 * rth;
  double s, c;
  mk_sincos(th, &s, &c);
  const double it2 = rth * rth;
  const double A = it2 * (1.0 - (th * s * mk_rcp(2.0 * (1.0 - c))));
  Ji = rodrigues_like(xi.w, -0.5, A);
  Qm = se3_Q_closed(xi, t2, rth, th, s, c);
}

// Gauss-Jordan elimination, without pivoting, of the block of H on the free variables of the
// row (H symmetric positive definite, so every pivot is positive): lane i holds row i.  With
// pivot j every other free row subtracts f_ij times the pivot row; the multipliers and the
// reciprocal of the lane's final diagonal are kept, so that a right-hand side is solved for by
// repeating the eliminations on it (co_apply).  Rows and pivots of active variables take no part.
template <int NC>
struct CoFactor {
  double f[NC];
  double invd;
};

template <int NC>
MK_CO void co_factor(const double (&H)[NC], const double (&onehot)[NC], uint32_t act, bool valid, CoFactor<NC>& F) {
  const int l = co_lane();
  double A[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) A[k] = H[k];
  const uint32_t freem = ~act & ((1u << NC) - 1u);
  // bit j: this lane's row is eliminated with pivot j (both free, j != l)
  const uint32_t w = ((freem >> l) & 1u) ? (freem & ~(1u << l)) : 0u;
  co_for<0, NC>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const double rp = mk_rcp(co_bcast<j>(A[j]));
    const double f = ((w >> j) & 1u) ? A[j] * rp : 0.0;
    F.f[j] = f;
    co_for<j + 1, NC>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      A[k] = fma(-f, co_bcast<j>(A[k]), A[k]);
    });
  });
  double d = 0.0;
#pragma unroll
  for (int k = 0; k < NC; ++k) d = fma(onehot[k], A[k], d);
  F.invd = valid ? mk_rcp(d) : 0.0;
}

template <int NC>
MK_CO double co_apply(const CoFactor<NC>& F, double rhs) {
  co_for<0, NC>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    rhs = fma(-F.f[j], co_bcast<j>(rhs), rhs);
  });
  return rhs * F.invd;
}

// (H x)_l for the row's vector x.
template <int NC>
MK_CO double co_symv(const double (&H)[NC], double x) {
  double y = 0.0;
  co_for<0, NC>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    y = fma(H[k], co_bcast<k>(x), y);
  });
  return y;
}

// min 1/2 x'Hx + c'x  s.t. lo <= x <= hi over the row: the active-set method of box_qp
// (mk_fused.hip) -- phase 1 guesses the active set, phase 2 is the Goldfarb-Idnani dual method
// from the S-pair phase 1 ends in -- with one variable per lane.  The active set and its sides
// are 16-bit masks, the same in every lane of the row.  hdiag = H_ll.
// Returns 0 ok / 1 iteration limit / 2 infeasible.
template <int NC, bool REFINE>
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
      const double rhs = -c - co_symv<NC>(H, xa);
      co_factor<NC>(H, onehot, act, valid, F);
      double xf = co_apply<NC>(F, rhs);
      if (REFINE) {
        // one step of iterative refinement on the free block
        const double res = rhs - co_symv<NC>(H, a ? 0.0 : xf);
        xf += co_apply<NC>(F, res);
      }
      x = a ? xa : xf;
      const double g = c + co_symv<NC>(H, x);
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
      double worst;
      int who;
      co_argmax(fr ? fmax(vh, vl) : -INFINITY, worst, who);
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
      const double g = c + co_symv<NC>(H, x);
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
      co_factor<NC>(H, onehot, act, valid, F);
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
// Jc is consumed.  Returns the status; dq of the chain dof in x, of the lane's first other dof
// in xd; dofs d = l + 16 r (r >= 1) outside the chain are handled through `rest`, called once
// per such dof as rest(dof constants, dq).
template <int NC, bool REFINE, class RestQ, class RestOut>
MK_CO int co_solve(const MkFusedDesc* __restrict__ D, const CoLane<NC>& C, CoProblem<NC>& P, double (&Jc)[6], Twist e,
                   double dt, double damping, bool warm, double& x, double& xd, RestQ&& rest_q, RestOut&& rest_out) {
  M3 Ji, Qm;
  co_ljacinv_parts(e, Ji, Qm);
  double cost[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) cost[r] = D->cost[r];
  // Jw = diag(cost) (-jlog) J_b for this lane's column (frame_task.py:146, task.py:125-127)
  double Jw[6];
  {
    const V3 bl = v3(Jc[0], Jc[1], Jc[2]), ba = v3(Jc[3], Jc[4], Jc[5]);
    const V3 ta = mul(Ji, ba);
    const V3 tl = mul(Ji, bl - mul(Qm, ta));
    Jw[0] = -cost[0] * tl.x; Jw[1] = -cost[1] * tl.y; Jw[2] = -cost[2] * tl.z;
    Jw[3] = -cost[3] * ta.x; Jw[4] = -cost[4] * ta.y; Jw[5] = -cost[5] * ta.z;
  }
  const double g = D->gain;
  const double ew[6] = {cost[0] * (-g * e.v.x), cost[1] * (-g * e.v.y), cost[2] * (-g * e.v.z),
                        cost[3] * (-g * e.w.x), cost[4] * (-g * e.w.y), cost[5] * (-g * e.w.z)};
  const double diag = damping + D->lm * (ew[0] * ew[0] + ew[1] * ew[1] + ew[2] * ew[2] + ew[3] * ew[3] +
                                         ew[4] * ew[4] + ew[5] * ew[5]);
  // row l of H = Jw'Jw + (diag + W) I
  double H[NC];
  co_for<0, NC>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r) s = fma(Jw[r], co_bcast<k>(Jw[r]), s);
    H[k] = fma(C.onehot[k], diag + C.sd.W, s);
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
  int status = co_box_qp<NC, REFINE>(H, hdiag, C.onehot, c, lo, hi, C.valid, x, P.act, P.upper, warm);
  // non-finite chain q or target (the error, hence every entry of c, is NaN), a non-finite
  // coordinate outside the chain, or damping = 0 with a vanishing Levenberg-Marquardt term on a
  // dof without task cost (singular P, which the reference's QP backend rejects,
  // mink/solve_ik.py:101-103)
  const bool nonfinite = co_ballot(C.valid && !(fabs(diag + c) < INFINITY)) != 0u;
  if (P.input_bad || nonfinite || (D->needs_diag && !(diag > 0.0))) {
    status = 3;
    x = 0.0;
  }
  // ---- dofs outside the chain: diagonal Hessian, box limits -> clamp ----
  const int nd = D->nd;
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
template <int NC, bool REFINE>
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
  const Twist e = frame_error(W.q, W.t, q2m(W.q), Qt, pt);
  double x, xd;
  double* vrow = v_out + pb * nv;
  const int status = co_solve<NC, REFINE>(
      D, C, P, Jc, e, dt, damping, false, x, xd, [&](const MkDofC& dc) { return qrow[dc.qadr]; },
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
// integrate_inplace -> frame-task error -> exit test) -- for problem g, with the
// configuration in registers.
template <int NC, bool REFINE>
MK_CO void coop_iterate_body(const MkFusedDesc* __restrict__ D, double* __restrict__ q,
                             const double* __restrict__ targets, int target_stride, double dt, double damping, int B,
                             int max_iters, double pos_tol, double ori_tol, double* __restrict__ v_out,
                             int32_t* __restrict__ iters_out, double* __restrict__ err_out,
                             int32_t* __restrict__ status_out, long g) {
  const int l = co_lane();
  const bool live = g < B;
  const long pb = live ? g : B - 1;
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
  int status = 0, iters = 0;
  bool done = false;
  double vx = 0.0, vd = 0.0;
  for (int it = 0; it <= max_iters; ++it) {
    if (!done) {
      double Jc[6];
      CoPose W;
      co_chain<NC>(C, P.qc, Jc, W);
      e = frame_error(W.q, W.t, q2m(W.q), Qt, pt);
      // exit test of the callers' loop, on the configuration reached by the previous solve
      const bool conv = it > 0 && sqrt(dot(e.v, e.v)) <= pos_tol && sqrt(dot(e.w, e.w)) <= ori_tol;
      if (conv || it == max_iters) {
        done = true;
      } else {
        double x, xd;
        const int st = co_solve<NC, REFINE>(
            D, C, P, Jc, e, dt, damping, it > 0, x, xd, [&](const MkDofC& dc) { return qrow[dc.qadr]; },
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
    if (C.dof >= 0) qrow[C.qadr] = P.qc;
    if (P.has_dd) qrow[P.dd.qadr] = P.qd;
    if (iters > 0) co_free_renorm(D, qrow);
    if (vrow) {
      if (C.dof >= 0) vrow[C.dof] = vx;
      if (iters > 0) {
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

#if !defined(MK_COOP_NO_KERNELS)
template <int NC, bool REFINE>
__global__ __launch_bounds__(64) void coop_step_kernel(const MkFusedDesc* __restrict__ D, const double* __restrict__ q,
                                                       const double* __restrict__ targets, int target_stride, double dt,
                                                       double damping, int B, double* __restrict__ v_out,
                                                       int32_t* __restrict__ status_out) {
  coop_step_body<NC, REFINE>(D, q, targets, target_stride, dt, damping, B, v_out, status_out,
                             (long)blockIdx.x * 4 + (threadIdx.x >> 4));
}

template <int NC, bool REFINE>
__global__ __launch_bounds__(64) void coop_iterate_kernel(const MkFusedDesc* __restrict__ D, double* __restrict__ q,
                                                          const double* __restrict__ targets, int target_stride,
                                                          double dt, double damping, int B, int max_iters, double pos_tol,
                                                          double ori_tol, double* __restrict__ v_out,
                                                          int32_t* __restrict__ iters_out, double* __restrict__ err_out,
                                                          int32_t* __restrict__ status_out) {
  coop_iterate_body<NC, REFINE>(D, q, targets, target_stride, dt, damping, B, max_iters, pos_tol, ori_tol, v_out,
                                iters_out, err_out, status_out, (long)blockIdx.x * 4 + (threadIdx.x >> 4));
}
#endif

}  // namespace

#if !defined(MK_COOP_NO_KERNELS)
// ---- host-side launchers (called from mk_api.hip) ----
#define MK_COOP_CASES(M) M(4) M(6) M(7) M(8) M(10) M(12)

hipError_t mk_coop_launch_step(int nc, bool refine, const MkFusedDesc* d_desc, const double* q, const double* targets,
                               int target_stride, double dt, double damping, int B, double* v, int32_t* status,
                               hipStream_t stream) {
  const int blocks = (B + 3) / 4;
  switch (nc) {
#define MK_CASE(N)                                                                                                    \
  case N:                                                                                                             \
    if (refine)                                                                                                       \
      hipLaunchKernelGGL((coop_step_kernel<N, true>), dim3(blocks), dim3(64), 0, stream, d_desc, q, targets,          \
                         target_stride, dt, damping, B, v, status);                                                   \
    else                                                                                                              \
      hipLaunchKernelGGL((coop_step_kernel<N, false>), dim3(blocks), dim3(64), 0, stream, d_desc, q, targets,         \
                         target_stride, dt, damping, B, v, status);                                                   \
    break;
    MK_COOP_CASES(MK_CASE)
#undef MK_CASE
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t mk_coop_launch_iterate(int nc, bool refine, const MkFusedDesc* d_desc, double* q, const double* targets,
                                  int target_stride, double dt, double damping, int B, int max_iters, double pos_tol,
                                  double ori_tol, double* v, int32_t* iters, double* err, int32_t* status,
                                  hipStream_t stream) {
  const int blocks = (B + 3) / 4;
  switch (nc) {
#define MK_CASE(N)                                                                                                    \
  case N:                                                                                                             \
    if (refine)                                                                                                       \
      hipLaunchKernelGGL((coop_iterate_kernel<N, true>), dim3(blocks), dim3(64), 0, stream, d_desc, q, targets,       \
                         target_stride, dt, damping, B, max_iters, pos_tol, ori_tol, v, iters, err, status);          \
    else                                                                                                              \
      hipLaunchKernelGGL((coop_iterate_kernel<N, false>), dim3(blocks), dim3(64), 0, stream, d_desc, q, targets,      \
                         target_stride, dt, damping, B, max_iters, pos_tol, ori_tol, v, iters, err, status);          \
    break;
    MK_COOP_CASES(MK_CASE)
#undef MK_CASE
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

#endif  // MK_COOP_NO_KERNELS
