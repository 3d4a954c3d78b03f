// Fused differential-IK kernels: one IK problem per lane, 64 problems per wave.
//
// Per problem and per step the kernel does, entirely in registers + one LDS tile:
//   kinematics of the chain to the task frame and the body-frame frame Jacobian on the
//   chain dofs, in one backward pass     (mj_kinematics, mj_jacSite, configuration.py:143-153)
//   e = log(T_wb^-1 T_wt), J = -jlog(T_wt^-1 T_wb) J_b          (frame_task.py:95-146)
//   H = damping I + sum_t (Jw'Jw + mu I), c = -sum_t ew'Jw       (task.py:125-138, solve_ik.py:13-22)
//   box limits from ConfigurationLimit / VelocityLimit          (configuration_limit.py:98-124,
//                                                                 velocity_limit.py:96-101)
//   exact QP solve, v = dq / dt                                 (solve_ik.py:101-105)
//   optionally q <- q (+) dt v and the loop of teleop_mouse.py:105-121.
//
// Structure used: with a single frame task whose chain has NC 1-dof joints, H is a
// dense NC x NC block on the chain dofs plus a diagonal on all other dofs, and the
// limits are per-dof boxes; dofs outside the chain therefore decouple into clamps and
// the dense block is solved by a dual active-set method (Goldfarb-Idnani iterates)
// with the active bounds expressed as lane masks, so 64 problems with different
// active sets run one instruction stream.
// This file is compiled twice: as it is (float64, every kernel), and with MK_F32 from
// mk_fused32.hip: the step, loop and rollout kernels of problems without collision rows in
// float32 arithmetic, under launcher names of their own.
#if defined(MK_F32)
#define mk_fused_launch_step mk_fused32_launch_step
#define mk_fused_launch_iterate mk_fused32_launch_iterate
#define mk_fused_launch_move_to mk_fused32_launch_move_to
#endif
#include "mk_device.h"
#include "mk_generic.h"
#if !defined(MK_F32)
#include "mk_geom.h"
// the cooperative device functions (one problem per 16-lane row): the loop kernel with a
// hand-off finishes its last problems with them
#define MK_COOP_NO_KERNELS
#include "mk_coop.hip"
#undef MK_COOP_NO_KERNELS
#endif
#include "mk_options.h"

#include <algorithm>
#include <cstdlib>

// One wave per workgroup; register budget / occupancy target of the fused kernels.
#ifndef MK_KERNEL_ATTR
#define MK_KERNEL_ATTR __launch_bounds__(64)
#endif

namespace {

// Optional instrumentation of the host test build only (never in device code).
#if defined(MK_QP_STATS) && !defined(__HIP_DEVICE_COMPILE__)
static long g_qp_stats[5];  // solves, phase-1 rounds, phase-2 steps, factorizations, (spare)
#define MK_STAT(i) (++g_qp_stats[i])
#else
#define MK_STAT(i) ((void)0)
#endif

MK_DEV constexpr int tri(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

// Read-only view of a packed lower-triangular matrix with a per-lane stride: on the
// GPU the Hessian of each lane lives in LDS as element e of lane l at p[e * 64 + l]
// (conflict-free b64 reads, compile-time offsets); the host test build uses stride 1.
struct HView {
  const real* p;
  int stride;
  MK_DEV real operator()(int e) const { return p[e * stride]; }
};

// Cholesky factor of H with the rows/columns in `act` replaced by identity.
// L holds the strict lower triangle, Ld the reciprocals of the diagonal.
// Returns false in float64.  In float32: whether a pivot came out below 1e-6 of its diagonal
// entry, i.e. was lost to rounding (a Hessian whose condition number float32 cannot carry).
template <int N>
MK_DEV bool chol_masked(HView H, uint32_t act, real (&L)[N * (N + 1) / 2], real (&Ld)[N]) {
  MK_STAT(3);
  bool lost = false;
  real mk[N];
#pragma unroll
  for (int i = 0; i < N; ++i) mk[i] = ((act >> i) & 1u) ? 0.0 : 1.0;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    real s = mk[j] * H(tri(j, j)) + (1.0 - mk[j]);
    const real hjj = s;
#pragma unroll
    for (int k = 0; k < j; ++k) s -= L[tri(j, k)] * L[tri(j, k)];
    if (MK_IS_F32) lost |= !(s > 1e-6 * hjj);
    s = fmax(s, MK_TINY);
    const real r = mk_rsqrt(s);
    Ld[j] = r;
    const real rm = r * mk[j];   // a masked column is zero below its unit diagonal
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      // rows of masked variables are zero: L[i][k] = 0 for k < j, so only the H term needs mk[i]
      real t = H(tri(i, j)) * mk[i];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= L[tri(i, k)] * L[tri(j, k)];
      L[tri(i, j)] = t * rm;
    }
  }
  return lost;
}

// In-place solve of (L L') y = b.
template <int N>
MK_DEV void chol_solve(const real (&L)[N * (N + 1) / 2], const real (&Ld)[N], real (&b)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    real s = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= L[tri(i, k)] * b[k];
    b[i] = s * Ld[i];
  }
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    real s = b[i];
#pragma unroll
    for (int k = i + 1; k < N; ++k) s -= L[tri(k, i)] * b[k];
    b[i] = s * Ld[i];
  }
}

// y = H x for the packed symmetric H, reading every stored entry once.
template <int N>
MK_DEV void symv(HView H, const real (&x)[N], real (&y)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) y[i] = H(tri(i, i)) * x[i];
#pragma unroll
  for (int i = 1; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < i; ++j) {
      const real h = H(tri(i, j));
      y[i] += h * x[j];
      y[j] += h * x[i];
    }
  }
}

// min 1/2 x'Hx + c'x  s.t. lo <= x <= hi, H positive definite (dense N x N).
//
// Active-set method on lane masks, exact for strictly convex problems.  An "S-pair" is
// an active set A with x the minimiser on the face {x_A at its bounds} and multipliers
// u_A >= 0.
//   phase 1  starts from a guess of A (cold: the bounds violated by the diagonal estimate
//            -c_i / H_ii; warm: the active set of the previous IK step of the same
//            problem) and repeats: solve on the face, drop
//            every bound whose multiplier is negative and (for the first few rounds)
//            add every violated bound.  A round that changes nothing has found the
//            optimum outright.  After kGuessRounds (3 warm, 6 cold) the additions stop, so the set only
//            shrinks and ends in an S-pair.
//   phase 2  is the Goldfarb-Idnani dual method from that S-pair: pick the most violated
//            inactive bound p and move along the direction that keeps A tight until p
//            is satisfied (full step: p joins A) or a multiplier reaches zero (partial
//            step: that bound leaves A).  It terminates for any strictly convex problem.
// Every change of A refactors H with the rows/columns of A replaced by identity,
// which keeps 64 different active sets on one instruction stream.
// act/upper: in = guess (used when warm), out = final active set and its sides.
// Returns 0 ok / 1 iteration limit / 2 infeasible / 3 (float32 only) a factorisation lost a pivot
// to rounding: the answer is not to be trusted.
template <int N>
MK_DEV int box_qp(HView H, const real (&c)[N], const real (&lo)[N], const real (&hi)[N], real (&x)[N],
                  uint32_t& act_io, uint32_t& upper_io, bool warm) {
  // bound violations below kTol count as satisfied (x is a step dq of a few 1e-2 at most)
  const real kTol = MK_IS_F32 ? 1e-7 : 1e-14;
  // a cold start has many bounds to find and profits from more guessing rounds; a warm
  // start that does not settle quickly is usually cycling and is better off in phase 2
  const int kGuessRounds = warm ? 3 : 6;
  MK_STAT(0);
  real L[N * (N + 1) / 2], Ld[N];
  uint32_t act = 0u, upper = upper_io;
  bool bad = false, lost = false;
#pragma unroll
  for (int i = 0; i < N; ++i) bad |= !(lo[i] <= hi[i]);
  if (warm) {
    act = act_io;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const real b = ((upper >> i) & 1u) ? hi[i] : lo[i];
      if (!(fabs(b) < INFINITY)) act &= ~(1u << i);
    }
  }
  if (!warm) {
    // cold start: the guess is the set of bounds that the diagonal estimate -c_i / H_ii
    // violates (about half the rounds of starting from the empty set)
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const real xi = -c[i] * mk_rcp(H(tri(i, i)));
      if (xi > hi[i]) { act |= (1u << i); upper |= (1u << i); }
      if (xi < lo[i]) { act |= (1u << i); upper &= ~(1u << i); }
    }
  }
  // ---- phase 1 ----
  bool settled = bad;
  // (the cap is the same for warm and cold lanes, so the loop stays wave-uniform when the two
  // kinds share a wave; a lane is settled at most N rounds after its additions stop)
#if defined(MK_AB_ROUNDCAP)
  for (int round = 0; round < kGuessRounds + N + 2; ++round) {
#else
  for (int round = 0; round < 6 + N + 2; ++round) {
#endif
    if (!settled) {
      MK_STAT(1);
      // H is re-read from LDS every round on purpose: hoisting the loads would pin the
      // whole matrix in registers next to the factor
      asm volatile("" ::: "memory");
      lost |= chol_masked<N>(H, act, L, Ld);
      real xa[N], hx[N];
#pragma unroll
      for (int i = 0; i < N; ++i)
        xa[i] = ((act >> i) & 1u) ? (((upper >> i) & 1u) ? hi[i] : lo[i]) : 0.0;
      symv<N>(H, xa, hx);
#pragma unroll
      for (int i = 0; i < N; ++i) x[i] = ((act >> i) & 1u) ? xa[i] : -c[i] - hx[i];
      chol_solve<N>(L, Ld, x);
      symv<N>(H, x, hx);
      uint32_t drop = 0u, add = 0u, add_up = 0u;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const real g = c[i] + hx[i];
        const bool a = (act >> i) & 1u;
        const real ui = ((upper >> i) & 1u) ? -g : g;
        if (a && ui < 0.0) drop |= (1u << i);
        if (!a && x[i] - hi[i] > kTol) { add |= (1u << i); add_up |= (1u << i); }
        if (!a && lo[i] - x[i] > kTol) add |= (1u << i);
      }
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
  // ---- phase 2 (L is valid for `act` here) ----
  int status = bad ? 2 : 0;
  bool done = bad;
  int p = -1;
  real sgn = 0.0, bound = 0.0, up_new = 0.0;
  // multipliers of the active bounds: phase 1 only needs their signs, so they are formed
  // here, once, and only if some lane of the wave enters phase 2 at all
  real u[N];
  bool have_u = false;
  for (int it = 0; it < 6 * N + 8; ++it) {
    if (!done && p < 0) {
      real worst = kTol;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const bool free_i = !((act >> i) & 1u);
        const real vh = x[i] - hi[i], vl = lo[i] - x[i];
        if (free_i && vh > worst) { worst = vh; p = i; sgn = 1.0; bound = hi[i]; }
        if (free_i && vl > worst) { worst = vl; p = i; sgn = -1.0; bound = lo[i]; }
      }
      up_new = 0.0;
      if (p < 0) done = true;
    }
    if (mk_wave_none(!done)) break;
    if (!have_u) {   // wave-uniform
      have_u = true;
      asm volatile("" ::: "memory");
      real hx[N];
      symv<N>(H, x, hx);
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const real g = c[i] + hx[i];
        const real ui = ((upper >> i) & 1u) ? -g : g;
        u[i] = (((act >> i) & 1u) && ui > 0.0) ? ui : 0.0;
      }
    }
    if (!done) {
      MK_STAT(2);
      asm volatile("" ::: "memory");
      real z[N], r[N];
#pragma unroll
      for (int i = 0; i < N; ++i) z[i] = (i == p) ? -sgn : 0.0;
      chol_solve<N>(L, Ld, z);
      symv<N>(H, z, r);
      // dual step direction of the active bounds and the blocking one
      real t1 = INFINITY;
      int kb = -1;
      real xp = 0.0, zp = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const bool a = (act >> k) & 1u;
        r[k] = ((upper >> k) & 1u) ? r[k] : -r[k];
        if (a && r[k] > 0.0) {
          real t = u[k] * mk_rcp(r[k]);
          if (t < t1) { t1 = t; kb = k; }
        }
        if (k == p) { xp = x[k]; zp = z[k]; }
      }
      const real t2 = (bound - xp) * mk_rcp(zp);
      const bool full = t2 <= t1;
      const real t = full ? t2 : t1;
      up_new += t;
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const bool a = (act >> k) & 1u;
        x[k] += t * z[k];
        if (a) u[k] = fmax(u[k] - t * r[k], 0.0);
        if (full && k == p) { x[k] = bound; u[k] = up_new; }
        if (!full && k == kb) u[k] = 0.0;
      }
      if (full) {
        act |= (1u << p);
        if (sgn > 0.0) upper |= (1u << p); else upper &= ~(1u << p);
        p = -1;
      } else {
        act &= ~(1u << kb);
      }
      lost |= chol_masked<N>(H, act, L, Ld);
    }
  }
  if (!done) status = 1;
  if (MK_IS_F32) {
    // float32: rounding may leave x a few ulps outside a bound it sits on
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = fmin(fmax(x[i], lo[i]), hi[i]);
    if (lost && status == 0) status = 3;
  }
  act_io = act;
  upper_io = upper;
  return status;
}

// Chain kinematics of one problem: the body-frame Jacobian of the task frame on the
// chain dofs (rows linear; angular -- what Configuration.get_frame_jacobian returns),
// the pose of the task frame and the frame-task error e = log(T_wb^-1 T_wt).
template <int NC>
struct ChainJac {
  real J[6][NC];
  Twist e;
};

// One backward pass from the task frame to the world.  (Q, t) is the pose of the task
// frame in the current joint frame and R = mat(Q).  Joints are in canonical form (about
// / along the z axis of their own frame, see MkSegC), so a hinge moves the task frame
// with the twist (z x t, z) and a slide with (z, 0) of that frame; expressed in the task
// frame these are (t.x R[1] - t.y R[0], R[2]) and (R[2], 0) in terms of the rows of R.
// Then the joint displacement and the fixed transform to the previous joint frame are
// applied on the left.  This is mj_kinematics + mj_jacSite + the rotation into the body
// frame (mink/configuration.py:143-153) without world-frame anchors / axes as
// intermediates.  Hinges, slides and padding slots share one instruction stream (the
// `hinge` factor of the slot selects), so the pass is straight-line code.
template <int NC, bool CAP = false>
MK_DEV void chain_jac_pose(const MkFusedDesc* __restrict__ D, const real (&qc)[NC], ChainJac<NC>& cj, Q4& Q, V3& t,
                           M3& R, int cap_slot = 0, Q4* Qcap = nullptr, V3* tcap = nullptr) {
  Q = Q4{D->post_quat[0], D->post_quat[1], D->post_quat[2], D->post_quat[3]};
  t = v3(D->post_pos[0], D->post_pos[1], D->post_pos[2]);
  R = q2m(Q);
  MkSegC s = D->segc[NC - 1];
#pragma unroll
  for (int k = NC - 1; k >= 0; --k) {
    if (CAP && k == cap_slot) {   // pose of the task frame in the frame behind joint cap_slot (wave-uniform)
      *Qcap = Q;
      *tcap = t;
    }
    const MkSegC nx = D->segc[k > 0 ? k - 1 : 0];   // requested one slot ahead
    const real h = s.hinge;
    const real th = h * qc[k];       // hinge angle (0 for a slide)
    real sn, cs;
    mk_sincos(0.5 * th, &sn, &cs);
    const real hx = h * t.x, hy = h * t.y, sl = s.slide;
    cj.J[0][k] = fma(hx, R.m[3], fma(-hy, R.m[0], sl * R.m[6]));
    cj.J[1][k] = fma(hx, R.m[4], fma(-hy, R.m[1], sl * R.m[7]));
    cj.J[2][k] = fma(hx, R.m[5], fma(-hy, R.m[2], sl * R.m[8]));
    cj.J[3][k] = h * R.m[6];
    cj.J[4][k] = h * R.m[7];
    cj.J[5][k] = h * R.m[8];
    t.z = fma(sl, qc[k], t.z);         // slide displacement
    // fixed transform to the previous joint frame times the joint rotation about z
    const Q4 Qs = Q4{s.pq[0] * cs - s.pq[3] * sn, s.pq[1] * cs + s.pq[2] * sn, s.pq[2] * cs - s.pq[1] * sn,
                     s.pq[3] * cs + s.pq[0] * sn};
    t = qrot(Qs, t) + v3(s.pp[0], s.pp[1], s.pp[2]);
    Q = qmul(Qs, Q);
    MK_SCHED_FENCE();
    MK_PIN4(Q.w, Q.x, Q.y, Q.z);
    MK_PIN3(t.x, t.y, t.z);
    // the column is pinned too: it is only consumed on the not-yet-converged path, and the
    // optimiser would otherwise sink its arithmetic (and with it the pose and the constants
    // of every slot) into that branch
    MK_PIN3(cj.J[0][k], cj.J[1][k], cj.J[2][k]);
    MK_PIN3(cj.J[3][k], cj.J[4][k], cj.J[5][k]);
    R = q2m(Q);
    s = nx;
  }
  // (Q, t, R) is now the pose of the task frame in the world
  if (CAP && cap_slot < 0) {
    *Qcap = Q;
    *tcap = t;
  }
}

template <int NC>
MK_DEV void chain_jac(const MkFusedDesc* __restrict__ D, const real (&qc)[NC], Q4 Qt, V3 pt, ChainJac<NC>& cj) {
  Q4 Q;
  V3 t;
  M3 R;
  chain_jac_pose<NC>(D, qc, cj, Q, t, R);
  cj.e = frame_error(Q, t, R, Qt, pt);
}

// World pose (Q, t) of the task frame alone: the recursion of chain_jac without columns.
template <int NC>
MK_DEV void chain_pose(const MkFusedDesc* __restrict__ D, const real (&qc)[NC], Q4& Q, V3& t) {
  Q = Q4{D->post_quat[0], D->post_quat[1], D->post_quat[2], D->post_quat[3]};
  t = v3(D->post_pos[0], D->post_pos[1], D->post_pos[2]);
  MkSegC s = D->segc[NC - 1];
#pragma unroll
  for (int k = NC - 1; k >= 0; --k) {
    const MkSegC nx = D->segc[k > 0 ? k - 1 : 0];
    real sn, cs;
    mk_sincos(0.5 * (s.hinge * qc[k]), &sn, &cs);
    t.z = fma(s.slide, qc[k], t.z);
    const Q4 Qs = Q4{s.pq[0] * cs - s.pq[3] * sn, s.pq[1] * cs + s.pq[2] * sn, s.pq[2] * cs - s.pq[1] * sn,
                     s.pq[3] * cs + s.pq[0] * sn};
    t = qrot(Qs, t) + v3(s.pp[0], s.pp[1], s.pp[2]);
    Q = qmul(Qs, Q);
    MK_SCHED_FENCE();
    MK_PIN4(Q.w, Q.x, Q.y, Q.z);
    MK_PIN3(t.x, t.y, t.z);
    s = nx;
  }
}

template <int NC>
MK_DEV void store_chain_q(const MkFusedDesc* __restrict__ D, real* row, const real (&qc)[NC]) {
#pragma unroll
  for (int k = 0; k < NC; ++k)
    if (D->seg_dof[k] >= 0) row[D->seg_qadr[k]] = qc[k];
}

// Chain coordinates of one problem from its q row; padding slots read as 0 so that they
// neither rotate nor translate.
template <int NC>
MK_DEV void load_chain_q(const MkFusedDesc* __restrict__ D, const real* row, real (&qc)[NC]) {
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const real v = row[D->seg_qadr[k]];
    qc[k] = (D->seg_dof[k] >= 0) ? v : 0.0;
  }
}

// True when a coordinate outside the chain (scalar dofs, free-joint poses) is not finite.
// Those coordinates only change by the kernels' own finite updates, so one look per launch
// is enough.
MK_DEV bool rest_nonfinite(const MkFusedDesc* __restrict__ D, const real* row) {
  real acc = 0.0;
  for (int d = 0; d < D->nd; ++d) acc += 0.0 * row[D->ddof[d].qadr];
  for (int f = 0; f < D->nfree; ++f)
    for (int k = 0; k < 7; ++k) acc += 0.0 * row[D->free_qadr[f] + k];
  return !(acc == 0.0);
}

// QP of one problem from its chain kinematics: the dense block of H on the chain dofs goes
// to the lane's LDS column (hstore), its linear term c and box bounds lo / hi to registers.
// cj.J is consumed (it becomes the weighted task Jacobian).  Returns the diagonal every dof
// shares: damping plus the Levenberg-Marquardt terms.  `row` is the lane's q.
template <int NC, bool REL>
MK_DEV real assemble_chain_qp(const MkFusedDesc* __restrict__ D, const real* row, const real (&qc)[NC],
                                ChainJac<NC>& cj, real dt, real damping, real (&c)[NC], real (&lo)[NC],
                                real (&hi)[NC], real* hstore, int hstride, const MkFusedExt* __restrict__ X,
                                Twist e_rel, real& hacc) {
  // second frame term (RelativeFrameTask, mink/tasks/relative_frame_task.py:106-142): its block of
  // H goes to the lane's LDS column first, its share of c and of the diagonal to registers
  real c_rel[NC], diag_rel = 0.0;
  if (REL) {
    M3 Jir, Bmr;
    se3_ljacinv_blocks(Twist{neg(e_rel.v), neg(e_rel.w)}, Jir, Bmr);   // jlog(T_rt^-1 T_rf) = ljacinv(-e)
    real rc[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) rc[r] = X->rel_cost[r];
    const int rs = X->rel_slot;
    real Jr[6][NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const real m = (k > rs) ? 1.0 : 0.0;   // joints in front of the root move root and frame alike
      const V3 bl = v3(cj.J[0][k], cj.J[1][k], cj.J[2][k]);
      const V3 ba = v3(cj.J[3][k], cj.J[4][k], cj.J[5][k]);
      const V3 tl = mul(Jir, bl) + mul(Bmr, ba);
      const V3 ta = mul(Jir, ba);
      Jr[0][k] = m * rc[0] * tl.x; Jr[1][k] = m * rc[1] * tl.y; Jr[2][k] = m * rc[2] * tl.z;
      Jr[3][k] = m * rc[3] * ta.x; Jr[4][k] = m * rc[4] * ta.y; Jr[5][k] = m * rc[5] * ta.z;
    }
    const real gr = X->rel_gain;
    const real ewr[6] = {rc[0] * (-gr * e_rel.v.x), rc[1] * (-gr * e_rel.v.y), rc[2] * (-gr * e_rel.v.z),
                           rc[3] * (-gr * e_rel.w.x), rc[4] * (-gr * e_rel.w.y), rc[5] * (-gr * e_rel.w.z)};
    diag_rel = X->rel_lm * (ewr[0] * ewr[0] + ewr[1] * ewr[1] + ewr[2] * ewr[2] + ewr[3] * ewr[3] + ewr[4] * ewr[4] +
                            ewr[5] * ewr[5]);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        real sr = 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r) sr += Jr[r][i] * Jr[r][j];
        hstore[tri(i, j) * hstride] = sr;
      }
      real sr = 0.0;
#pragma unroll
      for (int r = 0; r < 6; ++r) sr += ewr[r] * Jr[r][i];
      c_rel[i] = -sr;
    }
    asm volatile("" ::: "memory");
  }
  const Twist e = cj.e;
  M3 Ji, Bm;
  se3_ljacinv_blocks(e, Ji, Bm);
  real cost[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) cost[r] = D->cost[r];
  // Jw = diag(cost) (-jlog) J_b, in place (frame_task.py:146, task.py:125-127)
  real (&Jw)[6][NC] = cj.J;
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const V3 bl = v3(Jw[0][k], Jw[1][k], Jw[2][k]);
    const V3 ba = v3(Jw[3][k], Jw[4][k], Jw[5][k]);
    const V3 tl = mul(Ji, bl) + mul(Bm, ba);
    const V3 ta = mul(Ji, ba);
    Jw[0][k] = -cost[0] * tl.x; Jw[1][k] = -cost[1] * tl.y; Jw[2][k] = -cost[2] * tl.z;
    Jw[3][k] = -cost[3] * ta.x; Jw[4][k] = -cost[4] * ta.y; Jw[5][k] = -cost[5] * ta.z;
  }
  const real g = D->gain;
  real ew[6] = {cost[0] * (-g * e.v.x), cost[1] * (-g * e.v.y), cost[2] * (-g * e.v.z),
                  cost[3] * (-g * e.w.x), cost[4] * (-g * e.w.y), cost[5] * (-g * e.w.z)};
  real diag = damping + D->lm * (ew[0] * ew[0] + ew[1] * ew[1] + ew[2] * ew[2] + ew[3] * ew[3] +
                                   ew[4] * ew[4] + ew[5] * ew[5]);
  if (REL) diag += diag_rel;
  // Levenberg-Marquardt terms of posture tasks need the error over every dof (rare).
  if (D->any_posture_lm) {
    for (int t = 0; t < D->nposture; ++t) {
      const MkPosture& P = D->posture[t];
      if (P.lm > 0.0) {
        real s2 = 0.0;
        for (int i = 0; i < D->nv; ++i) {
          if (D->dof[i].kind != MK_DOF_SCALAR) continue;
          real ewi = P.cost[i] * (-P.gain * (P.target[i] - row[D->dof[i].qadr]));
          s2 += ewi * ewi;
        }
        diag += P.lm * s2;
      }
    }
  }

  // one row of H per slot, together with the slot's posture terms and box bounds; padding
  // slots carry W = 1, A = B = 0 and infinite bounds: a decoupled entry with solution 0
  MkDofC dc = D->sdof[0];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const MkDofC nx = D->sdof[i + 1 < NC ? i + 1 : i];   // requested one slot ahead
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      real s = 0.0;
#pragma unroll
      for (int r = 0; r < 6; ++r) s += Jw[r][i] * Jw[r][j];
      if (j == i) s += diag + dc.W;
      if (REL) s += hstore[tri(i, j) * hstride];
      // straight to the lane's LDS column: the solver re-reads H from there, and the
      // registers are needed for the factor
#if defined(MK_ABLATE) && MK_ABLATE == 2
      hacc += s;
#else
      hstore[tri(i, j) * hstride] = s;
#endif
    }
    real s = 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r) s += ew[r] * Jw[r][i];
    c[i] = dc.B * qc[i] - dc.A - s;
    if (REL) c[i] += c_rel[i];
    const real vb = dt * dc.vmax;
    hi[i] = fmin(dc.gain * (dc.upper - qc[i]), vb);
    lo[i] = fmax(-(dc.gain * (qc[i] - dc.lower)), -vb);
    MK_SCHED_FENCE();
    MK_PIN3(c[i], hi[i], lo[i]);
    dc = nx;
  }
  return diag;
}

// Builds and solves the QP of one problem from its chain kinematics. `row` is the
// lane's q row in LDS. On return x holds dq of the chain dofs; the other dofs are
// handled in place:
//   INTEGRATE == false: row[dof] <- v[dof] for every non-chain dof (q is consumed)
//   INTEGRATE == true : row[qadr] <- q (+) dt v for every non-chain dof, and their
//                       velocity goes to vrow[dof] when vrow is not null
template <int NC, bool INTEGRATE, bool REL = false>
MK_DEV int solve_problem(const MkFusedDesc* __restrict__ D, real* row, const real (&qc)[NC],
                         ChainJac<NC>& cj, real dt, real inv_dt, real damping, real (&x)[NC],
                         double* vrow, uint32_t& act, uint32_t& upper, bool warm, real* hstore,
                         int hstride, bool input_bad = false, const MkFusedExt* __restrict__ X = nullptr,
                         Twist e_rel = Twist{V3{0, 0, 0}, V3{0, 0, 0}}) {
  real c[NC], lo[NC], hi[NC];
  real hacc = 0.0;
  const real diag = assemble_chain_qp<NC, REL>(D, row, qc, cj, dt, damping, c, lo, hi, hstore, hstride, X, e_rel, hacc);
  MkDofC dc;

#if defined(MK_ABLATE) && MK_ABLATE == 2
  // tuning build only: skip the QP, keep the assembled H / c / bounds alive
  int status = 0;
  {
#pragma unroll
    for (int i = 0; i < NC; ++i) x[i] = fmin(fmax(-c[i] * 1e-3 + 1e-30 * hacc, lo[i]), hi[i]);
  }
#else
  // keep the compiler from forwarding the Hessian stores back into registers
  asm volatile("" ::: "memory");
  int status = box_qp<NC>(HView{hstore, hstride}, c, lo, hi, x, act, upper, warm);
#endif
  {
    // non-finite chain q or target: the task error, hence every entry of c, is NaN
    // (input_bad: a non-finite coordinate outside the chain, found by the caller)
    // (also: damping = 0 with a vanishing Levenberg-Marquardt term leaves the dofs that carry
    // no task cost with a zero Hessian entry -- the reference's QP backend rejects such a
    // singular P, mink/solve_ik.py:101-103)
    // (float32 build also: the solver lost a pivot of the factorisation to rounding)
    if (input_bad || !(fabs(diag + c[0]) < INFINITY) || (D->needs_diag && !(diag > 0.0)) || (MK_IS_F32 && status == 3)) {
      status = 3;
#pragma unroll
      for (int i = 0; i < NC; ++i) x[i] = 0.0;
    }
  }

  // ---- dofs outside the chain: diagonal Hessian, box limits -> clamp ----
  // (the constants of the next dof are fetched while the current one is computed)
  const int nd = D->nd;
  dc = D->ddof[0];
  if (status == 3) {
    // flagged problem: nothing moves (rare, so the common path pays one branch)
    for (int d = 0; d < nd; ++d) {
      const MkDofC z = D->ddof[d];
      if (!INTEGRATE) row[z.dof] = 0.0; else if (vrow) vrow[z.dof] = 0.0;
    }
  } else
  for (int d = 0; d < nd; ++d) {
    const MkDofC nx = D->ddof[d + 1 < MK_MAX_NV ? d + 1 : d];
    const real qi = row[dc.qadr];
    const real vb = dt * dc.vmax;
    const real h = fmin(dc.gain * (dc.upper - qi), vb);
    const real l = fmax(-(dc.gain * (qi - dc.lower)), -vb);
    if (!(l <= h)) status = 2;
    const real xi = fmin(fmax(-(dc.B * qi - dc.A) * mk_rcp(diag + dc.W), l), h);
    const real vi = xi * inv_dt;
    if (INTEGRATE) {
      row[dc.qadr] = qi + dt * vi;
      if (vrow) vrow[dc.dof] = (double)vi;
    } else {
      row[dc.dof] = vi;
    }
    dc = nx;
  }
  // free-joint dofs: zero Jacobian columns and zeroed posture rows -> v = 0
  for (int f = 0; f < D->nfree; ++f) {
    for (int k = 0; k < 6; ++k) {
      if (!INTEGRATE) row[D->free_dofadr[f] + k] = 0.0; else if (vrow) vrow[D->free_dofadr[f] + k] = 0.0;
    }
  }
  if (INTEGRATE) {
    // mj_integratePos renormalises free-joint quaternions even at zero velocity.
    for (int f = 0; f < D->nfree; ++f) {
      real* qq = row + D->free_qadr[f] + 3;
      Q4 nq = qnormalize(Q4{qq[0], qq[1], qq[2], qq[3]});
      qq[0] = nq.w; qq[1] = nq.x; qq[2] = nq.y; qq[3] = nq.z;
    }
  }
  return status;
}

// Coalesced copy between a [rows, width] global block and the per-lane LDS rows.
template <bool TO_LDS>
MK_DEV void tile_copy(real* tile, int stride, double* g, int rows, int width) {
  const int lane = threadIdx.x;
  int r = lane / width, cidx = lane % width;
  const int dr = 64 / width, dc = 64 % width;
  const int total = rows * width;
  for (int i = lane; i < total; i += 64) {
    if (TO_LDS) tile[r * stride + cidx] = g[i]; else g[i] = (double)tile[r * stride + cidx];
    r += dr;
    cidx += dc;
    if (cidx >= width) { cidx -= width; ++r; }
  }
}

template <int NC>
__global__ MK_KERNEL_ATTR void fused_step_kernel(const MkFusedDesc* __restrict__ D,
                                                        const double* __restrict__ q,
                                                        const double* __restrict__ targets, int target_stride,
                                                        real dt, real damping, int B,
                                                        double* __restrict__ v_out, int32_t* __restrict__ status_out) {
  extern __shared__ real tile[];
  const int nq = D->nq, nv = D->nv;
  const int stride = nq | 1;
  const real inv_dt = 1.0 / dt;
  const int lane = threadIdx.x;
  const long base = (long)blockIdx.x * 64;
  const int rows = (int)min((long)64, (long)B - base);
  tile_copy<true>(tile, stride, const_cast<double*>(q) + base * nq, rows, nq);
  __syncthreads();
  const bool valid = lane < rows;
  real* row = tile + lane * stride;
  real* hlds = tile + 64 * stride + lane;   // packed Hessian, element e at hlds[e * 64]
  if (!valid) {
    for (int i = 0; i < nq; ++i) row[i] = tile[i];
  }
  const long pb = valid ? base + lane : base;
  const double* tg = targets + pb * target_stride + D->target_off;
  Q4 Qt = Q4{tg[0], tg[1], tg[2], tg[3]};
  V3 pt = v3(tg[4], tg[5], tg[6]);
  real qc[NC], x[NC];
  load_chain_q<NC>(D, row, qc);
  ChainJac<NC> cj;
  chain_jac<NC>(D, qc, Qt, pt, cj);
  uint32_t act = 0u, upper = 0u;
  int status = solve_problem<NC, false>(D, row, qc, cj, dt, inv_dt, damping, x, nullptr, act, upper, false,
                                        hlds, 64, rest_nonfinite(D, row));
#pragma unroll
  for (int k = 0; k < NC; ++k)
    if (D->seg_dof[k] >= 0) row[D->seg_dof[k]] = x[k] * inv_dt;
  if (valid) status_out[pb] = status;
  __syncthreads();
  tile_copy<false>(tile, stride, v_out + base * nv, rows, nv);
}

// Per-lane copy of n doubles with the loads of eight elements in flight together (a plain
// loop waits for every element's round trip in turn; nothing else hides that latency at one
// wave per SIMD).
template <typename TD, typename TS>
MK_DEV void lane_copy(TD* __restrict__ dst, const TS* __restrict__ src, int n) {
  int i = 0;
  for (; i + 8 <= n; i += 8) {
    TS t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = src[i + k];
#pragma unroll
    for (int k = 0; k < 8; ++k) dst[i + k] = (TD)t[k];
  }
  TS t[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) t[k] = (i + k < n) ? src[i + k] : TS(0.0);
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (i + k < n) dst[i + k] = (TD)t[k];
}

#if !defined(MK_F32)
// Pose composition helpers of the collision screen.
struct PoseQT {
  Q4 q;
  V3 t;
};
MK_DEV PoseQT pose_then(PoseQT a, Q4 bq, V3 bt) { return PoseQT{qmul(a.q, bq), a.t + qrot(a.q, bt)}; }

// World pose of the frame behind chain joint `slot` (forward composition of the canonical
// segments 0..slot at the lane's q): where a collision body on an inner link of the chain hangs.
MK_DEV PoseQT chain_frame_pose(const MkFusedDesc* __restrict__ D, const real* row, int slot) {
  PoseQT p = PoseQT{Q4{1, 0, 0, 0}, v3(0, 0, 0)};
  for (int i = 0; i <= slot; ++i) {
    const MkSegC& S = D->segc[i];
    p = pose_then(p, Q4{S.pq[0], S.pq[1], S.pq[2], S.pq[3]}, v3(S.pp[0], S.pp[1], S.pp[2]));
    const real qv = D->seg_dof[i] >= 0 ? row[D->seg_qadr[i]] : 0.0;
    real sn, cs;
    mk_sincos(0.5 * (S.hinge * qv), &sn, &cs);
    p = pose_then(p, Q4{cs, 0.0, 0.0, sn}, v3(0.0, 0.0, S.slide * qv));
  }
  return p;
}

// True when some CollisionAvoidanceLimit pair of the problem may be within its detection
// distance: conservative lower bounds of the pair distances (bounding spheres; the support
// extent of a box against a plane; the separating-axis gaps of two boxes) -- the same bounds
// behind which the general kernel skips a pair, so a problem that passes has no finite
// collision row there either.  (Qw, tw) is the world pose of the task frame, `row` the
// lane's q.
MK_DEV bool collision_in_range(const MkFusedDesc* __restrict__ D, const MkFusedExt* __restrict__ X, const real* row,
                               Q4 Qw, V3 tw, real* scr, int ss) {
  // frame behind the chain's last joint: task frame o post^-1
  const Q4 postq = Q4{D->post_quat[0], D->post_quat[1], D->post_quat[2], D->post_quat[3]};
  PoseQT tip;
  tip.q = qmul(Qw, qconj(postq));
  tip.t = tw - qrot(tip.q, v3(D->post_pos[0], D->post_pos[1], D->post_pos[2]));
  const int np = X->ncpair;
  // Coarse bounds first, from the positions the lane has already: a geom lies within G.reach of
  // the frame that carries its body.  A wave none of whose problems has a pair inside these
  // bounds skips the poses of the bodies and geoms below (for TidyBot, four gripper pads behind
  // three or four finger joints each).
  {
    bool maybe = false;
    auto anchor = [&](const MkCollGeom& G) -> V3 {
      const MkCollBody& CB = X->cbody[G.body];
      if (CB.kind == MK_CB_TIP) return tip.t;
      if (CB.kind == MK_CB_FREE) {
        const real* f = row + CB.free_qadr;
        return v3(f[0], f[1], f[2]);
      }
      return v3(G.wc[0], G.wc[1], G.wc[2]);
    };
    for (int k = 0; k < np; ++k) {
      const MkCollPair& P = X->cpair[k];
      const MkCollGeom& G1 = X->cgeom[P.g1];
      const MkCollGeom& G2 = X->cgeom[P.g2];
      if (G1.reach < 0.0 || G2.reach < 0.0) {   // (the same for all lanes)
        maybe = true;
        continue;
      }
      real lower;
      if (G1.type == 0) {
        lower = dot(anchor(G2) - v3(G1.wc[0], G1.wc[1], G1.wc[2]), v3(G1.wn[0], G1.wn[1], G1.wn[2])) - G2.reach;
      } else {
        const V3 d = anchor(G2) - anchor(G1);
        lower = sqrt(dot(d, d)) - G1.reach - G2.reach;
      }
      maybe |= !(lower >= P.d_detect);   // also true for a non-finite bound
    }
    if (mk_wave_none(maybe)) return false;
  }
  // per geom the world centre of its bounding sphere (element 3 g + c of the lane's scratch column,
  // scr[e * ss]); behind them the orientation of the planes and boxes (4 each, by G.oslot)
  const int nb = X->ncbody, ng = X->ncgeom;
  for (int b = 0; b < nb; ++b) {
    const MkCollBody& CB = X->cbody[b];
    PoseQT p = PoseQT{Q4{1, 0, 0, 0}, v3(0, 0, 0)};
    if (CB.kind == MK_CB_TIP) {
      p = CB.slot >= D->nc - 1 ? tip : chain_frame_pose(D, row, CB.slot);
      for (int sidx = 0; sidx < CB.nseg; ++sidx) {
        const MkCollSeg& S = CB.seg[sidx];
        p = pose_then(p, Q4{S.pq[0], S.pq[1], S.pq[2], S.pq[3]}, v3(S.pp[0], S.pp[1], S.pp[2]));
        const real qv = row[S.qadr];
        real sn, cs;
        mk_sincos(0.5 * (S.hinge * qv), &sn, &cs);
        p = pose_then(p, Q4{cs, 0.0, 0.0, sn}, v3(0.0, 0.0, S.slide * qv));
      }
    } else if (CB.kind == MK_CB_FREE) {
      const real* f = row + CB.free_qadr;
      p.t = v3(f[0], f[1], f[2]);
      p.q = qnormalize(Q4{f[3], f[4], f[5], f[6]});
    }
    p = pose_then(p, Q4{CB.tq[0], CB.tq[1], CB.tq[2], CB.tq[3]}, v3(CB.tp[0], CB.tp[1], CB.tp[2]));
    for (int g = CB.g0; g < CB.g0 + CB.ng; ++g) {
      const MkCollGeom& G = X->cgeom[g];
      const PoseQT w = pose_then(p, Q4{G.lq[0], G.lq[1], G.lq[2], G.lq[3]}, v3(G.lp[0], G.lp[1], G.lp[2]));
      const V3 c = w.t + qrot(w.q, v3(G.bc[0], G.bc[1], G.bc[2]));
      real* sc = scr + 3 * g * ss;
      sc[0] = c.x; sc[ss] = c.y; sc[2 * ss] = c.z;
      if (G.oslot >= 0) {
        real* so = scr + (3 * ng + 4 * G.oslot) * ss;
        so[0] = w.q.w; so[ss] = w.q.x; so[2 * ss] = w.q.y; so[3 * ss] = w.q.z;
      }
    }
  }
  bool near = false;
  for (int k = 0; k < np; ++k) {
    const int g1 = X->cpair[k].g1, g2 = X->cpair[k].g2;
    const real d_detect = X->cpair[k].d_detect;
    const MkCollGeom& G1 = X->cgeom[g1];
    const MkCollGeom& G2 = X->cgeom[g2];
    // centres of the bounding spheres (the geom centre but for meshes); orientations where kept
    const real *s1 = scr + 3 * g1 * ss, *s2 = scr + 3 * g2 * ss;
    const V3 ac = v3(s1[0], s1[ss], s1[2 * ss]), bcw = v3(s2[0], s2[ss], s2[2 * ss]);
    PoseQT a = PoseQT{Q4{1, 0, 0, 0}, ac}, b = PoseQT{Q4{1, 0, 0, 0}, bcw};
    if (G1.oslot >= 0) {
      const real* so = scr + (3 * ng + 4 * G1.oslot) * ss;
      a.q = Q4{so[0], so[ss], so[2 * ss], so[3 * ss]};
    }
    if (G2.oslot >= 0) {
      const real* so = scr + (3 * ng + 4 * G2.oslot) * ss;
      b.q = Q4{so[0], so[ss], so[2 * ss], so[3 * ss]};
    }
    real lower;
    if (G1.type == 0) {   // plane: height of the other geom's centre minus its extent along the normal
      const M3 R1 = q2m(a.q);
      const V3 n = v3(R1.m[2], R1.m[5], R1.m[8]);
      real ext = G2.rbound;
      if (G2.type == 6) {
        const M3 R2 = q2m(b.q);
        const V3 nl = mulT(R2, n);
        ext = fabs(nl.x) * G2.size[0] + fabs(nl.y) * G2.size[1] + fabs(nl.z) * G2.size[2];
      }
      lower = dot(bcw - a.t, n) - ext;
    } else {
      const V3 d = bcw - ac;
      lower = sqrt(dot(d, d)) - G1.rbound - G2.rbound;
      if (G1.type == 6 && G2.type == 6 && lower < d_detect) {
        // two boxes: the largest gap along the 15 candidate separating axes, in the frame of box 1
        // (C = R1'R2, t = R1'd: an axis u has extent sum_k |R'u|_k h_k on a box, and for the
        // face normals and the edge cross products R1'u and R2'u are columns, rows and cross
        // products of unit vectors with columns of C)
        const M3 R1 = q2m(a.q), R2 = q2m(b.q);
        const real h1[3] = {G1.size[0], G1.size[1], G1.size[2]}, h2[3] = {G2.size[0], G2.size[1], G2.size[2]};
        real C[3][3], t[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const V3 ci = v3(R1.m[i], R1.m[3 + i], R1.m[6 + i]);
          t[i] = dot(ci, d);
#pragma unroll
          for (int j = 0; j < 3; ++j) C[i][j] = dot(ci, v3(R2.m[j], R2.m[3 + j], R2.m[6 + j]));
        }
        real gap = -INFINITY;
#pragma unroll
        for (int i = 0; i < 3; ++i)   // face normals of box 1
          gap = fmax(gap, fabs(t[i]) - h1[i] - (h2[0] * fabs(C[i][0]) + h2[1] * fabs(C[i][1]) + h2[2] * fabs(C[i][2])));
#pragma unroll
        for (int j = 0; j < 3; ++j)   // face normals of box 2
          gap = fmax(gap, fabs(t[0] * C[0][j] + t[1] * C[1][j] + t[2] * C[2][j]) - h2[j] -
                              (h1[0] * fabs(C[0][j]) + h1[1] * fabs(C[1][j]) + h1[2] * fabs(C[2][j])));
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
          for (int j = 0; j < 3; ++j) {   // edge i of box 1 x edge j of box 2; |e_i x C_j|^2 without cancellation
            const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            const real n2 = C[i1][j] * C[i1][j] + C[i2][j] * C[i2][j];
            const real num = fabs(t[i2] * C[i1][j] - t[i1] * C[i2][j]) - (h1[i1] * fabs(C[i2][j]) + h1[i2] * fabs(C[i1][j])) -
                             (h2[j1] * fabs(C[i][j2]) + h2[j2] * fabs(C[i][j1]));
            if (n2 >= 1e-12) gap = fmax(gap, num * mk_rsqrt(n2));
          }
        }
        lower = fmax(lower, gap);
      }
    }
    near |= !(lower >= d_detect);   // also true for a non-finite bound
  }
  return near;
}

// ---- general rows in the lane-per-problem form ------------------------------------------
// CollisionAvoidanceLimit rows of one lane's problem, bound h[k]: row k over all nv dofs, element
// (k, dof) at g[(k * nv + dof) * stride].  With at most two rows the entries at the chain's dofs
// are kept apart, slot i of the chain (dof seg_dof[i]) at gc[(k * ncs + i) * cstride]: the solve
// reads the chain part in every sweep, and the kernel then gives it the faster memory; row k's
// other entries are those of row idx[k] of the storage g (the rows of the problem as they were
// built, of which the solve takes a selection).
template <int M>
struct LaneRows {
  static constexpr bool kSplit = M <= 2;
  real* g;
  real* gc;
  const int* seg_dof;
  int stride, cstride, nv, ncs, n;
  int nzero;   // rows [0, nzero) of the storage are zero on entry of collision_rows; it clears the others as it takes them
  int idx[M];
  real h[M];
  MK_DEV real& at(int k, int dof) const { return g[((kSplit ? idx[k] : k) * nv + dof) * stride]; }
  // entry of row k at chain slot i (zero for a slot without a dof)
  MK_DEV real chain(int k, int i) const {
    if constexpr (kSplit) {
      return gc[(k * ncs + i) * cstride];
    } else {
      const int dof = seg_dof[i];
      return dof >= 0 ? g[(k * nv + dof) * stride] : real(0.0);
    }
  }
  MK_DEV void chain_add(int k, int i, real val) const {   // (slot i has a dof)
    if constexpr (kSplit) gc[(k * ncs + i) * cstride] += val;
    else g[(k * nv + seg_dof[i]) * stride] += val;
  }
};

// World pose of collision body b; `tip` is the frame behind the chain's last joint.
MK_DEV PoseQT coll_body_pose(const MkFusedDesc* __restrict__ D, const MkFusedExt* __restrict__ X, const real* row,
                             PoseQT tip, int b) {
  const MkCollBody& CB = X->cbody[b];
  PoseQT p = PoseQT{Q4{1, 0, 0, 0}, v3(0, 0, 0)};
  if (CB.kind == MK_CB_TIP) {
    p = CB.slot >= D->nc - 1 ? tip : chain_frame_pose(D, row, CB.slot);
    for (int sidx = 0; sidx < CB.nseg; ++sidx) {
      const MkCollSeg& S = CB.seg[sidx];
      p = pose_then(p, Q4{S.pq[0], S.pq[1], S.pq[2], S.pq[3]}, v3(S.pp[0], S.pp[1], S.pp[2]));
      const real qv = row[S.qadr];
      real sn, cs;
      mk_sincos(0.5 * (S.hinge * qv), &sn, &cs);
      p = pose_then(p, Q4{cs, 0.0, 0.0, sn}, v3(0.0, 0.0, S.slide * qv));
    }
  } else if (CB.kind == MK_CB_FREE) {
    const real* f = row + CB.free_qadr;
    p.t = v3(f[0], f[1], f[2]);
    p.q = qnormalize(Q4{f[3], f[4], f[5], f[6]});
  }
  return pose_then(p, Q4{CB.tq[0], CB.tq[1], CB.tq[2], CB.tq[3]}, v3(CB.tp[0], CB.tp[1], CB.tp[2]));
}

// Adds s * n' Jp(point) to row m, Jp the translational Jacobian of `point` rigidly attached to
// collision body b (mj_jac as used by compute_contact_normal_jacobian,
// mink/limits/collision_avoidance_limit.py:83-110).  On the chain dofs the body-frame Jacobian of
// the task frame (J, still unweighted; task frame at (tw, Rw)) gives the motion of every point
// behind a joint: n' Rw (v_j + w_j x r) = (Rw'n).v_j + (r x Rw'n).w_j with r = Rw'(point - tw).
template <int NC, int M>
MK_DEV void add_point_row(const MkFusedDesc* __restrict__ D, const MkFusedExt* __restrict__ X, const real* row,
                          const real (&J)[6][NC], V3 tw, const M3& Rw, PoseQT tip, int b, V3 point, V3 n, real s,
                          const LaneRows<M>& R, int m) {
  const MkCollBody& CB = X->cbody[b];
  if (CB.kind == MK_CB_TIP) {
    // (a joint moves every point behind it the same way whichever link the point is on, so the
    // task frame's columns serve any chain body: the columns of the joints behind its slot vanish)
    const V3 nt = mulT(Rw, n), r = mulT(Rw, point - tw);
    const V3 mt = cross(r, nt);
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int dof = D->seg_dof[j];
      const real val = nt.x * J[0][j] + nt.y * J[1][j] + nt.z * J[2][j] + mt.x * J[3][j] + mt.y * J[4][j] + mt.z * J[5][j];
      if (dof >= 0 && j <= CB.slot) R.chain_add(m, j, s * val);
    }
    PoseQT f = CB.slot >= D->nc - 1 ? tip : chain_frame_pose(D, row, CB.slot);
    for (int sidx = 0; sidx < CB.nseg; ++sidx) {
      const MkCollSeg& S = CB.seg[sidx];
      f = pose_then(f, Q4{S.pq[0], S.pq[1], S.pq[2], S.pq[3]}, v3(S.pp[0], S.pp[1], S.pp[2]));
      const V3 z = qrot(f.q, v3(0, 0, 1));
      const real val = S.hinge * dot(n, cross(z, point - f.t)) + S.slide * dot(n, z);
      R.at(m, S.dof) += s * val;
      const real qv = row[S.qadr];
      real sn, cs;
      mk_sincos(0.5 * (S.hinge * qv), &sn, &cs);
      f = pose_then(f, Q4{cs, 0.0, 0.0, sn}, v3(0.0, 0.0, S.slide * qv));
    }
  } else if (CB.kind == MK_CB_FREE) {
    // free joint: translation along the world axes, rotation about the body's own axes
    const real* fq = row + CB.free_qadr;
    const M3 Rb = q2m(qnormalize(Q4{fq[3], fq[4], fq[5], fq[6]}));
    const V3 w = mulT(Rb, cross(point - v3(fq[0], fq[1], fq[2]), n));
    const int d0 = CB.free_dofadr;
    R.at(m, d0) += s * n.x; R.at(m, d0 + 1) += s * n.y; R.at(m, d0 + 2) += s * n.z;
    R.at(m, d0 + 3) += s * w.x; R.at(m, d0 + 4) += s * w.y; R.at(m, d0 + 5) += s * w.z;
  }
}

// Rows of the CollisionAvoidanceLimit pairs of one problem that are within their detection
// distance (mink/limits/collision_avoidance_limit.py:187-210): G[k] = -n'(Jp2 - Jp1),
// h[k] = gain (dist - d_min) / dt + relax (relax at dist <= d_min).  Rows [0, R.nzero) of R.g must be zero on entry.
// J is the unweighted body-frame Jacobian of the task frame at (Qw, tw, Rw).  Returns false when
// the lane form cannot hold the problem: more than MK_LANE_ROWS rows in range, or a pair without a
// closed form (ellipsoid, cylinder, mesh) that may be in range.
template <int NC, int M>
MK_DEV bool collision_rows(const MkFusedDesc* __restrict__ D, const MkFusedExt* __restrict__ X, const real* row,
                           const real (&J)[6][NC], Q4 Qw, V3 tw, const M3& Rw, real dt, LaneRows<M>& R) {
  const Q4 postq = Q4{D->post_quat[0], D->post_quat[1], D->post_quat[2], D->post_quat[3]};
  PoseQT tip;
  tip.q = qmul(Qw, qconj(postq));
  tip.t = tw - qrot(tip.q, v3(D->post_pos[0], D->post_pos[1], D->post_pos[2]));
  bool ok = true;
  const int np = X->ncpair;
  // pose of the body walked last from the chain (consecutive pairs usually share it; the body
  // index is the same for all lanes, so the test is a scalar branch)
  int cached = -1;
  PoseQT cpose = tip;
  auto body_pose = [&](int b) -> PoseQT {
    if (X->cbody[b].kind != MK_CB_TIP) return coll_body_pose(D, X, row, tip, b);
    if (b != cached) {
      cpose = coll_body_pose(D, X, row, tip, b);
      cached = b;
    }
    return cpose;
  };
  for (int k = 0; k < np; ++k) {
    const MkCollPair P = X->cpair[k];
    const MkCollGeom& G1 = X->cgeom[P.g1];
    const MkCollGeom& G2 = X->cgeom[P.g2];
    const PoseQT a = pose_then(body_pose(G1.body), Q4{G1.lq[0], G1.lq[1], G1.lq[2], G1.lq[3]},
                               v3(G1.lp[0], G1.lp[1], G1.lp[2]));
    const PoseQT b = pose_then(body_pose(G2.body), Q4{G2.lq[0], G2.lq[1], G2.lq[2], G2.lq[3]},
                               v3(G2.lp[0], G2.lp[1], G2.lp[2]));
    const M3 R1 = q2m(a.q);
    const V3 s2 = v3(G2.size[0], G2.size[1], G2.size[2]);
    // bounding spheres first (the bound behind which the general kernel skips the pair, too)
    const V3 bcw = b.t + qrot(b.q, v3(G2.bc[0], G2.bc[1], G2.bc[2]));
    real lower;
    if (G1.type == 0) {
      lower = dot(bcw - a.t, col(R1, 2)) - G2.rbound;
    } else {
      const V3 d = bcw - (a.t + qrot(a.q, v3(G1.bc[0], G1.bc[1], G1.bc[2])));
      lower = sqrt(dot(d, d)) - G1.rbound - G2.rbound;
    }
    if (lower >= P.d_detect) continue;
#if defined(MK_ROWS_ABLATE) && MK_ROWS_ABLATE == 1
    continue;   // tuning build only: no exact distances, so no rows
#endif
    const bool closed1 = G1.type == 0 || G1.type == 2 || G1.type == 3 || G1.type == 6;
    const bool closed2 = G2.type == 2 || G2.type == 3 || G2.type == 6;
    if (!closed1 || !closed2) { ok = false; continue; }   // support-mapping pair: the general kernel
    Contact ct;
    const bool active = geom_distance_lane(G1.type, a.t, R1, v3(G1.size[0], G1.size[1], G1.size[2]), G2.type, b.t,
                                           q2m(b.q), s2, P.d_detect, ct);
    if (!active) continue;
#if defined(MK_ROWS_ABLATE) && MK_ROWS_ABLATE == 2
    // tuning build only: the exact distances are computed and dropped
    asm volatile("" ::"v"(ct.dist), "v"(ct.p1.x), "v"(ct.p1.y), "v"(ct.p1.z), "v"(ct.p2.x), "v"(ct.p2.y), "v"(ct.p2.z));
    continue;
#endif
    if (R.n >= M) { ok = false; continue; }
    const int m = R.n++;
    if (m >= R.nzero)
      for (int dof = 0; dof < R.nv; ++dof) R.at(m, dof) = 0.0;
    const V3 dn = ct.p2 - ct.p1;
    const real nn = sqrt(dot(dn, dn));
    const V3 nrm = nn < MK_MINVAL ? v3(1, 0, 0) : (1.0 / nn) * dn;
    const real hval = ct.dist > P.d_min ? P.gain * (ct.dist - P.d_min) / dt + P.relax : P.relax;
#pragma unroll
    for (int i = 0; i < M; ++i)
      if (i == m) R.h[i] = hval;
    add_point_row<NC, M>(D, X, row, J, tw, Rw, tip, G2.body, ct.p2, nrm, -1.0, R, m);
    add_point_row<NC, M>(D, X, row, J, tw, Rw, tip, G1.body, ct.p1, nrm, 1.0, R, m);
  }
  return ok;
}

// min 1/2 v'Hv + c'v  s.t. box bounds, G v <= h for the structured problem: H = the dense chain
// block (LDS) + a diagonal on every other dof, at most MK_LANE_ROWS general rows.
//
// Dual iteration on the multipliers mu >= 0 of the general rows.  For fixed mu the problem is the
// box-constrained one with linear term c + G'mu, which separates as before (box_qp on the chain
// block, clamps elsewhere) and is solved exactly.  Its solution is piecewise linear in mu: on the
// piece given by the current active box bounds, the row residuals move as r(mu') = r - S (mu' - mu)
// with S = G Z (Z'HZ)^-1 Z'G' over the free dofs Z.  The step solves the linear complementarity
// problem mu' >= 0, r(mu') <= 0, mu'.r(mu') = 0 of that piece exactly (enumeration of the 2^rows
// active sets), then the box problem is solved again.  When the Karush-Kuhn-Tucker conditions
// of the rows hold at a box solution, that solution is the optimum of the whole QP.  The iteration
// visits pieces and normally ends in one to three steps; a step that lands on another piece
// without reducing the natural residual sum_k min(mu_k, -r_k)^2 is halved.  On a piece where every dof of a
// violated row sits at a bound, S has a zero and the piece gives no step; the row's multiplier
// then moves to the point where the first of those bounds lets go (a ratio test on the bound
// multipliers).  Termination is not guaranteed: the caller hands a problem that does not finish
// in kOuter steps to the general kernel.
// Returns true with x (chain dofs) and the other dofs' dq in vrow[dof] (already divided by dt).
template <int NC, int M>
MK_DEV bool rows_qp(const MkFusedDesc* __restrict__ D, const real* row, HView H, const real (&c)[NC],
                    const real (&lo)[NC], const real (&hi)[NC], real diag, real dt, real inv_dt,
                    const LaneRows<M>& R, real (&x)[NC], real* vrow) {
  constexpr int kOuter = 16;
  constexpr real kTol = 1e-13;
  real mu[M], mu_prev[M], merit_prev = INFINITY;
#pragma unroll
  for (int k = 0; k < M; ++k) mu[k] = mu_prev[k] = 0.0;
  uint32_t act = 0u, upper = 0u;
  bool done = false, good = false;
  const int nd = D->nd, nrows = R.n;
  const real rdiag = mk_rcp(diag);
  for (int it = 0; it < kOuter; ++it) {
    if (!done) {
      // ---- box problem at mu ----
      real cm[NC];
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        real sft = 0.0;
#pragma unroll
        for (int k = 0; k < M; ++k) sft += mu[k] * R.chain(k, i);
        cm[i] = c[i] + sft;
      }
      asm volatile("" ::: "memory");
      const int st = box_qp<NC>(H, cm, lo, hi, x, act, upper, it > 0);
      real r[M];
#pragma unroll
      for (int k = 0; k < M; ++k) r[k] = k < nrows ? -R.h[k] : 0.0;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
#pragma unroll
        for (int k = 0; k < M; ++k) r[k] += R.chain(k, i) * x[i];
      }
      uint64_t clamped = 0ull;
      bool infeasible = st != 0;
#pragma unroll 4
      for (int d = 0; d < nd; ++d) {
        const MkDofC dc = D->ddof[d];
        const real qi = row[dc.qadr];
        const real vb = dt * dc.vmax;
        const real h = fmin(dc.gain * (dc.upper - qi), vb);
        const real l = fmax(-(dc.gain * (qi - dc.lower)), -vb);
        infeasible |= !(l <= h);
        real g[M], sft = 0.0;
#pragma unroll
        for (int k = 0; k < M; ++k) { g[k] = R.at(k, dc.dof); sft += mu[k] * g[k]; }
        const real raw = -(dc.B * qi - dc.A + sft) * mk_rcp(diag + dc.W);
        const real y = fmin(fmax(raw, l), h);
        if (raw < l || raw > h) clamped |= 1ull << d;
#pragma unroll
        for (int k = 0; k < M; ++k) r[k] += g[k] * y;
        vrow[dc.dof] = y * inv_dt;
      }
      for (int f = 0; f < D->nfree; ++f) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int dof = D->free_dofadr[f] + j;
          real g[M], sft = 0.0;
#pragma unroll
          for (int k = 0; k < M; ++k) { g[k] = R.at(k, dof); sft += mu[k] * g[k]; }
          const real z = -sft * rdiag;
#pragma unroll
          for (int k = 0; k < M; ++k) r[k] += g[k] * z;
          vrow[dof] = z * inv_dt;
        }
      }
      // ---- optimality of the rows ----
      bool kkt = true;
#pragma unroll
      for (int k = 0; k < M; ++k) {
        const real tol = kTol * fmax(1.0, fabs(R.h[k]));
        if (k < nrows) kkt &= (mu[k] > 0.0) ? (fabs(r[k]) <= tol) : (r[k] <= tol);
      }
      // natural residual of the rows' complementarity conditions: the step control's merit
      real merit = 0.0;
#pragma unroll
      for (int k = 0; k < M; ++k) {
        const real rho = fmin(mu[k], -r[k]);
        if (k < nrows) merit += rho * rho;
      }
      if (infeasible) {
        done = true;
      } else if (kkt) {
        done = true;
        good = true;
      } else if (!(merit < merit_prev)) {
        // the step left the piece it was computed on and did not help: halve it
#pragma unroll
        for (int k = 0; k < M; ++k) mu[k] = mu_prev[k] + 0.5 * (mu[k] - mu_prev[k]);
      } else {
        merit_prev = merit;
#pragma unroll
        for (int k = 0; k < M; ++k) mu_prev[k] = mu[k];
        // ---- S on the current piece ----
        real L[NC * (NC + 1) / 2], Ld[NC];
        asm volatile("" ::: "memory");
        chol_masked<NC>(H, act, L, Ld);
        real S[M][M];
#pragma unroll
        for (int k = 0; k < M; ++k)
#pragma unroll
          for (int l = 0; l < M; ++l) S[k][l] = 0.0;
#pragma unroll
        for (int k = 0; k < M; ++k) {
          real sk[NC];
#pragma unroll
          for (int i = 0; i < NC; ++i) sk[i] = ((act >> i) & 1u) ? 0.0 : R.chain(k, i);
          chol_solve<NC>(L, Ld, sk);
#pragma unroll
          for (int i = 0; i < NC; ++i) {
            const bool fr = !((act >> i) & 1u);
#pragma unroll
            for (int l = 0; l < M; ++l)
              if (l >= k) S[k][l] += (fr ? R.chain(l, i) : 0.0) * sk[i];
          }
        }
#pragma unroll 4
        for (int d = 0; d < nd; ++d) {
          const MkDofC dc = D->ddof[d];
          const real iw = ((clamped >> d) & 1ull) ? 0.0 : mk_rcp(diag + dc.W);
          real g[M];
#pragma unroll
          for (int k = 0; k < M; ++k) g[k] = R.at(k, dc.dof);
#pragma unroll
          for (int k = 0; k < M; ++k)
#pragma unroll
            for (int l = 0; l < M; ++l)
              if (l >= k) S[k][l] += g[k] * g[l] * iw;
        }
        for (int f = 0; f < D->nfree; ++f) {
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            const int dof = D->free_dofadr[f] + j;
            real g[M];
#pragma unroll
            for (int k = 0; k < M; ++k) g[k] = R.at(k, dof);
#pragma unroll
            for (int k = 0; k < M; ++k)
#pragma unroll
              for (int l = 0; l < M; ++l)
                if (l >= k) S[k][l] += g[k] * g[l] * rdiag;
          }
        }
#pragma unroll
        for (int k = 0; k < M; ++k)
#pragma unroll
          for (int l = 0; l < k; ++l) S[k][l] = S[l][k];
        // ---- complementarity problem of the piece: mu' >= 0, w = S mu' - b >= 0, mu'.w = 0 ----
        real b[M];
#pragma unroll
        for (int k = 0; k < M; ++k) {
          b[k] = r[k];
#pragma unroll
          for (int l = 0; l < M; ++l) b[k] += S[k][l] * mu[l];
        }
        bool found = false;
        real best[M];
#pragma unroll
        for (int k = 0; k < M; ++k) best[k] = 0.0;
        // the sets are tried starting from the likely one -- the rows that carry a multiplier or are
        // violated now -- so that most lanes are done after the first
        int w0 = 0;
#pragma unroll
        for (int k = 0; k < M; ++k)
          if (k < nrows && (mu[k] > 0.0 || r[k] > 0.0)) w0 |= 1 << k;
        for (int trial = 0; trial < (1 << M); ++trial) {
          if (mk_wave_none(!found && (trial >> nrows) == 0)) continue;   // no lane of the wave needs this set
          if (found || (trial >> nrows) != 0) continue;
          const int w = trial ^ w0;
          // rows outside the set: unit row / column, zero right-hand side
          real A[M][M], rhs[M];
#pragma unroll
          for (int k = 0; k < M; ++k) {
            const bool ink = (w >> k) & 1;
            rhs[k] = ink ? b[k] : 0.0;
#pragma unroll
            for (int l = 0; l < M; ++l) {
              const bool inl = (w >> l) & 1;
              A[k][l] = (ink && inl) ? S[k][l] : (k == l ? 1.0 : 0.0);
            }
          }
          // Cholesky solve of the masked (symmetric positive definite) system
          real Lc[M][M], m3[M];
          bool spd = true;
#pragma unroll
          for (int j = 0; j < M; ++j) {
            real dj = A[j][j];
#pragma unroll
            for (int k = 0; k < j; ++k) dj -= Lc[j][k] * Lc[j][k];
            spd &= dj > 1e-12 * A[j][j];   // dependent rows on this piece
            const real rj = 1.0 / sqrt(fmax(dj, MK_TINY));
            Lc[j][j] = rj;
#pragma unroll
            for (int i = j + 1; i < M; ++i) {
              real t = A[i][j];
#pragma unroll
              for (int k = 0; k < j; ++k) t -= Lc[i][k] * Lc[j][k];
              Lc[i][j] = t * rj;
            }
          }
          if (!spd) continue;
#pragma unroll
          for (int i = 0; i < M; ++i) {
            real t = rhs[i];
#pragma unroll
            for (int k = 0; k < i; ++k) t -= Lc[i][k] * m3[k];
            m3[i] = t * Lc[i][i];
          }
#pragma unroll
          for (int i = M - 1; i >= 0; --i) {
            real t = m3[i];
#pragma unroll
            for (int k = i + 1; k < M; ++k) t -= Lc[k][i] * m3[k];
            m3[i] = t * Lc[i][i];
          }
          bool valid = true;
#pragma unroll
          for (int k = 0; k < M; ++k) {
            const bool ink = (w >> k) & 1;
            real wk = -b[k];
#pragma unroll
            for (int l = 0; l < M; ++l) wk += S[k][l] * m3[l];
            if (ink) valid &= m3[k] >= 0.0;
            else if (k < nrows) valid &= wk >= -kTol * fmax(1.0, fabs(R.h[k]));
          }
          if (valid) {
            found = true;
#pragma unroll
            for (int k = 0; k < M; ++k) best[k] = ((w >> k) & 1) ? m3[k] : 0.0;
          }
        }
        if (!found) {
          // No step on this piece: a violated row whose dofs all sit at bounds has no curvature
          // (S_kk = 0).  Its multiplier then grows until the first of those bounds lets go: with x
          // fixed, the multiplier u_i of an active bound moves at the rate -+G_ki, a clamped dof's
          // unconstrained value at the rate -G_kd / w_d (ratio test).  A little beyond that point
          // the box problem frees the dof and the next piece has curvature.
          asm volatile("" ::: "memory");
          real hx[NC];
          symv<NC>(H, x, hx);
          int kstep = -1;
          real worst = 0.0, tstep = 0.0;
#pragma unroll
          for (int k = 0; k < M; ++k) {
            const real tol = kTol * fmax(1.0, fabs(R.h[k]));
            real tau = INFINITY;
#pragma unroll
            for (int i = 0; i < NC; ++i) {
              const real gki = R.chain(k, i);
              const bool up = (upper >> i) & 1u;
              const real gi = cm[i] + hx[i];
              const real ui = fmax(up ? -gi : gi, 0.0), slope = up ? -gki : gki;
              if (((act >> i) & 1u) && slope < 0.0) tau = fmin(tau, ui / -slope);
            }
            for (int d = 0; d < nd; ++d) {
              const MkDofC dc = D->ddof[d];
              const real qi = row[dc.qadr];
              const real vb = dt * dc.vmax;
              const real h = fmin(dc.gain * (dc.upper - qi), vb);
              const real l = fmax(-(dc.gain * (qi - dc.lower)), -vb);
              real sft = 0.0;
#pragma unroll
              for (int kk = 0; kk < M; ++kk) sft += mu[kk] * R.at(kk, dc.dof);
              const real w = diag + dc.W, gkd = R.at(k, dc.dof);
              const real raw = -(dc.B * qi - dc.A + sft) / w;
              if (raw > h && gkd > 0.0) tau = fmin(tau, (raw - h) * w / gkd);
              if (raw < l && gkd < 0.0) tau = fmin(tau, (l - raw) * w / -gkd);
            }
            // the most violated row moves (one at a time: the others' ratio tests assume x fixed)
            if (k < nrows && r[k] > tol && tau < INFINITY && r[k] > worst) {
              worst = r[k];
              kstep = k;
              tstep = tau;
            }
          }
#pragma unroll
          for (int k = 0; k < M; ++k)
            if (k == kstep) mu[k] += tstep * (1.0 + 1e-6) + 1e-14;
          merit_prev = INFINITY;   // a step of this kind is taken as it is
          if (kstep < 0) done = true;
        } else {
#pragma unroll
          for (int k = 0; k < M; ++k) mu[k] = best[k];
        }
      }
    }
    if (mk_wave_none(!done)) break;
  }
  return good;
}

// One pending problem of the staged solve in the lane form: chain kinematics, the collision rows
// within range, QP assembly and the solve with those rows.  q is the problem's row of the batch
// (read in place), hstore / gstore the lane's storage for H and the rows (M * nv elements), cstore
// that of the rows' chain part where it is kept apart (M * NC elements, M <= 2).  Returns true when
// vrow [nv] holds the problem's velocity; false leaves the problem to the general kernel.
template <int NC, int M = MK_LANE_ROWS>
MK_DEV bool rows_problem(const MkFusedDesc* __restrict__ D, const MkFusedExt* __restrict__ X, const real* row,
                         const real* tg_row, real dt, real inv_dt, real damping, real* vrow, real* hstore,
                         int hstride, real* cstore, int cstride, real* gstore, int gstride) {
  const real* tg = tg_row + D->target_off;
  const Q4 Qt = Q4{tg[0], tg[1], tg[2], tg[3]};
  const V3 pt = v3(tg[4], tg[5], tg[6]);
  real qc[NC], x[NC];
  load_chain_q<NC>(D, row, qc);
  ChainJac<NC> cj;
  Q4 Qw, Qcap = Q4{1, 0, 0, 0};
  V3 tw, tcap = v3(0, 0, 0);
  M3 Rw;
  chain_jac_pose<NC, true>(D, qc, cj, Qw, tw, Rw, X->has_rel ? X->rel_slot : NC, &Qcap, &tcap);
  cj.e = frame_error(Qw, tw, Rw, Qt, pt);
  // The rows are built for up to MB pairs in range.  The form that solves with fewer (M = 2) takes
  // the M rows with the smallest bounds -- the closest pairs -- and checks the others at its
  // solution afterwards: a row that holds there does not change the optimum, so the answer is that
  // of the whole problem; if one does not hold, the problem is left to the next stage.
  constexpr bool kSplit = LaneRows<M>::kSplit;
  constexpr int MB = kSplit ? MK_LANE_ROWS : M;
  LaneRows<MB> RA;
  RA.g = gstore;
  RA.stride = gstride;
  RA.gc = cstore;
  RA.cstride = cstride;
  RA.seg_dof = D->seg_dof;
  RA.nv = D->nv;
  RA.ncs = NC;
  RA.n = 0;
#pragma unroll
  for (int k = 0; k < MB; ++k) {
    RA.h[k] = INFINITY;
    RA.idx[k] = k;
  }
  // (the solve reads all of its M rows, in range or not; the rows it does not read are cleared when they are taken)
  RA.nzero = M;
  for (int i = 0; i < M * D->nv; ++i) gstore[i * gstride] = 0.0;
  const bool fits = collision_rows<NC, MB>(D, X, row, cj.J, Qw, tw, Rw, dt, RA);
  // (a wave none of whose problems fits this many rows leaves them all to the next stage at once)
  if (mk_wave_none(fits)) return false;
  LaneRows<M> R;
  if constexpr (!kSplit) {
    R = RA;
  } else {
    static_assert(!kSplit || M == 2, "the selection below picks two rows");
    R.g = gstore;
    R.stride = gstride;
    R.gc = cstore;
    R.cstride = cstride;
    R.seg_dof = D->seg_dof;
    R.nv = D->nv;
    R.ncs = NC;
    R.n = min(RA.n, M);
    int i0 = 0, i1 = 1;
    if (RA.n > M) {
      real h0 = INFINITY, h1 = INFINITY;
#pragma unroll
      for (int k = 0; k < MB; ++k) {
        const real hk = RA.h[k];   // (INFINITY beyond the rows in range)
        if (hk < h0) { h1 = h0; i1 = i0; h0 = hk; i0 = k; }
        else if (hk < h1) { h1 = hk; i1 = k; }
      }
      if (i1 < i0) { const int t = i0; i0 = i1; i1 = t; }   // (in pair order, like a problem with two rows)
    }
    R.idx[0] = i0;
    R.idx[1] = i1;
#pragma unroll
    for (int k = 0; k < MB; ++k) {
      if (k == i0) R.h[0] = RA.h[k];
      if (k == i1) R.h[1] = RA.h[k];
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int dof = D->seg_dof[i];
      cstore[i * cstride] = dof >= 0 ? gstore[(i0 * R.nv + dof) * gstride] : real(0.0);
      cstore[(NC + i) * cstride] = dof >= 0 ? gstore[(i1 * R.nv + dof) * gstride] : real(0.0);
    }
  }
  real c[NC], lo[NC], hi[NC], hacc = 0.0, diag;
  if (X->has_rel) {
    const real* tr = tg_row + X->rel_target_off;
    const Q4 Qrt = Q4{tr[0], tr[1], tr[2], tr[3]};
    const V3 prt = v3(tr[4], tr[5], tr[6]);
    const Q4 xqi = qconj(Q4{X->rel_xq[0], X->rel_xq[1], X->rel_xq[2], X->rel_xq[3]});
    const Q4 Qrf = qmul(xqi, Qcap);
    const V3 trf = qrot(xqi, tcap - v3(X->rel_xp[0], X->rel_xp[1], X->rel_xp[2]));
    const M3 Rt = q2m(Qrt);
    const Twist e_rel = se3_log(qmul(qconj(Qrt), Qrf), mulT(Rt, trf - prt));
    diag = assemble_chain_qp<NC, true>(D, row, qc, cj, dt, damping, c, lo, hi, hstore, hstride, X, e_rel, hacc);
  } else {
    diag = assemble_chain_qp<NC, false>(D, row, qc, cj, dt, damping, c, lo, hi, hstore, hstride, X,
                                        Twist{V3{0, 0, 0}, V3{0, 0, 0}}, hacc);
  }
  asm volatile("" ::: "memory");
  bool good = rows_qp<NC, M>(D, row, HView{hstore, hstride}, c, lo, hi, diag, dt, inv_dt, R, x, vrow);
  good &= fits && (diag > 0.0);
  if constexpr (kSplit) {
    // the rows the solve left out, at its solution (chain dofs in x, the others as written to vrow)
    const bool more = good && RA.n > M;
    if (!mk_wave_none(more)) {
      real r[MB];
#pragma unroll
      for (int k = 0; k < MB; ++k) {
        r[k] = -RA.h[k];
#pragma unroll
        for (int i = 0; i < NC; ++i) r[k] += RA.chain(k, i) * x[i];
      }
      // (dofs outside, rows inside, a few dofs at a time: the loads of a sweep are independent)
#pragma unroll 4
      for (int d = 0; d < D->nd; ++d) {
        const int dof = D->ddof[d].dof;
        const real y = vrow[dof] * dt;
#pragma unroll
        for (int k = 0; k < MB; ++k) r[k] += RA.at(k, dof) * y;
      }
      for (int f = 0; f < D->nfree; ++f) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int dof = D->free_dofadr[f] + j;
          const real y = vrow[dof] * dt;
#pragma unroll
          for (int k = 0; k < MB; ++k) r[k] += RA.at(k, dof) * y;
        }
      }
      bool holds = true;
#pragma unroll
      for (int k = 0; k < MB; ++k)
        if (k < RA.n && k != R.idx[0] && k != R.idx[1]) holds &= r[k] <= 1e-13 * fmax(1.0, fabs(RA.h[k]));
      if (more) good = holds;
    }
  }
  if (good) {
#pragma unroll
    for (int k = 0; k < NC; ++k)
      if (D->seg_dof[k] >= 0) vrow[D->seg_dof[k]] = x[k] * inv_dt;
  }
  return good;
}

// Second stage of the staged solve: the problems the structured kernel left pending (a collision
// pair may be in range), 64 of them per wave in the lane form.  Every wave scans its share of the
// status array in chunks of 64 entries (chunk c belongs to block c mod gridDim.x), queues the
// pending indices in LDS and works through the queue 64 problems at a time.  A problem it
// finishes gets its velocity row and status 0; one it cannot finish stays pending.
template <int NC, int M>
__global__ MK_KERNEL_ATTR void fused_rows_kernel(const MkFusedDesc* __restrict__ D, const MkFusedExt* __restrict__ X,
                                                 const double* __restrict__ q, const double* __restrict__ targets,
                                                 int target_stride, real dt, real damping, int B,
                                                 double* __restrict__ v_out, int32_t* status_out) {
  extern __shared__ real tile[];
  constexpr int NH = NC * (NC + 1) / 2;
  const int nq = D->nq, nv = D->nv;
  const int lane = threadIdx.x;
  real* hlds = tile + lane;
  // The form for at most two rows keeps only their chain part in LDS and the entries at the
  // other dofs in the lane's private memory: H and the chain part of two rows are 38 KB per wave
  // for a ten-joint chain, which lets every SIMD of a compute unit hold a wave, where H and two
  // whole rows of a 24-dof model (53 KB) allow three -- and the pending problems of a large batch
  // then take two rounds of batches instead of one.  The private entries are read in sweeps over
  // the dofs, all lanes at the same element, which the private segment's lane-interleaved layout
  // serves as contiguous lines.  The form for more rows, which few problems need, has them in LDS.
  constexpr bool kSplit = LaneRows<M>::kSplit;
  real gpriv[kSplit ? MK_LANE_ROWS * MK_ROWS_MAX_NV : 1];   // (all rows of a problem: the solve selects M of them)
  real* cstore = tile + NH * 64 + lane;
  real* gstore = kSplit ? gpriv : tile + NH * 64 + lane;
  const int gstride = kSplit ? 1 : 64;
  int32_t* queue = reinterpret_cast<int32_t*>(tile + (NH + (kSplit ? M * NC : M * nv)) * 64);   // 128 entries
  const real inv_dt = 1.0 / dt;
  const int nchunks = (B + 63) / 64;
  int n = 0;
  for (int ch = blockIdx.x;; ch += gridDim.x) {
    const bool more = ch < nchunks;
    if (more) {
      const long idx = (long)ch * 64 + lane;
      const bool pend = idx < B && (status_out[idx] & MK_STATUS_PENDING) != 0;
      const uint64_t mask = __ballot(pend);
      if (pend) queue[n + __popcll(mask & ((1ull << lane) - 1ull))] = (int32_t)idx;
      n += __popcll(mask);
    }
    __syncthreads();
    while (n >= 64 || (!more && n > 0)) {
      const int take = min(n, 64);
      const bool valid = lane < take;
      const long pb = queue[valid ? lane : 0];
      // lanes without a problem of their own repeat the first one, stores included (same values)
      const bool good = rows_problem<NC, M>(D, X, q + pb * nq, targets + pb * target_stride, dt, inv_dt, damping,
                                         v_out + pb * nv, hlds, 64, cstore, 64, gstore, gstride);
      if (valid && good) status_out[pb] &= ~MK_STATUS_PENDING;   // the low bits: status accumulated by a loop
      const int rest = n - take;
      const int32_t mv = lane < rest ? queue[64 + lane] : 0;
      __syncthreads();
      if (lane < rest) queue[lane] = mv;
      n = rest;
      __syncthreads();
    }
    if (!more) break;
  }
}

// fused_step_kernel with the extension descriptor: an optional RelativeFrameTask term and the
// collision screen.  A problem with a collision pair in range gets MK_STATUS_PENDING in its
// status entry (its velocity row is then provisional) and is solved by the later stages.
//
// With `iters` given the launch is one pass of the callers' loop (teleop_mouse.py:105-121) for
// problems outside the one-launch path: it first does the loop's bookkeeping for pass `it` --
// frame-task error of the current q to err_out, exit test (from the second pass on), solve count
// or done bit to iters -- and then solves only the problems still in the loop, leaving velocity
// and status of the others as they are; status accumulates (maximum over the passes) in its low
// bits next to the pending bit.
template <int NC>
__global__ MK_KERNEL_ATTR void fused_step_ext_kernel(const MkFusedDesc* __restrict__ D,
                                                     const MkFusedExt* __restrict__ X,
                                                     const double* __restrict__ q,
                                                     const double* __restrict__ targets, int target_stride,
                                                     real dt, real damping, int B,
                                                     real* v_out, int32_t* status_out, int32_t* iters,
                                                     double* __restrict__ err_out, int it, int max_iters,
                                                     real pos_tol, real ori_tol) {
  extern __shared__ real tile[];
  const int nq = D->nq, nv = D->nv;
  const int stride = nq | 1;
  const real inv_dt = 1.0 / dt;
  const int lane = threadIdx.x;
  const long base = (long)blockIdx.x * 64;
  const int rows = (int)min((long)64, (long)B - base);
  tile_copy<true>(tile, stride, const_cast<double*>(q) + base * nq, rows, nq);
  __syncthreads();
  const bool valid = lane < rows;
  real* row = tile + lane * stride;
  real* hlds = tile + 64 * stride + lane;
  if (!valid) {
    for (int i = 0; i < nq; ++i) row[i] = tile[i];
  }
  const long pb = valid ? base + lane : base;
  const double* tg = targets + pb * target_stride + D->target_off;
  const Q4 Qt = Q4{tg[0], tg[1], tg[2], tg[3]};
  const V3 pt = v3(tg[4], tg[5], tg[6]);
  real qc[NC], x[NC];
  load_chain_q<NC>(D, row, qc);
  ChainJac<NC> cj;
  Q4 Qw, Qcap = Q4{1, 0, 0, 0};
  V3 tw, tcap = v3(0, 0, 0);
  M3 Rw;
  chain_jac_pose<NC, true>(D, qc, cj, Qw, tw, Rw, X->has_rel ? X->rel_slot : NC, &Qcap, &tcap);
  cj.e = frame_error(Qw, tw, Rw, Qt, pt);
  bool skip = false;
  if (iters) {
    const int32_t cur = iters[pb];
    const bool was_done = (cur & MK_ITER_DONE_BIT) != 0;
    const bool conv = it > 0 && sqrt(dot(cj.e.v, cj.e.v)) <= pos_tol && sqrt(dot(cj.e.w, cj.e.w)) <= ori_tol;
    const bool done_now = conv || it == max_iters;
    if (valid && !was_done) {
      real* eo = err_out + pb * 6;
      eo[0] = cj.e.v.x; eo[1] = cj.e.v.y; eo[2] = cj.e.v.z; eo[3] = cj.e.w.x; eo[4] = cj.e.w.y; eo[5] = cj.e.w.z;
      iters[pb] = done_now ? (cur | MK_ITER_DONE_BIT) : it + 1;
    }
    skip = was_done || done_now;
    if (mk_wave_none(valid && !skip)) return;   // every problem of the block has left the loop
  }
  bool bad = rest_nonfinite(D, row);
  // collision screen first: it reads the lane's q, which the solve consumes
  // (its scratch is the lane's column of the region that holds H afterwards)
  const bool pending = X->ncpair > 0 && collision_in_range(D, X, row, Qw, tw, hlds, 64);
  uint32_t act = 0u, upper = 0u;
  int status;
  if (X->has_rel) {
    // T_root,frame = X^-1 o (pose carried at the root's slot); e = log(T_rt^-1 T_rf)
    const real* tr = targets + pb * target_stride + X->rel_target_off;
    const Q4 Qrt = Q4{tr[0], tr[1], tr[2], tr[3]};
    const V3 prt = v3(tr[4], tr[5], tr[6]);
    bad |= !(fabs(tr[0] + tr[1] + tr[2] + tr[3] + tr[4] + tr[5] + tr[6]) < INFINITY);
    const Q4 xqi = qconj(Q4{X->rel_xq[0], X->rel_xq[1], X->rel_xq[2], X->rel_xq[3]});
    const Q4 Qrf = qmul(xqi, Qcap);
    const V3 trf = qrot(xqi, tcap - v3(X->rel_xp[0], X->rel_xp[1], X->rel_xp[2]));
    const M3 Rt = q2m(Qrt);
    const Twist e_rel = se3_log(qmul(qconj(Qrt), Qrf), mulT(Rt, trf - prt));
    status = solve_problem<NC, false, true>(D, row, qc, cj, dt, inv_dt, damping, x, nullptr, act, upper, false, hlds, 64,
                                            bad, X, e_rel);
  } else {
    status = solve_problem<NC, false, false>(D, row, qc, cj, dt, inv_dt, damping, x, nullptr, act, upper, false, hlds, 64,
                                             bad);
  }
#pragma unroll
  for (int k = 0; k < NC; ++k)
    if (D->seg_dof[k] >= 0) row[D->seg_dof[k]] = x[k] * inv_dt;
  const int32_t flag = (pending && status != 3) ? MK_STATUS_PENDING : 0;
  if (!iters) {
    if (valid) status_out[pb] = flag ? flag : status;
  } else if (valid) {
    if (skip) {
      // out of the loop: the tile copy below rewrites the velocity this problem already has
      for (int i = 0; i < nv; ++i) row[i] = v_out[pb * nv + i];
    } else {
      status_out[pb] = max(status_out[pb] & ~MK_STATUS_PENDING, status) | flag;
    }
  }
  __syncthreads();
  tile_copy<false>(tile, stride, v_out + base * nv, rows, nv);
}

// HANDOFF: a wave whose problems have all but `handoff_live` of them left the loop stops there and
// marks the rest (MK_ITER_CONT_BIT in iters_out, which must be given): with the exit test the
// problems of a wave finish at different iterations, and the last few would hold the launch for
// their remaining sequential solves; the cooperative form (coop_iterate_body, mk_coop.hip)
// finishes them at 16 lanes per problem -- in this same launch (the wave takes its own marked
// problems, four at a time, one per 16-lane row, once its rows are back in memory); the
// streaming and rollout kernels leave theirs to coop_resume_kernel / coop_move_to_resume_kernel.
#endif  // !MK_F32

template <int NC, bool HANDOFF = false>
__global__ MK_KERNEL_ATTR void fused_iterate_kernel(const MkFusedDesc* __restrict__ D_arg,
                                                           double* __restrict__ q,
                                                           const double* __restrict__ targets, int target_stride,
                                                           real dt, real damping, int B, int max_iters,
                                                           real pos_tol, real ori_tol,
                                                           double* __restrict__ v_out, int32_t* __restrict__ iters_out,
                                                           double* __restrict__ err_out,
                                                           int32_t* __restrict__ status_out, int handoff_live = 0) {
  const MkFusedDesc* D = D_arg;
  extern __shared__ real tile[];
  const int nq = D->nq, nv = D->nv;
  const int stride = nq | 1;
  const real inv_dt = 1.0 / dt;
  const int lane = threadIdx.x;
  const long base = (long)blockIdx.x * 64;
  const int rows = (int)min((long)64, (long)B - base);
  tile_copy<true>(tile, stride, q + base * nq, rows, nq);
  __syncthreads();
  const bool valid = lane < rows;
  real* row = tile + lane * stride;
  real* hlds = tile + 64 * stride + lane;   // packed Hessian, element e at hlds[e * 64]
  if (!valid) {
    for (int i = 0; i < nq; ++i) row[i] = tile[i];
  }
  const long pb = valid ? base + lane : base;
  const double* tg = targets + pb * target_stride + D->target_off;
  Q4 Qt = Q4{tg[0], tg[1], tg[2], tg[3]};
  V3 pt = v3(tg[4], tg[5], tg[6]);
  real qc[NC], x[NC];
  load_chain_q<NC>(D, row, qc);
#pragma unroll
  for (int k = 0; k < NC; ++k) x[k] = 0.0;
  const bool sync_rows = D->any_posture_lm != 0;   // only that path reads chain q from the rows
  const bool input_bad = rest_nonfinite(D, row);
  Twist e = Twist{v3(0, 0, 0), v3(0, 0, 0)};
  int status = 0, iters = 0;
  bool done = false, handed = false;
  uint32_t act = 0u, upper = 0u;  // active bounds of the previous step: warm start of the next
  double* vrow = (valid && v_out) ? v_out + pb * nv : nullptr;
  for (int it = 0; it <= max_iters; ++it) {
    const MkFusedDesc* __restrict__ D = D_arg + mk_opaque_zero();   // same descriptor, re-read per iteration
    if (!done) {
      ChainJac<NC> cj;
      chain_jac<NC>(D, qc, Qt, pt, cj);
      e = cj.e;
      // exit test of the callers' loop, evaluated on the configuration reached by
      // the previous solve (interactive_scripts/teleop_mouse.py:117-121)
      const bool conv = it > 0 && sqrt(dot(e.v, e.v)) <= pos_tol && sqrt(dot(e.w, e.w)) <= ori_tol;
      bool hand = false;
      if constexpr (HANDOFF) {
        // the lanes of this branch that are about to solve: few enough (and not before the first
        // solve, whose cold start the lane form amortises best), they go to the cooperative kernel
        const bool go = !(conv || it == max_iters);
        hand = go && it > 0 && __popcll(__ballot(go)) <= handoff_live;
      }
      if (conv || it == max_iters) {
        done = true;
      } else if (HANDOFF && hand) {
        done = true;
        handed = true;
      } else {
#if defined(MK_ABLATE) && MK_ABLATE == 1
        // tuning build only: forward kinematics + error, no QP build / solve
        int st = 0;
#pragma unroll
        for (int k = 0; k < NC; ++k) x[k] = 1e-4 * (cj.e.v.x + cj.e.w.y + cj.J[0][k] + cj.J[4][k]);
#else
        int st = solve_problem<NC, true>(D, row, qc, cj, dt, inv_dt, damping, x, vrow, act, upper, it > 0, hlds, 64,
                                         input_bad);
#endif
        status = max(status, st);
        iters = it + 1;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          real vk = x[k] * inv_dt;
          x[k] = vk;
          qc[k] = qc[k] + dt * vk;
        }
        if (sync_rows) store_chain_q<NC>(D, row, qc);
      }
    }
    if (mk_wave_none(!done)) break;
  }
  store_chain_q<NC>(D, row, qc);
  if (vrow) {
#pragma unroll
    for (int k = 0; k < NC; ++k)
      if (D->seg_dof[k] >= 0) vrow[D->seg_dof[k]] = (double)x[k];
  }
  if (valid) {
    status_out[pb] = status;
    if (iters_out) iters_out[pb] = (HANDOFF && handed) ? (iters | MK_ITER_CONT_BIT) : iters;
    if (err_out) {
      double* eo = err_out + pb * 6;
      eo[0] = (double)e.v.x; eo[1] = (double)e.v.y; eo[2] = (double)e.v.z;
      eo[3] = (double)e.w.x; eo[4] = (double)e.w.y; eo[5] = (double)e.w.z;
    }
  }
  __syncthreads();
  tile_copy<false>(tile, stride, q + base * nq, rows, nq);
#if !defined(MK_F32)
  if constexpr (HANDOFF) {
    uint64_t marked = __ballot(valid && handed);   // wave-uniform; at most handoff_live bits
    if (marked != 0ull) {
      // the rows, counters and statuses written above are read back by other lanes of this wave
      __threadfence();
      __syncthreads();
      while (marked != 0ull) {   // four problems at a time, one per 16-lane row
        uint64_t m = marked;
        for (int k = 0; k < (lane >> 4); ++k) m &= m - 1ull;
        if (m != 0ull) {
          const long pr = base + (__ffsll((unsigned long long)m) - 1);
          const int it0 = iters_out[pr] & ~MK_ITER_CONT_BIT;
          coop_iterate_body<NC>(D_arg, q, targets, target_stride, (double)dt, (double)damping, max_iters,
                                (double)pos_tol, (double)ori_tol, v_out, iters_out, err_out, status_out, pr, true, it0,
                                status_out[pr]);
        }
        for (int k = 0; k < 4 && marked != 0ull; ++k) marked &= marked - 1ull;
      }
    }
  }
#endif
}

// The callers' loop for batches much larger than the machine, with early exit: a persistent
// wave owns a contiguous chunk of problems and every LANE works through problems of the
// chunk one after the other -- a lane whose problem has left the loop (converged, or out of
// iterations) writes it back and takes the next one, so a slow problem holds one lane, not
// 64.  The arithmetic of a problem is the same instruction sequence as in
// fused_iterate_kernel, whatever lane and company it runs in: results are bit-identical.
// The next problem index is wave-uniform state (one wave per block), so handing out
// problems needs neither atomics nor memory.
//
// HANDOFF: once a wave has handed out its whole chunk and no more than `handoff_live` of its
// problems are still in the loop, those leave the kernel marked MK_ITER_CONT_BIT in iters_out
// (which must be given) and finish on the cooperative kernel (coop_resume_kernel, mk_coop.hip),
// as in fused_iterate_kernel.
template <int NC, bool HANDOFF = false>
__global__ MK_KERNEL_ATTR void fused_iterate_stream_kernel(const MkFusedDesc* __restrict__ D_arg,
                                                           double* __restrict__ q,
                                                           const double* __restrict__ targets, int target_stride,
                                                           real dt, real damping, int B, int chunk, int refill_min,
                                                           int max_iters, real pos_tol, real ori_tol,
                                                           double* __restrict__ v_out, int32_t* __restrict__ iters_out,
                                                           double* __restrict__ err_out,
                                                           int32_t* __restrict__ status_out, int handoff_live = 0) {
  const MkFusedDesc* D = D_arg;
  extern __shared__ real tile[];
  const int nq = D->nq, nv = D->nv;
  const int stride = nq | 1;
  const real inv_dt = 1.0 / dt;
  const int lane = threadIdx.x;
  real* row = tile + lane * stride;
  real* hlds = tile + 64 * stride + lane;
  long next = (long)blockIdx.x * chunk;                       // wave-uniform: first problem not yet handed out
  const long last = min((long)B, next + (long)chunk);
  const bool sync_rows = D->any_posture_lm != 0;
  long pid = -1;                                              // -1: nothing yet, -2: chunk exhausted
  bool live = false, input_bad = false, handed = false;
  real qc[NC], x[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) { qc[k] = 0.0; x[k] = 0.0; }
  Q4 Qt = Q4{1, 0, 0, 0};
  V3 pt = v3(0, 0, 0);
  Twist e = Twist{v3(0, 0, 0), v3(0, 0, 0)};
  int status = 0, iters = 0, it = 0;
  uint32_t act = 0u, upper = 0u;
  for (;;) {
    const bool want = !live && pid != -2;
    const uint64_t wm = __ballot(want);
    // lanes are handed new problems once enough of them wait (or nothing else is left to do):
    // a hand-over costs the wave a memory round trip whatever the number of lanes in it
    if (wm != 0ull && (__popcll(wm) >= refill_min || __ballot(live) == 0ull)) {   // wave-uniform
      if (want && pid >= 0) {
        // the finished problem goes back to its rows
        store_chain_q<NC>(D, row, qc);
        lane_copy(q + pid * nq, row, nq);
        if (v_out) {
          double* vr = v_out + pid * nv;
#pragma unroll
          for (int k = 0; k < NC; ++k)
            if (D->seg_dof[k] >= 0) vr[D->seg_dof[k]] = (double)x[k];
        }
        status_out[pid] = status;
        if (iters_out) iters_out[pid] = (HANDOFF && handed) ? (iters | MK_ITER_CONT_BIT) : iters;
        if (err_out) {
          double* eo = err_out + pid * 6;
          eo[0] = (double)e.v.x; eo[1] = (double)e.v.y; eo[2] = (double)e.v.z;
          eo[3] = (double)e.w.x; eo[4] = (double)e.w.y; eo[5] = (double)e.w.z;
        }
      }
      const long mine = next + __popcll(wm & ((1ull << lane) - 1ull));
      next += __popcll(wm);
      if (want) {
        if (mine < last) {
          pid = mine;
          lane_copy(row, q + pid * nq, nq);
          const double* tg = targets + pid * target_stride + D->target_off;
          Qt = Q4{tg[0], tg[1], tg[2], tg[3]};
          pt = v3(tg[4], tg[5], tg[6]);
          load_chain_q<NC>(D, row, qc);
#pragma unroll
          for (int k = 0; k < NC; ++k) x[k] = 0.0;
          input_bad = rest_nonfinite(D, row);
          e = Twist{v3(0, 0, 0), v3(0, 0, 0)};
          status = 0; iters = 0; it = 0;
          act = 0u; upper = 0u;
          live = true;
          handed = false;
        } else {
          pid = -2;
        }
      }
    }
    if (__ballot(live) == 0ull) break;
    const MkFusedDesc* __restrict__ Di = D_arg + mk_opaque_zero();   // same descriptor, re-read per pass
    // the wave's tail: nothing left to hand out and few problems still in the loop (wave-uniform)
    const bool tail = HANDOFF && next >= last && __popcll(__ballot(live)) <= handoff_live;
    if (live) {
      ChainJac<NC> cj;
      chain_jac<NC>(Di, qc, Qt, pt, cj);
      e = cj.e;
      const bool conv = it > 0 && sqrt(dot(e.v, e.v)) <= pos_tol && sqrt(dot(e.w, e.w)) <= ori_tol;
      if (conv || it == max_iters) {
        live = false;
      } else if (HANDOFF && tail) {
        handed = true;
        live = false;
      } else {
        double* vrow = v_out ? v_out + pid * nv : nullptr;
        const int st = solve_problem<NC, true>(Di, row, qc, cj, dt, inv_dt, damping, x, vrow, act, upper, it > 0, hlds, 64,
                                               input_bad);
        status = max(status, st);
        iters = it + 1;
        ++it;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          const real vk = x[k] * inv_dt;
          x[k] = vk;
          qc[k] = qc[k] + dt * vk;
        }
        if (sync_rows) store_chain_q<NC>(Di, row, qc);
      }
    }
  }
}

// Waypoint reaching (envs/mj_env.py:193-276 with envs/robot_utils.py:113-160) as one launch.
// Per problem the reference runs, for up to max_steps outer steps: the pose of the task
// frame; the waypoint generator (bounded linear step towards the goal position, half-way
// slerp towards the goal orientation, reached test on the position distance and the x-y-z
// Euler angles of goal * current^-1); then the callers' IK loop towards the commanded pose;
// and stops after the step in which the generator reported the goal reached.
//
// Here the two nested loops of a problem are one flat sequence of "ticks" (one IK iteration
// each), and every lane runs its own sequence: a tick evaluates the chain once, closes the
// lane's outer step if its IK loop has ended on this pose (exit test or iteration cap),
// opens the next one on the same pose (waypoint generator), and solves.  Lanes whose problem
// is finished take the next problem of the wave's chunk, as in fused_iterate_stream_kernel.
// A problem's arithmetic does not depend on the lane or the company it runs in.
//
// HANDOFF: once a wave has handed out its whole chunk and no more than `handoff_live` of its
// rollouts are still running, each of those leaves the kernel at its next outer-step boundary
// (where its state is just q and the three counters) with MK_ITER_CONT_BIT set in its steps_out
// entry; the cooperative kernel (coop_move_to_resume_kernel, mk_coop.hip) runs it to the end at
// 16 lanes per rollout instead of holding a 64-lane wave for its remaining sequential solves.
// steps_out and solves_out must be given.
template <int NC, bool HANDOFF = false>
__global__ MK_KERNEL_ATTR void fused_move_to_kernel(const MkFusedDesc* __restrict__ D_arg, double* __restrict__ q,
                                                    const double* __restrict__ goals, real dt, real damping,
                                                    int B, int chunk, int refill_min, int max_steps, int max_iters,
                                                    real pos_tol, real ori_tol, real wp_pos_thr,
                                                    real wp_pos_step, real wp_rot_thr,
                                                    int32_t* __restrict__ steps_out,
                                                    int32_t* __restrict__ reached_out,
                                                    int32_t* __restrict__ solves_out,
                                                    int32_t* __restrict__ status_out, int handoff_live = 0) {
  const MkFusedDesc* D = D_arg;
  extern __shared__ real tile[];
  const int nq = D->nq;
  const int stride = nq | 1;
  const real inv_dt = 1.0 / dt;
  const int lane = threadIdx.x;
  real* row = tile + lane * stride;
  real* hlds = tile + 64 * stride + lane;
  long next = (long)blockIdx.x * chunk;
  const long last = min((long)B, next + (long)chunk);
  const bool sync_rows = D->any_posture_lm != 0;
  long pid = -1;                                   // -1: nothing yet, -2: chunk exhausted
  bool live = false, input_bad = false, now = false, reached = false, in_step = false, handed = false;
  real qc[NC], x[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) { qc[k] = 0.0; x[k] = 0.0; }
  // (the goal pose is read again at every outer step instead of being held: fourteen registers
  // less across the solver, which is what kept this kernel's spills in scratch memory)
  Q4 Qt = Q4{1, 0, 0, 0};
  V3 pt = v3(0, 0, 0);
  int status = 0, steps = 0, solves = 0, it = 0;
  uint32_t act = 0u, upper = 0u;
  for (;;) {
    const bool want = !live && pid != -2;
    const uint64_t wm = __ballot(want);
    if (wm != 0ull && (__popcll(wm) >= refill_min || __ballot(live) == 0ull)) {   // wave-uniform
      if (want && pid >= 0) {
        store_chain_q<NC>(D, row, qc);
        lane_copy(q + pid * nq, row, nq);
        status_out[pid] = status;
        if (steps_out) steps_out[pid] = (HANDOFF && handed) ? (steps | MK_ITER_CONT_BIT) : steps;
        if (reached_out) reached_out[pid] = reached ? 1 : 0;
        if (solves_out) solves_out[pid] = solves;
      }
      const long mine = next + __popcll(wm & ((1ull << lane) - 1ull));
      next += __popcll(wm);
      if (want) {
        if (mine < last) {
          pid = mine;
          lane_copy(row, q + pid * nq, nq);
          load_chain_q<NC>(D, row, qc);
          input_bad = rest_nonfinite(D, row);
          status = 0; steps = 0; solves = 0; it = 0;
          act = 0u; upper = 0u;
          now = false; reached = false; in_step = false; handed = false;
          live = max_steps > 0;     // no outer step at all: the problem goes straight back
        } else {
          pid = -2;
        }
      }
    }
    if (__ballot(live) == 0ull) {
      if (__ballot(!live && pid >= 0) == 0ull) break;   // nothing left to write back either
      continue;
    }
    const MkFusedDesc* __restrict__ Di = D_arg + mk_opaque_zero();
    // the wave's tail: nothing left to hand out and few rollouts still running (wave-uniform)
    const bool tail = HANDOFF && next >= last && __popcll(__ballot(live)) <= handoff_live;
    if (live) {
      ChainJac<NC> cj;
      Q4 Q;
      V3 t;
      M3 R;
      chain_jac_pose<NC>(Di, qc, cj, Q, t, R);
      // does the IK loop of the lane's current outer step end on this pose?
      bool begin = !in_step;
      if (in_step) {
        cj.e = frame_error(Q, t, R, Qt, pt);
        const bool conv = it > 0 && sqrt(dot(cj.e.v, cj.e.v)) <= pos_tol && sqrt(dot(cj.e.w, cj.e.w)) <= ori_tol;
        if (conv || it == max_iters) {
          ++steps;
          if (now) {
            reached = true;
            live = false;
          } else if (steps == max_steps) {
            live = false;
          } else {
            begin = true;
          }
        }
      }
      if (HANDOFF && tail && begin) {
        // outer-step boundary in the tail: the cooperative kernel takes the rollout from here
        handed = true;
        live = false;
        begin = false;
      }
      if (begin) {
        // the next outer step opens on the same pose: waypoint generator, then the error
        // towards the new command
        const double* gp = goals + pid * 7;
        const Q4 Qg = Q4{gp[0], gp[1], gp[2], gp[3]};
        const V3 pg = v3(gp[4], gp[5], gp[6]);
        now = waypoint_step(Q, t, Qg, pg, wp_pos_thr, wp_pos_step, wp_rot_thr, Qt, pt);
        it = 0;
        in_step = true;
        if (max_iters == 0) {
          // no IK iteration per step: the pose never changes, so every step repeats this one
          if (now) { ++steps; reached = true; } else { steps = max_steps; }
          live = false;
        }
        cj.e = frame_error(Q, t, R, Qt, pt);
      }
      if (live) {
        const int st = solve_problem<NC, true>(Di, row, qc, cj, dt, inv_dt, damping, x, nullptr, act, upper,
                                               it > 0 || steps > 0, hlds, 64, input_bad);
        status = max(status, st);
        ++it;
        ++solves;
#pragma unroll
        for (int k = 0; k < NC; ++k) qc[k] = qc[k] + dt * (x[k] * inv_dt);
        if (sync_rows) store_chain_q<NC>(Di, row, qc);
      }
    }
  }
}

}  // namespace

// Host-side launchers (called from mk_api.hip).
// Dynamic LDS beyond the 64 KB default (long q rows with a 12-dof chain) needs the kernel's
// limit raised, once per kernel instantiation and device; the CU has 160 KB.
template <auto Kernel>
static hipError_t allow_lds(size_t lds) {
  static MkLdsGrant grant;   // one per instantiation of this template, i.e. per kernel
  return mk_grant_lds(reinterpret_cast<const void*>(Kernel), lds, grant);
}

template <int NC>
static hipError_t launch_step(const MkFusedDesc* d_desc, int nq, const double* q, const double* targets,
                              int target_stride, double dt, double damping, int B, double* v, int32_t* status,
                              hipStream_t stream) {
  const int blocks = (B + 63) / 64;
  const size_t lds = (size_t)64 * ((nq | 1) + NC * (NC + 1) / 2) * sizeof(real);
  if (hipError_t e = allow_lds<fused_step_kernel<NC>>(lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(fused_step_kernel<NC>, dim3(blocks), dim3(64), lds, stream, d_desc, q, targets, target_stride,
                     dt, damping, B, v, status);
  return hipGetLastError();
}

template <int NC>
static hipError_t launch_iterate(const MkFusedDesc* d_desc, int nq, double* q, const double* targets,
                                 int target_stride, double dt, double damping, int B, int max_iters, double pos_tol,
                                 double ori_tol, double* v, int32_t* iters, double* err, int32_t* status,
                                 hipStream_t stream, int handoff_live) {
  const int blocks = (B + 63) / 64;
  const size_t lds = (size_t)64 * ((nq | 1) + NC * (NC + 1) / 2) * sizeof(real);
#if !defined(MK_F32)   // (the cooperative kernels that take a hand-off are float64 only)
  if (handoff_live > 0) {
    if (hipError_t e = allow_lds<fused_iterate_kernel<NC, true>>(lds); e != hipSuccess) return e;
    hipLaunchKernelGGL((fused_iterate_kernel<NC, true>), dim3(blocks), dim3(64), lds, stream, d_desc, q, targets,
                       target_stride, dt, damping, B, max_iters, pos_tol, ori_tol, v, iters, err, status, handoff_live);
    return hipGetLastError();
  }
#endif
  if (hipError_t e = allow_lds<fused_iterate_kernel<NC, false>>(lds); e != hipSuccess) return e;
  hipLaunchKernelGGL((fused_iterate_kernel<NC, false>), dim3(blocks), dim3(64), lds, stream, d_desc, q, targets,
                     target_stride, dt, damping, B, max_iters, pos_tol, ori_tol, v, iters, err, status, 0);
  return hipGetLastError();
}

// Wave slots of the current device at one wave per SIMD (the occupancy of these kernels).
static int wave_slots() {
  static std::atomic<int> cached[32];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return 1024;
  int v = cached[dev].load(std::memory_order_relaxed);
  if (v == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    v = 4 * cus;
    cached[dev].store(v, std::memory_order_relaxed);
  }
  return v;
}

template <int NC>
static hipError_t launch_iterate_stream(const MkFusedDesc* d_desc, int nq, double* q, const double* targets,
                                        int target_stride, double dt, double damping, int B, int blocks, int max_iters,
                                        double pos_tol, double ori_tol, double* v, int32_t* iters, double* err,
                                        int32_t* status, hipStream_t stream, int handoff_live) {
  const int chunk = (B + blocks - 1) / blocks;
  const int refill_min = (int)mk_opt(MK_OPT_STREAM_REFILL);   // tuning
  const size_t lds = (size_t)64 * ((nq | 1) + NC * (NC + 1) / 2) * sizeof(real);
#if !defined(MK_F32)
  if (handoff_live > 0) {
    if (hipError_t e = allow_lds<fused_iterate_stream_kernel<NC, true>>(lds); e != hipSuccess) return e;
    hipLaunchKernelGGL((fused_iterate_stream_kernel<NC, true>), dim3(blocks), dim3(64), lds, stream, d_desc, q, targets,
                       target_stride, dt, damping, B, chunk, refill_min, max_iters, pos_tol, ori_tol, v, iters, err,
                       status, handoff_live);
    return hipGetLastError();
  }
#endif
  if (hipError_t e = allow_lds<fused_iterate_stream_kernel<NC, false>>(lds); e != hipSuccess) return e;
  hipLaunchKernelGGL((fused_iterate_stream_kernel<NC, false>), dim3(blocks), dim3(64), lds, stream, d_desc, q, targets,
                     target_stride, dt, damping, B, chunk, refill_min, max_iters, pos_tol, ori_tol, v, iters, err,
                     status, 0);
  return hipGetLastError();
}

hipError_t mk_fused_launch_step(int nc, const MkFusedDesc* d_desc, int nq, const double* q, const double* targets,
                                int target_stride, double dt, double damping, int B, double* v, int32_t* status,
                                hipStream_t stream) {
  switch (nc) {
#define MK_CASE(N) \
  case N: return launch_step<N>(d_desc, nq, q, targets, target_stride, dt, damping, B, v, status, stream);
    MK_CASE(4) MK_CASE(6) MK_CASE(7) MK_CASE(8) MK_CASE(10) MK_CASE(12)
#undef MK_CASE
    default: return hipErrorInvalidValue;
  }
}

#if !defined(MK_F32)
template <int NC>
static hipError_t launch_step_ext(const MkFusedDesc* d_desc, const MkFusedExt* d_ext, int nq, const double* q,
                                  const double* targets, int target_stride, double dt, double damping, int B, double* v,
                                  int32_t* status, hipStream_t stream, int screen_doubles, int32_t* iters, double* err,
                                  int it, int max_iters, double pos_tol, double ori_tol) {
  const int blocks = (B + 63) / 64;
  // per lane: its q, and H or (before it) the scratch of the collision screen
  const size_t lds = (size_t)64 * ((nq | 1) + std::max(NC * (NC + 1) / 2, screen_doubles)) * sizeof(real);
  if (hipError_t e = allow_lds<fused_step_ext_kernel<NC>>(lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(fused_step_ext_kernel<NC>, dim3(blocks), dim3(64), lds, stream, d_desc, d_ext, q, targets,
                     target_stride, dt, damping, B, v, status, iters, err, it, max_iters, pos_tol, ori_tol);
  return hipGetLastError();
}

hipError_t mk_fused_launch_step_ext(int nc, const MkFusedDesc* d_desc, const MkFusedExt* d_ext, int nq, const double* q,
                                    const double* targets, int target_stride, double dt, double damping, int B,
                                    double* v, int32_t* status, hipStream_t stream, int screen_doubles, int32_t* iters,
                                    double* err, int it, int max_iters, double pos_tol, double ori_tol) {
  switch (nc) {
#define MK_CASE(N)                                                                                                  \
  case N:                                                                                                           \
    return launch_step_ext<N>(d_desc, d_ext, nq, q, targets, target_stride, dt, damping, B, v, status, stream,       \
                              screen_doubles, iters, err, it, max_iters, pos_tol, ori_tol);
    MK_CASE(4) MK_CASE(6) MK_CASE(7) MK_CASE(8) MK_CASE(10) MK_CASE(12)
#undef MK_CASE
    default: return hipErrorInvalidValue;
  }
}

// One launch of the rows stage for problems with at most M rows in range.
template <int NC, int M>
static hipError_t launch_rows_m(const MkFusedDesc* d_desc, const MkFusedExt* d_ext, int nv, const double* q,
                                const double* targets, int target_stride, double dt, double damping, int B, double* v,
                                int32_t* status, hipStream_t stream) {
  // per wave: H and the rows (the chain part of up to two rows) of 64 problems and the queue of pending indices
  const size_t lds = (size_t)64 * (NC * (NC + 1) / 2 + (M <= 2 ? M * NC : M * nv)) * sizeof(real) + 128 * sizeof(int32_t);
  // a model with more dofs than a lane's private rows hold, or too wide for one wave's rows to fit
  // the LDS of a compute unit, skips this stage: its pending problems all go to the general kernel
  if (nv > MK_ROWS_MAX_NV || lds > (size_t)160 * 1024) return hipSuccess;
  if (hipError_t e = allow_lds<fused_rows_kernel<NC, M>>(lds); e != hipSuccess) return e;
  // as many resident waves per compute unit as the LDS allows (at most one per SIMD); each scans
  // its chunks of the status array
  const int per_cu = std::max(1, std::min(4, (int)(((size_t)160 * 1024) / lds)));
  const int blocks = std::max(1, std::min((B + 63) / 64, wave_slots() / 4 * per_cu));
  hipLaunchKernelGGL((fused_rows_kernel<NC, M>), dim3(blocks), dim3(64), lds, stream, d_desc, d_ext, q, targets,
                     target_stride, dt, damping, B, v, status);
  return hipGetLastError();
}

// The rows stage in two launches: the form that solves with two rows first -- a 2 x 2 instead of
// a 4 x 4 complementarity problem per step and far fewer registers; it finishes the problems
// with at most two rows in range and those with more whose two closest pairs decide the answer --
// then, for what that leaves pending, the form with MK_LANE_ROWS rows.
template <int NC>
static hipError_t launch_rows(const MkFusedDesc* d_desc, const MkFusedExt* d_ext, int nv, const double* q,
                              const double* targets, int target_stride, double dt, double damping, int B, double* v,
                              int32_t* status, hipStream_t stream) {
  if (hipError_t e = launch_rows_m<NC, 2>(d_desc, d_ext, nv, q, targets, target_stride, dt, damping, B, v, status, stream);
      e != hipSuccess)
    return e;
  if (mk_opt(MK_OPT_LANE_ROWS) == 2) return hipSuccess;   // (tests: the two-row form alone)
  return launch_rows_m<NC, MK_LANE_ROWS>(d_desc, d_ext, nv, q, targets, target_stride, dt, damping, B, v, status, stream);
}

hipError_t mk_fused_launch_rows(int nc, const MkFusedDesc* d_desc, const MkFusedExt* d_ext, int nv, const double* q,
                                const double* targets, int target_stride, double dt, double damping, int B, double* v,
                                int32_t* status, hipStream_t stream) {
  switch (nc) {
#define MK_CASE(N) \
  case N: return launch_rows<N>(d_desc, d_ext, nv, q, targets, target_stride, dt, damping, B, v, status, stream);
    MK_CASE(4) MK_CASE(6) MK_CASE(7) MK_CASE(8) MK_CASE(10) MK_CASE(12)
#undef MK_CASE
    default: return hipErrorInvalidValue;
  }
}

#endif  // !MK_F32

// handoff_live > 0 (and an exit test, and `iters` given): the kernels may leave up to that many
// problems per wave marked MK_ITER_CONT_BIT for mk_coop_launch_resume; *handed tells whether
// this launch can have left any.
hipError_t mk_fused_launch_iterate(int nc, const MkFusedDesc* d_desc, int nq, double* q, const double* targets,
                                   int target_stride, double dt, double damping, int B, int max_iters,
                                   double pos_tol, double ori_tol, double* v, int32_t* iters, double* err,
                                   int32_t* status, hipStream_t stream, int handoff_live, bool* handed) {
  if (!(handoff_live > 0 && iters != nullptr && pos_tol >= 0.0 && ori_tol >= 0.0 && max_iters > 2)) handoff_live = 0;
  if (MK_IS_F32) handoff_live = 0;
  if (handed) *handed = handoff_live > 0;
  // With an exit test and more problems than twice the lanes of the machine, lanes stream
  // through the problems of a chunk (one persistent wave per slot); otherwise one block of 64
  // problems per wave with coalesced tile copies.
  int slots = wave_slots();
  bool stream_path = (long)B >= 2L * 64 * slots;
  // option "stream_blocks" = n (tests, tuning): take the streaming kernel with n persistent
  // waves whatever the batch size; 0 switches it off
  if (const long n = mk_opt(MK_OPT_STREAM_BLOCKS); n >= 0) {
    stream_path = n > 0;
    if (n > 0) slots = (int)n;
  }
  if (stream_path && (pos_tol >= 0.0 && ori_tol >= 0.0) && max_iters > 1) {
    switch (nc) {
#define MK_CASE(N)                                                                                              \
  case N:                                                                                                       \
    return launch_iterate_stream<N>(d_desc, nq, q, targets, target_stride, dt, damping, B, slots, max_iters,  \
                                    pos_tol, ori_tol, v, iters, err, status, stream, handoff_live);
      MK_CASE(4) MK_CASE(6) MK_CASE(7) MK_CASE(8) MK_CASE(10) MK_CASE(12)
#undef MK_CASE
      default: return hipErrorInvalidValue;
    }
  }
  // (the block-per-64 kernel finishes its marked problems itself)
  if (handed) *handed = false;
  switch (nc) {
#define MK_CASE(N)                                                                                         \
  case N:                                                                                                  \
    return launch_iterate<N>(d_desc, nq, q, targets, target_stride, dt, damping, B, max_iters, pos_tol,   \
                             ori_tol, v, iters, err, status, stream, handoff_live);
    MK_CASE(4) MK_CASE(6) MK_CASE(7) MK_CASE(8) MK_CASE(10) MK_CASE(12)
#undef MK_CASE
    default: return hipErrorInvalidValue;
  }
}

template <int NC>
static hipError_t launch_move_to(const MkFusedDesc* d_desc, int nq, double* q, const double* goals, double dt,
                                 double damping, int B, int max_steps, int max_iters, double pos_tol, double ori_tol,
                                 double wp_pos_thr, double wp_pos_step, double wp_rot_thr, int32_t* steps,
                                 int32_t* reached, int32_t* solves, int32_t* status, hipStream_t stream,
                                 int handoff_live) {
  // one persistent wave per wave slot of the device at most; with fewer problems than lanes
  // every lane gets exactly one problem
  int slots = wave_slots();
  if (const long n = mk_opt(MK_OPT_STREAM_BLOCKS); n > 0) slots = (int)n;
  const int blocks = std::max(1, std::min((B + 63) / 64, slots));
  const int chunk = (B + blocks - 1) / blocks;
  const int refill_min = (int)mk_opt(MK_OPT_STREAM_REFILL);   // tuning
  const size_t lds = (size_t)64 * ((nq | 1) + NC * (NC + 1) / 2) * sizeof(real);
#if !defined(MK_F32)
  if (handoff_live > 0) {
    if (hipError_t e = allow_lds<fused_move_to_kernel<NC, true>>(lds); e != hipSuccess) return e;
    hipLaunchKernelGGL((fused_move_to_kernel<NC, true>), dim3(blocks), dim3(64), lds, stream, d_desc, q, goals, dt,
                       damping, B, chunk, refill_min, max_steps, max_iters, pos_tol, ori_tol, wp_pos_thr, wp_pos_step,
                       wp_rot_thr, steps, reached, solves, status, handoff_live);
    return hipGetLastError();
  }
#endif
  if (hipError_t e = allow_lds<fused_move_to_kernel<NC, false>>(lds); e != hipSuccess) return e;
  hipLaunchKernelGGL((fused_move_to_kernel<NC, false>), dim3(blocks), dim3(64), lds, stream, d_desc, q, goals, dt,
                     damping, B, chunk, refill_min, max_steps, max_iters, pos_tol, ori_tol, wp_pos_thr, wp_pos_step,
                     wp_rot_thr, steps, reached, solves, status, 0);
  return hipGetLastError();
}

hipError_t mk_fused_launch_move_to(int nc, const MkFusedDesc* d_desc, int nq, double* q, const double* goals,
                                   double dt, double damping, int B, int max_steps, int max_iters, double pos_tol,
                                   double ori_tol, double wp_pos_thr, double wp_pos_step, double wp_rot_thr,
                                   int32_t* steps, int32_t* reached, int32_t* solves, int32_t* status,
                                   hipStream_t stream, int handoff_live, bool* handed) {
  // (the marks of the hand-off live in `steps`, and the resumed rollout continues its counters)
  if (!(handoff_live > 0 && steps != nullptr && solves != nullptr && max_iters > 0)) handoff_live = 0;
  if (MK_IS_F32) handoff_live = 0;
  if (handed) *handed = handoff_live > 0;
  switch (nc) {
#define MK_CASE(N)                                                                                              \
  case N:                                                                                                       \
    return launch_move_to<N>(d_desc, nq, q, goals, dt, damping, B, max_steps, max_iters, pos_tol, ori_tol,     \
                             wp_pos_thr, wp_pos_step, wp_rot_thr, steps, reached, solves, status, stream,      \
                             handoff_live);
    MK_CASE(4) MK_CASE(6) MK_CASE(7) MK_CASE(8) MK_CASE(10) MK_CASE(12)
#undef MK_CASE
    default: return hipErrorInvalidValue;
  }
}
