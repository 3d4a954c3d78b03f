// Path-walking forward kinematics shared by the generic kernels: a frame's pose is
// obtained by advancing a (position, quaternion, rotation) state through the bodies
// on the path from the root to the frame's body -- no per-body storage.  Restates
// mj_kinematics for one chain (MuJoCo semantics: body offset, then the body's joints
// in order, quaternion renormalised at the end of every body).
#pragma once
#include "../../include/mink_amd.h"
#include "mk_device.h"
#include "mk_generic.h"

struct BodyState {
  V3 p;
  Q4 Q;
  M3 R;
};

MK_DEV BodyState body_state_identity() {
  BodyState s;
  s.p = v3(0, 0, 0);
  s.Q = Q4{1, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 9; ++i) s.R.m[i] = (i % 4 == 0) ? 1.0 : 0.0;
  return s;
}

enum { MK_COL_LIN = 0, MK_COL_ROT = 1 };

// Advances `st` from the parent frame through body `b`.  When JAC, calls
// emit(dof, kind, axis, anchor) for each dof of the body: a MK_COL_LIN dof moves a
// point along `axis`; a MK_COL_ROT dof rotates it about `axis` through `anchor`
// (world frame), i.e. Jp = axis x (point - anchor), Jr = axis.
template <bool JAC, class Emit>
MK_DEV void advance_body(const MkDevModel& m, const double* __restrict__ q, int b, BodyState& st, Emit& emit) {
  const int ja = m.body_jntadr[b], jn = m.body_jntnum[b];
  if (jn == 1 && m.jnt_type[ja] == MK_JNT_FREE) {
    const double* qq = q + m.jnt_qposadr[ja];
    st.p = v3(qq[0], qq[1], qq[2]);
    st.Q = qnormalize(Q4{qq[3], qq[4], qq[5], qq[6]});
    st.R = q2m(st.Q);
    if (JAC) {
      const int d0 = m.jnt_dofadr[ja];
      emit(d0 + 0, MK_COL_LIN, v3(1, 0, 0), st.p);
      emit(d0 + 1, MK_COL_LIN, v3(0, 1, 0), st.p);
      emit(d0 + 2, MK_COL_LIN, v3(0, 0, 1), st.p);
      for (int k = 0; k < 3; ++k) emit(d0 + 3 + k, MK_COL_ROT, v3(st.R.m[k], st.R.m[3 + k], st.R.m[6 + k]), st.p);
    }
    return;
  }
  st.p = st.p + mul(st.R, v3(m.body_pos[3 * b], m.body_pos[3 * b + 1], m.body_pos[3 * b + 2]));
  st.Q = qmul(st.Q, Q4{m.body_quat[4 * b], m.body_quat[4 * b + 1], m.body_quat[4 * b + 2], m.body_quat[4 * b + 3]});
  st.R = q2m(st.Q);
  int ball_dof = -1;  // ball-joint columns use the body's final orientation
  V3 ball_anchor = v3(0, 0, 0);
  for (int j = ja; j < ja + jn; ++j) {
    const int t = m.jnt_type[j];
    const int qa = m.jnt_qposadr[j], d0 = m.jnt_dofadr[j];
    const V3 jpos = v3(m.jnt_pos[3 * j], m.jnt_pos[3 * j + 1], m.jnt_pos[3 * j + 2]);
    const V3 axl = v3(m.jnt_axis[3 * j], m.jnt_axis[3 * j + 1], m.jnt_axis[3 * j + 2]);
    const V3 ax = mul(st.R, axl);
    const V3 an = st.p + mul(st.R, jpos);
    if (t == MK_JNT_SLIDE) {
      st.p = st.p + (q[qa] - m.qpos0[qa]) * ax;
      if (JAC) emit(d0, MK_COL_LIN, ax, an);
    } else {
      Q4 ql;
      if (t == MK_JNT_BALL) {
        ql = qnormalize(Q4{q[qa], q[qa + 1], q[qa + 2], q[qa + 3]});
        ball_dof = d0;
        ball_anchor = an;
      } else {
        double sn, cs;
        sincos(0.5 * (q[qa] - m.qpos0[qa]), &sn, &cs);
        ql = Q4{cs, sn * axl.x, sn * axl.y, sn * axl.z};
        if (JAC) emit(d0, MK_COL_ROT, ax, an);
      }
      st.Q = qmul(st.Q, ql);
      st.R = q2m(st.Q);
      st.p = an - mul(st.R, jpos);
    }
  }
  st.Q = qnormalize(st.Q);
  st.R = q2m(st.Q);
  if (JAC && ball_dof >= 0) {
    for (int k = 0; k < 3; ++k)
      emit(ball_dof + k, MK_COL_ROT, v3(st.R.m[k], st.R.m[3 + k], st.R.m[6 + k]), ball_anchor);
  }
}

struct NoEmit {
  MK_DEV void operator()(int, int, V3, V3) {}
};
