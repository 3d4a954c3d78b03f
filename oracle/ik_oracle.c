/*
 * TEST INFRASTRUCTURE ONLY -- plain-C restatement of the reference's differential
 * IK path (dense, one problem at a time), used as (1) a fast checker for large
 * batches in tests/ and (2) the CPU baseline ("port") timed by bench.py.  Nothing
 * under mink_amd/ links or loads this file.
 *
 * It follows, step by step and with dense nv x nv matrices, what the reference does
 * per solve_ik call:
 *   mink/configuration.py:53-64   update(): mj_kinematics over the whole tree
 *   mink/configuration.py:112-155 get_frame_jacobian (mj_jac + rotation to the frame)
 *   mink/tasks/frame_task.py:95-146, mink/lie/se3.py:159-249, mink/lie/so3.py:176-226
 *   mink/tasks/posture_task.py:87-142, mink/tasks/damping_task.py:18-20
 *   mink/tasks/task.py:105-138   (H, c) of each task
 *   mink/limits/configuration_limit.py:69-124, mink/limits/velocity_limit.py:71-101
 *   mink/solve_ik.py:13-105      stack, solve, v = dq / dt
 *   mink/configuration.py:228-236 integrate_inplace (mj_integratePos)
 * and the QP backend (qpsolvers 4.4.0 -> quadprog 0.1.13, absent here) by the
 * published Goldfarb-Idnani dual active-set algorithm, with an explicit active list.
 * MuJoCo / quadprog themselves are not available: PARITY UNPINNED against them; this
 * file is pinned against oracle/ik_oracle.py, which is pinned against golden vectors
 * of the reference's own Python (tests/golden).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define JNT_FREE 0
#define JNT_BALL 1
#define JNT_SLIDE 2
#define JNT_HINGE 3
#define FRAME_BODY 0
#define FRAME_GEOM 1
#define FRAME_SITE 2
#define MINVAL 1e-15
#define LIE_EPS 1e-10
#define MAXVAL 1e10
#define MAXNV 64
#define MAXROWS 256
#define MAXBODY 128

typedef struct {
  int32_t nq, nv, nbody, njnt, ngeom, nsite;
  const int32_t *body_parentid, *body_jntadr, *body_jntnum, *body_mocapid;
  const double *body_pos, *body_quat, *body_mass, *body_ipos;
  const int32_t *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid, *jnt_limited;
  const double *jnt_pos, *jnt_axis, *jnt_range, *qpos0;
  const int32_t *dof_parentid, *dof_jntid;
  const int32_t* site_bodyid;
  const double *site_pos, *site_quat;
  const int32_t *geom_bodyid, *geom_type;
  const double *geom_size, *geom_pos, *geom_quat;
} or_model;

typedef struct {
  int32_t frame_type, frame_id; /* frame_id < 0: no frame task */
  double frame_cost[6], frame_gain, frame_lm;
  int32_t nposture;
  const double* posture_cost;   /* [nposture * nv] */
  const double* posture_target; /* [nposture * nq] */
  const double* posture_gain;   /* [nposture] */
  const double* posture_lm;     /* [nposture] */
  int32_t has_cfg_limit;
  double cfg_gain, cfg_min_dist;
  int32_t nvel;
  const int32_t* vel_dof; /* [nvel] */
  const double* vel_max;  /* [nvel] */
} or_problem;

/* ------------------------------------------------------------- small math -- */
static void q_mul(const double* a, const double* b, double* r) {
  double t0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double t1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double t2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double t3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = t0; r[1] = t1; r[2] = t2; r[3] = t3;
}
static void q_normalize(double* q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static void q_to_mat(const double* q, double* m) {
  double q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
  double q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
  double q12 = q[1] * q[2], q13 = q[1] * q[3], q23 = q[2] * q[3];
  m[0] = q00 + q11 - q22 - q33; m[1] = 2 * (q12 - q03); m[2] = 2 * (q13 + q02);
  m[3] = 2 * (q12 + q03); m[4] = q00 - q11 + q22 - q33; m[5] = 2 * (q23 - q01);
  m[6] = 2 * (q13 - q02); m[7] = 2 * (q23 + q01); m[8] = q00 - q11 - q22 + q33;
}
static void m_vec(const double* m, const double* v, double* r) {
  double a = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  double b = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  double c = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = a; r[1] = b; r[2] = c;
}
static void mt_vec(const double* m, const double* v, double* r) {
  double a = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
  double b = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
  double c = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = a; r[1] = b; r[2] = c;
}
static void m_mul(const double* a, const double* b, double* r) {
  double t[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  memcpy(r, t, sizeof(t));
}
static void cross3(const double* a, const double* b, double* r) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static void skew3(const double* w, double* m) {
  m[0] = 0; m[1] = -w[2]; m[2] = w[1];
  m[3] = w[2]; m[4] = 0; m[5] = -w[0];
  m[6] = -w[1]; m[7] = w[0]; m[8] = 0;
}
/* I + a S + b S^2, S = skew(w) */
static void eye_s_s2(const double* w, double a, double b, double* m) {
  double S[9], S2[9];
  skew3(w, S);
  m_mul(S, S, S2);
  for (int i = 0; i < 9; ++i) m[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * S[i] + b * S2[i];
}

/* ------------------------------------------------------------------- lie -- */
static void so3_log(const double* q, double* w) {
  double n2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3], f;
  if (n2 < LIE_EPS) {
    f = 2.0 / q[0] - 2.0 / 3.0 * n2 / (q[0] * q[0] * q[0]);
  } else {
    double n = sqrt(n2);
    if (fabs(q[0]) < LIE_EPS) f = (q[0] > 0.0 ? 1.0 : -1.0) * M_PI / n;
    else f = 2.0 * atan2(q[0] < 0 ? -n : n, fabs(q[0])) / n;
  }
  w[0] = f * q[1]; w[1] = f * q[2]; w[2] = f * q[3];
}
/* log of the transform (q, t) -> xi = (v; w) */
static void se3_log(const double* q, const double* t, double* xi) {
  double w[3], Vi[9], coef;
  so3_log(q, w);
  double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  if (t2 < LIE_EPS) coef = 1.0 / 12.0;
  else {
    double th = sqrt(t2), h = 0.5 * th;
    coef = (1.0 - th * cos(h) / (2.0 * sin(h))) / t2;
  }
  eye_s_s2(w, -0.5, coef, Vi);
  m_vec(Vi, t, xi);
  xi[3] = w[0]; xi[4] = w[1]; xi[5] = w[2];
}
/* 6x6 SE3.ljacinv(xi), row-major */
static void se3_ljacinv(const double* xi, double* J) {
  memset(J, 0, 36 * sizeof(double));
  const double* w = xi + 3;
  double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  if (t2 < LIE_EPS) {
    for (int i = 0; i < 6; ++i) J[7 * i] = 1.0;
    return;
  }
  double th = sqrt(t2), s = sin(th), c = cos(th);
  double A = (1.0 / t2) * (1.0 - (th * s / (2.0 * (1.0 - c))));
  double Ji[9], Q[9], V[9], W[9], VW[9], WV[9], WVW[9], VWW[9], t1[9], t2m[9], B[9];
  eye_s_s2(w, -0.5, A, Ji);
  double Bc = (th - s) / (t2 * th);
  double Cc = (1.0 - t2 / 2.0 - c) / (t2 * t2);
  double Dc = (2.0 * th - 3.0 * s + th * c) / (2.0 * t2 * t2 * th);
  skew3(xi, V);
  skew3(w, W);
  m_mul(V, W, VW);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) WV[3 * i + j] = VW[3 * j + i];
  m_mul(WV, W, WVW);
  m_mul(VW, W, VWW);
  m_mul(WVW, W, t1);
  m_mul(W, WVW, t2m);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      int k = 3 * i + j, kt = 3 * j + i;
      Q[k] = 0.5 * V[k] + Bc * (WV[k] + VW[k] + WVW[k]) - Cc * (VWW[k] - VWW[kt] - 3.0 * WVW[k]) +
             Dc * (t1[k] + t2m[k]);
    }
  m_mul(Ji, Q, B);
  m_mul(B, Ji, B);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      J[6 * i + j] = Ji[3 * i + j];
      J[6 * i + 3 + j] = -B[3 * i + j];
      J[6 * (i + 3) + 3 + j] = Ji[3 * i + j];
    }
}

/* ------------------------------------------------------------ kinematics -- */
typedef struct {
  double xpos[MAXBODY][3], xquat[MAXBODY][4], xmat[MAXBODY][9];
  double xanchor[MAXNV][3], xaxis[MAXNV][3]; /* per joint */
} or_data;

/* mj_kinematics for bodies 1..nbody-1 (mocap bodies keep their model pose). */
static void kinematics(const or_model* m, const double* q, or_data* d) {
  memset(d->xpos[0], 0, 3 * sizeof(double));
  d->xquat[0][0] = 1; d->xquat[0][1] = d->xquat[0][2] = d->xquat[0][3] = 0;
  q_to_mat(d->xquat[0], d->xmat[0]);
  for (int b = 1; b < m->nbody; ++b) {
    double pos[3], quat[4], tmp[3], mat[9];
    int ja = m->body_jntadr[b], jn = m->body_jntnum[b];
    if (m->body_mocapid[b] >= 0) {
      memcpy(pos, m->body_pos + 3 * b, sizeof(pos));
      memcpy(quat, m->body_quat + 4 * b, sizeof(quat));
    } else if (jn == 1 && m->jnt_type[ja] == JNT_FREE) {
      const double* qq = q + m->jnt_qposadr[ja];
      memcpy(pos, qq, sizeof(pos));
      memcpy(quat, qq + 3, sizeof(quat));
      q_normalize(quat);
      memcpy(d->xanchor[ja], pos, sizeof(pos));
      memcpy(d->xaxis[ja], m->jnt_axis + 3 * ja, sizeof(pos));
    } else {
      int p = m->body_parentid[b];
      m_vec(d->xmat[p], m->body_pos + 3 * b, pos);
      for (int i = 0; i < 3; ++i) pos[i] += d->xpos[p][i];
      q_mul(d->xquat[p], m->body_quat + 4 * b, quat);
      for (int j = ja; j < ja + jn; ++j) {
        int qa = m->jnt_qposadr[j], t = m->jnt_type[j];
        q_to_mat(quat, mat);
        m_vec(mat, m->jnt_axis + 3 * j, d->xaxis[j]);
        m_vec(mat, m->jnt_pos + 3 * j, tmp);
        for (int i = 0; i < 3; ++i) d->xanchor[j][i] = tmp[i] + pos[i];
        if (t == JNT_SLIDE) {
          double dq = q[qa] - m->qpos0[qa];
          for (int i = 0; i < 3; ++i) pos[i] += d->xaxis[j][i] * dq;
        } else {
          double ql[4];
          if (t == JNT_BALL) {
            memcpy(ql, q + qa, sizeof(ql));
            q_normalize(ql);
          } else {
            double ang = q[qa] - m->qpos0[qa];
            if (ang == 0.0) { ql[0] = 1; ql[1] = ql[2] = ql[3] = 0; }
            else {
              double s = sin(0.5 * ang);
              ql[0] = cos(0.5 * ang);
              ql[1] = s * m->jnt_axis[3 * j]; ql[2] = s * m->jnt_axis[3 * j + 1]; ql[3] = s * m->jnt_axis[3 * j + 2];
            }
          }
          q_mul(quat, ql, quat);
          q_to_mat(quat, mat);
          m_vec(mat, m->jnt_pos + 3 * j, tmp);
          for (int i = 0; i < 3; ++i) pos[i] = d->xanchor[j][i] - tmp[i];
        }
      }
    }
    q_normalize(quat);
    memcpy(d->xpos[b], pos, sizeof(pos));
    memcpy(d->xquat[b], quat, sizeof(quat));
    q_to_mat(quat, d->xmat[b]);
  }
}

static int frame_pose(const or_model* m, const or_data* d, int ftype, int fid, double* pos, double* quat) {
  int body;
  const double *lp = 0, *lq = 0;
  if (ftype == FRAME_BODY) body = fid;
  else if (ftype == FRAME_SITE) { body = m->site_bodyid[fid]; lp = m->site_pos + 3 * fid; lq = m->site_quat + 4 * fid; }
  else { body = m->geom_bodyid[fid]; lp = m->geom_pos + 3 * fid; lq = m->geom_quat + 4 * fid; }
  if (!lp) {
    memcpy(pos, d->xpos[body], 3 * sizeof(double));
    memcpy(quat, d->xquat[body], 4 * sizeof(double));
  } else {
    m_vec(d->xmat[body], lp, pos);
    for (int i = 0; i < 3; ++i) pos[i] += d->xpos[body][i];
    q_mul(d->xquat[body], lq, quat);
  }
  return body;
}

/* mj_jac: world-aligned Jp, Jr (3 x nv each, row-major) of `point` fixed to `body`. */
static void jac_point(const or_model* m, const or_data* d, const double* point, int body, double* jp, double* jr) {
  int nv = m->nv;
  memset(jp, 0, 3 * nv * sizeof(double));
  memset(jr, 0, 3 * nv * sizeof(double));
  while (body && m->body_jntnum[body] == 0) body = m->body_parentid[body];
  if (!body) return;
  int jlast = m->body_jntadr[body] + m->body_jntnum[body] - 1;
  int t = m->jnt_type[jlast];
  int i = m->jnt_dofadr[jlast] + (t == JNT_FREE ? 5 : t == JNT_BALL ? 2 : 0);
  while (i >= 0) {
    int j = m->dof_jntid[i];
    int jt = m->jnt_type[j], k = i - m->jnt_dofadr[j];
    double off[3], cr[3];
    for (int a = 0; a < 3; ++a) off[a] = point[a] - d->xanchor[j][a];
    if (jt == JNT_SLIDE) {
      for (int a = 0; a < 3; ++a) jp[a * nv + i] = d->xaxis[j][a];
    } else if (jt == JNT_HINGE) {
      cross3(d->xaxis[j], off, cr);
      for (int a = 0; a < 3; ++a) { jr[a * nv + i] = d->xaxis[j][a]; jp[a * nv + i] = cr[a]; }
    } else if (jt == JNT_FREE && k < 3) {
      jp[k * nv + i] = 1.0;
    } else {
      int kk = jt == JNT_FREE ? k - 3 : k;
      const double* xm = d->xmat[m->jnt_bodyid[j]];
      double ax[3] = {xm[kk], xm[3 + kk], xm[6 + kk]};
      cross3(ax, off, cr);
      for (int a = 0; a < 3; ++a) { jr[a * nv + i] = ax[a]; jp[a * nv + i] = cr[a]; }
    }
    i = m->dof_parentid[i];
  }
}

/* --------------------------------------------------------------- build QP -- */
/* Dense P [nv*nv], c [nv], G [rows*nv], h [rows]; returns the number of rows. e6
 * receives the frame-task error. */
static int build_qp(const or_model* m, const or_problem* p, const double* q, const double* target, double dt,
                    double damping, double* P, double* c, double* G, double* h, double* e6) {
  int nv = m->nv;
  or_data dd;
  or_data* d = &dd;
  kinematics(m, q, d);
  memset(P, 0, (size_t)nv * nv * sizeof(double));
  memset(c, 0, nv * sizeof(double));
  for (int i = 0; i < nv; ++i) P[i * nv + i] = damping;
  if (p->frame_id >= 0) {
    double ps[3], Qs[4], Rs[9], jp[3 * MAXNV], jr[3 * MAXNV];
    int body = frame_pose(m, d, p->frame_type, p->frame_id, ps, Qs);
    q_to_mat(Qs, Rs);
    jac_point(m, d, ps, body, jp, jr);
    /* e = log(T_wb^-1 T_wt); J = -ljacinv(e) J_b (jlog(T_tb) = ljacinv(-log T_tb) = ljacinv(e)) */
    double Qi[4] = {Qs[0], -Qs[1], -Qs[2], -Qs[3]}, qbt[4], dp[3], tbt[3], e[6], Jl[36];
    q_mul(Qi, target, qbt);
    for (int i = 0; i < 3; ++i) dp[i] = target[4 + i] - ps[i];
    mt_vec(Rs, dp, tbt);
    se3_log(qbt, tbt, e);
    memcpy(e6, e, sizeof(e));
    se3_ljacinv(e, Jl);
    double Jb[6 * MAXNV], Jw[6 * MAXNV], ew[6], mu = 0.0;
    for (int i = 0; i < nv; ++i) {
      double a[3] = {jp[i], jp[nv + i], jp[2 * nv + i]}, b[3] = {jr[i], jr[nv + i], jr[2 * nv + i]}, ra[3], rb[3];
      mt_vec(Rs, a, ra);
      mt_vec(Rs, b, rb);
      for (int r = 0; r < 3; ++r) { Jb[r * nv + i] = ra[r]; Jb[(3 + r) * nv + i] = rb[r]; }
    }
    for (int r = 0; r < 6; ++r) {
      for (int i = 0; i < nv; ++i) {
        double s = 0.0;
        for (int k = 0; k < 6; ++k) s += Jl[6 * r + k] * Jb[k * nv + i];
        Jw[r * nv + i] = p->frame_cost[r] * (-s);
      }
      ew[r] = p->frame_cost[r] * (-p->frame_gain * e[r]);
      mu += ew[r] * ew[r];
    }
    mu *= p->frame_lm;
    for (int i = 0; i < nv; ++i) {
      for (int j = 0; j < nv; ++j) {
        double s = 0.0;
        for (int r = 0; r < 6; ++r) s += Jw[r * nv + i] * Jw[r * nv + j];
        P[i * nv + j] += s;
      }
      P[i * nv + i] += mu;
      double s = 0.0;
      for (int r = 0; r < 6; ++r) s += ew[r] * Jw[r * nv + i];
      c[i] -= s;
    }
  } else {
    memset(e6, 0, 6 * sizeof(double));
  }
  /* dof -> (qpos address, is scalar joint) */
  int qadr[MAXNV], scalar[MAXNV];
  for (int j = 0; j < m->njnt; ++j) {
    int t = m->jnt_type[j], w = t == JNT_FREE ? 6 : t == JNT_BALL ? 3 : 1;
    for (int k = 0; k < w; ++k) {
      qadr[m->jnt_dofadr[j] + k] = m->jnt_qposadr[j] + k;
      scalar[m->jnt_dofadr[j] + k] = (t == JNT_SLIDE || t == JNT_HINGE);
    }
  }
  for (int t = 0; t < p->nposture; ++t) {
    const double* cost = p->posture_cost + (size_t)t * nv;
    const double* tq = p->posture_target + (size_t)t * m->nq;
    double ewv[MAXNV], mu = 0.0;
    for (int i = 0; i < nv; ++i) {
      /* free-joint dofs are zeroed by the reference; ball joints are not handled here */
      double e = scalar[i] ? tq[qadr[i]] - q[qadr[i]] : 0.0;
      double cs = scalar[i] ? cost[i] : 0.0;
      ewv[i] = cs * (-p->posture_gain[t] * e);
      mu += ewv[i] * ewv[i];
    }
    mu *= p->posture_lm[t];
    for (int i = 0; i < nv; ++i) {
      double cs = scalar[i] ? cost[i] : 0.0;
      P[i * nv + i] += cs * cs + mu;
      c[i] += ewv[i] * cs; /* -ew * Jw with Jw = -cost */
    }
  }
  int rows = 0;
  if (p->has_cfg_limit) {
    int nlim = 0, idx[MAXNV];
    for (int j = 0; j < m->njnt; ++j)
      if (m->jnt_type[j] != JNT_FREE && m->jnt_limited[j]) idx[nlim++] = j;
    for (int s = 0; s < 2; ++s)
      for (int k = 0; k < nlim; ++k) {
        int j = idx[k], dof = m->jnt_dofadr[j], qa = m->jnt_qposadr[j];
        double lo = m->jnt_range[2 * j] + p->cfg_min_dist, hi = m->jnt_range[2 * j + 1] - p->cfg_min_dist;
        memset(G + (size_t)rows * nv, 0, nv * sizeof(double));
        G[(size_t)rows * nv + dof] = s == 0 ? 1.0 : -1.0;
        h[rows] = s == 0 ? p->cfg_gain * (hi - q[qa]) : p->cfg_gain * (q[qa] - lo);
        ++rows;
      }
  }
  for (int s = 0; s < 2 && p->nvel > 0; ++s)
    for (int k = 0; k < p->nvel; ++k) {
      memset(G + (size_t)rows * nv, 0, nv * sizeof(double));
      G[(size_t)rows * nv + p->vel_dof[k]] = s == 0 ? 1.0 : -1.0;
      h[rows] = dt * p->vel_max[k];
      ++rows;
    }
  return rows;
}

/* --------------------------------------------------------------- dense QP -- */
static int chol(const double* A, int n, double* L) {
  for (int j = 0; j < n; ++j) {
    double s = A[j * n + j];
    for (int k = 0; k < j; ++k) s -= L[j * n + k] * L[j * n + k];
    if (!(s > 0.0)) return -1;
    L[j * n + j] = sqrt(s);
    for (int i = j + 1; i < n; ++i) {
      double t = A[i * n + j];
      for (int k = 0; k < j; ++k) t -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = t / L[j * n + j];
    }
  }
  return 0;
}
static void chol_solve(const double* L, int n, double* b) {
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i * n + k] * b[k];
    b[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * b[k];
    b[i] = s / L[i * n + i];
  }
}

/* The minimiser on the final active set, recomputed from the data: x = -H^-1 (c + N u) with
 * (N' H^-1 N) u = -(b + N' H^-1 c), then one pass of iterative refinement on the KKT system with residuals
 * from P, c, G, h themselves.  The dual iteration reaches the same point through a chain of updates of x and
 * u whose rounding errors add up (1e-9 relative at cond(P) ~ 1e6); this restatement is the checker, so its
 * answer should not be the less accurate one.  Leaves x alone if the active rows turn out dependent. */
static int g_polish = 1; /* or_set_polish(0): the plain dual iteration, as timed for the CPU baseline */

static void polish_qp(const double* P, const double* c, const double* G, const double* h, int n, const double* L,
                      const int* act, int na, double* x) {
  double HiN[MAXNV][MAXNV], M[MAXNV * MAXNV], LM[MAXNV * MAXNV], u[MAXNV], w[MAXNV], xn[MAXNV];
  for (int a = 0; a < na; ++a) {
    memcpy(HiN[a], G + (size_t)act[a] * n, n * sizeof(double));
    chol_solve(L, n, HiN[a]);
  }
  for (int a = 0; a < na; ++a)
    for (int b = 0; b <= a; ++b) {
      const double* ga = G + (size_t)act[a] * n;
      double s = 0.0;
      for (int i = 0; i < n; ++i) s += ga[i] * HiN[b][i];
      M[a * na + b] = M[b * na + a] = s;
    }
  if (na > 0 && chol(M, na, LM) != 0) return;
  for (int i = 0; i < n; ++i) { xn[i] = 0.0; }
  for (int a = 0; a < na; ++a) u[a] = 0.0;
  for (int pass = 0; pass < 3; ++pass) {
    /* residuals of the current (xn, u): r = P xn + c + N u, s = N' xn - b */
    double r[MAXNV], sres[MAXNV], du[MAXNV];
    for (int i = 0; i < n; ++i) {
      double t = c[i];
      for (int j = 0; j < n; ++j) t += P[i * n + j] * xn[j];
      for (int a = 0; a < na; ++a) t += G[(size_t)act[a] * n + i] * u[a];
      r[i] = t;
    }
    for (int a = 0; a < na; ++a) {
      const double* ga = G + (size_t)act[a] * n;
      double t = -h[act[a]];
      for (int i = 0; i < n; ++i) t += ga[i] * xn[i];
      sres[a] = t;
    }
    /* du = M^-1 (s - N' H^-1 r), dx = -H^-1 r - H^-1 N du */
    memcpy(w, r, n * sizeof(double));
    chol_solve(L, n, w);
    for (int a = 0; a < na; ++a) {
      const double* ga = G + (size_t)act[a] * n;
      double t = sres[a];
      for (int i = 0; i < n; ++i) t -= ga[i] * w[i];
      du[a] = t;
    }
    if (na > 0) chol_solve(LM, na, du);
    for (int i = 0; i < n; ++i) {
      double t = w[i];
      for (int a = 0; a < na; ++a) t += HiN[a][i] * du[a];
      xn[i] -= t;
    }
    for (int a = 0; a < na; ++a) u[a] += du[a];
  }
  for (int i = 0; i < n; ++i)
    if (!isfinite(xn[i])) return;
  /* the polished point must still be the same vertex: multipliers not negative beyond rounding */
  for (int a = 0; a < na; ++a)
    if (u[a] < -1e-9 * (1.0 + fabs(u[a]))) return;
  memcpy(x, xn, n * sizeof(double));
}

/* Goldfarb-Idnani dual active set. Returns 0 ok, 1 iteration limit, 2 infeasible. */
static int solve_qp(const double* P, const double* c, const double* G, const double* h, int n, int m, double* x) {
  double L[MAXNV * MAXNV];
  if (chol(P, n, L) != 0) return 3;
  for (int i = 0; i < n; ++i) x[i] = -c[i];
  chol_solve(L, n, x);
  if (m == 0) return 0;
  int act[MAXNV], na = 0;
  double u[MAXNV];
  double HiN[MAXNV][MAXNV]; /* H^-1 n_k of each active constraint */
  double rn[MAXROWS];
  unsigned char is_act[MAXROWS];
  memset(is_act, 0, m);
  for (int j = 0; j < m; ++j) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += G[(size_t)j * n + i] * G[(size_t)j * n + i];
    rn[j] = sqrt(s);
  }
  for (int it = 0; it < 10 * (m + n); ++it) {
    int p = -1;
    double worst = 1e-12;
    for (int j = 0; j < m; ++j) {
      if (is_act[j] || !(rn[j] > 0.0)) continue;
      double s = -h[j];
      for (int i = 0; i < n; ++i) s += G[(size_t)j * n + i] * x[i];
      s /= rn[j];
      if (s > worst) { worst = s; p = j; }
    }
    if (p < 0) {
      if (g_polish) polish_qp(P, c, G, h, n, L, act, na, x);
      return 0;
    }
    const double* np = G + (size_t)p * n;
    double up = 0.0;
    for (;;) {
      double Hin[MAXNV], z[MAXNV], r[MAXNV], M[MAXNV * MAXNV], LM[MAXNV * MAXNV];
      memcpy(Hin, np, n * sizeof(double));
      chol_solve(L, n, Hin);
      double nHn = 0.0;
      for (int i = 0; i < n; ++i) nHn += np[i] * Hin[i];
      if (na > 0) {
        for (int a = 0; a < na; ++a) {
          const double* ga = G + (size_t)act[a] * n;
          for (int b = 0; b <= a; ++b) {
            double s = 0.0;
            for (int i = 0; i < n; ++i) s += ga[i] * HiN[b][i];
            M[a * na + b] = M[b * na + a] = s;
          }
          double s = 0.0;
          for (int i = 0; i < n; ++i) s += ga[i] * Hin[i];
          r[a] = s;
        }
        if (chol(M, na, LM) != 0) return 3;
        chol_solve(LM, na, r);
      }
      for (int i = 0; i < n; ++i) {
        double s = Hin[i];
        for (int a = 0; a < na; ++a) s -= HiN[a][i] * r[a];
        z[i] = -s;
      }
      double nz = 0.0;
      for (int i = 0; i < n; ++i) nz -= np[i] * z[i];
      int dependent = nz <= 1e-13 * nHn;
      double t1 = INFINITY;
      int kb = -1;
      for (int a = 0; a < na; ++a)
        if (r[a] > 0.0 && u[a] / r[a] < t1) { t1 = u[a] / r[a]; kb = a; }
      double viol = -h[p];
      for (int i = 0; i < n; ++i) viol += np[i] * x[i];
      double t2 = dependent ? INFINITY : viol / nz;
      double t = t1 < t2 ? t1 : t2;
      if (!isfinite(t)) return 2;
      if (!dependent)
        for (int i = 0; i < n; ++i) x[i] += t * z[i];
      for (int a = 0; a < na; ++a) u[a] -= t * r[a];
      up += t;
      if (!dependent && t2 <= t1) {
        act[na] = p;
        u[na] = up;
        memcpy(HiN[na], Hin, n * sizeof(double));
        is_act[p] = 1;
        ++na;
        break;
      }
      is_act[act[kb]] = 0;
      for (int a = kb; a < na - 1; ++a) {
        act[a] = act[a + 1];
        u[a] = u[a + 1];
        memcpy(HiN[a], HiN[a + 1], n * sizeof(double));
      }
      --na;
    }
  }
  return 1;
}

/* mj_integratePos */
static void integrate_pos(const or_model* m, double* q, const double* v, double dt) {
  for (int j = 0; j < m->njnt; ++j) {
    int t = m->jnt_type[j];
    double* qq = q + m->jnt_qposadr[j];
    const double* vv = v + m->jnt_dofadr[j];
    if (t == JNT_HINGE || t == JNT_SLIDE) { qq[0] += dt * vv[0]; continue; }
    if (t == JNT_FREE) {
      for (int i = 0; i < 3; ++i) qq[i] += dt * vv[i];
      qq += 3;
      vv += 3;
    }
    double n = sqrt(vv[0] * vv[0] + vv[1] * vv[1] + vv[2] * vv[2]);
    double ax[3] = {1, 0, 0};
    if (n >= MINVAL) { ax[0] = vv[0] / n; ax[1] = vv[1] / n; ax[2] = vv[2] / n; } else n = 0.0;
    double ang = dt * n, rot[4] = {1, 0, 0, 0};
    if (ang != 0.0) {
      double s = sin(0.5 * ang);
      rot[0] = cos(0.5 * ang); rot[1] = s * ax[0]; rot[2] = s * ax[1]; rot[3] = s * ax[2];
    }
    q_normalize(qq);
    q_mul(qq, rot, qq);
  }
}

/* ------------------------------------------------------------ entry points -- */
int or_build_qp(const or_model* m, const or_problem* p, const double* q, const double* target, double dt,
                double damping, double* P, double* c, double* G, double* h, double* e6) {
  return build_qp(m, p, q, target, dt, damping, P, c, G, h, e6);
}

/* 1 (default): answers are polished on their final active set (checker); 0: the dual iteration alone. */
void or_set_polish(int on) { g_polish = on; }

int or_solve_qp(const double* P, const double* c, const double* G, const double* h, int n, int m, double* x) {
  return solve_qp(P, c, G, h, n, m, x);
}

static int solve_one(const or_model* m, const or_problem* p, const double* q, const double* target, double dt,
                     double damping, double* v, double* e6) {
  int nv = m->nv;
  double P[MAXNV * MAXNV], c[MAXNV], G[MAXROWS * MAXNV], h[MAXROWS], x[MAXNV];
  int rows = build_qp(m, p, q, target, dt, damping, P, c, G, h, e6);
  int st = solve_qp(P, c, G, h, nv, rows, x);
  for (int i = 0; i < nv; ++i) v[i] = x[i] / dt;
  return st;
}

void or_solve_ik_batch(const or_model* m, const or_problem* p, const double* q, const double* targets, double dt,
                       double damping, int B, double* v, int32_t* status, int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int b = 0; b < B; ++b) {
    double e6[6];
    status[b] = solve_one(m, p, q + (size_t)b * m->nq, targets + (size_t)b * 7, dt, damping, v + (size_t)b * m->nv, e6);
  }
}

/* The callers' loop: up to max_iters x (solve_ik, integrate_inplace, error check). */
void or_iterate_batch(const or_model* m, const or_problem* p, double* q, const double* targets, double dt,
                      double damping, int B, int max_iters, double pos_tol, double ori_tol, int32_t* iters,
                      double* err, int32_t* status, int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int b = 0; b < B; ++b) {
    double* qb = q + (size_t)b * m->nq;
    const double* tg = targets + (size_t)b * 7;
    double v[MAXNV], e6[6] = {0};
    int st = 0, n = 0;
    for (int it = 0; it < max_iters; ++it) {
      int s = solve_one(m, p, qb, tg, dt, damping, v, e6);
      if (s > st) st = s;
      integrate_pos(m, qb, v, dt);
      n = it + 1;
      /* frame-task error at the new configuration */
      or_data dd;
      or_data* d = &dd;
      double ps[3], Qs[4], Rs[9];
      kinematics(m, qb, d);
      frame_pose(m, d, p->frame_type, p->frame_id, ps, Qs);
      q_to_mat(Qs, Rs);
      double Qi[4] = {Qs[0], -Qs[1], -Qs[2], -Qs[3]}, qbt[4], dp[3], tbt[3];
      q_mul(Qi, tg, qbt);
      for (int i = 0; i < 3; ++i) dp[i] = tg[4 + i] - ps[i];
      mt_vec(Rs, dp, tbt);
      se3_log(qbt, tbt, e6);
      if (sqrt(e6[0] * e6[0] + e6[1] * e6[1] + e6[2] * e6[2]) <= pos_tol &&
          sqrt(e6[3] * e6[3] + e6[4] * e6[4] + e6[5] * e6[5]) <= ori_tol)
        break;
    }
    status[b] = st;
    if (iters) iters[b] = n;
    if (err) memcpy(err + (size_t)b * 6, e6, sizeof(e6));
  }
}
