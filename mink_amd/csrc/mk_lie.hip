// Batched SO3 / SE3 operations of mink.lie (mink/lie/so3.py, se3.py, base.py) on the device:
// one group element (or tangent) per lane.  The operations are built from the same device
// functions the IK kernels use for their frame tasks (so3_log, so3_exp, se3_log,
// se3_ljacinv_blocks in mk_device.h), so every branch of those -- small angle, |w| ~ 0,
// w < 0, the theta^2 switch of the SE3 Jacobians -- can be driven directly from the host.
#include "../../include/mink_amd.h"
#include "mk_device.h"

namespace {

MK_DEV M3 eye3() {
  M3 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.m[i] = (i % 4 == 0) ? 1.0 : 0.0;
  return r;
}

// SO3.ljac (mink/lie/so3.py:200-212): the switch is on theta, not theta^2.
MK_DEV M3 so3_ljac(V3 w) {
  const double t2 = dot(w, w);
  const double th = sqrt(t2);
  double A, B;
  if (th < MK_LIE_EPS) {
    A = 0.5 * (1.0 - t2 / 12.0 * (1.0 - t2 / 30.0 * (1.0 - t2 / 56.0)));
    B = (1.0 / 6.0) * (1.0 - t2 / 20.0 * (1.0 - t2 / 42.0 * (1.0 - t2 / 72.0)));
  } else {
    double s, c;
    mk_sincos(th, &s, &c);
    A = (1.0 - c) / t2;
    B = (th - s) / (t2 * th);
  }
  return rodrigues_like(w, A, B);
}

// SO3.ljacinv (mink/lie/so3.py:214-226).
MK_DEV M3 so3_ljacinv(V3 w) {
  const double t2 = dot(w, w);
  const double th = sqrt(t2);
  double A;
  if (th < MK_LIE_EPS) {
    A = (1.0 / 12.0) * (1.0 + t2 / 60.0 * (1.0 + t2 / 42.0 * (1.0 + t2 / 40.0)));
  } else {
    double s, c;
    mk_sincos(th, &s, &c);
    A = (1.0 / t2) * (1.0 - (th * s / (2.0 * (1.0 - c))));
  }
  return rodrigues_like(w, -0.5, A);
}

// mju_mat2Quat: four branches on the trace / largest diagonal entry, then normalised; the
// sign of w is not canonicalised (mink/lie/so3.py:79-84).
MK_DEV Q4 mat2quat(const M3& m) {
  Q4 q;
  const double m0 = m.m[0], m4 = m.m[4], m8 = m.m[8];
  if (m0 + m4 + m8 > 0.0) {
    q.w = 0.5 * sqrt(1.0 + m0 + m4 + m8);
    q.x = 0.25 * (m.m[7] - m.m[5]) / q.w;
    q.y = 0.25 * (m.m[2] - m.m[6]) / q.w;
    q.z = 0.25 * (m.m[3] - m.m[1]) / q.w;
  } else if (m0 > m4 && m0 > m8) {
    q.x = 0.5 * sqrt(1.0 + m0 - m4 - m8);
    q.w = 0.25 * (m.m[7] - m.m[5]) / q.x;
    q.y = 0.25 * (m.m[1] + m.m[3]) / q.x;
    q.z = 0.25 * (m.m[2] + m.m[6]) / q.x;
  } else if (m4 > m8) {
    q.y = 0.5 * sqrt(1.0 - m0 + m4 - m8);
    q.w = 0.25 * (m.m[2] - m.m[6]) / q.y;
    q.x = 0.25 * (m.m[1] + m.m[3]) / q.y;
    q.z = 0.25 * (m.m[5] + m.m[7]) / q.y;
  } else {
    q.z = 0.5 * sqrt(1.0 - m0 - m4 + m8);
    q.w = 0.25 * (m.m[3] - m.m[1]) / q.z;
    q.x = 0.25 * (m.m[2] + m.m[6]) / q.z;
    q.y = 0.25 * (m.m[5] + m.m[7]) / q.z;
  }
  return qnormalize(q);
}

struct Pose {
  Q4 q;
  V3 t;
};

// SO3.apply as the reference writes it, q (0, v) q^-1 with q^-1 the conjugate
// (mink/lie/so3.py:143-146); equal to the rotation by q for unit quaternions.
MK_DEV V3 qrot_ref(Q4 q, V3 v) {
  const Q4 r = qmul(qmul(q, Q4{0.0, v.x, v.y, v.z}), qconj(q));
  return v3(r.x, r.y, r.z);
}

MK_DEV Pose pose_mul(Pose a, Pose b) { return Pose{qmul(a.q, b.q), qrot_ref(a.q, b.t) + a.t}; }
MK_DEV Pose pose_inv(Pose a) {
  const Q4 qi = qconj(a.q);
  return Pose{qi, neg(qrot_ref(qi, a.t))};
}

// SE3.exp (mink/lie/se3.py:112-134).
MK_DEV Pose se3_exp(Twist xi) {
  const Q4 q = so3_exp(xi.w);
  const double t2 = dot(xi.w, xi.w);
  M3 V;
  if (t2 < MK_LIE_EPS) {
    V = q2m(q);
  } else {
    const double th = sqrt(t2);
    double s, c;
    mk_sincos(th, &s, &c);
    V = rodrigues_like(xi.w, (1.0 - c) / t2, (th - s) / (t2 * th));
  }
  return Pose{q, mul(V, xi.v)};
}

// SE3.ljac / SE3.ljacinv (mink/lie/se3.py:199-218) as a row-major 6 x 6.
MK_DEV void se3_ljac6(Twist xi, double* out) {
  M3 J, Q;
  const double t2 = dot(xi.w, xi.w);
  if (t2 < MK_LIE_EPS) {
    J = eye3();
#pragma unroll
    for (int i = 0; i < 9; ++i) Q.m[i] = 0.0;
  } else {
    const double rth = mk_rsqrt(t2), th = t2 * rth;
    double s, c;
    mk_sincos(th, &s, &c);
    J = so3_ljac(xi.w);
    Q = se3_Q_closed(xi, t2, rth, th, s, c);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      out[6 * i + j] = J.m[3 * i + j];
      out[6 * i + 3 + j] = Q.m[3 * i + j];
      out[6 * (i + 3) + j] = 0.0;
      out[6 * (i + 3) + 3 + j] = J.m[3 * i + j];
    }
}

MK_DEV void se3_ljacinv6(Twist xi, double* out) {
  M3 Ji, B;
  se3_ljacinv_blocks(xi, Ji, B);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      out[6 * i + j] = Ji.m[3 * i + j];
      out[6 * i + 3 + j] = B.m[3 * i + j];
      out[6 * (i + 3) + j] = 0.0;
      out[6 * (i + 3) + 3 + j] = Ji.m[3 * i + j];
    }
}

MK_DEV Twist tneg(Twist x) { return Twist{neg(x.v), neg(x.w)}; }
MK_DEV Pose rd_pose(const double* p) { return Pose{Q4{p[0], p[1], p[2], p[3]}, v3(p[4], p[5], p[6])}; }
MK_DEV Q4 rd_quat(const double* p) { return Q4{p[0], p[1], p[2], p[3]}; }
MK_DEV V3 rd_v3(const double* p) { return v3(p[0], p[1], p[2]); }
MK_DEV Twist rd_twist(const double* p) { return Twist{v3(p[0], p[1], p[2]), v3(p[3], p[4], p[5])}; }
MK_DEV void wr_pose(double* o, Pose p) {
  o[0] = p.q.w; o[1] = p.q.x; o[2] = p.q.y; o[3] = p.q.z; o[4] = p.t.x; o[5] = p.t.y; o[6] = p.t.z;
}
MK_DEV void wr_quat(double* o, Q4 q) { o[0] = q.w; o[1] = q.x; o[2] = q.y; o[3] = q.z; }
MK_DEV void wr_v3(double* o, V3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }
MK_DEV void wr_twist(double* o, Twist t) { wr_v3(o, t.v); wr_v3(o + 3, t.w); }
MK_DEV void wr_m3(double* o, const M3& m) {
#pragma unroll
  for (int i = 0; i < 9; ++i) o[i] = m.m[i];
}

struct OpShape {
  int wa, wb, wo;
};

// widths (doubles per element) of the two inputs and the output of each operation
__host__ __device__ inline OpShape op_shape(int op) {
  switch (op) {
    case MK_SO3_EXP: return {3, 0, 4};
    case MK_SO3_LOG: return {4, 0, 3};
    case MK_SO3_MULTIPLY: return {4, 4, 4};
    case MK_SO3_INVERSE: return {4, 0, 4};
    case MK_SO3_NORMALIZE: return {4, 0, 4};
    case MK_SO3_APPLY: return {4, 3, 3};
    case MK_SO3_AS_MATRIX: return {4, 0, 9};
    case MK_SO3_FROM_MATRIX: return {9, 0, 4};
    case MK_SO3_LJAC: return {3, 0, 9};
    case MK_SO3_LJACINV: return {3, 0, 9};
    case MK_SO3_RJAC: return {3, 0, 9};
    case MK_SO3_RJACINV: return {3, 0, 9};
    case MK_SO3_JLOG: return {4, 0, 9};
    case MK_SO3_RPLUS: return {4, 3, 4};
    case MK_SO3_LPLUS: return {4, 3, 4};
    case MK_SO3_RMINUS: return {4, 4, 3};
    case MK_SO3_LMINUS: return {4, 4, 3};
    case MK_SO3_AS_RPY: return {4, 0, 3};
    case MK_SO3_FROM_RPY: return {3, 0, 4};
    case MK_SE3_EXP: return {6, 0, 7};
    case MK_SE3_LOG: return {7, 0, 6};
    case MK_SE3_MULTIPLY: return {7, 7, 7};
    case MK_SE3_INVERSE: return {7, 0, 7};
    case MK_SE3_NORMALIZE: return {7, 0, 7};
    case MK_SE3_APPLY: return {7, 3, 3};
    case MK_SE3_AS_MATRIX: return {7, 0, 16};
    case MK_SE3_FROM_MATRIX: return {16, 0, 7};
    case MK_SE3_ADJOINT: return {7, 0, 36};
    case MK_SE3_LJAC: return {6, 0, 36};
    case MK_SE3_LJACINV: return {6, 0, 36};
    case MK_SE3_RJAC: return {6, 0, 36};
    case MK_SE3_RJACINV: return {6, 0, 36};
    case MK_SE3_JLOG: return {7, 0, 36};
    case MK_SE3_RPLUS: return {7, 6, 7};
    case MK_SE3_LPLUS: return {7, 6, 7};
    case MK_SE3_RMINUS: return {7, 7, 6};
    case MK_SE3_LMINUS: return {7, 7, 6};
    default: return {0, 0, 0};
  }
}

__global__ __launch_bounds__(256) void lie_kernel(int op, const double* __restrict__ a_all,
                                                  const double* __restrict__ b_all, int B,
                                                  double* __restrict__ out_all) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const OpShape sh = op_shape(op);
  const double* a = a_all + i * sh.wa;
  const double* b = b_all ? b_all + i * sh.wb : nullptr;
  double* o = out_all + i * sh.wo;
  switch (op) {
    // ---- SO3 (mink/lie/so3.py) ----
    case MK_SO3_EXP: wr_quat(o, so3_exp(rd_v3(a))); break;
    case MK_SO3_LOG: wr_v3(o, so3_log(rd_quat(a))); break;
    case MK_SO3_MULTIPLY: wr_quat(o, qmul(rd_quat(a), rd_quat(b))); break;
    case MK_SO3_INVERSE: wr_quat(o, qconj(rd_quat(a))); break;
    case MK_SO3_NORMALIZE: {
      const Q4 q = rd_quat(a);
      const double r = 1.0 / sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
      wr_quat(o, Q4{q.w * r, q.x * r, q.y * r, q.z * r});
      break;
    }
    case MK_SO3_APPLY: wr_v3(o, qrot_ref(rd_quat(a), rd_v3(b))); break;
    case MK_SO3_AS_MATRIX: wr_m3(o, q2m(rd_quat(a))); break;
    case MK_SO3_FROM_MATRIX: {
      M3 m;
#pragma unroll
      for (int k = 0; k < 9; ++k) m.m[k] = a[k];
      wr_quat(o, mat2quat(m));
      break;
    }
    case MK_SO3_LJAC: wr_m3(o, so3_ljac(rd_v3(a))); break;
    case MK_SO3_LJACINV: wr_m3(o, so3_ljacinv(rd_v3(a))); break;
    case MK_SO3_RJAC: wr_m3(o, so3_ljac(neg(rd_v3(a)))); break;
    case MK_SO3_RJACINV: wr_m3(o, so3_ljacinv(neg(rd_v3(a)))); break;
    case MK_SO3_JLOG: wr_m3(o, so3_ljacinv(neg(so3_log(rd_quat(a))))); break;   // base.py:155
    case MK_SO3_RPLUS: wr_quat(o, qmul(rd_quat(a), so3_exp(rd_v3(b)))); break;   // base.py:109
    case MK_SO3_LPLUS: wr_quat(o, qmul(so3_exp(rd_v3(b)), rd_quat(a))); break;   // base.py:117
    case MK_SO3_RMINUS: wr_v3(o, so3_log(qmul(qconj(rd_quat(b)), rd_quat(a)))); break;   // base.py:113
    case MK_SO3_LMINUS: wr_v3(o, so3_log(qmul(rd_quat(a), qconj(rd_quat(b))))); break;   // base.py:121
    case MK_SO3_AS_RPY: wr_v3(o, quat_rpy(rd_quat(a))); break;
    case MK_SO3_FROM_RPY: {   // Rz(yaw) Ry(pitch) Rx(roll), so3.py:66-77
      const V3 r = rd_v3(a);
      wr_quat(o, qmul(qmul(so3_exp(v3(0, 0, r.z)), so3_exp(v3(0, r.y, 0))), so3_exp(v3(r.x, 0, 0))));
      break;
    }
    // ---- SE3 (mink/lie/se3.py) ----
    case MK_SE3_EXP: wr_pose(o, se3_exp(rd_twist(a))); break;
    case MK_SE3_LOG: {
      const Pose p = rd_pose(a);
      wr_twist(o, se3_log(p.q, p.t));
      break;
    }
    case MK_SE3_MULTIPLY: wr_pose(o, pose_mul(rd_pose(a), rd_pose(b))); break;
    case MK_SE3_INVERSE: wr_pose(o, pose_inv(rd_pose(a))); break;
    case MK_SE3_NORMALIZE: {
      Pose p = rd_pose(a);
      const double r = 1.0 / sqrt(p.q.w * p.q.w + p.q.x * p.q.x + p.q.y * p.q.y + p.q.z * p.q.z);
      p.q = Q4{p.q.w * r, p.q.x * r, p.q.y * r, p.q.z * r};
      wr_pose(o, p);
      break;
    }
    case MK_SE3_APPLY: {
      const Pose p = rd_pose(a);
      wr_v3(o, qrot_ref(p.q, rd_v3(b)) + p.t);
      break;
    }
    case MK_SE3_AS_MATRIX: {
      const Pose p = rd_pose(a);
      const M3 R = q2m(p.q);
      const double t[3] = {p.t.x, p.t.y, p.t.z};
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) o[4 * r + c] = R.m[3 * r + c];
        o[4 * r + 3] = t[r];
      }
      o[12] = 0.0; o[13] = 0.0; o[14] = 0.0; o[15] = 1.0;
      break;
    }
    case MK_SE3_FROM_MATRIX: {
      M3 m;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) m.m[3 * r + c] = a[4 * r + c];
      wr_pose(o, Pose{mat2quat(m), v3(a[3], a[7], a[11])});
      break;
    }
    case MK_SE3_ADJOINT: {   // [[R, [t]x R], [0, R]], se3.py:187-194
      const Pose p = rd_pose(a);
      const M3 R = q2m(p.q);
      const M3 TR = mmul(skew(p.t), R);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          o[6 * r + c] = R.m[3 * r + c];
          o[6 * r + 3 + c] = TR.m[3 * r + c];
          o[6 * (r + 3) + c] = 0.0;
          o[6 * (r + 3) + 3 + c] = R.m[3 * r + c];
        }
      break;
    }
    case MK_SE3_LJAC: se3_ljac6(rd_twist(a), o); break;
    case MK_SE3_LJACINV: se3_ljacinv6(rd_twist(a), o); break;
    case MK_SE3_RJAC: se3_ljac6(tneg(rd_twist(a)), o); break;
    case MK_SE3_RJACINV: se3_ljacinv6(tneg(rd_twist(a)), o); break;
    case MK_SE3_JLOG: {   // rjacinv(log(T)) = ljacinv(-log(T)), base.py:151-155
      const Pose p = rd_pose(a);
      se3_ljacinv6(tneg(se3_log(p.q, p.t)), o);
      break;
    }
    case MK_SE3_RPLUS: wr_pose(o, pose_mul(rd_pose(a), se3_exp(rd_twist(b)))); break;
    case MK_SE3_LPLUS: wr_pose(o, pose_mul(se3_exp(rd_twist(b)), rd_pose(a))); break;
    case MK_SE3_RMINUS: {
      const Pose p = pose_mul(pose_inv(rd_pose(b)), rd_pose(a));
      wr_twist(o, se3_log(p.q, p.t));
      break;
    }
    case MK_SE3_LMINUS: {
      const Pose p = pose_mul(rd_pose(a), pose_inv(rd_pose(b)));
      wr_twist(o, se3_log(p.q, p.t));
      break;
    }
    default: break;
  }
}

}  // namespace

int mk_lie_shape(int op, int* wa, int* wb, int* wo) {
  const OpShape sh = op_shape(op);
  *wa = sh.wa; *wb = sh.wb; *wo = sh.wo;
  return sh.wo > 0;
}

hipError_t mk_lie_launch(int op, const double* a, const double* b, int B, double* out, hipStream_t s) {
  hipLaunchKernelGGL(lie_kernel, dim3((B + 255) / 256), dim3(256), 0, s, op, a, b, B, out);
  return hipGetLastError();
}
