// Host side of the C ABI declared in include/mink_amd.h: opaque handles, validation,
// flattening of model / task / limit descriptors into the device descriptors the
// kernels read, and kernel dispatch.  No device memory is allocated by the solve
// calls themselves; handles own the (small) descriptor buffers.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <algorithm>
#include <vector>

#include "../../include/mink_amd.h"
#include "mk_device.h"
#include "mk_generic.h"
#include "mk_options.h"

#include <atomic>

// ---- process-wide switches (mk_options.h) ----
namespace {
struct MkOptionEntry {
  const char* name;
  long initial;
};
const MkOptionEntry kOptions[MK_OPT_COUNT] = {
    {"staged", 1},        {"lane_rows", 1},     {"general_stage", 1}, {"staged_loop", 1},        {"stream_blocks", -1},
    {"stream_refill", 16}, {"coop", -1},        {"coop_max_batch", 16384}, {"handoff", -1},
    {"f32", 0},
};
std::atomic<long>* option_table() {
  static std::atomic<long> table[MK_OPT_COUNT];
  static const bool init = [] {
    for (int i = 0; i < MK_OPT_COUNT; ++i) table[i].store(kOptions[i].initial, std::memory_order_relaxed);
    return true;
  }();
  (void)init;
  return table;
}
}  // namespace
long mk_opt(MkOption o) { return option_table()[o].load(std::memory_order_relaxed); }

// ---- launchers implemented in mk_coop.hip ----
hipError_t mk_coop_launch_step(int nc, const MkFusedDesc* d_desc, const double* q, const double* targets,
                               int target_stride, double dt, double damping, int B, double* v, int32_t* status,
                               hipStream_t stream);
hipError_t mk_coop_launch_iterate(int nc, const MkFusedDesc* d_desc, double* q, const double* targets,
                                  int target_stride, double dt, double damping, int B, int max_iters, double pos_tol,
                                  double ori_tol, double* v, int32_t* iters, double* err, int32_t* status,
                                  hipStream_t stream);
hipError_t mk_coop_launch_move_to(int nc, const MkFusedDesc* d_desc, double* q, const double* goals, double dt,
                                  double damping, int B, int max_steps, int max_iters, double pos_tol, double ori_tol,
                                  double wp_pos_thr, double wp_pos_step, double wp_rot_thr, int32_t* steps,
                                  int32_t* reached, int32_t* solves, int32_t* status, hipStream_t stream);
hipError_t mk_coop_launch_move_to_resume(int nc, const MkFusedDesc* d_desc, double* q, const double* goals, double dt,
                                         double damping, int B, int max_steps, int max_iters, double pos_tol,
                                         double ori_tol, double wp_pos_thr, double wp_pos_step, double wp_rot_thr,
                                         int32_t* steps, int32_t* reached, int32_t* solves, int32_t* status,
                                         int max_blocks, hipStream_t stream);
hipError_t mk_coop_launch_resume(int nc, const MkFusedDesc* d_desc, double* q, const double* targets, int target_stride,
                                 double dt, double damping, int B, int max_iters, double pos_tol, double ori_tol,
                                 double* v, int32_t* iters, double* err, int32_t* status, int max_blocks,
                                 hipStream_t stream);

// ---- launchers implemented in mk_fused.hip ----
hipError_t mk_fused_launch_step(int nc, const MkFusedDesc* d_desc, int nq, const double* q, const double* targets,
                                int target_stride, double dt, double damping, int B, double* v, int32_t* status,
                                hipStream_t stream);
hipError_t mk_fused_launch_step_ext(int nc, const MkFusedDesc* d_desc, const MkFusedExt* d_ext, int nq, const double* q,
                                    const double* targets, int target_stride, double dt, double damping, int B,
                                    double* v, int32_t* status, hipStream_t stream, int screen_doubles,
                                    int32_t* iters = nullptr, double* err = nullptr, int it = 0, int max_iters = 0,
                                    double pos_tol = 0.0, double ori_tol = 0.0);
hipError_t mk_fused_launch_rows(int nc, const MkFusedDesc* d_desc, const MkFusedExt* d_ext, int nv, const double* q,
                                const double* targets, int target_stride, double dt, double damping, int B, double* v,
                                int32_t* status, hipStream_t stream);
hipError_t mk_fused_launch_iterate(int nc, const MkFusedDesc* d_desc, int nq, double* q, const double* targets,
                                   int target_stride, double dt, double damping, int B, int max_iters,
                                   double pos_tol, double ori_tol, double* v, int32_t* iters, double* err,
                                   int32_t* status, hipStream_t stream, int handoff_live, bool* handed);
hipError_t mk_fused_launch_move_to(int nc, const MkFusedDesc* d_desc, int nq, double* q, const double* goals,
                                   double dt, double damping, int B, int max_steps, int max_iters, double pos_tol,
                                   double ori_tol, double wp_pos_thr, double wp_pos_step, double wp_rot_thr,
                                   int32_t* steps, int32_t* reached, int32_t* solves, int32_t* status,
                                   hipStream_t stream, int handoff_live, bool* handed);

// ---- the same three in float32 arithmetic (mk_fused32.hip); q, targets and v stay float64 ----
hipError_t mk_fused32_launch_step(int nc, const MkFusedDesc* d_desc, int nq, const double* q, const double* targets,
                                  int target_stride, double dt, double damping, int B, double* v, int32_t* status,
                                  hipStream_t stream);
hipError_t mk_fused32_launch_iterate(int nc, const MkFusedDesc* d_desc, int nq, double* q, const double* targets,
                                     int target_stride, double dt, double damping, int B, int max_iters,
                                     double pos_tol, double ori_tol, double* v, int32_t* iters, double* err,
                                     int32_t* status, hipStream_t stream, int handoff_live, bool* handed);
hipError_t mk_fused32_launch_move_to(int nc, const MkFusedDesc* d_desc, int nq, double* q, const double* goals,
                                     double dt, double damping, int B, int max_steps, int max_iters, double pos_tol,
                                     double ori_tol, double wp_pos_thr, double wp_pos_step, double wp_rot_thr,
                                     int32_t* steps, int32_t* reached, int32_t* solves, int32_t* status,
                                     hipStream_t stream, int handoff_live, bool* handed);

// ---- implemented in mk_lie.hip ----
int mk_lie_shape(int op, int* wa, int* wb, int* wo);
hipError_t mk_lie_launch(int op, const double* a, const double* b, int B, double* out, hipStream_t s);

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define MK_HIP(expr)                                                                     \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) return fail(MK_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

template <typename T>
std::vector<T> copy_in(const T* p, size_t n) {
  return p ? std::vector<T>(p, p + n) : std::vector<T>(n, T());
}

struct HostModel {
  int nq, nv, nbody, njnt, ngeom, nsite;
  std::vector<int32_t> body_parentid, body_jntadr, body_jntnum, body_mocapid;
  std::vector<double> body_pos, body_quat, body_mass, body_ipos;
  std::vector<int32_t> jnt_type, jnt_qposadr, jnt_dofadr, jnt_bodyid, jnt_limited;
  std::vector<double> jnt_pos, jnt_axis, jnt_range, qpos0;
  std::vector<int32_t> dof_parentid, dof_jntid;
  std::vector<int32_t> site_bodyid;
  std::vector<double> site_pos, site_quat;
  std::vector<int32_t> geom_bodyid, geom_type;
  std::vector<double> geom_size, geom_pos, geom_quat;
  int nmesh, nmeshvert;
  std::vector<int32_t> geom_dataid, mesh_vertadr, mesh_vertnum;
  std::vector<double> mesh_vert, mesh_center, mesh_radius;
};

void hq_mul(const double* a, const double* b, double* r) {
  double t[4] = {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                 a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                 a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
                 a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
  memcpy(r, t, sizeof(t));
}
void hq_normalize(double* q) {
  double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < 1e-15) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  for (int i = 0; i < 4; ++i) q[i] /= n;
}
void hq_rotate(const double* q, const double* v, double* r) {
  double q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
  double q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
  double q12 = q[1] * q[2], q13 = q[1] * q[3], q23 = q[2] * q[3];
  double m[9] = {q00 + q11 - q22 - q33, 2 * (q12 - q03), 2 * (q13 + q02),
                 2 * (q12 + q03), q00 - q11 + q22 - q33, 2 * (q23 - q01),
                 2 * (q13 - q02), 2 * (q23 + q01), q00 - q11 - q22 + q33};
  double t[3] = {m[0] * v[0] + m[1] * v[1] + m[2] * v[2], m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
                 m[6] * v[0] + m[7] * v[1] + m[8] * v[2]};
  memcpy(r, t, sizeof(t));
}

// pose <- pose o (pos, quat)
void compose(double* ppos, double* pquat, const double* pos, const double* quat) {
  double r[3];
  hq_rotate(pquat, pos, r);
  for (int i = 0; i < 3; ++i) ppos[i] += r[i];
  hq_mul(pquat, quat, pquat);
  hq_normalize(pquat);
}

}  // namespace

struct mk_model {
  HostModel h;
  MkDevModel dev;          // device pointers (owned)
  std::vector<void*> allocs;
  int device;
};

struct mk_problem {
  const mk_model* model;
  mk_problem_info info;
  bool fused;
  int nc_kernel;            // template size used by the fused kernels
  MkFusedDesc* d_fused;     // device copy
  MkFusedDesc h_fused;
  bool staged;              // structured kernel first, general kernel for problems with a collision pair in range
  MkFusedExt h_ext;
  MkFusedExt* d_ext;
  bool dense;               // the dense (general) path can serve this problem
  std::string dense_why;    // why not, otherwise
  MkDenseProblem h_dense;
  MkDenseProblem* d_dense;
  std::vector<int> task_k;  // error dimension of each task
  std::vector<int> task_type;
  std::vector<void*> allocs;
};

namespace {

// A handle's tables live on the device that was current when it was created; a launch from
// another device would hand the kernels pointers of a foreign address space.
int check_device(const mk_model* m) {
  int cur = -1;
  MK_HIP(hipGetDevice(&cur));
  if (cur != m->device)
    return fail(MK_ERR_INVALID, "the handle was created on device %d but the current device is %d", m->device, cur);
  return MK_OK;
}
#define MK_ON_DEVICE(model)                  \
  do {                                       \
    const int _rc = check_device(model);     \
    if (_rc != MK_OK) return _rc;            \
  } while (0)

template <typename T>
int upload(std::vector<void*>& allocs, const std::vector<T>& v, const T** out) {
  void* p = nullptr;
  size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  MK_HIP(hipMalloc(&p, bytes));
  allocs.push_back(p);
  if (!v.empty()) MK_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = static_cast<const T*>(p);
  return MK_OK;
}

int frame_body_and_local(const HostModel& m, int frame_type, int frame_id, int* body, double* pos, double* quat) {
  pos[0] = pos[1] = pos[2] = 0.0;
  quat[0] = 1.0; quat[1] = quat[2] = quat[3] = 0.0;
  if (frame_type == MK_FRAME_BODY) {
    if (frame_id < 0 || frame_id >= m.nbody) return fail(MK_ERR_INVALID, "body id %d out of range", frame_id);
    *body = frame_id;
  } else if (frame_type == MK_FRAME_SITE) {
    if (frame_id < 0 || frame_id >= m.nsite) return fail(MK_ERR_INVALID, "site id %d out of range", frame_id);
    *body = m.site_bodyid[frame_id];
    memcpy(pos, &m.site_pos[3 * frame_id], 3 * sizeof(double));
    memcpy(quat, &m.site_quat[4 * frame_id], 4 * sizeof(double));
  } else if (frame_type == MK_FRAME_GEOM) {
    if (frame_id < 0 || frame_id >= m.ngeom) return fail(MK_ERR_INVALID, "geom id %d out of range", frame_id);
    *body = m.geom_bodyid[frame_id];
    memcpy(pos, &m.geom_pos[3 * frame_id], 3 * sizeof(double));
    memcpy(quat, &m.geom_quat[4 * frame_id], 4 * sizeof(double));
  } else {
    return fail(MK_ERR_INVALID, "unknown frame type %d", frame_type);
  }
  // A mocap body is placed by the caller's mocap_pos / mocap_quat (mj_kinematics reads them
  // from MjData), which the kernels never see: frames on it are the host's business.
  if (m.body_mocapid[*body] >= 0)
    return fail(MK_ERR_UNSUPPORTED, "frame %d of type %d is attached to a mocap body, whose pose is not part of q", frame_id, frame_type);
  return MK_OK;
}

// Can the kernels compute the distance of geoms g1, g2?  Closed forms exist for plane / sphere /
// capsule / box pairs (except plane - plane); every other pair of convex geoms (ellipsoid,
// cylinder, mesh with a hull, and those against the primitives) goes by support mappings.
// MuJoCo geom types: plane 0, hfield 1, sphere 2, capsule 3, ellipsoid 4, cylinder 5, box 6, mesh 7.
bool pair_supported(const HostModel& m, int g1, int g2, bool* convex, std::string& why) {
  int t1 = m.geom_type[g1], t2 = m.geom_type[g2];
  if (t1 > t2) { std::swap(t1, t2); std::swap(g1, g2); }
  auto usable = [&](int g) {
    const int t = m.geom_type[g];
    if (t == 0 || (t >= 2 && t <= 6)) return true;
    return t == 7 && m.geom_dataid[g] >= 0 && m.mesh_vertnum[m.geom_dataid[g]] > 0;
  };
  if (!usable(g1) || !usable(g2)) { why = "geom without a distance routine (height field, or a mesh without hull vertices in the model)"; return false; }
  if (t1 == 0 && t2 == 0) { why = "plane - plane pair"; return false; }
  const bool closed = (t1 == 0 && (t2 == 2 || t2 == 3 || t2 == 6)) || (t1 == 2 && (t2 == 2 || t2 == 3 || t2 == 6)) ||
                      (t1 == 3 && (t2 == 3 || t2 == 6)) || (t1 == 6 && t2 == 6);
  if (convex) *convex = !closed;
  return true;
}

const int kFusedSizes[] = {4, 6, 7, 8, 10, 12};

// Brings 1-dof joint j to canonical form.  (ppos, pquat) is the fixed transform accumulated
// since the previous joint; on return (pq, pp) is the fixed transform in front of the joint
// (its qpos0 folded in) and (ppos, pquat) the one behind it.  A joint with unit axis a and
// anchor p in its body frame is F Jz(q - qpos0) F^-1 with F = (translation p, rotation taking
// z to a) and Jz the motion about / along z.
bool canonical_joint(const HostModel& m, int j, double* ppos, double* pquat, double* pq, double* pp, std::string& why) {
  const int jt = m.jnt_type[j];
  double a[3] = {m.jnt_axis[3 * j], m.jnt_axis[3 * j + 1], m.jnt_axis[3 * j + 2]};
  const double an = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  if (!(an > MK_MINVAL)) { why = "zero joint axis"; return false; }
  for (double& c : a) c /= an;
  // F: shortest rotation taking z to a (half-turn about x when a = -z), anchored at the joint for hinges
  double fq[4] = {1.0 + a[2], -a[1], a[0], 0.0};
  if (1.0 + a[2] < 1e-12) { fq[0] = 0.0; fq[1] = 1.0; fq[2] = 0.0; }
  hq_normalize(fq);
  const double zero3[3] = {0, 0, 0};
  const double* fp = (jt == MK_JNT_HINGE) ? &m.jnt_pos[3 * j] : zero3;
  compose(ppos, pquat, fp, fq);
  const double q0 = m.qpos0[m.jnt_qposadr[j]];
  if (q0 != 0.0) {  // Jz(-qpos0)
    const double zq[4] = {std::cos(-0.5 * q0), 0.0, 0.0, std::sin(-0.5 * q0)};
    const double zp[3] = {0.0, 0.0, -q0};
    const double id[4] = {1, 0, 0, 0};
    if (jt == MK_JNT_HINGE) compose(ppos, pquat, zero3, zq); else compose(ppos, pquat, zp, id);
  }
  memcpy(pq, pquat, 4 * sizeof(double));
  memcpy(pp, ppos, 3 * sizeof(double));
  // behind the joint: F^-1 = (rotation conj(fq), translation -R(conj(fq)) fp)
  const double fqi[4] = {fq[0], -fq[1], -fq[2], -fq[3]};
  double r[3];
  hq_rotate(fqi, fp, r);
  ppos[0] = -r[0]; ppos[1] = -r[1]; ppos[2] = -r[2];
  memcpy(pquat, fqi, sizeof(fqi));
  return true;
}

struct BodyAnchor {   // body frame (after its joints) in the frame behind chain slot `slot` (-1: the world)
  int slot;
  double pos[3], quat[4];
};

// Tries to express the problem in the structured form the fused kernels solve.
// Returns false (with `why`) when the generic dense path is needed instead.
// With X given, one RelativeFrameTask on the frame task's frame and CollisionAvoidanceLimits
// are accepted as well and described in X (see MkFusedExt); *needs_ext tells whether X is used.
bool build_fused(const HostModel& m, const mk_task_desc* tasks, int ntask, const mk_limit_desc* limits, int nlimit,
                 MkFusedDesc& D, int* nc_kernel, std::string& why, MkFusedExt* X = nullptr, bool* needs_ext = nullptr) {
  memset(&D, 0, sizeof(D));
  if (X) memset(X, 0, sizeof(*X));
  if (needs_ext) *needs_ext = false;
  if (m.nv > MK_MAX_NV || m.nq > MK_MAX_NQ) { why = "model too large"; return false; }
  int frame_task = -1, rel_task = -1, nposture = 0, off = 0, target_off = 0;
  for (int t = 0; t < ntask; ++t) {
    if (tasks[t].type == MK_TASK_FRAME) {
      if (frame_task >= 0) { why = "more than one frame task"; return false; }
      frame_task = t;
      target_off = off;
      off += 7;
    } else if (tasks[t].type == MK_TASK_RELATIVE_FRAME && X) {
      if (rel_task >= 0) { why = "more than one relative frame task"; return false; }
      rel_task = t;
      X->has_rel = 1;
      X->rel_target_off = off;
      off += 7;
    } else if (tasks[t].type == MK_TASK_POSTURE) {
      if (++nposture > MK_MAX_POSTURE) { why = "too many posture tasks"; return false; }
    } else {
      why = "task type needs the dense path";
      return false;
    }
  }
  if (frame_task < 0) { why = "no frame task"; return false; }
  bool any_collision = false;
  for (int l = 0; l < nlimit; ++l)
    if (limits[l].type == MK_LIMIT_COLLISION) {
      if (!X) { why = "collision limit adds general rows"; return false; }
      any_collision = true;
    }
  if (needs_ext) *needs_ext = any_collision || rel_task >= 0;
  for (int j = 0; j < m.njnt; ++j)
    if (m.jnt_type[j] == MK_JNT_BALL) { why = "ball joint"; return false; }

  D.nq = m.nq;
  D.nv = m.nv;
  D.target_off = target_off;
  // dof table
  for (int j = 0; j < m.njnt; ++j) {
    const int a = m.jnt_dofadr[j], qa = m.jnt_qposadr[j];
    if (m.jnt_type[j] == MK_JNT_FREE) {
      if (D.nfree >= 8) { why = "too many free joints"; return false; }
      D.free_dofadr[D.nfree] = a;
      D.free_qadr[D.nfree++] = qa;
      for (int k = 0; k < 6; ++k) {
        D.dof[a + k].qadr = qa + (k < 3 ? k : 3);
        D.dof[a + k].kind = k < 3 ? MK_DOF_FREE_LIN : MK_DOF_FREE_ROT;
        D.dof[a + k].slot = -1;
      }
    } else {
      D.dof[a].qadr = qa;
      D.dof[a].kind = MK_DOF_SCALAR;
      D.dof[a].slot = -1;
    }
  }
  // chain of the frame task
  const mk_task_desc& ft = tasks[frame_task];
  int body;
  double lpos[3], lquat[4];
  if (frame_body_and_local(m, ft.frame_type, ft.frame_id, &body, lpos, lquat) != MK_OK) { why = g_last_error; return false; }
  std::vector<int> path;
  for (int b = body; b != 0; b = m.body_parentid[b]) path.insert(path.begin(), b);
  // Walk the chain and bring every joint to canonical form.  A joint with unit axis a and
  // anchor j in its body frame is F Jz(q - qpos0) F^-1 with F = (translation j, rotation
  // taking z to a) and Jz the motion about / along z; F and Jz(-qpos0) go into the fixed
  // transform in front of the joint, F^-1 into the one behind it (`carry`).
  double ppos[3] = {0, 0, 0}, pquat[4] = {1, 0, 0, 0};   // fixed transform since the previous joint
  int nc = 0;
  std::vector<BodyAnchor> anchor(m.nbody, BodyAnchor{-2, {0, 0, 0}, {1, 0, 0, 0}});   // -2: not on the path
  for (int b : path) {
    if (m.body_mocapid[b] >= 0) { why = "frame attached to a mocap body"; return false; }
    compose(ppos, pquat, &m.body_pos[3 * b], &m.body_quat[4 * b]);
    const int jn = m.body_jntnum[b], ja = m.body_jntadr[b];
    for (int j = ja; j < ja + jn; ++j) {
      const int jt = m.jnt_type[j];
      if (jt != MK_JNT_HINGE && jt != MK_JNT_SLIDE) { why = "free/ball joint on the task chain"; return false; }
      if (nc >= MK_MAX_CHAIN) { why = "task chain longer than the fused kernels support"; return false; }
      MkSegC& sc = D.segc[nc];
      if (!canonical_joint(m, j, ppos, pquat, sc.pq, sc.pp, why)) { why += " on the task chain"; return false; }
      sc.hinge = (jt == MK_JNT_HINGE) ? 1.0 : 0.0;
      sc.slide = 1.0 - sc.hinge;
      D.seg_qadr[nc] = m.jnt_qposadr[j];
      D.seg_dof[nc] = m.jnt_dofadr[j];
      D.dof[m.jnt_dofadr[j]].slot = nc;
      ++nc;
    }
    anchor[b].slot = nc - 1;
    memcpy(anchor[b].pos, ppos, sizeof(ppos));
    memcpy(anchor[b].quat, pquat, sizeof(pquat));
  }
  if (nc == 0) { why = "task frame has no moving joints"; return false; }
  compose(ppos, pquat, lpos, lquat);
  memcpy(D.post_pos, ppos, sizeof(ppos));
  memcpy(D.post_quat, pquat, sizeof(pquat));
  int nk = -1;
  for (int s : kFusedSizes)
    if (s >= nc) { nk = s; break; }
  if (nk < 0) { why = "task chain longer than the fused kernels support"; return false; }
  for (int k = nc; k < nk; ++k) {  // padding slots: identity transform, no motion
    D.segc[k].pq[0] = 1.0;
    D.seg_qadr[k] = 0;
    D.seg_dof[k] = -1;
  }
  D.nc = nc;
  *nc_kernel = nk;
  if (ft.ncost != 6) { why = "frame task cost must have 6 entries"; return false; }
  memcpy(D.cost, ft.cost, 6 * sizeof(double));
  D.gain = ft.gain;
  D.lm = ft.lm_damping;
  // posture-like tasks
  for (int t = 0; t < ntask; ++t) {
    if (tasks[t].type != MK_TASK_POSTURE) continue;
    if (tasks[t].ncost != m.nv || !tasks[t].target_q) { why = "posture task needs nv costs and a target"; return false; }
    MkPosture& P = D.posture[D.nposture++];
    P.gain = tasks[t].gain;
    P.lm = tasks[t].lm_damping;
    for (int i = 0; i < m.nv; ++i) {
      const bool scalar = D.dof[i].kind == MK_DOF_SCALAR;
      P.cost[i] = scalar ? tasks[t].cost[i] : 0.0;
      P.target[i] = scalar ? tasks[t].target_q[D.dof[i].qadr] : 0.0;
    }
  }
  // box limits
  auto add_term = [&](int dof, const MkBoundTerm& bt) -> bool {
    MkDof& dd = D.dof[dof];
    if (dd.nterm >= MK_MAX_TERMS) return false;
    dd.term[dd.nterm++] = bt;
    return true;
  };
  for (int l = 0; l < nlimit; ++l) {
    const mk_limit_desc& L = limits[l];
    if (L.type == MK_LIMIT_CONFIGURATION) {
      for (int j = 0; j < m.njnt; ++j) {
        if (m.jnt_type[j] == MK_JNT_FREE || !m.jnt_limited[j]) continue;
        MkBoundTerm bt{MK_TERM_CFG, 0, m.jnt_range[2 * j] + L.min_distance_from_limits,
                       m.jnt_range[2 * j + 1] - L.min_distance_from_limits, L.gain};
        if (!add_term(m.jnt_dofadr[j], bt)) { why = "too many limits on one dof"; return false; }
      }
    } else if (L.type == MK_LIMIT_VELOCITY) {
      for (int k = 0; k < L.n; ++k) {
        const int dof = L.index[k];
        if (dof < 0 || dof >= m.nv || D.dof[dof].kind != MK_DOF_SCALAR) { why = "velocity limit on an unsupported dof"; return false; }
        MkBoundTerm bt{MK_TERM_VEL, 0, L.vmax[k], 0.0, 0.0};
        if (!add_term(dof, bt)) { why = "too many limits on one dof"; return false; }
      }
    }
  }
  // packed per-dof constants (what the kernels read on the lm_damping == 0 path)
  auto pack = [&](int dof, MkDofC& c) -> bool {
    const MkDof& dd = D.dof[dof];
    c.A = c.B = c.W = 0.0;
    c.lower = -INFINITY; c.upper = INFINITY; c.gain = 1.0; c.vmax = INFINITY;
    c.qadr = dd.qadr;
    c.dof = dof;
    for (int t = 0; t < D.nposture; ++t) {
      const MkPosture& P = D.posture[t];
      const double w = P.cost[dof] * P.cost[dof];
      c.A += P.gain * w * P.target[dof];
      c.B += P.gain * w;
      c.W += w;
    }
    bool has_cfg = false, has_vel = false;
    for (int t = 0; t < dd.nterm; ++t) {
      const MkBoundTerm& bt = dd.term[t];
      if (bt.kind == MK_TERM_CFG) {
        if (has_cfg) return false;
        has_cfg = true;
        c.lower = bt.a; c.upper = bt.b; c.gain = bt.gain;
      } else {
        if (has_vel) { c.vmax = std::min(c.vmax, bt.a); continue; }
        has_vel = true;
        c.vmax = bt.a;
      }
    }
    return true;
  };
  D.nd = 0;
  D.any_posture_lm = 0;
  for (int t = 0; t < D.nposture; ++t)
    if (D.posture[t].lm > 0.0) D.any_posture_lm = 1;
  for (int k = 0; k < nk; ++k) {
    if (D.seg_dof[k] < 0) {  // padding slot: decoupled unit entry with infinite bounds -> solution 0
      MkDofC& c = D.sdof[k];
      c.A = c.B = 0.0; c.W = 1.0;
      c.lower = -INFINITY; c.upper = INFINITY; c.gain = 1.0; c.vmax = INFINITY;
      c.qadr = 0; c.dof = -1;
      continue;
    }
    if (!pack(D.seg_dof[k], D.sdof[k])) { why = "several configuration limits on one dof"; return false; }
  }
  for (int i = 0; i < m.nv; ++i) {
    if (D.dof[i].slot >= 0 || D.dof[i].kind != MK_DOF_SCALAR) continue;
    if (!pack(i, D.ddof[D.nd++])) { why = "several configuration limits on one dof"; return false; }
  }
  // dofs whose Hessian entry is damping + mu and nothing else: free-joint dofs (zero Jacobian
  // columns, posture rows zeroed) and scalar dofs outside the chain without posture cost
  D.needs_diag = D.nfree > 0 ? 1 : 0;
  for (int d = 0; d < D.nd; ++d)
    if (!(D.ddof[d].W > 0.0)) D.needs_diag = 1;
  if (!X) return true;

  // ---- extension: relative frame term ----
  if (rel_task >= 0) {
    const mk_task_desc& rt = tasks[rel_task];
    if (rt.frame_type != ft.frame_type || rt.frame_id != ft.frame_id) { why = "relative frame task on another frame than the frame task"; return false; }
    if (rt.ncost != 6) { why = "frame task cost must have 6 entries"; return false; }
    int rb;
    double rpos[3], rquat[4];
    if (frame_body_and_local(m, rt.root_type, rt.root_id, &rb, rpos, rquat) != MK_OK) { why = g_last_error; return false; }
    // climb from the root's body through welded bodies to a body of the chain (or the world)
    double xp[3] = {rpos[0], rpos[1], rpos[2]}, xq[4] = {rquat[0], rquat[1], rquat[2], rquat[3]};
    int r = rb;
    while (r != 0 && anchor[r].slot == -2) {
      if (m.body_jntnum[r] > 0 || m.body_mocapid[r] >= 0) { why = "root frame of the relative task moves off the task chain"; return false; }
      double np[3] = {m.body_pos[3 * r], m.body_pos[3 * r + 1], m.body_pos[3 * r + 2]};
      double nq4[4] = {m.body_quat[4 * r], m.body_quat[4 * r + 1], m.body_quat[4 * r + 2], m.body_quat[4 * r + 3]};
      compose(np, nq4, xp, xq);
      memcpy(xp, np, sizeof(np));
      memcpy(xq, nq4, sizeof(nq4));
      r = m.body_parentid[r];
    }
    double bp[3] = {0, 0, 0}, bq[4] = {1, 0, 0, 0};
    int slot = -1;
    if (r != 0) {
      slot = anchor[r].slot;
      memcpy(bp, anchor[r].pos, sizeof(bp));
      memcpy(bq, anchor[r].quat, sizeof(bq));
    }
    compose(bp, bq, xp, xq);
    X->rel_slot = slot;
    memcpy(X->rel_xp, bp, sizeof(bp));
    memcpy(X->rel_xq, bq, sizeof(bq));
    memcpy(X->rel_cost, rt.cost, 6 * sizeof(double));
    X->rel_gain = rt.gain;
    X->rel_lm = rt.lm_damping;
  }

  // ---- extension: collision screen ----
  std::vector<int> body_slot(m.nbody, -1);
  auto coll_body = [&](int gb) -> int {
    if (body_slot[gb] >= 0) return body_slot[gb];
    if (X->ncbody >= MK_MAX_CBODY) { why = "too many bodies with collision geoms"; return -1; }
    MkCollBody CB;
    memset(&CB, 0, sizeof(CB));
    // bodies from gb upwards until a body of the chain, a free-joint body or the world
    std::vector<int> up;
    int b = gb;
    while (b != 0 && anchor[b].slot == -2 && !(m.body_jntnum[b] == 1 && m.jnt_type[m.body_jntadr[b]] == MK_JNT_FREE)) {
      up.insert(up.begin(), b);
      b = m.body_parentid[b];
    }
    double pp[3] = {0, 0, 0}, pq[4] = {1, 0, 0, 0};
    if (b != 0 && anchor[b].slot >= 0) {
      // carried by the frame behind chain joint `slot` (the last one: the body the task frame is on)
      CB.kind = MK_CB_TIP;
      CB.slot = anchor[b].slot;
      memcpy(pp, anchor[b].pos, sizeof(pp));
      memcpy(pq, anchor[b].quat, sizeof(pq));
    } else if (b != 0 && anchor[b].slot == -1) {
      // on the path in front of the first joint: fixed in the world
      CB.kind = MK_CB_WORLD;
      memcpy(pp, anchor[b].pos, sizeof(pp));
      memcpy(pq, anchor[b].quat, sizeof(pq));
    } else if (b == 0) {
      CB.kind = MK_CB_WORLD;
    } else {
      CB.kind = MK_CB_FREE;
      CB.free_qadr = m.jnt_qposadr[m.body_jntadr[b]];
      CB.free_dofadr = m.jnt_dofadr[m.body_jntadr[b]];
    }
    for (int c : up) {
      if (m.body_mocapid[c] >= 0) { why = "collision geom on a mocap body"; return -1; }
      compose(pp, pq, &m.body_pos[3 * c], &m.body_quat[4 * c]);
      for (int j = m.body_jntadr[c]; j < m.body_jntadr[c] + m.body_jntnum[c]; ++j) {
        const int jt = m.jnt_type[j];
        if (jt != MK_JNT_HINGE && jt != MK_JNT_SLIDE) { why = "free/ball joint below a collision body"; return -1; }
        if (CB.kind == MK_CB_WORLD) { why = "collision geom on a separate articulated branch"; return -1; }
        if (CB.nseg >= MK_MAX_CSEG) { why = "too many joints between the chain and a collision body"; return -1; }
        MkCollSeg& S = CB.seg[CB.nseg++];
        if (!canonical_joint(m, j, pp, pq, S.pq, S.pp, why)) return -1;
        S.hinge = (jt == MK_JNT_HINGE) ? 1.0 : 0.0;
        S.slide = 1.0 - S.hinge;
        S.qadr = m.jnt_qposadr[j];
        S.dof = m.jnt_dofadr[j];
        // (coarse screen: the fixed offset in front of the joint, and the travel of a slide joint)
        if (CB.reach >= 0.0) {
          CB.reach += std::sqrt(S.pp[0] * S.pp[0] + S.pp[1] * S.pp[1] + S.pp[2] * S.pp[2]);
          if (jt == MK_JNT_SLIDE) {
            if (m.jnt_limited[j]) CB.reach += std::max(std::fabs(m.jnt_range[2 * j]), std::fabs(m.jnt_range[2 * j + 1]));
            else CB.reach = -1.0;
          }
        }
      }
    }
    if (CB.kind == MK_CB_FREE && CB.nseg > 0) { why = "joints below a free body"; return -1; }
    memcpy(CB.tp, pp, sizeof(pp));
    memcpy(CB.tq, pq, sizeof(pq));
    if (CB.kind == MK_CB_WORLD) CB.reach = 0.0;   // (tp is its world position: the geoms' centres are taken as they are)
    else if (CB.reach >= 0.0) CB.reach += std::sqrt(pp[0] * pp[0] + pp[1] * pp[1] + pp[2] * pp[2]);
    X->cbody[X->ncbody] = CB;
    return body_slot[gb] = X->ncbody++;
  };
  std::vector<int> geom_slot(m.ngeom, -1);
  int n_orient = 0;
  auto coll_geom = [&](int g) -> int {
    if (geom_slot[g] >= 0) return geom_slot[g];
    if (X->ncgeom >= MK_MAX_CGEOM) { why = "too many collision geoms"; return -1; }
    const int bs = coll_body(m.geom_bodyid[g]);
    if (bs < 0) return -1;
    MkCollGeom& G = X->cgeom[X->ncgeom];
    G.body = bs;
    G.type = m.geom_type[g];
    memcpy(G.lp, &m.geom_pos[3 * g], 24);
    memcpy(G.lq, &m.geom_quat[4 * g], 32);
    memcpy(G.size, &m.geom_size[3 * g], 24);
    const double* sz = G.size;
    G.bc[0] = G.bc[1] = G.bc[2] = 0.0;
    G.oslot = -1;
    if (G.type == 0 || G.type == 6) G.oslot = n_orient++;
    switch (G.type) {
      case 2: G.rbound = sz[0]; break;
      case 3: G.rbound = sz[0] + sz[1]; break;
      case 4: G.rbound = std::max(sz[0], std::max(sz[1], sz[2])); break;
      case 5: G.rbound = std::sqrt(sz[0] * sz[0] + sz[1] * sz[1]); break;
      case 6: G.rbound = std::sqrt(sz[0] * sz[0] + sz[1] * sz[1] + sz[2] * sz[2]); break;
      case 7: {
        const int k = m.geom_dataid[g];
        memcpy(G.bc, &m.mesh_center[3 * k], 24);
        G.rbound = m.mesh_radius[k];
        break;
      }
      default: G.rbound = 0.0;
    }
    return geom_slot[g] = X->ncgeom++;
  };
  for (int l = 0; l < nlimit; ++l) {
    const mk_limit_desc& L = limits[l];
    if (L.type != MK_LIMIT_COLLISION) continue;
    for (int k = 0; k < L.n; ++k) {
      int g1 = L.index[2 * k], g2 = L.index[2 * k + 1];
      if (g1 < 0 || g1 >= m.ngeom || g2 < 0 || g2 >= m.ngeom) { why = "geom id out of range"; return false; }
      if (m.geom_type[g1] > m.geom_type[g2]) std::swap(g1, g2);
      // pairs without a closed form (ellipsoid, cylinder, mesh) are screened by their bounding
      // spheres here and, when in range, solved by the general kernel
      if (!pair_supported(m, g1, g2, nullptr, why)) return false;
      if (X->ncpair >= MK_MAX_CPAIR) { why = "too many collision pairs"; return false; }
      const int s1 = coll_geom(g1), s2 = coll_geom(g2);
      if (s1 < 0 || s2 < 0) return false;
      MkCollPair& P = X->cpair[X->ncpair++];
      P.g1 = s1;
      P.g2 = s2;
      P.d_detect = L.collision_detection_distance;
      P.d_min = L.minimum_distance_from_collisions;
      P.gain = L.gain;
      P.relax = L.bound_relaxation;
    }
  }
  X->norient = n_orient;
  // geoms in the order of their bodies, so that a body names its geoms as a range
  std::vector<int> order(X->ncgeom), where(X->ncgeom);
  for (int g = 0; g < X->ncgeom; ++g) order[g] = g;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return X->cgeom[a].body < X->cgeom[b].body; });
  MkCollGeom sorted[MK_MAX_CGEOM];
  for (int g = 0; g < X->ncgeom; ++g) { sorted[g] = X->cgeom[order[g]]; where[order[g]] = g; }
  for (int b = 0; b < X->ncbody; ++b) X->cbody[b].g0 = X->cbody[b].ng = 0;
  for (int g = 0; g < X->ncgeom; ++g) {
    X->cgeom[g] = sorted[g];
    MkCollBody& CB = X->cbody[sorted[g].body];
    if (CB.ng++ == 0) CB.g0 = g;
  }
  for (int k = 0; k < X->ncpair; ++k) {
    X->cpair[k].g1 = where[X->cpair[k].g1];
    X->cpair[k].g2 = where[X->cpair[k].g2];
  }
  // coarse bounds of the screen (see MkCollGeom)
  for (int g = 0; g < X->ncgeom; ++g) {
    MkCollGeom& G = X->cgeom[g];
    const MkCollBody& CB = X->cbody[G.body];
    const double off = std::sqrt(G.lp[0] * G.lp[0] + G.lp[1] * G.lp[1] + G.lp[2] * G.lp[2]) +
                       std::sqrt(G.bc[0] * G.bc[0] + G.bc[1] * G.bc[1] + G.bc[2] * G.bc[2]);
    G.wc[0] = G.wc[1] = G.wc[2] = 0.0;
    G.wn[0] = G.wn[1] = 0.0;
    G.wn[2] = 1.0;
    if (CB.kind == MK_CB_WORLD) {
      double wp[3], wq[4];
      memcpy(wp, CB.tp, sizeof(wp));
      memcpy(wq, CB.tq, sizeof(wq));
      compose(wp, wq, G.lp, G.lq);
      double c[3];
      const double ez[3] = {0.0, 0.0, 1.0};
      hq_rotate(wq, G.bc, c);
      for (int i = 0; i < 3; ++i) G.wc[i] = wp[i] + c[i];
      hq_rotate(wq, ez, G.wn);
      G.reach = G.rbound;
    } else if (G.type == 0 || CB.reach < 0.0 || (CB.kind == MK_CB_TIP && CB.slot < D.nc - 1)) {
      G.reach = -1.0;
    } else {
      G.reach = CB.reach + off + G.rbound;
    }
  }
  return true;
}

// Describes the problem for the dense (wave-per-problem) kernels. Device arrays go to
// `allocs`. Returns false with `why` when a feature is outside what they implement.
bool build_dense(const HostModel& m, const mk_task_desc* tasks, int ntask, const mk_limit_desc* limits, int nlimit,
                 MkDenseProblem& D, std::vector<int>& task_k, std::vector<void*>& allocs, std::string& why) {
  memset(&D, 0, sizeof(D));
  if (m.nv > MK_DENSE_MAX_NV) { why = "more than 64 dofs"; return false; }
  if (ntask > MK_DENSE_MAX_TASKS || nlimit > MK_DENSE_MAX_LIMITS) { why = "too many tasks or limits"; return false; }
  D.ntask = ntask;
  D.nlimit = nlimit;
  int off = 0;
  for (int t = 0; t < ntask; ++t) {
    const mk_task_desc& T = tasks[t];
    MkDenseTask& O = D.task[t];
    O.type = T.type;
    O.gain = T.gain;
    O.lm = T.lm_damping;
    O.target_off = off;
    if (T.type == MK_TASK_FRAME || T.type == MK_TASK_RELATIVE_FRAME) {
      if (T.ncost != 6) { why = "frame task cost must have 6 entries"; return false; }
      memcpy(O.cost, T.cost, 6 * sizeof(double));
      if (frame_body_and_local(m, T.frame_type, T.frame_id, &O.body, O.lpos, O.lquat) != MK_OK) { why = g_last_error; return false; }
      if (T.type == MK_TASK_RELATIVE_FRAME &&
          frame_body_and_local(m, T.root_type, T.root_id, &O.root_body, O.rpos, O.rquat) != MK_OK) { why = g_last_error; return false; }
      O.k = 6;
      off += 7;
    } else if (T.type == MK_TASK_POSTURE) {
      if (T.ncost != m.nv || !T.target_q) { why = "posture task needs nv costs and a target"; return false; }
      std::vector<double> pc(T.cost, T.cost + m.nv), pt(T.target_q, T.target_q + m.nq);
      if (upload(allocs, pc, &O.pcost) != MK_OK || upload(allocs, pt, &O.ptarget) != MK_OK) { why = g_last_error; return false; }
      O.k = m.nv;
    } else {  // com: subtree of body 1 (mink/tasks/com_task.py:82,96)
      if (T.ncost != 3) { why = "com task cost must have 3 entries"; return false; }
      memcpy(O.cost, T.cost, 3 * sizeof(double));
      std::vector<int32_t> bodies;
      for (int b = 1; b < m.nbody; ++b) {
        int x = b;
        while (x > 1) x = m.body_parentid[x];
        if (x == 1) bodies.push_back(b);
      }
      O.nbodies = (int)bodies.size();
      if (upload(allocs, bodies, &O.bodies) != MK_OK) { why = g_last_error; return false; }
      O.k = 3;
      off += 3;
    }
    task_k.push_back(O.k);
  }
  D.target_stride = off;
  int rows = 0;
  for (int l = 0; l < nlimit; ++l) {
    const mk_limit_desc& L = limits[l];
    MkDenseLimit& O = D.limit[l];
    O.type = L.type;
    O.row0 = rows;
    if (L.type == MK_LIMIT_CONFIGURATION) {
      std::vector<int32_t> idx;
      for (int j = 0; j < m.njnt; ++j) {
        if (m.jnt_type[j] == MK_JNT_FREE || !m.jnt_limited[j]) continue;
        if (m.jnt_type[j] == MK_JNT_BALL) { why = "configuration limit on a ball joint"; return false; }
        idx.push_back(j);
      }
      O.n = (int)idx.size();
      O.gain = L.gain;
      O.min_dist = L.min_distance_from_limits;
      if (upload(allocs, idx, &O.index) != MK_OK) { why = g_last_error; return false; }
      rows += 2 * O.n;
    } else if (L.type == MK_LIMIT_VELOCITY) {
      std::vector<int32_t> idx(L.index, L.index + L.n);
      std::vector<double> vm(L.vmax, L.vmax + L.n);
      for (int d : idx)
        if (d < 0 || d >= m.nv) { why = "velocity limit dof out of range"; return false; }
      O.n = L.n;
      if (upload(allocs, idx, &O.index) != MK_OK || upload(allocs, vm, &O.vmax) != MK_OK) { why = g_last_error; return false; }
      rows += 2 * O.n;
    } else {
      std::vector<int32_t> idx(L.index, L.index + 2 * L.n);
      for (int k = 0; k < L.n; ++k) {
        const int g1 = idx[2 * k], g2 = idx[2 * k + 1];
        if (g1 < 0 || g1 >= m.ngeom || g2 < 0 || g2 >= m.ngeom) { why = "geom id out of range"; return false; }
        if (m.body_mocapid[m.geom_bodyid[g1]] >= 0 || m.body_mocapid[m.geom_bodyid[g2]] >= 0) { why = "collision geom on a mocap body, whose pose is not part of q"; return false; }
        std::string why_pair;
        if (!pair_supported(m, g1, g2, nullptr, why_pair)) { why = why_pair; return false; }
      }
      O.n = L.n;
      O.gain = L.gain;
      O.d_min = L.minimum_distance_from_collisions;
      O.d_detect = L.collision_detection_distance;
      O.relax = L.bound_relaxation;
      if (upload(allocs, idx, &O.index) != MK_OK) { why = g_last_error; return false; }
      rows += O.n;
      D.n_general += O.n;
    }
  }
  D.n_rows = rows;
  // flattened forms of the limits for the kernels: per-dof bounds, per-pair records
  {
    int ncfg = 0;
    for (int l = 0; l < nlimit; ++l) ncfg += limits[l].type == MK_LIMIT_CONFIGURATION;
    if (ncfg <= 1) {
      std::vector<MkDenseDofBound> bd(m.nv);
      for (int i = 0; i < m.nv; ++i) bd[i] = MkDenseDofBound{-INFINITY, INFINITY, 1.0, INFINITY, 0, 0};
      for (int l = 0; l < nlimit; ++l) {
        const mk_limit_desc& L = limits[l];
        if (L.type == MK_LIMIT_CONFIGURATION) {
          for (int j = 0; j < m.njnt; ++j) {
            if (m.jnt_type[j] == MK_JNT_FREE || !m.jnt_limited[j]) continue;
            MkDenseDofBound& b = bd[m.jnt_dofadr[j]];
            b.lower = m.jnt_range[2 * j] + L.min_distance_from_limits;
            b.upper = m.jnt_range[2 * j + 1] - L.min_distance_from_limits;
            b.gain = L.gain;
            b.qadr = m.jnt_qposadr[j];
            b.has_cfg = 1;
          }
        } else if (L.type == MK_LIMIT_VELOCITY) {
          for (int k = 0; k < L.n; ++k) bd[L.index[k]].vmax = std::min(bd[L.index[k]].vmax, L.vmax[k]);
        }
      }
      if (upload(allocs, bd, &D.dof_bound) != MK_OK) { why = g_last_error; return false; }
    }
    std::vector<MkDensePair> pr;
    std::vector<uint64_t> mask(m.nbody, 0);
    for (int b = 1; b < m.nbody; ++b) {
      mask[b] = mask[m.body_parentid[b]];
      for (int j = m.body_jntadr[b]; j >= 0 && j < m.body_jntadr[b] + m.body_jntnum[b]; ++j) {
        const int t = m.jnt_type[j], w = t == MK_JNT_FREE ? 6 : t == MK_JNT_BALL ? 3 : 1;
        for (int k = 0; k < w; ++k) mask[b] |= 1ull << (m.jnt_dofadr[j] + k);
      }
    }
    for (int l = 0; l < nlimit; ++l) {
      const mk_limit_desc& L = limits[l];
      if (L.type != MK_LIMIT_COLLISION) continue;
      for (int k = 0; k < L.n; ++k) {
        int g1 = L.index[2 * k], g2 = L.index[2 * k + 1];
        MkDensePair P;
        memset(&P, 0, sizeof(P));
        P.swap = m.geom_type[g1] > m.geom_type[g2];
        if (P.swap) std::swap(g1, g2);
        P.t1 = m.geom_type[g1]; P.t2 = m.geom_type[g2];
        P.b1 = m.geom_bodyid[g1]; P.b2 = m.geom_bodyid[g2];
        P.row = D.limit[l].row0 + k;
        P.mask1 = mask[P.b1]; P.mask2 = mask[P.b2];
        memcpy(P.size1, &m.geom_size[3 * g1], 24); memcpy(P.size2, &m.geom_size[3 * g2], 24);
        memcpy(P.pos1, &m.geom_pos[3 * g1], 24); memcpy(P.pos2, &m.geom_pos[3 * g2], 24);
        memcpy(P.quat1, &m.geom_quat[4 * g1], 32); memcpy(P.quat2, &m.geom_quat[4 * g2], 32);
        P.gain = L.gain; P.d_min = L.minimum_distance_from_collisions;
        P.d_detect = L.collision_detection_distance; P.relax = L.bound_relaxation;
        bool cvx = false;
        std::string ignore;
        pair_supported(m, g1, g2, &cvx, ignore);
        P.convex = cvx ? 1 : 0;
        auto shape_info = [&](int g, double* bc, double* rb, int32_t* vadr, int32_t* vnum, double* mean, double* ext) {
          const int t = m.geom_type[g];
          const double* sz = &m.geom_size[3 * g];
          bc[0] = bc[1] = bc[2] = 0.0;
          mean[0] = mean[1] = mean[2] = 0.0;
          *vadr = 0; *vnum = 0;
          *ext = std::max(std::fabs(sz[0]), std::max(std::fabs(sz[1]), std::fabs(sz[2])));
          if (t == 2) *rb = sz[0];
          else if (t == 3) *rb = sz[0] + sz[1];
          else if (t == 4) *rb = *ext;
          else if (t == 5) *rb = std::sqrt(sz[0] * sz[0] + sz[1] * sz[1]);
          else if (t == 6) *rb = std::sqrt(sz[0] * sz[0] + sz[1] * sz[1] + sz[2] * sz[2]);
          else if (t == 7) {
            const int k = m.geom_dataid[g];
            *vadr = m.mesh_vertadr[k]; *vnum = m.mesh_vertnum[k];
            memcpy(bc, &m.mesh_center[3 * k], 24);
            *rb = m.mesh_radius[k];
            double e = 0.0;
            for (int i = 0; i < *vnum; ++i)
              for (int a = 0; a < 3; ++a) {
                const double x = m.mesh_vert[3 * (size_t)(*vadr + i) + a];
                mean[a] += x;
                e = std::max(e, std::fabs(x));
              }
            for (int a = 0; a < 3; ++a) mean[a] /= std::max(1, (int)*vnum);
            *ext = e;
          } else *rb = 0.0;
        };
        shape_info(g1, P.bc1, &P.rb1, &P.v1adr, &P.v1num, P.mean1, &P.ext1);
        shape_info(g2, P.bc2, &P.rb2, &P.v2adr, &P.v2num, P.mean2, &P.ext2);
        pr.push_back(P);
      }
    }
    D.npairs = (int)pr.size();
    if (!pr.empty() && upload(allocs, pr, &D.pairs) != MK_OK) { why = g_last_error; return false; }
  }
  int npath = 0;   // total length of the root -> body paths
  for (int b = 0; b < m.nbody; ++b)
    for (int x = b; x != 0; x = m.body_parentid[x]) ++npath;
  if (mk_dense_lds_bytes(m.nv, D.n_general, m.nq, D.target_stride, m.nbody, m.njnt, npath) == 0) { why = "problem does not fit the LDS budget of the dense kernels"; return false; }
  return true;
}

int count_rows(const HostModel& m, const mk_limit_desc* limits, int nlimit) {
  int rows = 0;
  for (int l = 0; l < nlimit; ++l) {
    const mk_limit_desc& L = limits[l];
    if (L.type == MK_LIMIT_CONFIGURATION) {
      for (int j = 0; j < m.njnt; ++j)
        if (m.jnt_type[j] != MK_JNT_FREE && m.jnt_limited[j]) rows += 2 * (m.jnt_type[j] == MK_JNT_BALL ? 3 : 1);
    } else if (L.type == MK_LIMIT_VELOCITY) {
      rows += 2 * L.n;
    } else if (L.type == MK_LIMIT_COLLISION) {
      rows += L.n;
    }
  }
  return rows;
}

int host_model_from_desc(const mk_model_desc* d, HostModel& h) {
  h.nq = d->nq; h.nv = d->nv; h.nbody = d->nbody; h.njnt = d->njnt; h.ngeom = d->ngeom; h.nsite = d->nsite;
  h.body_parentid = copy_in(d->body_parentid, h.nbody);
  h.body_jntadr = copy_in(d->body_jntadr, h.nbody);
  h.body_jntnum = copy_in(d->body_jntnum, h.nbody);
  h.body_mocapid = d->body_mocapid ? copy_in(d->body_mocapid, h.nbody) : std::vector<int32_t>(h.nbody, -1);
  h.body_pos = copy_in(d->body_pos, 3 * h.nbody);
  h.body_quat = copy_in(d->body_quat, 4 * h.nbody);
  h.body_mass = copy_in(d->body_mass, h.nbody);
  h.body_ipos = copy_in(d->body_ipos, 3 * h.nbody);
  h.jnt_type = copy_in(d->jnt_type, h.njnt);
  h.jnt_qposadr = copy_in(d->jnt_qposadr, h.njnt);
  h.jnt_dofadr = copy_in(d->jnt_dofadr, h.njnt);
  h.jnt_bodyid = copy_in(d->jnt_bodyid, h.njnt);
  h.jnt_limited = copy_in(d->jnt_limited, h.njnt);
  h.jnt_pos = copy_in(d->jnt_pos, 3 * h.njnt);
  h.jnt_axis = copy_in(d->jnt_axis, 3 * h.njnt);
  h.jnt_range = copy_in(d->jnt_range, 2 * h.njnt);
  h.qpos0 = copy_in(d->qpos0, h.nq);
  h.dof_parentid = copy_in(d->dof_parentid, h.nv);
  h.dof_jntid = copy_in(d->dof_jntid, h.nv);
  h.site_bodyid = copy_in(d->site_bodyid, h.nsite);
  h.site_pos = copy_in(d->site_pos, 3 * h.nsite);
  h.site_quat = copy_in(d->site_quat, 4 * h.nsite);
  h.geom_bodyid = copy_in(d->geom_bodyid, h.ngeom);
  h.geom_type = copy_in(d->geom_type, h.ngeom);
  h.geom_size = copy_in(d->geom_size, 3 * h.ngeom);
  h.geom_pos = copy_in(d->geom_pos, 3 * h.ngeom);
  h.geom_quat = copy_in(d->geom_quat, 4 * h.ngeom);
  h.nmesh = d->nmesh > 0 ? d->nmesh : 0;
  h.nmeshvert = d->nmeshvert > 0 ? d->nmeshvert : 0;
  h.geom_dataid = d->geom_dataid ? copy_in(d->geom_dataid, h.ngeom) : std::vector<int32_t>(h.ngeom, -1);
  h.mesh_vertadr = copy_in(d->mesh_vertadr, h.nmesh);
  h.mesh_vertnum = copy_in(d->mesh_vertnum, h.nmesh);
  h.mesh_vert = copy_in(d->mesh_vert, 3 * (size_t)h.nmeshvert);
  h.mesh_center = copy_in(d->mesh_center, 3 * (size_t)h.nmesh);
  h.mesh_radius = copy_in(d->mesh_radius, h.nmesh);
  for (int g = 0; g < h.ngeom; ++g)
    if (h.geom_dataid[g] >= h.nmesh) return fail(MK_ERR_INVALID, "geom %d: mesh id out of range", g);
  for (int k = 0; k < h.nmesh; ++k)
    if (h.mesh_vertadr[k] < 0 || h.mesh_vertnum[k] < 0 || h.mesh_vertadr[k] + h.mesh_vertnum[k] > h.nmeshvert)
      return fail(MK_ERR_INVALID, "mesh %d: vertex range", k);
  // consistency checks the kernels rely on
  int nq = 0, nv = 0;
  for (int j = 0; j < h.njnt; ++j) {
    const int t = h.jnt_type[j];
    if (t < 0 || t > 3 || h.jnt_qposadr[j] != nq || h.jnt_dofadr[j] != nv || h.jnt_bodyid[j] <= 0 ||
        h.jnt_bodyid[j] >= h.nbody) {
      return fail(MK_ERR_INVALID, "joint %d: inconsistent type/address/body", j);
    }
    nq += (t == MK_JNT_FREE ? 7 : t == MK_JNT_BALL ? 4 : 1);
    nv += (t == MK_JNT_FREE ? 6 : t == MK_JNT_BALL ? 3 : 1);
  }
  if (nq != h.nq || nv != h.nv) { return fail(MK_ERR_INVALID, "nq/nv do not match the joint table"); }
  for (int b = 1; b < h.nbody; ++b) {
    if (h.body_parentid[b] < 0 || h.body_parentid[b] >= b) { return fail(MK_ERR_INVALID, "body %d: parent must precede it", b); }
    const int ja = h.body_jntadr[b], jn = h.body_jntnum[b];
    if (jn < 0 || (jn > 0 && (ja < 0 || ja + jn > h.njnt))) { return fail(MK_ERR_INVALID, "body %d: joint range", b); }
    for (int j = ja; j < ja + jn; ++j)
      if (h.jnt_bodyid[j] != b || (h.jnt_type[j] == MK_JNT_FREE && jn != 1)) { return fail(MK_ERR_INVALID, "body %d: joint table", b); }
  }
  return MK_OK;
}

}  // namespace

extern "C" {

const char* mk_last_error(void) { return g_last_error.c_str(); }

int mk_model_create(const mk_model_desc* d, mk_model** out) {
  if (!d || !out) return fail(MK_ERR_INVALID, "null argument");
  if (d->nq <= 0 || d->nv <= 0 || d->nbody <= 0 || d->njnt <= 0) return fail(MK_ERR_INVALID, "empty model");
  mk_model* m = new mk_model();
  HostModel& h = m->h;
  {
    int rc0 = host_model_from_desc(d, h);
    if (rc0 != MK_OK) { delete m; return rc0; }
  }
  if (hipGetDevice(&m->device) != hipSuccess) { delete m; return fail(MK_ERR_HIP, "hipGetDevice failed"); }
  // paths root -> body, for the generic per-frame kernels
  std::vector<int32_t> path_adr(h.nbody), path_len(h.nbody), path_data;
  for (int b = 0; b < h.nbody; ++b) {
    std::vector<int> p;
    for (int x = b; x != 0; x = h.body_parentid[x]) p.insert(p.begin(), x);
    path_adr[b] = (int32_t)path_data.size();
    path_len[b] = (int32_t)p.size();
    path_data.insert(path_data.end(), p.begin(), p.end());
  }
  std::vector<int32_t> dof_qadr(h.nv), dof_scalar(h.nv), dof_ball(h.nv, -1);
  for (int j = 0; j < h.njnt; ++j) {
    const int t = h.jnt_type[j], w = t == MK_JNT_FREE ? 6 : t == MK_JNT_BALL ? 3 : 1;
    for (int k = 0; k < w; ++k) {
      dof_qadr[h.jnt_dofadr[j] + k] = h.jnt_qposadr[j] + k;
      dof_scalar[h.jnt_dofadr[j] + k] = (t == MK_JNT_HINGE || t == MK_JNT_SLIDE) ? 1 : 0;
      if (t == MK_JNT_BALL) dof_ball[h.jnt_dofadr[j] + k] = k;
    }
  }
  MkDevModel& dv = m->dev;
  memset(&dv, 0, sizeof(dv));
  dv.nq = h.nq; dv.nv = h.nv; dv.nbody = h.nbody; dv.njnt = h.njnt; dv.ngeom = h.ngeom;
  int rc = MK_OK;
#define UP(field) if (rc == MK_OK) rc = upload(m->allocs, h.field, &dv.field)
  UP(body_parentid); UP(body_jntadr); UP(body_jntnum); UP(body_mocapid); UP(body_pos); UP(body_quat);
  UP(jnt_type); UP(jnt_qposadr); UP(jnt_dofadr); UP(jnt_bodyid); UP(jnt_limited); UP(jnt_pos); UP(jnt_axis);
  UP(jnt_range); UP(qpos0); UP(body_mass); UP(body_ipos);
  UP(geom_bodyid); UP(geom_type); UP(geom_size); UP(geom_pos); UP(geom_quat); UP(mesh_vert);
#undef UP
  if (rc == MK_OK) rc = upload(m->allocs, dof_qadr, &dv.dof_qadr);
  if (rc == MK_OK) rc = upload(m->allocs, dof_scalar, &dv.dof_scalar);
  if (rc == MK_OK) rc = upload(m->allocs, dof_ball, &dv.dof_ball);
  if (rc == MK_OK) rc = upload(m->allocs, path_adr, &dv.path_adr);
  if (rc == MK_OK) rc = upload(m->allocs, path_len, &dv.path_len);
  if (rc == MK_OK) rc = upload(m->allocs, path_data, &dv.path_data);
  dv.npath = (int32_t)path_data.size();
  if (h.nv <= 64) {
    std::vector<uint64_t> mask(h.nbody, 0);   // bodies are ordered parents first
    for (int b = 1; b < h.nbody; ++b) {
      mask[b] = mask[h.body_parentid[b]];
      for (int j = h.body_jntadr[b]; j >= 0 && j < h.body_jntadr[b] + h.body_jntnum[b]; ++j) {
        const int t = h.jnt_type[j], w = t == MK_JNT_FREE ? 6 : t == MK_JNT_BALL ? 3 : 1;
        for (int k = 0; k < w; ++k) mask[b] |= 1ull << (h.jnt_dofadr[j] + k);
      }
    }
    if (rc == MK_OK) rc = upload(m->allocs, mask, &dv.body_dofmask);
  }
  if (rc != MK_OK) { mk_model_destroy(m); return rc; }
  *out = m;
  return MK_OK;
}

void mk_model_destroy(mk_model* m) {
  if (!m) return;
  for (void* p : m->allocs) (void)hipFree(p);
  delete m;
}

int mk_problem_create(const mk_model* model, const mk_task_desc* tasks, int32_t ntask, const mk_limit_desc* limits,
                      int32_t nlimit, mk_problem** out) {
  if (!model || !out || ntask < 0 || nlimit < 0 || (ntask > 0 && !tasks) || (nlimit > 0 && !limits))
    return fail(MK_ERR_INVALID, "null or negative argument");
  const HostModel& m = model->h;
  int stride = 0;
  for (int t = 0; t < ntask; ++t) {
    const mk_task_desc& T = tasks[t];
    if (!(T.gain >= 0.0 && T.gain <= 1.0)) return fail(MK_ERR_INVALID, "task %d: gain must be in [0, 1]", t);
    if (!(T.lm_damping >= 0.0)) return fail(MK_ERR_INVALID, "task %d: lm_damping must be >= 0", t);
    if (!T.cost) return fail(MK_ERR_INVALID, "task %d: cost is null", t);
    for (int i = 0; i < T.ncost; ++i)
      if (!(T.cost[i] >= 0.0)) return fail(MK_ERR_INVALID, "task %d: cost must be >= 0", t);
    if (T.type == MK_TASK_FRAME || T.type == MK_TASK_RELATIVE_FRAME) stride += 7;
    else if (T.type == MK_TASK_COM) stride += 3;
    else if (T.type != MK_TASK_POSTURE) return fail(MK_ERR_INVALID, "task %d: unknown type %d", t, T.type);
  }
  for (int l = 0; l < nlimit; ++l) {
    const mk_limit_desc& L = limits[l];
    if (L.type == MK_LIMIT_CONFIGURATION) {
      if (!(L.gain > 0.0 && L.gain <= 1.0)) return fail(MK_ERR_INVALID, "limit %d: gain must be in (0, 1]", l);
    } else if (L.type == MK_LIMIT_VELOCITY) {
      if (L.n < 0 || (L.n > 0 && (!L.index || !L.vmax))) return fail(MK_ERR_INVALID, "limit %d: velocity arrays", l);
    } else if (L.type == MK_LIMIT_COLLISION) {
      if (L.n < 0 || (L.n > 0 && !L.index)) return fail(MK_ERR_INVALID, "limit %d: geom pairs", l);
    } else {
      return fail(MK_ERR_INVALID, "limit %d: unknown type %d", l, L.type);
    }
  }
  mk_problem* p = new mk_problem();
  p->model = model;
  for (int t = 0; t < ntask; ++t) p->task_type.push_back(tasks[t].type);
  p->d_fused = nullptr;
  p->d_dense = nullptr;
  p->info.nq = m.nq;
  p->info.nv = m.nv;
  p->info.target_stride = stride;
  p->info.n_rows = count_rows(m, limits, nlimit);
  std::string why;
  p->fused = build_fused(m, tasks, ntask, limits, nlimit, p->h_fused, &p->nc_kernel, why);
  p->staged = false;
  p->d_ext = nullptr;
  if (!p->fused) {
    // the same structured form plus a relative frame term and / or a collision screen?
    std::string why2;
    bool needs = false;
    int nck = 0;
    // (option "staged" = 0: tests and timing of the general kernel alone)
    if (mk_opt(MK_OPT_STAGED) != 0 &&
        build_fused(m, tasks, ntask, limits, nlimit, p->h_fused, &nck, why2, &p->h_ext, &needs) && needs) {
      p->staged = true;
      p->nc_kernel = nck;
    }
  }
  p->info.fused = p->fused ? 1 : 0;
  p->info.staged = p->staged ? 1 : 0;
  p->info.n_chain = (p->fused || p->staged) ? p->h_fused.nc : 0;
  if (p->fused || p->staged) {
    void* d = nullptr;
    if (hipMalloc(&d, sizeof(MkFusedDesc)) != hipSuccess ||
        hipMemcpy(d, &p->h_fused, sizeof(MkFusedDesc), hipMemcpyHostToDevice) != hipSuccess) {
      if (d) (void)hipFree(d);
      delete p;
      return fail(MK_ERR_HIP, "could not upload the problem descriptor");
    }
    p->d_fused = static_cast<MkFusedDesc*>(d);
    p->allocs.push_back(d);
    if (p->staged) {
      void* dx = nullptr;
      if (hipMalloc(&dx, sizeof(MkFusedExt)) != hipSuccess ||
          hipMemcpy(dx, &p->h_ext, sizeof(MkFusedExt), hipMemcpyHostToDevice) != hipSuccess) {
        if (dx) (void)hipFree(dx);
        mk_problem_destroy(p);
        return fail(MK_ERR_HIP, "could not upload the problem descriptor");
      }
      p->d_ext = static_cast<MkFusedExt*>(dx);
      p->allocs.push_back(dx);
    }
  }
  p->dense = build_dense(m, tasks, ntask, limits, nlimit, p->h_dense, p->task_k, p->allocs, p->dense_why);
  if (p->dense) {
    void* d = nullptr;
    if (hipMalloc(&d, sizeof(MkDenseProblem)) != hipSuccess ||
        hipMemcpy(d, &p->h_dense, sizeof(MkDenseProblem), hipMemcpyHostToDevice) != hipSuccess) {
      if (d) (void)hipFree(d);
      mk_problem_destroy(p);
      return fail(MK_ERR_HIP, "could not upload the problem descriptor");
    }
    p->d_dense = static_cast<MkDenseProblem*>(d);
    p->allocs.push_back(d);
  }
  if (p->staged && !p->dense) p->staged = false, p->info.staged = 0;   // the second stage is the general kernel
  // (as in the rows kernel's launcher in mk_fused.hip: a lane holds rows of up to MK_ROWS_MAX_NV dofs, and the form
  // for MK_LANE_ROWS rows keeps H and the rows of 64 problems in LDS)
  p->info.staged_rows = (p->staged && m.nv <= MK_ROWS_MAX_NV &&
                         (size_t)64 * (p->nc_kernel * (p->nc_kernel + 1) / 2 + MK_LANE_ROWS * m.nv) * sizeof(double) +
                                 128 * sizeof(int32_t) <= (size_t)160 * 1024) ? 1 : 0;
  if (!p->fused && !p->dense) {
    std::string msg = "no kernel path serves this problem: fused: " + why + "; dense: " + p->dense_why;
    mk_problem_destroy(p);
    return fail(MK_ERR_UNSUPPORTED, "%s", msg.c_str());
  }
  *out = p;
  return MK_OK;
}

void mk_problem_destroy(mk_problem* p) {
  if (!p) return;
  for (void* a : p->allocs) (void)hipFree(a);
  delete p;
}

int mk_problem_get_info(const mk_problem* p, mk_problem_info* info) {
  if (!p || !info) return fail(MK_ERR_INVALID, "null argument");
  *info = p->info;
  return MK_OK;
}

int mk_problem_set_posture_target(mk_problem* p, int32_t task, const double* target_q, void* stream) {
  if (!p || !target_q) return fail(MK_ERR_INVALID, "null argument");
  if (task < 0 || task >= (int)p->task_type.size() || p->task_type[task] != MK_TASK_POSTURE)
    return fail(MK_ERR_INVALID, "task %d is not a posture task of this problem", task);
  MK_ON_DEVICE(p->model);
  const HostModel& m = p->model->h;
  for (int i = 0; i < m.nq; ++i)
    if (!(std::fabs(target_q[i]) < INFINITY)) return fail(MK_ERR_INVALID, "posture target is not finite");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (p->fused || p->staged) {
    // the k-th posture task of the list is posture[k]; its target enters the per-dof constant
    // A = sum_t gain_t cost_t^2 target_t of the chain slots and of the other scalar dofs
    int k = 0;
    for (int t = 0; t < task; ++t) k += p->task_type[t] == MK_TASK_POSTURE;
    MkFusedDesc& D = p->h_fused;
    MkPosture& P = D.posture[k];
    for (int i = 0; i < m.nv; ++i)
      if (D.dof[i].kind == MK_DOF_SCALAR) P.target[i] = target_q[D.dof[i].qadr];
    auto refresh = [&](MkDofC& c) {
      if (c.dof < 0) return;
      c.A = 0.0;
      for (int t = 0; t < D.nposture; ++t)
        c.A += D.posture[t].gain * D.posture[t].cost[c.dof] * D.posture[t].cost[c.dof] * D.posture[t].target[c.dof];
    };
    for (int s = 0; s < MK_MAX_CHAIN; ++s) refresh(D.sdof[s]);
    for (int d = 0; d < D.nd; ++d) refresh(D.ddof[d]);
    MK_HIP(hipMemcpyAsync(p->d_fused, &p->h_fused, sizeof(MkFusedDesc), hipMemcpyHostToDevice, st));
  }
  if (p->dense)
    MK_HIP(hipMemcpyAsync(const_cast<double*>(p->h_dense.task[task].ptarget), target_q, sizeof(double) * m.nq,
                          hipMemcpyHostToDevice, st));
  return MK_OK;
}

// Which form of the fused kernels serves a batch of B problems: the lane-per-problem kernels
// need 64 problems per wave to be worth their long sequential solves; below about one problem
// per four lanes of the machine the cooperative kernels (16 lanes per problem) finish sooner.
// Posture tasks with a Levenberg-Marquardt term stay with the lane form.
static bool use_coop(const mk_problem* p, int B) {
  if (!p->fused || p->h_fused.any_posture_lm) return false;
  const long mode = mk_opt(MK_OPT_COOP);
  if (mode == 0) return false;
  return mode > 0 || B <= mk_opt(MK_OPT_COOP_MAX_BATCH);
}

// Live problems per wave at which the lane-per-problem kernels pass them on to the cooperative
// kernels (0: never; posture tasks with a Levenberg-Marquardt term stay with the lane form).
static long handoff_threshold(const mk_problem* p) {
  long handoff = mk_opt(MK_OPT_HANDOFF);
  if (handoff < 0) handoff = 4;
  if (p->h_fused.any_posture_lm || mk_opt(MK_OPT_COOP) == 0) handoff = 0;
  return handoff;
}

// Option "f32": the fused lane-per-problem kernels in float32 arithmetic.  Only problems the
// fused path serves without collision rows have such kernels; anything else is refused rather
// than silently computed in float64.
static bool want_f32() { return mk_opt(MK_OPT_F32) != 0; }
static int f32_refusal(const mk_problem* p) {
  if (p && p->fused && !p->staged) return MK_OK;
  return fail(MK_ERR_UNSUPPORTED, "option f32 is set, but float32 kernels exist only for problems of the fused path "
                                  "without collision limits or relative frame tasks");
}

int mk_set_option(const char* name, int64_t value) {
  if (!name) return fail(MK_ERR_INVALID, "null option name");
  for (int i = 0; i < MK_OPT_COUNT; ++i)
    if (strcmp(name, kOptions[i].name) == 0) {
      option_table()[i].store((long)value, std::memory_order_relaxed);
      return MK_OK;
    }
  return fail(MK_ERR_INVALID, "unknown option '%s'", name);
}

int mk_get_option(const char* name, int64_t* value) {
  if (!name || !value) return fail(MK_ERR_INVALID, "null argument");
  for (int i = 0; i < MK_OPT_COUNT; ++i)
    if (strcmp(name, kOptions[i].name) == 0) {
      *value = option_table()[i].load(std::memory_order_relaxed);
      return MK_OK;
    }
  return fail(MK_ERR_INVALID, "unknown option '%s'", name);
}

int mk_solve_ik_batch(const mk_problem* p, const double* d_q, const double* d_targets, double dt, double damping,
                      int32_t B, double* d_v, int32_t* d_status, void* stream) {
  if (!p) return fail(MK_ERR_INVALID, "null problem");
  if (B < 0 || !(dt > 0.0) || !(damping >= 0.0)) return fail(MK_ERR_INVALID, "B, dt or damping out of range");
  if (B == 0) return MK_OK;
  MK_ON_DEVICE(p->model);
  if (!d_q || !d_v || !d_status) return fail(MK_ERR_INVALID, "null argument");
  if (p->info.target_stride > 0 && !d_targets) return fail(MK_ERR_INVALID, "targets are required");
  if (want_f32()) {
    if (int rc = f32_refusal(p); rc != MK_OK) return rc;
    MK_HIP(mk_fused32_launch_step(p->nc_kernel, p->d_fused, p->info.nq, d_q, d_targets, p->info.target_stride, dt,
                                  damping, B, d_v, d_status, static_cast<hipStream_t>(stream)));
    return MK_OK;
  }
  if (p->staged) {
    // Structured kernel for every problem; it marks the ones with a collision pair that may be
    // within its detection distance.  Those go through the lane-per-problem kernel that builds the
    // collision rows and solves the QP with them, and what that one leaves pending (more rows than
    // it keeps, no convergence) through the general kernel;
    // both overwrite velocity and status of the problems they finish.
    MK_HIP(mk_fused_launch_step_ext(p->nc_kernel, p->d_fused, p->d_ext, p->info.nq, d_q, d_targets, p->info.target_stride,
                                    dt, damping, B, d_v, d_status, static_cast<hipStream_t>(stream),
                                    3 * p->h_ext.ncgeom + 4 * p->h_ext.norient));
    const bool lane_rows = mk_opt(MK_OPT_LANE_ROWS) != 0;   // tuning / tests: skip the middle stage
    if (p->h_ext.ncpair > 0 && lane_rows && p->info.staged_rows)
      MK_HIP(mk_fused_launch_rows(p->nc_kernel, p->d_fused, p->d_ext, p->info.nv, d_q, d_targets, p->info.target_stride,
                                  dt, damping, B, d_v, d_status, static_cast<hipStream_t>(stream)));
    // diagnostic switch of the tests: without the last stage, the problems it would have solved
    // come back with status 1 (the pending bit never reaches the caller)
    const bool general_stage = mk_opt(MK_OPT_GENERAL_STAGE) != 0;
    if (p->h_ext.ncpair > 0 && general_stage)
      MK_HIP(mk_dense_launch(p->model->dev, p->d_dense, p->info.n_rows, p->h_dense.n_general, d_q, d_targets,
                             p->info.target_stride, dt, damping, B, nullptr, nullptr, nullptr, nullptr, d_v, d_status, -1,
                             nullptr, nullptr, static_cast<hipStream_t>(stream), nullptr, false, true));
    else if (p->h_ext.ncpair > 0)
      MK_HIP(mk_generic_resolve_pending(d_status, B, static_cast<hipStream_t>(stream)));
    return MK_OK;
  }
  if (!p->fused) {
    MK_HIP(mk_dense_launch(p->model->dev, p->d_dense, p->info.n_rows, p->h_dense.n_general, d_q, d_targets, p->info.target_stride, dt,
                           damping, B, nullptr, nullptr, nullptr, nullptr, d_v, d_status, -1, nullptr, nullptr,
                           static_cast<hipStream_t>(stream)));
    return MK_OK;
  }
  if (use_coop(p, B)) {
    MK_HIP(mk_coop_launch_step(p->nc_kernel, p->d_fused, d_q, d_targets,
                               p->info.target_stride, dt, damping, B, d_v, d_status, static_cast<hipStream_t>(stream)));
    return MK_OK;
  }
  MK_HIP(mk_fused_launch_step(p->nc_kernel, p->d_fused, p->info.nq, d_q, d_targets, p->info.target_stride, dt,
                              damping, B, d_v, d_status, static_cast<hipStream_t>(stream)));
  return MK_OK;
}

// The callers' loop (interactive_scripts/teleop_mouse.py:105-121) outside the one-launch path, as
// passes of kernels; no device memory is allocated: the optional outputs of the iterated entry
// hold the state of the loop.  `iters` carries a done bit (MK_ITER_DONE_BIT) while the loop runs
// -- a problem that enters with it set is left alone by every launch --, `err` the latest
// frame-task error, `v` the velocity of each problem's last solve, `status` accumulates.  Nothing
// synchronises, so all max_iters passes are issued.
static int iterate_passes(const mk_problem* p, double* d_q, const double* d_targets, double dt, double damping, int32_t B,
                          int32_t max_iters, double pos_tol, double ori_tol, double* d_v, int32_t* d_iters, double* d_err,
                          int32_t* d_status, hipStream_t st) {
  if (p->staged && mk_opt(MK_OPT_STAGED_LOOP) != 0) {
    // Staged problems: one pass = the structured kernel in its loop mode (error, exit test and
    // bookkeeping of the pass, then the solve of the problems still in the loop), the rows stage
    // and the general kernel for what is pending, and the integration.
    for (int it = 0; it <= max_iters; ++it) {
      MK_HIP(mk_fused_launch_step_ext(p->nc_kernel, p->d_fused, p->d_ext, p->info.nq, d_q, d_targets, p->info.target_stride,
                                      dt, damping, B, d_v, d_status, st, 3 * p->h_ext.ncgeom + 4 * p->h_ext.norient, d_iters,
                                      d_err, it, max_iters, pos_tol, ori_tol));
      if (it == max_iters) break;
      if (p->h_ext.ncpair > 0) {
        if (p->info.staged_rows)
          MK_HIP(mk_fused_launch_rows(p->nc_kernel, p->d_fused, p->d_ext, p->info.nv, d_q, d_targets, p->info.target_stride, dt,
                                    damping, B, d_v, d_status, st));
        MK_HIP(mk_dense_launch(p->model->dev, p->d_dense, p->info.n_rows, p->h_dense.n_general, d_q, d_targets,
                               p->info.target_stride, dt, damping, B, nullptr, nullptr, nullptr, nullptr, d_v, d_status, -1,
                               nullptr, nullptr, st, nullptr, true, true));
      }
      MK_HIP(mk_generic_integrate(p->model->dev, d_q, d_v, dt, B, st, d_iters));
    }
    return MK_OK;
  }
  // General path: the wave-per-problem kernels; the exit test reads the first frame task's error.
  int frame_k = -1;
  for (int t = 0; t < p->h_dense.ntask && frame_k < 0; ++t)
    if (p->h_dense.task[t].type == MK_TASK_FRAME) frame_k = t;
  for (int it = 0; it <= max_iters; ++it) {
    if (frame_k >= 0)
      MK_HIP(mk_dense_launch(p->model->dev, p->d_dense, p->info.n_rows, p->h_dense.n_general, d_q, d_targets,
                             p->info.target_stride, 1.0, 0.0, B, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                             frame_k, d_err, nullptr, st, d_iters, false));
    MK_HIP(mk_generic_iterate_mark(d_iters, d_err, B, it, max_iters, pos_tol, ori_tol, frame_k >= 0, st));
    if (it == max_iters) break;
    MK_HIP(mk_dense_launch(p->model->dev, p->d_dense, p->info.n_rows, p->h_dense.n_general, d_q, d_targets,
                           p->info.target_stride, dt, damping, B, nullptr, nullptr, nullptr, nullptr, d_v, d_status, -1,
                           nullptr, nullptr, st, d_iters, true));
    MK_HIP(mk_generic_integrate(p->model->dev, d_q, d_v, dt, B, st, d_iters));
  }
  return MK_OK;
}

int mk_solve_ik_iterate_batch(const mk_problem* p, double* d_q, const double* d_targets, double dt, double damping,
                              int32_t B, int32_t max_iters, double pos_tol, double ori_tol, double* d_v,
                              int32_t* d_iters, double* d_err, int32_t* d_status, void* stream) {
  if (!p) return fail(MK_ERR_INVALID, "null problem");
  if (B < 0 || max_iters < 0 || !(dt > 0.0) || !(damping >= 0.0)) return fail(MK_ERR_INVALID, "B, max_iters, dt or damping out of range");
  if (B == 0) return MK_OK;
  MK_ON_DEVICE(p->model);
  if (!d_q || !d_status) return fail(MK_ERR_INVALID, "null argument");
  if (p->info.target_stride > 0 && !d_targets) return fail(MK_ERR_INVALID, "targets are required");
  if (want_f32()) {
    if (int rc = f32_refusal(p); rc != MK_OK) return rc;
    MK_HIP(mk_fused32_launch_iterate(p->nc_kernel, p->d_fused, p->info.nq, d_q, d_targets, p->info.target_stride, dt,
                                     damping, B, max_iters, pos_tol, ori_tol, d_v, d_iters, d_err, d_status,
                                     static_cast<hipStream_t>(stream), 0, nullptr));
    return MK_OK;
  }
  if ((p->staged && mk_opt(MK_OPT_STAGED_LOOP) != 0) || !p->fused) {
    if (!p->staged && !p->dense) return fail(MK_ERR_UNSUPPORTED, "dense path unavailable: %s", p->dense_why.c_str());
    if (!d_v || !d_iters || !d_err)
      return fail(MK_ERR_INVALID, "outside the fused path d_v, d_iters and d_err hold the state of the loop and must be given");
    hipStream_t st = static_cast<hipStream_t>(stream);
    MK_HIP(hipMemsetAsync(d_iters, 0, sizeof(int32_t) * (size_t)B, st));
    MK_HIP(hipMemsetAsync(d_status, 0, sizeof(int32_t) * (size_t)B, st));
    MK_HIP(hipMemsetAsync(d_v, 0, sizeof(double) * (size_t)B * p->info.nv, st));
    MK_HIP(hipMemsetAsync(d_err, 0, sizeof(double) * (size_t)B * 6, st));
    if (int rc = iterate_passes(p, d_q, d_targets, dt, damping, B, max_iters, pos_tol, ori_tol, d_v, d_iters, d_err, d_status, st);
        rc != MK_OK)
      return rc;
    MK_HIP(mk_generic_iterate_finish(d_iters, B, st));
    return MK_OK;
  }
  if (use_coop(p, B)) {
    MK_HIP(mk_coop_launch_iterate(p->nc_kernel, p->d_fused, d_q, d_targets,
                                  p->info.target_stride, dt, damping, B, max_iters, pos_tol, ori_tol, d_v, d_iters, d_err,
                                  d_status, static_cast<hipStream_t>(stream)));
    return MK_OK;
  }
  // With an exit test the lane-per-problem kernel hands the last few live problems of each wave to
  // the cooperative kernel (second launch below); not for posture tasks with a Levenberg-Marquardt
  // term, which the cooperative kernels do not carry.
  const long handoff = handoff_threshold(p);
  bool handed = false;
  MK_HIP(mk_fused_launch_iterate(p->nc_kernel, p->d_fused, p->info.nq, d_q, d_targets, p->info.target_stride, dt,
                                 damping, B, max_iters, pos_tol, ori_tol, d_v, d_iters, d_err, d_status,
                                 static_cast<hipStream_t>(stream), (int)handoff, &handed));
  if (handed)
    MK_HIP(mk_coop_launch_resume(p->nc_kernel, p->d_fused, d_q, d_targets, p->info.target_stride, dt, damping, B, max_iters,
                                 pos_tol, ori_tol, d_v, d_iters, d_err, d_status, 2048, static_cast<hipStream_t>(stream)));
  return MK_OK;
}

int mk_move_to_batch(const mk_problem* p, double* d_q, const double* d_goals, const mk_waypoint_config* wp, double dt,
                     double damping, int32_t B, int32_t max_steps, int32_t max_iters, double pos_tol, double ori_tol,
                     int32_t* d_steps, int32_t* d_reached, int32_t* d_solves, int32_t* d_status, void* stream) {
  if (!p || !wp) return fail(MK_ERR_INVALID, "null problem or waypoint configuration");
  if (B < 0 || max_steps < 0 || max_iters < 0 || !(dt > 0.0) || !(damping >= 0.0))
    return fail(MK_ERR_INVALID, "B, max_steps, max_iters, dt or damping out of range");
  if (!(wp->pos_threshold >= 0.0) || !(wp->pos_step_size > 0.0) || !(wp->rot_threshold >= 0.0))
    return fail(MK_ERR_INVALID, "waypoint thresholds / step size out of range");
  if (B == 0) return MK_OK;
  MK_ON_DEVICE(p->model);
  if (!d_q || !d_goals || !d_status) return fail(MK_ERR_INVALID, "null argument");
  if (!p->fused) return fail(MK_ERR_UNSUPPORTED, "waypoint reaching needs a problem the fused path serves");
  if (p->info.target_stride != 7) return fail(MK_ERR_UNSUPPORTED, "waypoint reaching drives exactly one frame task");
  if (want_f32()) {
    if (int rc = f32_refusal(p); rc != MK_OK) return rc;
    MK_HIP(mk_fused32_launch_move_to(p->nc_kernel, p->d_fused, p->info.nq, d_q, d_goals, dt, damping, B, max_steps,
                                     max_iters, pos_tol, ori_tol, wp->pos_threshold, wp->pos_step_size, wp->rot_threshold,
                                     d_steps, d_reached, d_solves, d_status, static_cast<hipStream_t>(stream), 0, nullptr));
    return MK_OK;
  }
  if (use_coop(p, B)) {
    MK_HIP(mk_coop_launch_move_to(p->nc_kernel, p->d_fused, d_q, d_goals, dt, damping, B, max_steps, max_iters, pos_tol,
                                  ori_tol, wp->pos_threshold, wp->pos_step_size, wp->rot_threshold, d_steps, d_reached,
                                  d_solves, d_status, static_cast<hipStream_t>(stream)));
    return MK_OK;
  }
  // the last rollouts of every wave finish on the cooperative kernel (second launch)
  const long handoff = handoff_threshold(p);
  bool handed = false;
  MK_HIP(mk_fused_launch_move_to(p->nc_kernel, p->d_fused, p->info.nq, d_q, d_goals, dt, damping, B, max_steps, max_iters,
                                 pos_tol, ori_tol, wp->pos_threshold, wp->pos_step_size, wp->rot_threshold, d_steps,
                                 d_reached, d_solves, d_status, static_cast<hipStream_t>(stream), (int)handoff, &handed));
  if (handed)
    MK_HIP(mk_coop_launch_move_to_resume(p->nc_kernel, p->d_fused, d_q, d_goals, dt, damping, B, max_steps, max_iters,
                                         pos_tol, ori_tol, wp->pos_threshold, wp->pos_step_size, wp->rot_threshold,
                                         d_steps, d_reached, d_solves, d_status, 2048, static_cast<hipStream_t>(stream)));
  return MK_OK;
}

int mk_move_to_passes_batch(const mk_problem* p, int32_t task_index, double* d_q, const double* d_goals, double* d_targets,
                            const mk_waypoint_config* wp, double dt, double damping, int32_t B, int32_t max_steps,
                            int32_t max_iters, double pos_tol, double ori_tol, int32_t* d_steps, int32_t* d_reached,
                            int32_t* d_solves, int32_t* d_status, double* d_v, int32_t* d_iters, double* d_err,
                            void* stream) {
  if (!p || !wp) return fail(MK_ERR_INVALID, "null problem or waypoint configuration");
  if (B < 0 || max_steps < 0 || max_iters < 0 || !(dt > 0.0) || !(damping >= 0.0))
    return fail(MK_ERR_INVALID, "B, max_steps, max_iters, dt or damping out of range");
  if (!(wp->pos_threshold >= 0.0) || !(wp->pos_step_size > 0.0) || !(wp->rot_threshold >= 0.0))
    return fail(MK_ERR_INVALID, "waypoint thresholds / step size out of range");
  if (B == 0) return MK_OK;
  MK_ON_DEVICE(p->model);
  if (!d_q || !d_goals || !d_targets || !d_steps || !d_reached || !d_solves || !d_status || !d_v || !d_iters || !d_err)
    return fail(MK_ERR_INVALID, "null argument");
  if (want_f32()) return f32_refusal(nullptr);
  if (!p->dense) return fail(MK_ERR_UNSUPPORTED, "dense path unavailable: %s", p->dense_why.c_str());
  if (task_index < 0 || task_index >= p->h_dense.ntask || p->h_dense.task[task_index].type != MK_TASK_FRAME)
    return fail(MK_ERR_INVALID, "task_index must name a frame task of the problem");
  // (the exit test of the IK loop reads the first frame task's error: the reference's loop tests
  // the driven task, envs/mj_env.py:270-275)
  for (int t = 0; t < task_index; ++t)
    if (p->h_dense.task[t].type == MK_TASK_FRAME)
      return fail(MK_ERR_UNSUPPORTED, "the driven frame task must be the first frame task of the problem");
  const MkDenseTask& T = p->h_dense.task[task_index];
  hipStream_t st = static_cast<hipStream_t>(stream);
  MK_HIP(hipMemsetAsync(d_steps, 0, sizeof(int32_t) * (size_t)B, st));
  MK_HIP(hipMemsetAsync(d_reached, 0, sizeof(int32_t) * (size_t)B, st));
  MK_HIP(hipMemsetAsync(d_solves, 0, sizeof(int32_t) * (size_t)B, st));
  MK_HIP(hipMemsetAsync(d_status, 0, sizeof(int32_t) * (size_t)B, st));
  MK_HIP(hipMemsetAsync(d_iters, 0, sizeof(int32_t) * (size_t)B, st));
  MK_HIP(hipMemsetAsync(d_v, 0, sizeof(double) * (size_t)B * p->info.nv, st));
  MK_HIP(hipMemsetAsync(d_err, 0, sizeof(double) * (size_t)B * 6, st));
  for (int step = 0; step < max_steps; ++step) {
    MK_HIP(mk_generic_move_to_begin(p->model->dev, d_q, B, T.body, T.lpos, T.lquat, d_goals, wp->pos_threshold,
                                    wp->pos_step_size, wp->rot_threshold, d_targets, p->info.target_stride, T.target_off,
                                    d_iters, d_steps, d_reached, d_solves, step == 0, st));
    if (int rc = iterate_passes(p, d_q, d_targets, dt, damping, B, max_iters, pos_tol, ori_tol, d_v, d_iters, d_err, d_status,
                                st);
        rc != MK_OK)
      return rc;
  }
  MK_HIP(mk_generic_move_to_end(d_iters, d_solves, B, st));
  return MK_OK;
}

int mk_integrate_batch(const mk_model* m, double* d_q, const double* d_v, double dt, int32_t B, void* stream) {
  if (!m || B < 0) return fail(MK_ERR_INVALID, "null model or negative batch");
  if (B == 0) return MK_OK;
  MK_ON_DEVICE(m);
  if (!d_q || !d_v) return fail(MK_ERR_INVALID, "null argument");
  MK_HIP(mk_generic_integrate(m->dev, d_q, d_v, dt, B, static_cast<hipStream_t>(stream)));
  return MK_OK;
}

int mk_fk_batch(const mk_model* m, const double* d_q, int32_t B, int32_t frame_type, int32_t frame_id, double* d_pose,
                void* stream) {
  if (!m || B < 0) return fail(MK_ERR_INVALID, "null model or negative batch");
  if (B > 0 && (!d_q || !d_pose)) return fail(MK_ERR_INVALID, "null argument");
  int body;
  double lpos[3], lquat[4];
  int rc = frame_body_and_local(m->h, frame_type, frame_id, &body, lpos, lquat);
  if (rc != MK_OK) return rc;
  if (B == 0) return MK_OK;
  MK_ON_DEVICE(m);
  MK_HIP(mk_generic_frame(m->dev, d_q, B, body, lpos, lquat, d_pose, nullptr, static_cast<hipStream_t>(stream)));
  return MK_OK;
}

int mk_frame_jacobian_batch(const mk_model* m, const double* d_q, int32_t B, int32_t frame_type, int32_t frame_id,
                            double* d_jac, void* stream) {
  if (!m || B < 0) return fail(MK_ERR_INVALID, "null model or negative batch");
  if (B > 0 && (!d_q || !d_jac)) return fail(MK_ERR_INVALID, "null argument");
  int body;
  double lpos[3], lquat[4];
  int rc = frame_body_and_local(m->h, frame_type, frame_id, &body, lpos, lquat);
  if (rc != MK_OK) return rc;
  if (B == 0) return MK_OK;
  MK_ON_DEVICE(m);
  MK_HIP(mk_generic_frame(m->dev, d_q, B, body, lpos, lquat, nullptr, d_jac, static_cast<hipStream_t>(stream)));
  return MK_OK;
}

int mk_check_limits_batch(const mk_model* m, const double* d_q, double tol, int32_t B, int32_t* d_first, void* stream) {
  if (!m || B < 0) return fail(MK_ERR_INVALID, "null model or negative batch");
  if (B == 0) return MK_OK;
  MK_ON_DEVICE(m);
  if (!d_q || !d_first) return fail(MK_ERR_INVALID, "null argument");
  MK_HIP(mk_generic_check_limits(m->dev, d_q, tol, B, d_first, static_cast<hipStream_t>(stream)));
  return MK_OK;
}

int mk_build_qp_batch(const mk_problem* p, const double* d_q, const double* d_targets, double dt, double damping,
                      int32_t B, double* d_P, double* d_c, double* d_G, double* d_h, void* stream) {
  if (!p) return fail(MK_ERR_INVALID, "null problem");
  if (B < 0 || !(dt > 0.0) || !(damping >= 0.0)) return fail(MK_ERR_INVALID, "B, dt or damping out of range");
  if (B == 0) return MK_OK;
  MK_ON_DEVICE(p->model);
  if (!d_q || !d_P || !d_c) return fail(MK_ERR_INVALID, "null argument");
  if (p->info.n_rows > 0 && (!d_G || !d_h)) return fail(MK_ERR_INVALID, "G and h are required when the problem has limits");
  if (p->info.target_stride > 0 && !d_targets) return fail(MK_ERR_INVALID, "targets are required");
  if (!p->dense) return fail(MK_ERR_UNSUPPORTED, "dense path unavailable: %s", p->dense_why.c_str());
  MK_HIP(mk_dense_launch(p->model->dev, p->d_dense, p->info.n_rows, p->h_dense.n_general, d_q, d_targets, p->info.target_stride, dt, damping,
                         B, d_P, d_c, p->info.n_rows > 0 ? d_G : nullptr, d_h, nullptr, nullptr, -1, nullptr, nullptr,
                         static_cast<hipStream_t>(stream)));
  return MK_OK;
}

int mk_task_eval_batch(const mk_problem* p, int32_t task, const double* d_q, const double* d_targets, int32_t B,
                       double* d_err, double* d_jac, void* stream) {
  if (!p) return fail(MK_ERR_INVALID, "null problem");
  if (B < 0 || task < 0 || task >= p->h_dense.ntask) return fail(MK_ERR_INVALID, "batch or task index out of range");
  if (B == 0) return MK_OK;
  MK_ON_DEVICE(p->model);
  if (!d_q || (!d_err && !d_jac)) return fail(MK_ERR_INVALID, "null argument");
  if (p->info.target_stride > 0 && !d_targets) return fail(MK_ERR_INVALID, "targets are required");
  if (!p->dense) return fail(MK_ERR_UNSUPPORTED, "dense path unavailable: %s", p->dense_why.c_str());
  MK_HIP(mk_dense_launch(p->model->dev, p->d_dense, p->info.n_rows, p->h_dense.n_general, d_q, d_targets, p->info.target_stride, 1.0, 0.0, B,
                         nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, task, d_err, d_jac,
                         static_cast<hipStream_t>(stream)));
  return MK_OK;
}

int mk_qp_solve_batch(const double* d_P, const double* d_c, const double* d_G, const double* d_h, int32_t n, int32_t m,
                      int32_t B, double* d_x, int32_t* d_status, void* stream) {
  if (n <= 0 || m < 0 || B < 0) return fail(MK_ERR_INVALID, "n, m or B out of range");
  if (B == 0) return MK_OK;
  if (!d_P || !d_c || !d_x || !d_status || (m > 0 && (!d_G || !d_h))) return fail(MK_ERR_INVALID, "null argument");
  if (mk_dense_lds_bytes(n, m, 0, 0, 0, 0, 0) == 0) return fail(MK_ERR_UNSUPPORTED, "QP too large for the LDS-resident solver (n <= 64 and about (4 n^2 + m n) * 8 bytes <= 150 KB)");
  MK_HIP(mk_dense_qp_launch(d_P, d_c, d_G, d_h, n, m, B, d_x, d_status, static_cast<hipStream_t>(stream)));
  return MK_OK;
}

int mk_lie_op_shape(int32_t op, int32_t* wa, int32_t* wb, int32_t* wo) {
  if (!wa || !wb || !wo) return fail(MK_ERR_INVALID, "null argument");
  if (!mk_lie_shape(op, wa, wb, wo)) return fail(MK_ERR_INVALID, "unknown Lie operation %d", op);
  return MK_OK;
}

int mk_lie_batch(int32_t op, const double* d_a, const double* d_b, int32_t B, double* d_out, void* stream) {
  int wa, wb, wo;
  if (!mk_lie_shape(op, &wa, &wb, &wo)) return fail(MK_ERR_INVALID, "unknown Lie operation %d", op);
  if (B < 0) return fail(MK_ERR_INVALID, "negative batch");
  if (B == 0) return MK_OK;
  if (!d_a || !d_out || (wb > 0 && !d_b)) return fail(MK_ERR_INVALID, "null argument");
  MK_HIP(mk_lie_launch(op, d_a, wb > 0 ? d_b : nullptr, B, d_out, static_cast<hipStream_t>(stream)));
  return MK_OK;
}

}  // extern "C"
