"""Batched SO3 / SE3 value types of the host API.

Same interface as the reference's ``mink.lie`` (mink/lie/base.py, so3.py, se3.py) --
quaternions are (w, x, y, z), SE3 parameters (qw, qx, qy, qz, x, y, z), tangents
(v; omega), Taylor switches at 1e-10 -- with an arbitrary leading batch shape.

Elements held in GPU tensors are operated on by the HIP kernels (``mk_lie_batch``, one
element per lane, the same device functions the IK kernels use for their frame tasks).
Elements held in host tensors (single targets, fixtures, configuration-free code) are
evaluated by the torch expressions below; nothing on the IK path goes through those.
"""
from __future__ import annotations

import math

import numpy as np
import torch

EPS = 1e-10  # mink/lie/utils.py:4-8, float64


def _t(x, like=None) -> torch.Tensor:
    if isinstance(x, torch.Tensor):
        return x.to(torch.float64)
    dev = like.device if isinstance(like, torch.Tensor) else None
    return torch.as_tensor(np.asarray(x, dtype=np.float64), device=dev)


def get_epsilon(dtype) -> float:
    """Small-angle switch of the reference per floating type (mink/lie/utils.py:4-8)."""
    import numpy as np
    if isinstance(dtype, torch.dtype):
        return {torch.float32: 1e-5, torch.float64: 1e-10}[dtype]
    return {np.dtype("float32"): 1e-5, np.dtype("float64"): 1e-10}[np.dtype(dtype)]


def skew(w: torch.Tensor) -> torch.Tensor:
    w = _t(w)
    if w.shape[-1] != 3:
        raise AssertionError("skew expects (..., 3)")
    z = torch.zeros_like(w[..., 0])
    return torch.stack([torch.stack([z, -w[..., 2], w[..., 1]], -1),
                        torch.stack([w[..., 2], z, -w[..., 0]], -1),
                        torch.stack([-w[..., 1], w[..., 0], z], -1)], -2)


def _qmul(a, b):
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack([aw * bw - ax * bx - ay * by - az * bz,
                        aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx,
                        aw * bz + ax * by - ay * bx + az * bw], -1)


_MAT_SHAPE = {9: (3, 3), 16: (4, 4), 36: (6, 6)}


def _on_gpu(*ts) -> bool:
    return any(isinstance(t, torch.Tensor) and t.is_cuda for t in ts)


def _hip(op: str, a: torch.Tensor, b: torch.Tensor = None) -> torch.Tensor:
    """One mk_lie_batch call over the broadcast batch shape of a (and b)."""
    import ctypes as C

    from . import _lib as L
    lib = L.load_library()
    code = L.LIE_OPS[op]
    wa, wb, wo = C.c_int32(), C.c_int32(), C.c_int32()
    L.check(lib.mk_lie_op_shape(code, C.byref(wa), C.byref(wb), C.byref(wo)))
    wa, wb, wo = wa.value, wb.value, wo.value
    dev = a.device if a.is_cuda else b.device
    a = a.to(device=dev, dtype=torch.float64)
    if wa in _MAT_SHAPE:
        a = a.reshape(*a.shape[:-2], wa)
    batch = a.shape[:-1]
    if wb:
        b = b.to(device=dev, dtype=torch.float64)
        batch = torch.broadcast_shapes(batch, b.shape[:-1])
        b = b.expand(*batch, wb).reshape(-1, wb).contiguous()
    a = a.expand(*batch, wa).reshape(-1, wa).contiguous()
    n = a.shape[0]
    out = torch.empty((n, wo), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        L.check(lib.mk_lie_batch(code, a.data_ptr(), b.data_ptr() if wb else None, n, out.data_ptr(),
                                 C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return out.reshape(*batch, *_MAT_SHAPE.get(wo, (wo,)))


def _eye(n, like):
    return torch.eye(n, dtype=torch.float64, device=like.device).expand(*like.shape[:-1], n, n)


class MatrixLieGroup:
    """Operators shared by SO3 and SE3 (mink/lie/base.py:8-156)."""

    def __matmul__(self, other):
        if isinstance(other, (np.ndarray, torch.Tensor)):
            return self.apply(other)
        return self.multiply(other)

    def rplus(self, other):
        p = self.parameters()
        if _on_gpu(p):
            return type(self)(_hip(self._prefix + "rplus", p, _t(other, p)))
        return self @ type(self).exp(other)

    def rminus(self, other):
        p = self.parameters()
        if _on_gpu(p, other.parameters()):
            return _hip(self._prefix + "rminus", p, other.parameters())
        return (other.inverse() @ self).log()

    def lplus(self, other):
        p = self.parameters()
        if _on_gpu(p):
            return type(self)(_hip(self._prefix + "lplus", p, _t(other, p)))
        return type(self).exp(other) @ self

    def lminus(self, other):
        p = self.parameters()
        if _on_gpu(p, other.parameters()):
            return _hip(self._prefix + "lminus", p, other.parameters())
        return (self @ other.inverse()).log()

    def plus(self, other):
        return self.rplus(other)

    def minus(self, other):
        return self.rminus(other)

    @classmethod
    def rjac(cls, other):
        o = _t(other)
        if _on_gpu(o):
            return _hip(cls._prefix + "rjac", o)
        return cls.ljac(-o)

    @classmethod
    def rjacinv(cls, other):
        o = _t(other)
        if _on_gpu(o):
            return _hip(cls._prefix + "rjacinv", o)
        return cls.ljacinv(-o)

    def jlog(self):
        p = self.parameters()
        if _on_gpu(p):
            return _hip(self._prefix + "jlog", p)
        return type(self).rjacinv(self.log())


class SO3(MatrixLieGroup):
    matrix_dim, parameters_dim, tangent_dim, space_dim = 3, 4, 3, 3
    _prefix = "so3_"

    def __init__(self, wxyz):
        wxyz = _t(wxyz)
        if wxyz.shape[-1] != 4:
            raise ValueError(f"Expeced wxyz to be a length 4 vector but got {wxyz.shape[-1]}.")
        self.wxyz = wxyz

    def __repr__(self):
        return f"SO3(wxyz={np.round(self.wxyz.detach().cpu().numpy(), 5)})"

    def parameters(self):
        return self.wxyz

    def copy(self):
        return SO3(self.wxyz.clone())

    @classmethod
    def identity(cls, batch=(), device=None):
        q = torch.zeros(*batch, 4, dtype=torch.float64, device=device)
        q[..., 0] = 1.0
        return SO3(q)

    @classmethod
    def from_x_radians(cls, theta):
        t = _t(theta)
        return cls.exp(torch.stack([t, torch.zeros_like(t), torch.zeros_like(t)], -1))

    @classmethod
    def from_y_radians(cls, theta):
        t = _t(theta)
        return cls.exp(torch.stack([torch.zeros_like(t), t, torch.zeros_like(t)], -1))

    @classmethod
    def from_z_radians(cls, theta):
        t = _t(theta)
        return cls.exp(torch.stack([torch.zeros_like(t), torch.zeros_like(t), t], -1))

    @classmethod
    def from_rpy_radians(cls, roll, pitch, yaw):
        r, p, y = _t(roll), _t(pitch), _t(yaw)
        if _on_gpu(r, p, y):
            return SO3(_hip("so3_from_rpy", torch.stack(torch.broadcast_tensors(r, p, y), -1)))
        return cls.from_z_radians(yaw) @ cls.from_y_radians(pitch) @ cls.from_x_radians(roll)

    @classmethod
    def from_matrix(cls, matrix):
        """Quaternion of a rotation matrix with the branch structure of mju_mat2Quat."""
        m = _t(matrix)
        assert m.shape[-2:] == (3, 3)
        if _on_gpu(m):
            return SO3(_hip("so3_from_matrix", m))
        m00, m11, m22 = m[..., 0, 0], m[..., 1, 1], m[..., 2, 2]
        tr = m00 + m11 + m22
        c0 = tr > 0
        c1 = (~c0) & (m00 > m11) & (m00 > m22)
        c2 = (~c0) & (~c1) & (m11 > m22)
        one = torch.ones_like(tr)

        def branch(lead, a, b, c):
            s = 0.5 * torch.sqrt(torch.clamp(lead, min=1e-300))
            return s, 0.25 * a / s, 0.25 * b / s, 0.25 * c / s

        w0, x0, y0, z0 = branch(one + tr, m[..., 2, 1] - m[..., 1, 2], m[..., 0, 2] - m[..., 2, 0],
                                m[..., 1, 0] - m[..., 0, 1])
        x1, w1, y1, z1 = branch(one + m00 - m11 - m22, m[..., 2, 1] - m[..., 1, 2], m[..., 0, 1] + m[..., 1, 0],
                                m[..., 0, 2] + m[..., 2, 0])
        y2, w2, x2, z2 = branch(one - m00 + m11 - m22, m[..., 0, 2] - m[..., 2, 0], m[..., 0, 1] + m[..., 1, 0],
                                m[..., 1, 2] + m[..., 2, 1])
        z3, w3, x3, y3 = branch(one - m00 - m11 + m22, m[..., 1, 0] - m[..., 0, 1], m[..., 0, 2] + m[..., 2, 0],
                                m[..., 1, 2] + m[..., 2, 1])

        def pick(a0, a1, a2, a3):
            return torch.where(c0, a0, torch.where(c1, a1, torch.where(c2, a2, a3)))

        q = torch.stack([pick(w0, w1, w2, w3), pick(x0, x1, x2, x3), pick(y0, y1, y2, y3), pick(z0, z1, z2, z3)], -1)
        return SO3(q / torch.linalg.norm(q, dim=-1, keepdim=True))

    @classmethod
    def sample_uniform(cls, batch=(), generator=None, device=None):
        u = torch.rand(*batch, 3, dtype=torch.float64, generator=generator, device=device)
        u1, u2, u3 = u[..., 0], 2 * math.pi * u[..., 1], 2 * math.pi * u[..., 2]
        a, b = torch.sqrt(1 - u1), torch.sqrt(u1)
        return SO3(torch.stack([a * torch.sin(u2), a * torch.cos(u2), b * torch.sin(u3), b * torch.cos(u3)], -1))

    def as_matrix(self):
        w, x, y, z = self.wxyz.unbind(-1)
        if _on_gpu(self.wxyz):
            return _hip("so3_as_matrix", self.wxyz)
        return torch.stack([
            torch.stack([w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
            torch.stack([2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)], -1),
            torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z], -1)], -2)

    def compute_roll_radians(self):
        q0, q1, q2, q3 = self.wxyz.unbind(-1)
        return torch.atan2(2 * (q0 * q1 + q2 * q3), 1 - 2 * (q1**2 + q2**2))

    def compute_pitch_radians(self):
        q0, q1, q2, q3 = self.wxyz.unbind(-1)
        return torch.asin(2 * (q0 * q2 - q3 * q1))

    def compute_yaw_radians(self):
        q0, q1, q2, q3 = self.wxyz.unbind(-1)
        return torch.atan2(2 * (q0 * q3 + q1 * q2), 1 - 2 * (q2**2 + q3**2))

    def as_rpy_radians(self):
        if _on_gpu(self.wxyz):
            return tuple(_hip("so3_as_rpy", self.wxyz).unbind(-1))
        return (self.compute_roll_radians(), self.compute_pitch_radians(), self.compute_yaw_radians())

    def inverse(self):
        if _on_gpu(self.wxyz):
            return SO3(_hip("so3_inverse", self.wxyz))
        return SO3(self.wxyz * torch.tensor([1.0, -1.0, -1.0, -1.0], dtype=torch.float64, device=self.wxyz.device))

    def normalize(self):
        if _on_gpu(self.wxyz):
            return SO3(_hip("so3_normalize", self.wxyz))
        return SO3(self.wxyz / torch.linalg.norm(self.wxyz, dim=-1, keepdim=True))

    def apply(self, target):
        t = _t(target, self.wxyz)
        assert t.shape[-1] == 3
        if _on_gpu(self.wxyz, t):
            return _hip("so3_apply", self.wxyz, t)
        pad = torch.cat([torch.zeros_like(t[..., :1]), t], -1)
        return _qmul(_qmul(self.wxyz, pad), self.inverse().wxyz)[..., 1:]

    def multiply(self, other):
        if _on_gpu(self.wxyz, other.wxyz):
            return SO3(_hip("so3_multiply", self.wxyz, other.wxyz))
        return SO3(_qmul(self.wxyz, other.wxyz))

    @classmethod
    def exp(cls, tangent):
        w = _t(tangent)
        assert w.shape[-1] == 3
        if _on_gpu(w):
            return SO3(_hip("so3_exp", w))
        t2 = (w * w).sum(-1)
        small = t2 < EPS
        th = torch.sqrt(torch.where(small, torch.ones_like(t2), t2))
        re = torch.where(small, 1 - t2 / 8 + t2 * t2 / 384, torch.cos(0.5 * th))
        im = torch.where(small, 0.5 - t2 / 48 + t2 * t2 / 3840, torch.sin(0.5 * th) / th)
        return SO3(torch.cat([re[..., None], im[..., None] * w], -1))

    def log(self):
        if _on_gpu(self.wxyz):
            return _hip("so3_log", self.wxyz)
        w = self.wxyz[..., 0]
        v = self.wxyz[..., 1:]
        n2 = (v * v).sum(-1)
        small = n2 < EPS
        n = torch.sqrt(torch.where(small, torch.ones_like(n2), n2))
        ws = torch.where(small, w, torch.ones_like(w))
        taylor = 2.0 / ws - 2.0 / 3.0 * n2 / ws**3
        at = torch.atan2(torch.where(w < 0, -n, n), torch.abs(w))
        near_pi = torch.where(w > 0, torch.ones_like(w), -torch.ones_like(w)) * math.pi / n
        f = torch.where(small, taylor, torch.where(torch.abs(w) < EPS, near_pi, 2.0 * at / n))
        return f[..., None] * v

    def adjoint(self):
        return self.as_matrix()

    @classmethod
    def ljac(cls, other):
        w = _t(other)
        if _on_gpu(w):
            return _hip("so3_ljac", w)
        th = torch.sqrt((w * w).sum(-1))
        small = th < EPS
        t2 = th * th
        ths = torch.where(small, torch.ones_like(th), th)
        A = torch.where(small, 0.5 * (1 - t2 / 12 * (1 - t2 / 30 * (1 - t2 / 56))), (1 - torch.cos(ths)) / ths**2)
        B = torch.where(small, (1 / 6) * (1 - t2 / 20 * (1 - t2 / 42 * (1 - t2 / 72))), (ths - torch.sin(ths)) / ths**3)
        S = skew(w)
        return _eye(3, w) + A[..., None, None] * S + B[..., None, None] * (S @ S)

    @classmethod
    def ljacinv(cls, other):
        w = _t(other)
        if _on_gpu(w):
            return _hip("so3_ljacinv", w)
        th = torch.sqrt((w * w).sum(-1))
        small = th < EPS
        t2 = th * th
        ths = torch.where(small, torch.ones_like(th), th)
        A = torch.where(small, (1 / 12) * (1 + t2 / 60 * (1 + t2 / 42 * (1 + t2 / 40))),
                        (1 / ths**2) * (1 - (ths * torch.sin(ths) / (2 * (1 - torch.cos(ths))))))
        S = skew(w)
        return _eye(3, w) - 0.5 * S + A[..., None, None] * (S @ S)


def _getQ(c: torch.Tensor) -> torch.Tensor:
    """mink/lie/se3.py:222-249."""
    t2 = (c[..., 3:] * c[..., 3:]).sum(-1)
    small = t2 < EPS
    th = torch.sqrt(torch.where(small, torch.ones_like(t2), t2))
    s, co = torch.sin(th), torch.cos(th)
    B = torch.where(small, 1 / 6 + t2 / 120, (th - s) / (th**3))
    C = torch.where(small, -1 / 24 + t2 / 720, (1 - th * th / 2 - co) / (th**4))
    D = torch.where(small, torch.full_like(t2, -1 / 60), (2 * th - 3 * s + th * co) / (2 * th**5))
    V, W = skew(c[..., :3]), skew(c[..., 3:])
    VW = V @ W
    WV = VW.transpose(-1, -2)
    WVW = WV @ W
    VWW = VW @ W
    e = lambda x: x[..., None, None]  # noqa: E731
    return 0.5 * V + e(B) * (WV + VW + WVW) - e(C) * (VWW - VWW.transpose(-1, -2) - 3 * WVW) + e(D) * (WVW @ W + W @ WVW)


class SE3(MatrixLieGroup):
    matrix_dim, parameters_dim, tangent_dim, space_dim = 4, 7, 6, 3
    _prefix = "se3_"

    def __init__(self, wxyz_xyz):
        wxyz_xyz = _t(wxyz_xyz)
        if wxyz_xyz.shape[-1] != 7:
            raise ValueError(f"Expected wxyz_xyz to have 7 entries but got {wxyz_xyz.shape[-1]}.")
        self.wxyz_xyz = wxyz_xyz

    def __repr__(self):
        a = self.wxyz_xyz.detach().cpu().numpy()
        return f"SE3(wxyz={np.round(a[..., :4], 5)}, xyz={np.round(a[..., 4:], 5)})"

    def copy(self):
        return SE3(self.wxyz_xyz.clone())

    def parameters(self):
        return self.wxyz_xyz

    @classmethod
    def identity(cls, batch=(), device=None):
        p = torch.zeros(*batch, 7, dtype=torch.float64, device=device)
        p[..., 0] = 1.0
        return SE3(p)

    @classmethod
    def from_rotation_and_translation(cls, rotation: SO3, translation):
        t = _t(translation, rotation.wxyz)
        assert t.shape[-1] == 3
        q = rotation.wxyz
        shape = torch.broadcast_shapes(q.shape[:-1], t.shape[:-1])
        return SE3(torch.cat([q.expand(*shape, 4), t.expand(*shape, 3)], -1))

    @classmethod
    def from_rotation(cls, rotation: SO3):
        return cls.from_rotation_and_translation(rotation, torch.zeros_like(rotation.wxyz[..., :3]))

    @classmethod
    def from_translation(cls, translation):
        t = _t(translation)
        return cls.from_rotation_and_translation(SO3.identity(t.shape[:-1], t.device), t)

    @classmethod
    def from_matrix(cls, matrix):
        m = _t(matrix)
        assert m.shape[-2:] == (4, 4)
        if _on_gpu(m):
            return SE3(_hip("se3_from_matrix", m))
        return cls.from_rotation_and_translation(SO3.from_matrix(m[..., :3, :3]), m[..., :3, 3])

    @classmethod
    def from_mocap_id(cls, data, mocap_id: int):
        """Pose of a mocap body held by ``configuration.data`` (mink/lie/se3.py:77-82)."""
        data = getattr(data, "data", data)
        return cls.from_rotation_and_translation(SO3(data.mocap_quat[..., mocap_id, :]),
                                                 data.mocap_pos[..., mocap_id, :])

    @classmethod
    def from_mocap_name(cls, model, data, mocap_name: str):
        """mink/lie/se3.py:84-91."""
        from .utils import _mocap_id
        return cls.from_mocap_id(data, _mocap_id(model, mocap_name))

    @classmethod
    def sample_uniform(cls, batch=(), generator=None, device=None):
        t = 2 * torch.rand(*batch, 3, dtype=torch.float64, generator=generator, device=device) - 1
        return cls.from_rotation_and_translation(SO3.sample_uniform(batch, generator, device), t)

    def rotation(self) -> SO3:
        return SO3(self.wxyz_xyz[..., :4])

    def translation(self):
        return self.wxyz_xyz[..., 4:]

    def as_matrix(self):
        if _on_gpu(self.wxyz_xyz):
            return _hip("se3_as_matrix", self.wxyz_xyz)
        R = self.rotation().as_matrix()
        top = torch.cat([R, self.translation()[..., None]], -1)
        bottom = torch.zeros_like(top[..., :1, :])
        bottom[..., 0, 3] = 1.0
        return torch.cat([top, bottom], -2)

    @classmethod
    def exp(cls, tangent):
        xi = _t(tangent)
        assert xi.shape[-1] == 6
        if _on_gpu(xi):
            return SE3(_hip("se3_exp", xi))
        w = xi[..., 3:]
        rot = SO3.exp(w)
        t2 = (w * w).sum(-1)
        small = t2 < EPS
        t2s = torch.where(small, torch.ones_like(t2), t2)
        th = torch.sqrt(t2s)
        S = skew(w)
        V = _eye(3, w) + ((1 - torch.cos(th)) / t2s)[..., None, None] * S + \
            ((th - torch.sin(th)) / (t2s * th))[..., None, None] * (S @ S)
        V = torch.where(small[..., None, None], rot.as_matrix(), V)
        return cls.from_rotation_and_translation(rot, (V @ xi[..., :3, None])[..., 0])

    def inverse(self):
        if _on_gpu(self.wxyz_xyz):
            return SE3(_hip("se3_inverse", self.wxyz_xyz))
        Ri = self.rotation().inverse()
        return SE3.from_rotation_and_translation(Ri, -Ri.apply(self.translation()))

    def normalize(self):
        if _on_gpu(self.wxyz_xyz):
            return SE3(_hip("se3_normalize", self.wxyz_xyz))
        return SE3.from_rotation_and_translation(self.rotation().normalize(), self.translation())

    def apply(self, target):
        t = _t(target, self.wxyz_xyz)
        if _on_gpu(self.wxyz_xyz, t):
            return _hip("se3_apply", self.wxyz_xyz, t)
        return self.rotation().apply(t) + self.translation()

    def multiply(self, other):
        if _on_gpu(self.wxyz_xyz, other.wxyz_xyz):
            return SE3(_hip("se3_multiply", self.wxyz_xyz, other.wxyz_xyz))
        return SE3.from_rotation_and_translation(self.rotation() @ other.rotation(),
                                                 self.rotation().apply(other.translation()) + self.translation())

    def log(self):
        if _on_gpu(self.wxyz_xyz):
            return _hip("se3_log", self.wxyz_xyz)
        w = self.rotation().log()
        t2 = (w * w).sum(-1)
        small = t2 < EPS
        t2s = torch.where(small, torch.ones_like(t2), t2)
        th = torch.sqrt(t2s)
        S = skew(w)
        S2 = S @ S
        coef = torch.where(small, torch.full_like(t2, 1 / 12),
                           (1 - th * torch.cos(0.5 * th) / (2 * torch.sin(0.5 * th))) / t2s)
        Vinv = _eye(3, w) - 0.5 * S + coef[..., None, None] * S2
        return torch.cat([(Vinv @ self.translation()[..., None])[..., 0], w], -1)

    def adjoint(self):
        if _on_gpu(self.wxyz_xyz):
            return _hip("se3_adjoint", self.wxyz_xyz)
        R = self.rotation().as_matrix()
        top = torch.cat([R, skew(self.translation()) @ R], -1)
        bottom = torch.cat([torch.zeros_like(R), R], -1)
        return torch.cat([top, bottom], -2)

    @classmethod
    def ljac(cls, other):
        c = _t(other)
        if _on_gpu(c):
            return _hip("se3_ljac", c)
        w = c[..., 3:]
        small = (w * w).sum(-1) < EPS
        J = SO3.ljac(w)
        full = torch.cat([torch.cat([J, _getQ(c)], -1), torch.cat([torch.zeros_like(J), J], -1)], -2)
        return torch.where(small[..., None, None], _eye(6, c), full)

    @classmethod
    def ljacinv(cls, other):
        c = _t(other)
        if _on_gpu(c):
            return _hip("se3_ljacinv", c)
        w = c[..., 3:]
        small = (w * w).sum(-1) < EPS
        Ji = SO3.ljacinv(w)
        full = torch.cat([torch.cat([Ji, -Ji @ _getQ(c) @ Ji], -1), torch.cat([torch.zeros_like(Ji), Ji], -1)], -2)
        return torch.where(small[..., None, None], _eye(6, c), full)
