"""The SO3 / SE3 surface of mink.lie on the GPU, through the C ABI (mk_lie_batch): every
operation against the golden vectors recorded from the reference (tests/golden/lie.npz,
which holds inputs on each branch: near-identity, near-pi, w < 0, |w| ~ 0), the cases on
either side of the small-angle switches against the numpy restatement, and the group
axioms of the reference's tests/test_lie_axioms.py / test_lie_operations.py at size."""
import ctypes as C
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from mink_amd import _lib as L
from mink_amd.lie import SE3, SO3
from oracle import ik_oracle as o

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = dict(rtol=1e-12, atol=1e-12)


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "lie.npz"))


def lie(op, a, b=None):
    """Direct C-ABI call on numpy inputs."""
    lib = L.load_library()
    code = L.LIE_OPS[op]
    wa, wb, wo = C.c_int32(), C.c_int32(), C.c_int32()
    L.check(lib.mk_lie_op_shape(code, C.byref(wa), C.byref(wb), C.byref(wo)))
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1, wa.value)
    ta = torch.tensor(a, device=DEV)
    tb = None
    if wb.value:
        tb = torch.tensor(np.ascontiguousarray(b, dtype=np.float64).reshape(-1, wb.value), device=DEV)
    out = torch.empty((a.shape[0], wo.value), dtype=torch.float64, device=DEV)
    L.check(lib.mk_lie_batch(code, ta.data_ptr(), tb.data_ptr() if tb is not None else None, a.shape[0],
                             out.data_ptr(), None))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def quat_close(q, ref, tol):
    s = np.sign((q * ref).sum(-1, keepdims=True))
    np.testing.assert_allclose(q * s, ref, rtol=tol, atol=tol)


def test_golden_vectors_through_the_c_abi(g):
    T, T2, xi = g["T"], g["T2"], g["xi"]
    n = len(T)
    np.testing.assert_allclose(lie("se3_log", T), g["se3_log"], **TOL)
    np.testing.assert_allclose(lie("se3_jlog", T).reshape(n, 6, 6), g["se3_jlog"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(lie("se3_inverse", T), g["se3_inverse"], **TOL)
    np.testing.assert_allclose(lie("se3_adjoint", T).reshape(n, 6, 6), g["se3_adjoint"], **TOL)
    np.testing.assert_allclose(lie("se3_as_matrix", T).reshape(n, 4, 4), g["se3_matrix"], **TOL)
    np.testing.assert_allclose(lie("se3_multiply", T, T2), g["se3_multiply"], **TOL)
    np.testing.assert_allclose(lie("se3_rminus", T, T2), g["se3_rminus"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(lie("se3_lminus", T, T2), g["se3_lminus"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(lie("se3_exp", xi), g["se3_exp"], **TOL)
    np.testing.assert_allclose(lie("se3_rplus", T, xi), g["se3_rplus"], **TOL)
    np.testing.assert_allclose(lie("se3_lplus", T, xi), g["se3_lplus"], **TOL)
    np.testing.assert_allclose(lie("se3_ljac", xi).reshape(n, 6, 6), g["se3_ljac"], **TOL)
    np.testing.assert_allclose(lie("se3_ljacinv", xi).reshape(n, 6, 6), g["se3_ljacinv"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(lie("se3_apply", T, xi[:, :3]), g["se3_apply"], **TOL)
    R = T[:, :4]
    np.testing.assert_allclose(lie("so3_log", R), g["so3_log"], **TOL)
    np.testing.assert_allclose(lie("so3_exp", xi[:, 3:]), g["so3_exp"], **TOL)
    np.testing.assert_allclose(lie("so3_ljac", xi[:, 3:]).reshape(n, 3, 3), g["so3_ljac"], **TOL)
    np.testing.assert_allclose(lie("so3_ljacinv", xi[:, 3:]).reshape(n, 3, 3), g["so3_ljacinv"], **TOL)
    np.testing.assert_allclose(lie("so3_as_matrix", R).reshape(n, 3, 3), g["so3_matrix"], **TOL)
    np.testing.assert_allclose(lie("so3_apply", R, xi[:, :3]), g["so3_apply"], **TOL)
    quat_close(lie("so3_from_matrix", g["so3_matrix"]), g["so3_from_matrix"], 1e-11)
    np.testing.assert_allclose(lie("so3_from_rpy", g["rpy"]), g["so3_from_rpy"], **TOL)
    np.testing.assert_allclose(lie("so3_as_rpy", g["so3_from_rpy"]), g["so3_as_rpy"], **TOL)
    # the fixture does reach the branches it was built for
    w = T[:, 0]
    n2 = (T[:, 1:4] ** 2).sum(-1)
    assert (w < 0).any() and (n2 < 1e-10).any() and (np.abs(w) < 1e-3).any()
    t2 = (g["se3_log"][:, 3:] ** 2).sum(-1)
    assert (t2 < 1e-10).any() and (t2 > 9.0).any()


def test_host_classes_use_the_kernels_for_gpu_tensors(g):
    """The mink-named SE3 / SO3 classes on device tensors give the golden answers (they call
    mk_lie_batch) and keep a leading batch shape."""
    T, T2 = SE3(torch.tensor(g["T"], device=DEV)), SE3(torch.tensor(g["T2"], device=DEV))
    xi = torch.tensor(g["xi"], device=DEV)
    c = lambda t: t.cpu().numpy()  # noqa: E731
    np.testing.assert_allclose(c(T.log()), g["se3_log"], **TOL)
    np.testing.assert_allclose(c(T.jlog()), g["se3_jlog"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(c((T @ T2).wxyz_xyz), g["se3_multiply"], **TOL)
    np.testing.assert_allclose(c(T.minus(T2)), g["se3_rminus"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(c(T.plus(xi).wxyz_xyz), g["se3_rplus"], **TOL)
    np.testing.assert_allclose(c(T.inverse().wxyz_xyz), g["se3_inverse"], **TOL)
    np.testing.assert_allclose(c(T.adjoint()), g["se3_adjoint"], **TOL)
    np.testing.assert_allclose(c(T.as_matrix()), g["se3_matrix"], **TOL)
    np.testing.assert_allclose(c(SE3.exp(xi).wxyz_xyz), g["se3_exp"], **TOL)
    np.testing.assert_allclose(c(SE3.ljac(xi)), g["se3_ljac"], **TOL)
    np.testing.assert_allclose(c(T @ xi[:, :3]), g["se3_apply"], **TOL)
    R = T.rotation()
    np.testing.assert_allclose(c(R.log()), g["so3_log"], **TOL)
    np.testing.assert_allclose(c(R.as_matrix()), g["so3_matrix"], **TOL)
    np.testing.assert_allclose(c(SO3.exp(xi[:, 3:]).wxyz), g["so3_exp"], **TOL)
    # broadcasting of one element against a batch, and a 2-D batch shape
    one = SE3(torch.tensor(g["T"][3], device=DEV))
    np.testing.assert_allclose(c((one @ T2).wxyz_xyz), np.array([o.se3_multiply(g["T"][3], b) for b in g["T2"]]), **TOL)
    T55 = SE3(torch.tensor(g["T"][:54].reshape(6, 9, 7), device=DEV))
    assert T55.log().shape == (6, 9, 6) and T55.jlog().shape == (6, 9, 6, 6)
    np.testing.assert_allclose(c(T55.log()).reshape(54, 6), g["se3_log"][:54], **TOL)


def _unit(v):
    return v / np.linalg.norm(v)


def test_both_sides_of_every_switch_against_the_restatement():
    """Inputs placed just below and just above each Taylor / closed-form switch of
    mink/lie/so3.py:163,179,186,203,217 and mink/lie/se3.py:117,162,202,213,225."""
    rng = np.random.default_rng(5)
    quats, tangents = [], []
    for scale in (0.0, 1e-9, 0.99e-5, 1.01e-5, 1e-4, 1e-2):       # |v|^2 around 1e-10: so3 log switch
        ax = _unit(rng.normal(size=3))
        for sgn in (1.0, -1.0):
            q = np.concatenate([[sgn * np.sqrt(max(0.0, 1 - scale**2))], scale * ax])
            quats.append(q)
    for w in (0.0, 0.5e-10, -0.5e-10, 2e-10, -2e-10, 1e-6, -1e-6):  # |w| around 1e-10: the pi branch
        ax = _unit(rng.normal(size=3))
        quats.append(np.concatenate([[w], np.sqrt(1 - w * w) * ax]))
    for th in (0.0, 1e-12, 0.9e-10, 1.1e-10, 0.99e-5, 1.01e-5, 1e-3, 1.0, 3.0, np.pi - 1e-9, np.pi, 4.0, 6.0):
        tangents.append(np.concatenate([rng.uniform(-1, 1, 3), th * _unit(rng.normal(size=3))]))
    quats, tangents = np.array(quats), np.array(tangents)
    poses = np.concatenate([quats, rng.uniform(-1, 1, (len(quats), 3))], axis=1)
    np.testing.assert_allclose(lie("so3_log", quats), np.array([o.so3_log(q) for q in quats]), **TOL)
    np.testing.assert_allclose(lie("se3_log", poses), np.array([o.se3_log(p) for p in poses]), **TOL)
    np.testing.assert_allclose(lie("se3_jlog", poses).reshape(-1, 6, 6), np.array([o.se3_jlog(p) for p in poses]),
                               rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(lie("so3_exp", tangents[:, 3:]), np.array([o.so3_exp(t[3:]) for t in tangents]), **TOL)
    np.testing.assert_allclose(lie("se3_exp", tangents), np.array([o.se3_exp(t) for t in tangents]), **TOL)
    np.testing.assert_allclose(lie("se3_ljac", tangents).reshape(-1, 6, 6), np.array([o.se3_ljac(t) for t in tangents]),
                               rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(lie("se3_ljacinv", tangents).reshape(-1, 6, 6),
                               np.array([o.se3_ljacinv(t) for t in tangents]), rtol=1e-10, atol=1e-10)
    # exactly I6 below the theta^2 switch (mink/lie/se3.py:202-203, 213-214)
    small = (tangents[:, 3:] ** 2).sum(-1) < 1e-10
    assert small.sum() >= 4
    eye = np.eye(6).reshape(1, 36)
    assert np.array_equal(lie("se3_ljacinv", tangents[small]), np.repeat(eye, small.sum(), 0))
    assert np.array_equal(lie("se3_ljac", tangents[small]), np.repeat(eye, small.sum(), 0))
    assert np.array_equal(lie("se3_rjacinv", tangents[small]), np.repeat(eye, small.sum(), 0))


@pytest.mark.parametrize("G", [SO3, SE3])
def test_axioms_at_size(G):
    """Closure, identity, inverse, associativity, log/exp and plus/minus round trips and the
    jlog linearisation (reference: tests/test_lie_axioms.py:17-47, test_lie_operations.py:21-79)
    on 65536 uniformly sampled elements, all on the device."""
    n = 65536
    gen = torch.Generator(device="cpu").manual_seed(1)
    a, b, c3 = (G(G.sample_uniform((n,), gen).parameters().to(DEV)) for _ in range(3))
    k = 4

    def close(x, y, atol=1e-9):
        x, y = x.parameters(), y.parameters()
        s = torch.sign((x[:, :k] * y[:, :k]).sum(-1, keepdim=True))
        assert float((x[:, :k] * s - y[:, :k]).abs().max()) <= atol
        if x.shape[1] > k:
            assert float((x[:, k:] - y[:, k:]).abs().max()) <= atol

    ident = G.identity((n,), device=DEV)
    close(a @ ident, a)
    close(a @ a.inverse(), ident)
    close((a @ b) @ c3, a @ (b @ c3))
    close(a.inverse().inverse(), a, 1e-14)
    close(G.from_matrix(a.as_matrix()), a)
    close(G.exp(a.log()), a)
    close(a.rplus(b.rminus(a)), b)
    close(a.lplus(b.lminus(a)), b)
    n_q = torch.linalg.norm((a @ b).parameters()[:, :4], dim=-1)
    assert float((n_q - 1).abs().max()) <= 1e-12                      # closure: unit quaternions
    d = 1e-6 * torch.randn(n, G.tangent_dim, dtype=torch.float64, generator=gen).to(DEV)
    lhs = a.rplus(d).log()
    rhs = a.log() + (a.jlog() @ d[..., None])[..., 0]
    # away from the cut of the logarithm at pi, where a 1e-6 step can cross to the other sheet
    th = torch.linalg.norm(a.log()[:, -3:], dim=-1)
    ok = th < 3.1
    assert int(ok.sum()) > 0.9 * n
    assert float((lhs - rhs)[ok].abs().max()) <= 1e-7
    # adjoint identity T exp(x) T^-1 = exp(Ad_T x)
    x = 0.5 * torch.randn(n, G.tangent_dim, dtype=torch.float64, generator=gen).to(DEV)
    close(a @ G.exp(x) @ a.inverse(), G.exp((a.adjoint() @ x[..., None])[..., 0]))


def test_unknown_operation_is_refused():
    lib = L.load_library()
    a = torch.zeros((4, 8), dtype=torch.float64, device=DEV)
    assert lib.mk_lie_batch(31, a.data_ptr(), None, 4, a.data_ptr(), None) == L.MK_ERR_INVALID
    assert b"unknown Lie operation" in lib.mk_last_error()
