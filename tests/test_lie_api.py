"""Host-API SO3 / SE3 against the reference's golden vectors and its own group axioms
(reference: tests/test_lie_operations.py, tests/test_lie_axioms.py)."""
import os

import numpy as np
import pytest
import torch

from mink_amd.lie import SE3, SO3, skew


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "lie.npz"))


def n(x):
    return x.detach().cpu().numpy()


def test_batched_matches_reference(g):
    T, T2, xi = SE3(g["T"]), SE3(g["T2"]), torch.tensor(g["xi"])
    tol = dict(rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(n(T.log()), g["se3_log"], **tol)
    np.testing.assert_allclose(n(T.jlog()), g["se3_jlog"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(n(T.inverse().wxyz_xyz), g["se3_inverse"], **tol)
    np.testing.assert_allclose(n(T.adjoint()), g["se3_adjoint"], **tol)
    np.testing.assert_allclose(n(T.as_matrix()), g["se3_matrix"], **tol)
    np.testing.assert_allclose(n((T @ T2).wxyz_xyz), g["se3_multiply"], **tol)
    np.testing.assert_allclose(n(T.rminus(T2)), g["se3_rminus"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(n(T.lminus(T2)), g["se3_lminus"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(n(SE3.exp(xi).wxyz_xyz), g["se3_exp"], **tol)
    np.testing.assert_allclose(n(T.rplus(xi).wxyz_xyz), g["se3_rplus"], **tol)
    np.testing.assert_allclose(n(T.lplus(xi).wxyz_xyz), g["se3_lplus"], **tol)
    np.testing.assert_allclose(n(SE3.ljac(xi)), g["se3_ljac"], **tol)
    np.testing.assert_allclose(n(SE3.ljacinv(xi)), g["se3_ljacinv"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(n(T.apply(xi[:, :3])), g["se3_apply"], **tol)
    R = T.rotation()
    np.testing.assert_allclose(n(R.log()), g["so3_log"], **tol)
    np.testing.assert_allclose(n(SO3.exp(xi[:, 3:]).wxyz), g["so3_exp"], **tol)
    np.testing.assert_allclose(n(SO3.ljac(xi[:, 3:])), g["so3_ljac"], **tol)
    np.testing.assert_allclose(n(SO3.ljacinv(xi[:, 3:])), g["so3_ljacinv"], **tol)
    np.testing.assert_allclose(n(R.as_matrix()), g["so3_matrix"], **tol)
    np.testing.assert_allclose(n(R.apply(xi[:, :3])), g["so3_apply"], **tol)
    q = n(SO3.from_matrix(R.as_matrix()).wxyz)
    ref = g["so3_from_matrix"]
    s = np.sign((q * ref).sum(-1, keepdims=True))
    np.testing.assert_allclose(q * s, ref, rtol=1e-9, atol=1e-9)
    rpy = torch.tensor(g["rpy"])
    Rr = SO3.from_rpy_radians(rpy[:, 0], rpy[:, 1], rpy[:, 2])
    np.testing.assert_allclose(n(Rr.wxyz), g["so3_from_rpy"], **tol)
    np.testing.assert_allclose(np.stack([n(a) for a in Rr.as_rpy_radians()], -1), g["so3_as_rpy"], **tol)


@pytest.mark.parametrize("G", [SO3, SE3])
def test_axioms_and_operations(G):
    gen = torch.Generator().manual_seed(0)
    a, b, c = (G.sample_uniform((32,), gen) for _ in range(3))
    p = lambda x: x.parameters()  # noqa: E731

    def close(x, y, atol=1e-9):
        x, y = n(p(x)), n(p(y))
        k = 4
        s = np.sign((x[:, :k] * y[:, :k]).sum(-1, keepdims=True))
        np.testing.assert_allclose(x[:, :k] * s, y[:, :k], atol=atol)
        np.testing.assert_allclose(x[:, k:], y[:, k:], atol=atol)

    ident = G.identity((32,))
    close(a @ ident, a)
    close(a @ a.inverse(), ident)
    close((a @ b) @ c, a @ (b @ c))
    close(a.inverse().inverse(), a)
    close(G.from_matrix(a.as_matrix()), a)
    close(G.exp(a.log()), a)
    close(a.rplus(b.rminus(a)), b)
    close(a.lplus(b.lminus(a)), b)
    # jlog linearisation (reference tests/test_lie_operations.py:74-79)
    d = 1e-6 * torch.randn(32, G.tangent_dim, dtype=torch.float64, generator=gen)
    lhs = a.rplus(d).log()
    rhs = a.log() + (a.jlog() @ d[..., None])[..., 0]
    np.testing.assert_allclose(n(lhs), n(rhs), atol=1e-7)


def test_skew_and_single_element():
    w = torch.tensor([1.0, 2.0, 3.0])
    S = skew(w)
    np.testing.assert_allclose(n(S + S.T), 0)
    with pytest.raises(AssertionError):
        skew(torch.zeros(4))
    T = SE3(np.array([1.0, 0, 0, 0, 0.3, 0.2, 0.1]))
    assert torch.equal(T.jlog(), torch.eye(6, dtype=torch.float64))   # exactly I6 below the switch
    with pytest.raises(ValueError):
        SO3(np.zeros(3))
    c = T.copy()
    c.wxyz_xyz[0] = 5.0
    assert T.wxyz_xyz[0] == 1.0
