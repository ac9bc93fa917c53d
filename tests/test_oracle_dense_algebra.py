"""The oracle's small dense algebra against LAPACK (numpy, float64).

The compiled reference of oracle/_ref links a STAND-IN Eigen (oracle/ref_stubs/): its inverse() is the same restatement as the
oracle's, so that comparison cannot say whether the restatement inverts matrices.  numpy's LAPACK can: on Hessians of the
solvers' own kind (Gram matrices of steepest-descent images with the r, r^2 scaling of 6-, 12- and 12-(3D)-parameter shape
functions), on matrices that need every pivot exchange, and on warp-increment matrices close to the identity
(src/oc_icgn.cpp:210,290,759,831,1339,1439)."""
import numpy as np
import pytest

import oracle


def sd_hessian(dof, r, rng, n=None):
    """H = sum sd sd^T for the reference's shape functions on a (2r+1)^2 or (2r+1)^3 subset with random gradients."""
    if dof in (6, 12):
        y, x = np.mgrid[-r:r + 1, -r:r + 1].astype(np.float64)
        x, y = x.ravel(), y.ravel()
        gx, gy = rng.normal(0, 20, x.size), rng.normal(0, 20, x.size)
        if dof == 6:
            sd = np.stack([gx, gx * x, gx * y, gy, gy * x, gy * y])
        else:
            m = [np.ones_like(x), x, y, 0.5 * x * x, x * y, 0.5 * y * y]
            sd = np.stack([gx * k for k in m] + [gy * k for k in m])
    else:  # 3D, 12 parameters
        z, y, x = np.mgrid[-r:r + 1, -r:r + 1, -r:r + 1].astype(np.float64)
        x, y, z = x.ravel(), y.ravel(), z.ravel()
        g = [rng.normal(0, 20, x.size) for _ in range(3)]
        sd = np.stack([gk * k for gk in g for k in (np.ones_like(x), x, y, z)])
    return (sd @ sd.T).astype(np.float32)


def check_inverse(a, slack=64.0):
    a = np.asarray(a, np.float32)
    inv = oracle.inverse(a)
    a64 = a.astype(np.float64)
    ref = np.linalg.inv(a64)
    cond = np.linalg.cond(a64)
    eps = np.finfo(np.float32).eps
    n = a.shape[0]
    # forward error of a backward-stable solve in float32: <= c(n) * cond * eps, relative to the inverse's norm
    err = np.linalg.norm(inv.astype(np.float64) - ref) / np.linalg.norm(ref)
    assert err <= slack * n * cond * eps, (err, cond)
    # and it IS an inverse: residual relative to |A| |A^-1|
    res = np.linalg.norm(a64 @ inv.astype(np.float64) - np.eye(n)) / (np.linalg.norm(a64) * np.linalg.norm(ref))
    assert res <= slack * n * eps, res
    return err, cond


@pytest.mark.parametrize("dof,r", [(6, 8), (6, 16), (6, 30), (12, 12), (12, 20), ("3d", 8), ("3d", 16)])
def test_hessian_inverse_against_lapack(dof, r):
    rng = np.random.default_rng(1000 + r)
    for _ in range(5):
        h = sd_hessian(dof, r, rng)
        assert h.shape[0] in (6, 12)
        err, cond = check_inverse(h)
        assert cond > 10.0   # the r^2 ... r^4 scaling of the shape functions makes these matrices badly scaled: a real test


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 9, 12])
def test_random_matrices_and_pivoting(n):
    rng = np.random.default_rng(n)
    for _ in range(20):
        check_inverse(rng.normal(0, 1, (n, n)).astype(np.float32) + 0.1 * np.eye(n, dtype=np.float32), slack=256.0)
    # every column's largest entry sits below the diagonal: each elimination step has to exchange rows
    a = np.eye(n, dtype=np.float32)[::-1].copy() * 3.0 + rng.normal(0, 0.2, (n, n)).astype(np.float32)
    check_inverse(a, slack=256.0)
    # zero diagonal (a permutation matrix): unsolvable without pivoting
    p = np.roll(np.eye(n, dtype=np.float32), 1, axis=0) if n > 1 else np.eye(1, dtype=np.float32)
    assert np.array_equal(oracle.inverse(p), p.T)


def test_warp_increment_inverse_and_product():
    """W <- W * (dW)^-1 (src/oc_icgn.cpp:287-290, 828-831, 1436-1439): dW is the identity plus a small increment."""
    rng = np.random.default_rng(7)
    for n in (3, 4, 6):
        for _ in range(10):
            dw = (np.eye(n) + rng.normal(0, 1e-3, (n, n))).astype(np.float32)
            dw[-1, :] = 0.0
            dw[-1, -1] = 1.0
            w = (np.eye(n) + rng.normal(0, 1e-2, (n, n))).astype(np.float32)
            inv = oracle.inverse(dw)
            check_inverse(dw)
            got = oracle.mat_mul(w, inv)
            want = w.astype(np.float64) @ np.linalg.inv(dw.astype(np.float64))
            assert np.abs(got - want).max() <= 1e-6


def test_unsupported_size():
    with pytest.raises(ValueError):
        oracle.inverse(np.eye(13, dtype=np.float32))
