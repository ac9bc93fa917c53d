"""The oracle's FFTCC2D / FFTCC3D against an independent FFT (numpy's pocketfft, float64).

oracle/_ref compiles the reference's FFTCC loops against a STAND-IN fftw3.h, i.e. the same DFT restatement on both sides;
this file restates nothing of the oracle: windows are cut with numpy, the correlation surface is
irfftn(conj(rfftn(ref)) * rfftn(tar)), the peak is taken with the reference's first-maximum rule (src/oc_fftcc.cpp:246-262,
391-403).  Square, rectangular (the reference plans FFTW with (n0, n1) = (2rx, 2ry) over a buffer filled row by row -- restated
as-is, src/oc_fftcc.cpp:40-42 -- so a rectangular window is transformed as a 2rx x 2ry array) and cubic windows."""
import numpy as np
import pytest

import oracle
from oracle import P2, P3
from opencorr_amd import synth


def fftcc2d_numpy(ref, tar, rx, ry, x, y, gu, gv):
    sw, sh = 2 * rx, 2 * ry
    r0, c0 = int(np.float32(y) - ry), int(np.float32(x) - rx)
    rwin = ref[r0:r0 + sh, c0:c0 + sw].astype(np.float64)
    # target coordinates: (x + c - rx) + u truncated, per element (src/oc_fftcc.cpp:213-218)
    cols = (np.float32(x) + np.arange(sw, dtype=np.float32) - np.float32(rx) + np.float32(gu)).astype(np.int64)
    rows = (np.float32(y) + np.arange(sh, dtype=np.float32) - np.float32(ry) + np.float32(gv)).astype(np.int64)
    twin = tar[np.ix_(rows, cols)].astype(np.float64)
    rwin -= rwin.mean()
    twin -= twin.mean()
    # the buffer [r * sw + c] handed to a plan of dimensions (sw, sh)
    a = rwin.reshape(-1).reshape(sw, sh)
    b = twin.reshape(-1).reshape(sw, sh)
    surf = np.fft.irfftn(np.conj(np.fft.rfftn(a)) * np.fft.rfftn(b), s=a.shape, axes=(0, 1)).reshape(-1)
    idx = int(np.argmax(surf))          # numpy's argmax is the first maximum, like the reference's strict '>' scan
    du, dv = idx % sw, idx // sw
    if du > rx:
        du -= sw
    if dv > ry:
        dv -= sh
    zncc = surf[idx] / np.sqrt((rwin ** 2).sum() * (twin ** 2).sum())
    return du + gu, dv + gv, zncc, surf


@pytest.mark.parametrize("rx,ry", [(16, 16), (8, 8), (9, 9), (12, 20), (20, 12), (7, 15)])
def test_fftcc2d_against_numpy_fft(rx, ry):
    ref, tar = synth.speckle_pair_2d(220, 260, seed=100 + rx + 3 * ry)
    xs, ys = synth.poi_grid_2d(220, 260, 6, 5, max(rx, ry) + 8)
    pois = oracle.make_pois2d(xs, ys)
    rng = np.random.default_rng(rx * 31 + ry)
    pois[:, P2["u"]] = rng.integers(-2, 3, len(xs)).astype(np.float32)   # integer initial guesses displace the target window
    pois[:, P2["v"]] = rng.integers(-2, 3, len(xs)).astype(np.float32)
    guess = pois.copy()
    surf0 = oracle.fftcc2d(ref, tar, rx, ry, pois, want_surface=True)
    for i in range(len(xs)):
        u, v, zncc, surf = fftcc2d_numpy(ref, tar, rx, ry, xs[i], ys[i], guess[i, P2["u"]], guess[i, P2["v"]])
        assert pois[i, P2["u"]] == u and pois[i, P2["v"]] == v, (i, pois[i, P2["u"]], u, pois[i, P2["v"]], v)
        # the oracle forms means and norms like the reference: float32 running sums over the window (this side: float64)
        assert abs(pois[i, P2["zncc"]] - zncc) <= 1e-5
        if i == 0:   # the whole correlation surface of the first POI (unnormalised in the oracle, like FFTW's c2r)
            scale = np.abs(surf).max() * surf.size
            assert np.abs(surf0.reshape(-1) - surf * surf.size).max() <= 1e-6 * scale


def fftcc3d_numpy(ref, tar, r, x, y, z, g):
    n = 2 * r
    idx = [(np.float32(p) + np.arange(n, dtype=np.float32) - np.float32(r)).astype(np.int64) for p in (x, y, z)]
    tdx = [(np.float32(p) + np.arange(n, dtype=np.float32) - np.float32(r) + np.float32(gg)).astype(np.int64) for p, gg in zip((x, y, z), g)]
    rwin = ref[np.ix_(idx[2], idx[1], idx[0])].astype(np.float64)
    twin = tar[np.ix_(tdx[2], tdx[1], tdx[0])].astype(np.float64)
    rwin -= rwin.mean()
    twin -= twin.mean()
    surf = np.fft.irfftn(np.conj(np.fft.rfftn(rwin)) * np.fft.rfftn(twin), s=rwin.shape, axes=(0, 1, 2)).reshape(-1)
    k = int(np.argmax(surf))
    d = [k % n, (k // n) % n, k // (n * n)]
    d = [dd - n if dd > r else dd for dd in d]
    zncc = surf[k] / np.sqrt((rwin ** 2).sum() * (twin ** 2).sum())
    return [d[a] + g[a] for a in range(3)], zncc


@pytest.mark.parametrize("r", [8, 10, 16])
def test_fftcc3d_against_numpy_fft(r):
    dim = 4 * r + 24
    ref, tar = synth.speckle_pair_3d(dim, dim, dim, seed=300 + r)
    xs, ys, zs = synth.poi_grid_3d(dim, dim, dim, 2, 2, 2, r + 8)
    pois = oracle.make_pois3d(xs, ys, zs)
    rng = np.random.default_rng(r)
    for f in ("u", "v", "w"):
        pois[:, P3[f]] = rng.integers(-1, 2, len(xs)).astype(np.float32)
    guess = pois.copy()
    oracle.fftcc3d(ref, tar, r, r, r, pois)
    for i in range(len(xs)):
        g = [guess[i, P3["u"]], guess[i, P3["v"]], guess[i, P3["w"]]]
        d, zncc = fftcc3d_numpy(ref, tar, r, xs[i], ys[i], zs[i], g)
        assert [pois[i, P3["u"]], pois[i, P3["v"]], pois[i, P3["w"]]] == d
        assert abs(pois[i, P3["zncc"]] - zncc) <= 5e-5   # float32 running sums over 4 096 ... 32 768 voxels on the oracle's side
