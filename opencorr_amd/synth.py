"""Synthetic speckle images / volumes with an analytically known deformation.

Workloads of SURVEY.md section 8(d): Gaussian speckles (R = 2 px, 0.02 speckles/px,
8-bit gray levels) in the reference image; the target image is the SAME speckle
field rendered under a known first-order (optionally second-order) displacement
field plus Gaussian noise, so FFTCC/ICGN results can be checked against the
analytic parameters as well as against the CPU oracle.

The renderer splats every speckle over a (2*halo+1)^d window; it runs with NumPy
on the host or with torch on a GPU (``device="cuda"``) for the 4k/8k bench images.
"""
import numpy as np

# displacement of SURVEY 8(d): u = 2.3 + 1e-3 x' + 5e-4 y', v = -1.7 - 5e-4 x' + 2e-3 y'
DEFAULT_WARP_2D = dict(u=2.3, ux=1e-3, uy=5e-4, v=-1.7, vx=-5e-4, vy=2e-3)
DEFAULT_WARP_3D = dict(u=1.6, ux=1e-3, uy=5e-4, uz=-4e-4, v=-2.2, vx=-5e-4, vy=2e-3, vz=3e-4,
                       w=0.9, wx=4e-4, wy=-6e-4, wz=-1.5e-3)


def _splat(shape, centers, amps, radius, halo, device=None, chunk=1 << 18):
    """sum_k amps[k] * exp(-|x - centers[k]|^2 / radius^2) on an integer grid of `shape`."""
    nd = len(shape)
    if device is not None:
        import torch
        img = torch.zeros(int(np.prod(shape)), dtype=torch.float32, device=device)
        offs = torch.arange(-halo, halo + 1, device=device)
        grids = torch.meshgrid(*([offs] * nd), indexing="ij")
        win = torch.stack([g.reshape(-1) for g in grids], dim=1)  # (K, nd)
        for s in range(0, len(amps), chunk):
            c = torch.as_tensor(centers[s:s + chunk], dtype=torch.float64, device=device)
            a = torch.as_tensor(amps[s:s + chunk], dtype=torch.float32, device=device)
            base = torch.round(c).to(torch.int64)
            pos = base[:, None, :] + win[None, :, :]  # (n, K, nd)
            d2 = ((pos.to(torch.float64) - c[:, None, :]) ** 2).sum(-1)
            val = a[:, None] * torch.exp(-d2 / (radius * radius)).to(torch.float32)
            ok = torch.ones(pos.shape[:2], dtype=torch.bool, device=device)
            lin = torch.zeros(pos.shape[:2], dtype=torch.int64, device=device)
            for ax in range(nd):
                ok &= (pos[..., ax] >= 0) & (pos[..., ax] < shape[ax])
                lin = lin * shape[ax] + pos[..., ax].clamp(0, shape[ax] - 1)
            img.index_add_(0, lin[ok], val[ok])
        return img.reshape(shape)
    img = np.zeros(int(np.prod(shape)), dtype=np.float64)
    offs = np.arange(-halo, halo + 1)
    grids = np.meshgrid(*([offs] * nd), indexing="ij")
    win = np.stack([g.reshape(-1) for g in grids], axis=1)
    chunk = min(chunk, 1 << 15)
    for s in range(0, len(amps), chunk):
        c = np.asarray(centers[s:s + chunk], dtype=np.float64)
        a = np.asarray(amps[s:s + chunk], dtype=np.float64)
        base = np.round(c).astype(np.int64)
        pos = base[:, None, :] + win[None, :, :]
        d2 = ((pos - c[:, None, :]) ** 2).sum(-1)
        val = a[:, None] * np.exp(-d2 / (radius * radius))
        ok = np.ones(pos.shape[:2], dtype=bool)
        lin = np.zeros(pos.shape[:2], dtype=np.int64)
        for ax in range(nd):
            ok &= (pos[..., ax] >= 0) & (pos[..., ax] < shape[ax])
            lin = lin * shape[ax] + np.clip(pos[..., ax], 0, shape[ax] - 1)
        img += np.bincount(lin[ok], weights=val[ok], minlength=img.size)
    return img.reshape(shape).astype(np.float32)


def _finish(img, noise_sigma, rng, device):
    """background + noise, clipped and rounded to 8-bit gray levels (like the reference's BMP fixtures)."""
    if device is not None:
        import torch
        out = img + 20.0
        if noise_sigma > 0:
            g = torch.Generator(device=device)
            g.manual_seed(int(rng.integers(1 << 31)))
            out = out + noise_sigma * torch.randn(out.shape, generator=g, device=device, dtype=torch.float32)
        return torch.round(out.clamp(0.0, 255.0)).contiguous()
    out = img.astype(np.float64) + 20.0
    if noise_sigma > 0:
        out = out + noise_sigma * rng.standard_normal(out.shape)
    return np.round(np.clip(out, 0.0, 255.0)).astype(np.float32)


def speckle_pair_2d(height, width, seed=20260925, warp=None, second_order=None, noise_sigma=1.0, radius=2.0,
                    density=0.02, device=None):
    """Returns (ref, tar) float32 images (NumPy, or torch tensors on `device`).

    The target shows the reference speckles displaced by
        u(x', y') = u + ux x' + uy y' (+ 0.5 uxx x'^2 + uxy x'y' + 0.5 uyy y'^2), same for v,
    with (x', y') measured from the image centre.  Speckle centres are moved by the
    field; blob shape change (strain ~1e-3) is neglected, which is far below the
    noise floor and irrelevant for parity tests (both paths see the same images).
    """
    warp = dict(DEFAULT_WARP_2D if warp is None else warp)
    so = dict(uxx=0.0, uxy=0.0, uyy=0.0, vxx=0.0, vxy=0.0, vyy=0.0)
    if second_order:
        so.update(second_order)
    rng = np.random.default_rng(seed)
    n = int(round(density * height * width))
    cx = rng.uniform(-8, width + 8, n)
    cy = rng.uniform(-8, height + 8, n)
    amps = rng.uniform(100.0, 200.0, n)
    xc, yc = (width - 1) * 0.5, (height - 1) * 0.5
    xp, yp = cx - xc, cy - yc
    du = warp["u"] + warp["ux"] * xp + warp["uy"] * yp + 0.5 * so["uxx"] * xp * xp + so["uxy"] * xp * yp + 0.5 * so["uyy"] * yp * yp
    dv = warp["v"] + warp["vx"] * xp + warp["vy"] * yp + 0.5 * so["vxx"] * xp * xp + so["vxy"] * xp * yp + 0.5 * so["vyy"] * yp * yp
    halo = int(np.ceil(3.5 * radius))
    ref = _splat((height, width), np.stack([cy, cx], 1), amps, radius, halo, device)
    tar = _splat((height, width), np.stack([cy + dv, cx + du], 1), amps, radius, halo, device)
    ref = _finish(ref, noise_sigma, np.random.default_rng(seed + 1), device)
    tar = _finish(tar, noise_sigma, np.random.default_rng(seed + 2), device)
    return ref, tar


def expected_deformation_2d(xs, ys, height, width, warp=None, second_order=None):
    """Analytic u, ux, uy, v, vx, vy at POI positions (first-order part of the field)."""
    warp = dict(DEFAULT_WARP_2D if warp is None else warp)
    so = dict(uxx=0.0, uxy=0.0, uyy=0.0, vxx=0.0, vxy=0.0, vyy=0.0)
    if second_order:
        so.update(second_order)
    xp = np.asarray(xs, dtype=np.float64) - (width - 1) * 0.5
    yp = np.asarray(ys, dtype=np.float64) - (height - 1) * 0.5
    u = warp["u"] + warp["ux"] * xp + warp["uy"] * yp + 0.5 * so["uxx"] * xp * xp + so["uxy"] * xp * yp + 0.5 * so["uyy"] * yp * yp
    v = warp["v"] + warp["vx"] * xp + warp["vy"] * yp + 0.5 * so["vxx"] * xp * xp + so["vxy"] * xp * yp + 0.5 * so["vyy"] * yp * yp
    return u, v


def speckle_pair_3d(dz, dy, dx, seed=20260927, warp=None, noise_sigma=1.0, radius=2.5, density=0.004, device=None):
    """3D analogue: Gaussian blobs, first-order (affine) displacement, volumes indexed [z, y, x]."""
    warp = dict(DEFAULT_WARP_3D if warp is None else warp)
    rng = np.random.default_rng(seed)
    n = int(round(density * dz * dy * dx))
    cx = rng.uniform(-8, dx + 8, n)
    cy = rng.uniform(-8, dy + 8, n)
    cz = rng.uniform(-8, dz + 8, n)
    amps = rng.uniform(100.0, 200.0, n)
    xp, yp, zp = cx - (dx - 1) * 0.5, cy - (dy - 1) * 0.5, cz - (dz - 1) * 0.5
    du = warp["u"] + warp["ux"] * xp + warp["uy"] * yp + warp["uz"] * zp
    dv = warp["v"] + warp["vx"] * xp + warp["vy"] * yp + warp["vz"] * zp
    dw = warp["w"] + warp["wx"] * xp + warp["wy"] * yp + warp["wz"] * zp
    halo = int(np.ceil(3.0 * radius))
    ref = _splat((dz, dy, dx), np.stack([cz, cy, cx], 1), amps, radius, halo, device, chunk=1 << 14)
    tar = _splat((dz, dy, dx), np.stack([cz + dw, cy + dv, cx + du], 1), amps, radius, halo, device, chunk=1 << 14)
    ref = _finish(ref, noise_sigma, np.random.default_rng(seed + 1), device)
    tar = _finish(tar, noise_sigma, np.random.default_rng(seed + 2), device)
    return ref, tar


def poi_grid_2d(height, width, nx, ny, margin):
    """Regular nx x ny grid of integer POI positions inside `margin` px of the border (row-major queue)."""
    xs = np.round(np.linspace(margin, width - 1 - margin, nx)).astype(np.float32)
    ys = np.round(np.linspace(margin, height - 1 - margin, ny)).astype(np.float32)
    gx, gy = np.meshgrid(xs, ys)
    return gx.ravel(), gy.ravel()


def poi_grid_3d(dz, dy, dx, nx, ny, nz, margin):
    xs = np.round(np.linspace(margin, dx - 1 - margin, nx)).astype(np.float32)
    ys = np.round(np.linspace(margin, dy - 1 - margin, ny)).astype(np.float32)
    zs = np.round(np.linspace(margin, dz - 1 - margin, nz)).astype(np.float32)
    gz, gy, gx = np.meshgrid(zs, ys, xs, indexing="ij")
    return gx.ravel(), gy.ravel(), gz.ravel()
