"""ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the ReLU-Fields render hot path.

This file is NOT part of the product.  Only tests/, __graft_entry__.smoke() and the
``cpu_baseline`` leg of bench.py may import it; the product path (thr3ed_atom_amd) never does and
raises when its HIP library is missing.

It restates, function by function, the algorithm of the reference (akanimax/thr3ed_atom, paths
relative to /root/reference) in plain PyTorch-CPU tensor arithmetic.  The interpolation, scan and
transcendental arithmetic of the reference lives in a third-party dependency that is not vendored
(PyTorch ATen; the reference pins torch==1.11.0 at requirements.txt:2, this oracle is validated on
torch 2.10.0).  The ATen ``grid_sample`` algorithm (GridSampler.h:27-36 un-normalisation, 8-corner
accumulation order of grid_sampler_3d_cpu_impl) is restated explicitly in ``trilinear_recipe`` and
anchored on the reference's call sites (thre3d_reprs/voxels.py:292-322).

PARITY PINNING: the reference's own tests hold NO numeric vector for this path (they plot and
print a time).  The oracle is therefore pinned against outputs of the reference
itself: oracle/gen_golden.py imports /root/reference in the build container, runs the reference
functions and writes tests/golden/*.npz; tests/test_oracle_golden.py checks every function below
against those vectors (bit-exact where the recipe allows it).

All functions are dtype-generic: run them on float32 tensors for parity, on float64 tensors for
the "fp32 noise band" tolerance rule.
"""
from typing import Dict, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

ZERO_PLUS = 1e-10  # thre3d_atom/utils/constants.py:7
INFINITY = 1e10  # thre3d_atom/utils/constants.py:8

DENSITY_MODES = ("relu", "softplus", "abs", "identity")

# Real SH constants, PlenOctrees convention (rendering/volumetric/utils/spherical_harmonics.py:33-52)
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (
    -0.5900435899266435,
    2.890611442640554,
    -0.4570457994644658,
    0.3731763325901154,
    -0.4570457994644658,
    1.445305721320277,
    -0.5900435899266435,
)


# --------------------------------------------------------------------------------------------
# ray generation -- rendering/volumetric/utils/misc.py:12-50
# --------------------------------------------------------------------------------------------
def cast_rays(height: int, width: int, focal: float, rotation: Tensor, translation: Tensor) -> Tuple[Tensor, Tensor]:
    """Pixel-centre rays of a pinhole camera, always float32 (misc.py:31-32).

    Camera-space direction of pixel (row i, col j): ((j+.5 - W/2)/f, -(i+.5 - H/2)/f, -1); world
    direction d = R . dir (NOT normalised); origin = t for every pixel.  Returns [H, W, 3] x 2.
    The 3x3 product is written as explicit float32 multiply/adds in column order so that the HIP
    kernel can follow the same order."""
    rot = torch.as_tensor(rotation).to(torch.float32)
    trans = torch.as_tensor(translation).to(torch.float32).reshape(3)
    xs = torch.linspace(0.5, width - 0.5, width, dtype=torch.float32)
    ys = torch.linspace(0.5, height - 0.5, height, dtype=torch.float32)
    cx = ((xs - width * 0.5) / focal)[None, :].expand(height, width)
    cy = (-(ys - height * 0.5) / focal)[:, None].expand(height, width)
    cz = -torch.ones(height, width, dtype=torch.float32)
    dirs = torch.stack([rot[a, 0] * cx + rot[a, 1] * cy + rot[a, 2] * cz for a in range(3)], dim=-1)
    origins = trans.expand(height, width, 3)
    return origins, dirs


# --------------------------------------------------------------------------------------------
# sampling -- rendering/volumetric/sample.py
# --------------------------------------------------------------------------------------------
def sample_depths(
    num_rays: int,
    near,
    far,
    num_samples: int,
    t_rand: Optional[Tensor] = None,
    dtype=torch.float32,
) -> Tensor:
    """z[N, S] = near (1-t) + far t with t = linspace(0, 1, S)  (sample.py:39-54).
    ``near``/``far`` are scalars (CameraBounds) or [N, 1] tensors (per-ray bounds).
    With ``t_rand`` [N, S] (the reference draws torch.rand, sample.py:63) the samples are jittered
    inside the strata bounded by neighbouring mid-points (sample.py:57-64)."""
    near = torch.as_tensor(near, dtype=dtype).reshape(-1, 1).expand(num_rays, 1)
    far = torch.as_tensor(far, dtype=dtype).reshape(-1, 1).expand(num_rays, 1)
    t = torch.linspace(0.0, 1.0, num_samples, dtype=torch.float32).to(dtype)[None, :]
    z = near * (1.0 - t) + far * t
    if t_rand is not None:
        mids = 0.5 * (z[:, 1:] + z[:, :-1])
        upper = torch.cat([mids, z[:, -1:]], dim=-1)
        lower = torch.cat([z[:, :1], mids], dim=-1)
        z = lower + (upper - lower) * t_rand.to(dtype)
    return z


def ray_aabb_bounds(origins: Tensor, directions: Tensor, near: float, far: float, aabb) -> Tuple[Tensor, Tensor]:
    """Slab test of sample.py:71-184 ("optimized_sampling").

    Per axis: t = (plane - o) / (d + 1e-10), ordered so lo <= hi.  Axes are folded x, y, z with a
    running [max of lows, min of highs]; a ray "misses" when, at the time an axis is folded, the
    running interval and the axis interval are disjoint.  Missing rays get the camera (near, far).
    Finally everything is clamped to >= 0.  Hit intervals are NOT clipped to (near, far).
    Returns (bounds [N, 2], hit [N] bool)."""
    dtype = origins.dtype
    lo_run = hi_run = None
    hit = torch.ones(origins.shape[0], dtype=torch.bool)
    for axis, (pmin, pmax) in enumerate(aabb):
        den = directions[:, axis] + ZERO_PLUS
        ta = (pmin - origins[:, axis]) / den
        tb = (pmax - origins[:, axis]) / den
        swap = ta > tb
        lo = torch.where(swap, tb, ta)
        hi = torch.where(swap, ta, tb)
        if lo_run is None:
            lo_run, hi_run = lo, hi
            continue
        hit = hit & ~((lo_run > hi) | (lo > hi_run))
        lo_run = torch.where(lo > lo_run, lo, lo_run)
        hi_run = torch.where(hi < hi_run, hi, hi_run)
    near_t = torch.full_like(lo_run, near, dtype=dtype)
    far_t = torch.full_like(hi_run, far, dtype=dtype)
    lo_fin = torch.where(hit, lo_run, near_t).clamp(min=0.0)
    hi_fin = torch.where(hit, hi_run, far_t).clamp(min=0.0)
    return torch.stack([lo_fin, hi_fin], dim=-1), hit


# --------------------------------------------------------------------------------------------
# voxel grid -- thre3d_reprs/voxels.py
# --------------------------------------------------------------------------------------------
def make_aabb(grid_dims, voxel_size, grid_location=(0.0, 0.0, 0.0)):
    """centre +- dims * voxel / 2 per axis, in Python floats (voxels.py:187-212)."""
    out = []
    for n, v, c in zip(grid_dims, voxel_size, grid_location):
        half = (n * v) / 2
        out.append((c - half, c + half))
    return tuple(out)


def normalisation_constants(aabb) -> Tuple[Tuple[np.float32, np.float32], ...]:
    """float32 (scale, bias) per axis mapping the AABB to [-1, 1]
    (voxels.py:214-223 + utils/imaging_utils.py:58-63, slack=True branch)."""
    out = []
    for lo, hi in aabb:
        scale = (np.float32(1.0) - np.float32(-1.0)) / (np.float32(hi) - np.float32(lo))
        bias = np.float32(-1.0) - np.float32(lo) * scale
        out.append((np.float32(scale), np.float32(bias)))
    return tuple(out)


def normalise_points(points: Tensor, aabb) -> Tensor:
    consts = normalisation_constants(aabb)
    cols = [points[:, a] * float(consts[a][0]) + float(consts[a][1]) for a in range(3)]
    return torch.stack(cols, dim=-1)


def inside_aabb(points: Tensor, aabb) -> Tensor:
    """Strict inequalities on all three axes (voxels.py:252-274). [M] bool.
    The planes are compared in the dtype of the points (float32 planes for float32 points)."""
    m = torch.ones(points.shape[0], dtype=torch.bool)
    for a, (lo, hi) in enumerate(aabb):
        m = m & (points[:, a] > lo) & (points[:, a] < hi)
    return m


def trilinear_recipe(grid: Tensor, q: Tensor) -> Tensor:
    """Explicit restatement of ATen grid_sample(mode=bilinear/trilinear, padding=zeros,
    align_corners=False) for a channel-last grid[X, Y, Z, C] and normalised points q[M, 3] in
    (x, y, z) order -> [M, C].

    Continuous index per axis: i = ((q + 1) * size - 1) / 2 (GridSampler.h:27-36).  i0 = floor(i),
    i1 = i0 + 1; the weight of a corner is the product of the distances to the OPPOSITE corner,
    multiplied in (x, y, z) order; out starts at 0 and adds value * weight corner by corner in the
    order (dx,dy,dz) = 000,100,010,110,001,101,011,111, skipping corners outside the grid.
    Separate multiply and add (no fused multiply-add) -- on CPU this reproduces F.grid_sample bit
    for bit (tests/test_oracle_golden.py)."""
    X, Y, Z, C = grid.shape
    sizes = (X, Y, Z)
    idx = [((q[:, a] + 1.0) * sizes[a] - 1.0) / 2.0 for a in range(3)]
    i0 = [torch.floor(v) for v in idx]
    i1 = [v + 1.0 for v in i0]
    lo_w = [i1[a] - idx[a] for a in range(3)]  # weight of the lower corner along axis a
    hi_w = [idx[a] - i0[a] for a in range(3)]
    i0l = [v.long() for v in i0]
    flat = grid.reshape(-1, C)
    out = torch.zeros(q.shape[0], C, dtype=grid.dtype)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                ix, iy, iz = i0l[0] + dx, i0l[1] + dy, i0l[2] + dz
                wx = hi_w[0] if dx else lo_w[0]
                wy = hi_w[1] if dy else lo_w[1]
                wz = hi_w[2] if dz else lo_w[2]
                w = wx * wy * wz
                ok = (ix >= 0) & (ix < X) & (iy >= 0) & (iy < Y) & (iz >= 0) & (iz < Z)
                lin = (ix.clamp(0, X - 1) * Y + iy.clamp(0, Y - 1)) * Z + iz.clamp(0, Z - 1)
                contrib = flat[lin] * w[:, None]
                out = torch.where(ok[:, None], out + contrib, out)
    return out


def trilinear_aten(grid: Tensor, q: Tensor) -> Tensor:
    """The same interpolation through the ATen op the reference calls (voxels.py:296-303):
    used for the timed CPU baseline and to cross-check ``trilinear_recipe``."""
    vol = grid[None].permute(0, 4, 3, 2, 1)  # [1, C, Z, Y, X]
    out = torch.nn.functional.grid_sample(vol, q[None, None, None], align_corners=False)
    return out.permute(0, 2, 3, 4, 1).reshape(q.shape[0], grid.shape[-1])


def density_activation(raw_interp: Tensor, mode: str) -> Tensor:
    if mode == "relu":
        return torch.relu(raw_interp)
    if mode == "softplus":
        return torch.nn.functional.softplus(raw_interp)
    return raw_interp  # "abs" (pre-activation already applied) and "identity"


def voxel_grid_forward(
    densities: Tensor,
    features: Tensor,
    points: Tensor,
    aabb,
    density_scale: float,
    density_mode: str = "relu",
    interp: str = "recipe",
) -> Tensor:
    """VoxelGrid.forward (voxels.py:276-331): [M, 3] -> [M, F+1] = cat(features, density).

    density = post( interp( pre(D * rho) ) ): the scale rho multiplies the raw grid BEFORE
    interpolation and the non-linearity comes AFTER it -- that ordering is the ReLU field.
    mode "relu": pre = id, post = relu; "softplus": post = softplus; "abs": pre = |.|, post = id."""
    assert density_mode in DENSITY_MODES
    sampler = trilinear_recipe if interp == "recipe" else trilinear_aten
    q = normalise_points(points, aabb)
    pre = densities * density_scale
    if density_mode == "abs":
        pre = torch.abs(pre)
    sigma = density_activation(sampler(pre, q), density_mode)
    feats = sampler(features, q)
    return torch.cat([feats, sigma], dim=-1)


# --------------------------------------------------------------------------------------------
# spherical harmonics -- rendering/volumetric/utils/spherical_harmonics.py:64-116
# --------------------------------------------------------------------------------------------
def sh_basis(degree: int, v: Tensor) -> Tensor:
    """Signed real-SH basis Y_k(v) [.., K] for unit directions v[.., 3]; each entry is built with
    the reference's operation order ((constant * polynomial), sign folded in)."""
    assert 0 <= degree < 4, "only degrees 0..3 are supported"
    x, y, z = v[..., 0], v[..., 1], v[..., 2]
    cols = [torch.full_like(x, SH_C0)]
    if degree > 0:
        cols += [-(SH_C1 * y), SH_C1 * z, -(SH_C1 * x)]
    if degree > 1:
        xx, yy, zz = x * x, y * y, z * z
        xy, yz, xz = x * y, y * z, x * z
        cols += [
            SH_C2[0] * xy,
            SH_C2[1] * yz,
            SH_C2[2] * (2.0 * zz - xx - yy),
            SH_C2[3] * xz,
            SH_C2[4] * (xx - yy),
        ]
    if degree > 2:
        cols += [
            SH_C3[0] * y * (3 * xx - yy),
            SH_C3[1] * xy * z,
            SH_C3[2] * y * (4 * zz - xx - yy),
            SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy),
            SH_C3[4] * x * (4 * zz - xx - yy),
            SH_C3[5] * z * (xx - yy),
            SH_C3[6] * x * (xx - 3 * yy),
        ]
    return torch.stack(cols, dim=-1)


def evaluate_sh(degree: int, coeffs: Tensor, v: Tensor) -> Tensor:
    """coeffs[M, 3, K], unit v[M, 3] -> raw radiance [M, 3]: sum_k Y_k c_k accumulated in k order."""
    basis = sh_basis(degree, v)
    out = basis[:, None, 0] * coeffs[..., 0]
    for k in range(1, (degree + 1) ** 2):
        out = out + basis[:, None, k] * coeffs[..., k]
    return out


# --------------------------------------------------------------------------------------------
# point processing -- rendering/volumetric/process.py:20-96
# --------------------------------------------------------------------------------------------
def process_points(
    points: Tensor,  # [N, S, 3]
    directions: Tensor,  # [N, 3]
    densities: Tensor,
    features: Tensor,
    aabb,
    density_scale: float,
    density_mode: str = "relu",
    render_diffuse: bool = False,
    interp: str = "recipe",
) -> Tensor:
    """-> [N, S, 4] = (raw radiance rgb, sigma).  Feature layout is channel-major: feature index =
    colour * K + k (process.py:61,66).  ``render_diffuse`` keeps k = 0 only (process.py:59-63).
    Samples outside the AABB get radiance -1e10 (sigmoid -> exactly 0) and sigma 0 (process.py:80-84)."""
    N, S, _ = points.shape
    flat = points.reshape(-1, 3)
    interp_out = voxel_grid_forward(densities, features, flat, aabb, density_scale, density_mode, interp)
    coeffs = interp_out[:, :-1].reshape(flat.shape[0], 3, -1)
    sigma = interp_out[:, -1:]
    K = coeffs.shape[-1]
    if render_diffuse:
        coeffs, degree = coeffs[..., :1], 0
    else:
        degree = int(np.sqrt(K)) - 1
    v = directions / directions.norm(dim=-1, keepdim=True)
    v = v[:, None, :].expand(N, S, 3).reshape(-1, 3)
    radiance = evaluate_sh(degree, coeffs, v)
    inside = inside_aabb(flat, aabb)[:, None]
    radiance = torch.where(inside, radiance, torch.full_like(radiance, -INFINITY))
    sigma = torch.where(inside, sigma, torch.zeros_like(sigma))
    return torch.cat([radiance, sigma], dim=-1).reshape(N, S, 4)


# --------------------------------------------------------------------------------------------
# compositing -- rendering/volumetric/accumulate.py:31-113
# --------------------------------------------------------------------------------------------
def accumulate(processed: Tensor, depths: Tensor, directions: Tensor, white_bkgd: bool, density_noise: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """Front-to-back alpha compositing of processed[N, S, 4] at ray parameters depths[N, S].
    ``density_noise`` [N, S] = the reference's ``torch.randn(raw_density.shape) * stochastic_density_noise_std``
    (accumulate.py:58-62), added to the (activated, masked) density of EVERY sample before density2occupancy.

    delta_i = (z_{i+1} - z_i) |d|, delta_last = 1e10 |d|; alpha = 1 - exp(-sigma delta);
    T_i = prod_{j<i} (1 - alpha_j) (exclusive cumulative product, no epsilon); w = alpha T;
    colour = sum w sigmoid(R) (+ 1 - acc on white); acc = sum w; depth = sum w z (z is the
    parameter along the UN-normalised direction); disparity = 1 / max(1e-10, depth / acc), which is
    NaN for rays with acc == 0 (0/0 propagates through torch.maximum)."""
    radiance, sigma = processed[..., :3], processed[..., 3]
    dnorm = directions.norm(dim=-1, keepdim=True)
    gaps = depths[:, 1:] - depths[:, :-1]
    gaps = torch.cat([gaps, torch.full_like(depths[:, :1], INFINITY)], dim=-1) * dnorm
    if density_noise is not None:
        sigma = sigma + density_noise.to(sigma.dtype)
    alpha = 1.0 - torch.exp(-(sigma * gaps))
    ones = torch.ones_like(alpha[:, :1])
    trans = torch.cumprod(torch.cat([ones, 1.0 - alpha], dim=-1), dim=-1)[:, :-1]
    weights = alpha * trans
    colour = (torch.sigmoid(radiance) * weights[..., None]).sum(dim=-2)
    acc = weights.sum(dim=-1, keepdim=True)
    if white_bkgd:
        colour = colour + (1 - acc)
    depth = (depths * weights).sum(dim=-1, keepdim=True)
    disparity = 1.0 / torch.maximum(torch.full_like(acc, ZERO_PLUS), depth / acc)
    return {
        "colour": colour,
        "depth": depth,
        "acc": acc,
        "disparity": disparity,
        "alpha": alpha,
        "weights": weights,
        "deltas": gaps,
    }


# --------------------------------------------------------------------------------------------
# the composed procedure -- thre3d_reprs/renderers.py:48-102 + render_interface.py:103-134
# --------------------------------------------------------------------------------------------
def render(
    densities: Tensor,  # [X, Y, Z, 1]
    features: Tensor,  # [X, Y, Z, 3K]
    origins: Tensor,  # [N, 3]
    directions: Tensor,  # [N, 3]
    aabb,
    near: float,
    far: float,
    num_samples: int,
    density_scale: float = 1.0,
    density_mode: str = "relu",
    white_bkgd: bool = False,
    render_diffuse: bool = False,
    optimized_sampling: bool = False,
    t_rand: Optional[Tensor] = None,
    interp: str = "recipe",
    density_noise: Optional[Tensor] = None,
) -> Dict[str, Tensor]:
    """sample -> interpolate (+ReLU) -> SH -> mask -> composite for flat rays.  The dtype of
    ``origins`` selects the arithmetic (float32 / float64); grids are cast to it."""
    dtype = origins.dtype
    densities, features, directions = densities.to(dtype), features.to(dtype), directions.to(dtype)
    n = origins.shape[0]
    if optimized_sampling:
        bounds, _ = ray_aabb_bounds(origins, directions, near, far, aabb)
        z = sample_depths(n, bounds[:, :1], bounds[:, 1:], num_samples, t_rand, dtype)
    else:
        z = sample_depths(n, near, far, num_samples, t_rand, dtype)
    pts = origins[:, None, :] + directions[:, None, :] * z[:, :, None]
    processed = process_points(
        pts, directions, densities, features, aabb, density_scale, density_mode, render_diffuse, interp
    )
    out = accumulate(processed, z, directions, white_bkgd, density_noise)
    out["processed"] = processed
    out["z"] = z
    return out


def trilinear_upsample(volume: Tensor, output_size) -> Tensor:
    """[X, Y, Z, C] -> [X', Y', Z', C] by F.interpolate(trilinear, align_corners=False)
    (scale_voxel_grid_with_required_output_size, voxels.py:334-373)."""
    up = torch.nn.functional.interpolate(
        volume.permute(3, 0, 1, 2)[None],
        size=tuple(output_size),
        mode="trilinear",
        align_corners=False,
        recompute_scale_factor=False,
    )[0]
    return up.permute(1, 2, 3, 0)


def trilinear_upsample_recipe(volume, output_size, vector_width: int = 8):
    """The arithmetic behind ``trilinear_upsample`` spelled out in numpy (what rf_upsample_grid implements), pinned by the tests
    to F.interpolate on the CPU and to golden G3 bit for bit.  The reference hands ATen a CHANNELS-LAST volume
    (``unified.permute(3, 0, 1, 2)[None]``, voxels.py:352-361), so aten/src/ATen/native/cpu/UpSampleKernel.cpp
    (cpu_upsample_linear_channels_last, 3-d loop) applies:
      scale = float(in) / float(out);  src = fma(scale, dst + 0.5, -0.5) clamped at 0 (area_pixel_compute_source_index,
      contracted);  i0 = min(floor(src), in - 1);  i1 = i0 + (i0 < in - 1);  lambda1 = clamp(src - i0, 0, 1);  lambda0 = 1 - lambda1;
      corner weights w_dhw = (lambda_d * lambda_h) * lambda_w;
      the first C - C % 8 channels (8-wide vector body: interpolate(t, w, args...) = t * w + interpolate(args...)) are summed from
      the LAST corner to the first, the remaining channels (scalar tail) from the first to the last, every step a fused
      multiply-add (the innermost product of each chain is rounded on its own)."""
    v = np.asarray(volume, dtype=np.float32)
    f32 = np.float32

    def fma(a, b, c):  # exact: the product of two float32 values fits a float64
        return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(np.float32)

    def mul(a, b):
        return (a * b).astype(np.float32)

    def axis(out, size):
        scale = f32(size) / f32(out)
        dst = np.arange(out, dtype=np.float32)
        src = fma(scale, (dst + f32(0.5)).astype(np.float32), f32(-0.5))
        src = np.where(src < 0, f32(0), src).astype(np.float32)
        i0 = np.minimum(np.floor(src).astype(np.int64), size - 1)
        i1 = i0 + (i0 < size - 1)
        lam = np.clip((src - i0.astype(np.float32)).astype(np.float32), 0, 1).astype(np.float32)
        return (i0, (f32(1) - lam).astype(np.float32)), (i1, lam)

    ax, ay, az = (axis(o, n) for o, n in zip(output_size, v.shape[:3]))
    vals, wts = [], []
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                (ix, wx), (iy, wy), (iz, wz) = ax[dx], ay[dy], az[dz]
                vals.append(v[ix[:, None, None], iy[None, :, None], iz[None, None, :]])
                wts.append(mul(mul(wx[:, None, None], wy[None, :, None]), wz[None, None, :])[..., None])
    body = fma(vals[7], wts[7], mul(vals[6], wts[6]))
    for k in range(5, -1, -1):
        body = fma(vals[k], wts[k], body)
    tail = fma(vals[0], wts[0], mul(vals[1], wts[1]))
    for k in range(2, 8):
        tail = fma(vals[k], wts[k], tail)
    nv = v.shape[-1] - v.shape[-1] % vector_width
    return np.concatenate([body[..., :nv], tail[..., nv:]], axis=-1)


# --------------------------------------------------------------------------------------------
# checker for the build's own batch selection (rf_select_rays_and_pixels) -- not reference behaviour
# --------------------------------------------------------------------------------------------
def keyed_permutation(index: np.ndarray, domain: int, key: int) -> np.ndarray:
    """numpy restatement of the kernel's keyed bijection of [0, domain): 4-round Feistel network on
    ceil(log2(domain)) bits (at least 2) with cycle walking."""
    bits = 2
    while (1 << bits) < domain:
        bits += 1
    hr = bits // 2
    hl = bits - hr
    k0, k1 = key & 0xFFFFFFFF, (key >> 32) & 0xFFFFFFFF
    M32 = np.uint64(0xFFFFFFFF)

    def mix32(x):
        x = x & M32
        x ^= x >> np.uint64(16)
        x = (x * np.uint64(0x7FEB352D)) & M32
        x ^= x >> np.uint64(15)
        x = (x * np.uint64(0x846CA68B)) & M32
        x ^= x >> np.uint64(16)
        return x

    x = np.asarray(index, dtype=np.uint64).copy()
    todo = np.ones(x.shape, dtype=bool)
    while todo.any():
        xs = x[todo]
        L, R = xs >> np.uint64(hr), xs & np.uint64((1 << hr) - 1)
        for rnd in range(2):
            c1 = np.uint64((0x9E3779B9 * (2 * rnd + 1)) & 0xFFFFFFFF)
            c2 = np.uint64((0x85EBCA6B * (2 * rnd + 2)) & 0xFFFFFFFF)
            L = L ^ (mix32(R ^ np.uint64(k0) ^ c1) & np.uint64((1 << hl) - 1))
            R = R ^ (mix32(L ^ np.uint64(k1) ^ c2) & np.uint64((1 << hr) - 1))
        xs = (L << np.uint64(hr)) | R
        x[todo] = xs
        todo[todo] = xs >= np.uint64(domain)
    return x.astype(np.int64)


def keyed_jitter(key: int, first_ray: int, num_rays: int, num_samples: int) -> np.ndarray:
    """numpy restatement of the kernels' counter-based jitter (RF_FLAG_JITTER_KEYED; jitter_ray_seed / jitter_uniform in
    csrc/relu_field_kernels.hip): [num_rays, num_samples] float32 in [0, 1) on the 2^-24 lattice -- the table a caller would
    pass as ``t_rand`` to get the same render.  Not reference behaviour: the reference draws torch.rand(N, S) (sample.py:63);
    this is the checker of the build's replacement for that draw."""
    M32 = np.uint64(0xFFFFFFFF)

    def mix32(x):
        x = x & M32
        x ^= x >> np.uint64(16)
        x = (x * np.uint64(0x7FEB352D)) & M32
        x ^= x >> np.uint64(15)
        x = (x * np.uint64(0x846CA68B)) & M32
        x ^= x >> np.uint64(16)
        return x

    key = int(key) & 0xFFFFFFFFFFFFFFFF
    klo, khi = np.uint64(key & 0xFFFFFFFF), np.uint64(key >> 32)
    gray = np.arange(num_rays, dtype=np.uint64) + np.uint64(first_ray)
    a = mix32((gray & M32) ^ klo)
    b = mix32(((gray >> np.uint64(32)) + khi) & M32)
    seed = a ^ ((b * np.uint64(0x9E3779B9) + np.uint64(0x85EBCA6B)) & M32)
    s = np.arange(num_samples, dtype=np.uint64)
    h = mix32((seed[:, None] + ((s * np.uint64(0x9E3779B9)) & M32)[None, :]) & M32)
    return ((h >> np.uint64(8)).astype(np.float32) * np.float32(2.0**-24)).astype(np.float32)
