"""The render path at the granularity of the reference's plug-in points -- sampler, point processor, accumulator
(``RaySamplerFunction`` / ``PointProcessorFunction`` / ``AccumulatorFunction``, thre3d_atom/rendering/volumetric/render_interface.py:93-134)
-- for everything the fused kernels do not implement:

  * a ``density2occupancy`` / ``radiance_hdr_tone_map`` other than the defaults (thre3d_reprs/renderers.py:37-38),
  * ``stochastic_density_noise_std != 0`` (rendering/volumetric/accumulate.py:58-62; drawn with torch.randn like the reference, so the
    reference's own behaviour -- non-finite renders wherever sigma + noise < 0 on the last sample -- is reproduced, not "fixed"),
  * density / feature activations outside the kernels' four pairs, a radiance transfer function (thre3d_reprs/voxels.py:292-331),
  * the per-sample debug outputs of the accumulator (accumulate.py:96-107).

It is the reference's composition (render_interface.py:103-134) with the HIP interpolation in the middle: the grid is gathered by
rf_grid_query (the bit-exact grid_sample recipe, differentiable), everything around it is element-wise torch on the device, through
autograd.  Slower than the fused kernels by the per-sample tensors it materialises -- the price of arbitrary Python callables --, still
GPU-only: CPU tensors raise.  ``render_sh_voxel_grid`` routes here by itself when a configuration needs it.
"""
from typing import Callable, NamedTuple, Optional, Union

import numpy as np
import torch
from torch import Tensor

from .camera import CameraBounds
from .constants import (
    EXTRA_ACCUMULATED_WEIGHTS,
    EXTRA_DISPARITY,
    EXTRA_POINT_DENSITIES,
    EXTRA_POINT_DEPTHS,
    EXTRA_POINT_OCCUPANCIES,
    EXTRA_POINT_WEIGHTS,
    EXTRA_SAMPLE_INTERVALS,
    INFINITY,
    ZERO_PLUS,
)
from .render_interface import Rays, RenderOut


class SampledPointsOnRays(NamedTuple):
    points: Tensor  # [N, S, 3] for sampled points, [N, S, 4] = (raw r, g, b, density) once processed
    depths: Tensor  # [N, S] ray parameters


ProcessedPointsOnRays = SampledPointsOnRays
RaySamplerFunction = Callable[[Rays, CameraBounds, int], SampledPointsOnRays]
PointProcessorFunction = Callable[[SampledPointsOnRays, Rays], ProcessedPointsOnRays]
AccumulatorFunction = Callable[[ProcessedPointsOnRays, Rays], RenderOut]


def _require_device(t: Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what} must live on a HIP device (got {t.device}); the render path has no CPU fallback")


def render(rays: Rays, camera_bounds: CameraBounds, num_samples: int, sampler_fn: RaySamplerFunction, point_processor_fn: PointProcessorFunction,
           accumulator_fn: AccumulatorFunction) -> RenderOut:
    """sampler -> point processor -> accumulator on FLAT rays (render_interface.py:103-134)."""
    if rays.origins.dim() != 2 or rays.directions.dim() != 2:
        raise AssertionError("the render interface only works with FLAT rays")
    return accumulator_fn(point_processor_fn(sampler_fn(rays, camera_bounds, num_samples), rays), rays)


# ---------------------------------------------------------------------------------------------------------------------------
# samplers (rendering/volumetric/sample.py:15-68, 187-202)
# ---------------------------------------------------------------------------------------------------------------------------
def sample_uniform_points_on_rays(rays: Rays, bounds: Union[CameraBounds, Tensor], num_samples: int, perturb: bool = True,
                                  linear_disparity_sampling: bool = False, t_rand: Optional[Tensor] = None) -> SampledPointsOnRays:
    """``num_samples`` ray parameters per ray between the bounds -- uniform in depth, or in inverse depth -- optionally jittered inside
    their strata, and the points there.  ``bounds``: camera (near, far) or a per-ray [N, 2] tensor.  ``t_rand`` [N, S] replaces the
    torch.rand draw (tests)."""
    from .ops import t_vals_for

    o = rays.origins.reshape(-1, rays.origins.shape[-1])
    d = rays.directions.reshape(-1, rays.directions.shape[-1])
    _require_device(o, "ray origins")
    if isinstance(bounds, Tensor):
        near, far = bounds[:, :1].to(o.dtype), bounds[:, 1:].to(o.dtype)
    else:
        near = torch.full((o.shape[0], 1), float(np.float32(bounds.near)), dtype=o.dtype, device=o.device)
        far = torch.full((o.shape[0], 1), float(np.float32(bounds.far)), dtype=o.dtype, device=o.device)
    t = t_vals_for(num_samples, o.device).to(o.dtype)[None, :]  # linspace evaluated on the host (what the reference's CPU path sees)
    if linear_disparity_sampling:
        z = 1.0 / (1.0 / (near + ZERO_PLUS) * (1.0 - t) + 1.0 / far * t)
    else:
        z = near * (1.0 - t) + far * t
    if perturb:
        mid = 0.5 * (z[:, 1:] + z[:, :-1])
        hi = torch.cat([mid, z[:, -1:]], dim=-1)
        lo = torch.cat([z[:, :1], mid], dim=-1)
        u = torch.rand(z.shape, dtype=o.dtype, device=o.device) if t_rand is None else t_rand.to(o.dtype)
        z = lo + (hi - lo) * u
    return SampledPointsOnRays(o[:, None, :] + d[:, None, :] * z[:, :, None], z)


def sample_aabb_bound_uniform_points_on_rays(rays: Rays, bounds: CameraBounds, num_samples: int, aabb, perturb: bool = True,
                                             t_rand: Optional[Tensor] = None) -> SampledPointsOnRays:
    """the uniform sampler between each ray's entry and exit of the box (rf_ray_aabb_bounds: the slab test of sample.py:71-184; rays
    that miss keep the camera bounds)"""
    from .ops import ray_aabb_bounds_hip

    per_ray, _ = ray_aabb_bounds_hip(rays.origins, rays.directions, float(np.float32(bounds.near)), float(np.float32(bounds.far)), aabb)
    return sample_uniform_points_on_rays(rays, per_ray, num_samples, perturb=perturb, t_rand=t_rand)


# ---------------------------------------------------------------------------------------------------------------------------
# point processor (rendering/volumetric/process.py:20-96, utils/spherical_harmonics.py:64-116)
# ---------------------------------------------------------------------------------------------------------------------------
_SH_C0 = 0.28209479177387814
_SH_C1 = 0.4886025119029199
_SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658, 1.445305721320277, -0.5900435899266435)


def sh_basis(degree: int, viewdirs: Tensor) -> Tensor:
    """[M, (degree + 1)^2] signed real SH basis values at unit directions [M, 3] (the constants and signs of the reference's
    evaluate_spherical_harmonics; radiance = sum_k coeffs[..., k] * basis[..., k])."""
    if not 0 <= degree < 4:
        raise AssertionError(f"SH degree must be 0..3, got {degree}")
    x, y, z = viewdirs.unbind(-1)
    cols = [torch.full_like(x, _SH_C0)]
    if degree >= 1:
        cols += [-_SH_C1 * y, _SH_C1 * z, -_SH_C1 * x]
    if degree >= 2:
        xx, yy, zz = x * x, y * y, z * z
        cols += [_SH_C2[0] * (x * y), _SH_C2[1] * (y * z), _SH_C2[2] * (2.0 * zz - xx - yy), _SH_C2[3] * (x * z), _SH_C2[4] * (xx - yy)]
        if degree >= 3:
            cols += [_SH_C3[0] * y * (3.0 * xx - yy), _SH_C3[1] * (x * y) * z, _SH_C3[2] * y * (4.0 * zz - xx - yy), _SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy),
                     _SH_C3[4] * x * (4.0 * zz - xx - yy), _SH_C3[5] * z * (xx - yy), _SH_C3[6] * x * (xx - 3.0 * yy)]
    return torch.stack(cols, dim=-1)


def process_points_with_sh_voxel_grid(sampled_points: SampledPointsOnRays, rays: Rays, voxel_grid, render_diffuse: bool = False,
                                      parallel_points_chunk_size: Optional[int] = None) -> ProcessedPointsOnRays:
    """[N, S, 3] points -> [N, S, 4] = (raw radiance of the ray's viewing direction, density): the grid's forward at every point (any
    module mapping [M, 3] -> [M, 3 K + 1], channel-major SH coefficients: index = colour * K + k), SH evaluation, and the box mask
    (outside: radiance -1e10, which every sigmoid-like tone map sends to 0, and zero density)."""
    n, s, _ = sampled_points.points.shape
    flat = sampled_points.points.reshape(-1, 3)
    if parallel_points_chunk_size is None:
        interp = voxel_grid(flat)
    else:
        interp = torch.cat([voxel_grid(flat[i : i + parallel_points_chunk_size]) for i in range(0, flat.shape[0], parallel_points_chunk_size)], dim=0)
    coeffs = interp[:, :-1].reshape(flat.shape[0], 3, -1)
    density = interp[:, -1:]
    view = rays.directions / rays.directions.norm(dim=-1, keepdim=True)
    if render_diffuse:
        coeffs = coeffs[..., :1]
    degree = int(np.sqrt(coeffs.shape[-1])) - 1
    basis = sh_basis(degree, view)[:, None, :].expand(n, s, -1).reshape(flat.shape[0], 1, -1)
    radiance = (coeffs * basis).sum(dim=-1)
    inside = voxel_grid.test_inside_volume(flat)
    radiance = torch.where(inside, radiance, torch.full_like(radiance, -INFINITY))
    density = torch.where(inside, density, torch.zeros_like(density))
    return ProcessedPointsOnRays(torch.cat([radiance, density], dim=-1).reshape(n, s, 4), sampled_points.depths)


# ---------------------------------------------------------------------------------------------------------------------------
# accumulator (rendering/volumetric/accumulate.py:24-113)
# ---------------------------------------------------------------------------------------------------------------------------
def density2occupancy_pb(densities: Tensor, deltas: Tensor) -> Tensor:
    return 1.0 - torch.exp(-(densities * deltas))


def accumulate_radiance_density_on_rays(processed_points: ProcessedPointsOnRays, rays: Rays, stochastic_density_noise_std: float = 1.0,
                                        density2occupancy: Callable[[Tensor, Tensor], Tensor] = density2occupancy_pb,
                                        radiance_hdr_tone_map: Callable[[Tensor], Tensor] = torch.sigmoid, white_bkgd: bool = True,
                                        extra_debug_info: bool = False, density_noise: Optional[Tensor] = None) -> RenderOut:
    """front-to-back compositing of [N, S, 4] processed points.  Argument names, defaults and the noise draw are the reference's
    (torch.randn(N, S) * std, drawn even when std == 0); ``density_noise`` [N, S] replaces the draw (tests)."""
    values, z = processed_points
    _require_device(values, "processed points")
    raw_radiance, sigma = values[..., :-1], values[..., -1]
    step = torch.cat([z[..., 1:] - z[..., :-1], torch.full_like(z[..., :1], INFINITY)], dim=-1) * rays.directions.norm(dim=-1, keepdim=True)
    noise = torch.randn(sigma.shape, dtype=sigma.dtype, device=sigma.device) * stochastic_density_noise_std if density_noise is None else density_noise
    alpha = density2occupancy(sigma + noise, step)
    # transmittance in front of each sample.  The reference's torch.cumprod / torch.sum run on the CPU there: a sequential product and
    # pairwise sums.  On the device torch scans and reduces in another order, whose float32 roundings are coherent where the density
    # varies slowly -- 3..6e-5 on depth at 4096+ samples per ray against the CPU reference (tests/parity_fuzz.py, kind "longcomposed") --
    # so the scan and the three reductions are carried in float64 and rounded once (the products and sums of the float32 per-sample
    # values: as close to the reference's own float32 results as those are to the exact ones)
    wide = torch.float64 if alpha.dtype == torch.float32 else alpha.dtype
    through = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha], dim=-1).to(wide), dim=-1)[:, :-1].to(alpha.dtype)
    weights = alpha * through
    colour = (radiance_hdr_tone_map(raw_radiance) * weights[..., None]).sum(dim=-2, dtype=wide).to(alpha.dtype)
    acc = weights.sum(dim=-1, keepdim=True, dtype=wide).to(alpha.dtype)
    if white_bkgd:
        colour = colour + (1 - acc)
    depth = (z * weights).sum(dim=-1, keepdim=True, dtype=wide).to(alpha.dtype)
    disparity = 1.0 / torch.maximum(torch.full_like(acc, ZERO_PLUS), depth / acc)
    extra = {EXTRA_DISPARITY: disparity, EXTRA_ACCUMULATED_WEIGHTS: acc}
    if extra_debug_info:
        extra.update({EXTRA_POINT_DENSITIES: sigma, EXTRA_POINT_OCCUPANCIES: alpha, EXTRA_POINT_WEIGHTS: weights, EXTRA_POINT_DEPTHS: z, EXTRA_SAMPLE_INTERVALS: step})
    return RenderOut(colour=colour, depth=depth, extra=extra)


# ---------------------------------------------------------------------------------------------------------------------------
# the composed render procedure (thre3d_reprs/renderers.py:48-102)
# ---------------------------------------------------------------------------------------------------------------------------
def render_sh_voxel_grid_composed(voxel_grid, rays: Rays, render_config, parallel_points_chunk_size: Optional[int] = None, t_rand: Optional[Tensor] = None,
                                  extra_debug_info: bool = False, density_noise: Optional[Tensor] = None) -> RenderOut:
    """``render_sh_voxel_grid`` out of the three plug-ins above: any config, any activation callables (see the module docstring).
    ``t_rand`` / ``density_noise`` [N, S] replace the torch.rand / torch.randn draws (tests)."""
    _require_device(rays.origins, "rays")
    if render_config.optimized_sampling:
        sampler = lambda r, b, n: sample_aabb_bound_uniform_points_on_rays(r, b, n, voxel_grid.aabb, perturb=render_config.perturb_sampled_points, t_rand=t_rand)  # noqa: E731
    else:
        sampler = lambda r, b, n: sample_uniform_points_on_rays(r, b, n, perturb=render_config.perturb_sampled_points, t_rand=t_rand)  # noqa: E731
    processor = lambda pts, r: process_points_with_sh_voxel_grid(pts, r, voxel_grid, render_diffuse=render_config.render_diffuse,  # noqa: E731
                                                                 parallel_points_chunk_size=parallel_points_chunk_size)
    accumulator = lambda pts, r: accumulate_radiance_density_on_rays(  # noqa: E731
        pts, r, stochastic_density_noise_std=render_config.stochastic_density_noise_std, density2occupancy=render_config.density2occupancy,
        radiance_hdr_tone_map=render_config.radiance_hdr_tone_map, white_bkgd=render_config.white_bkgd, extra_debug_info=extra_debug_info,
        density_noise=density_noise)
    return render(rays, render_config.camera_bounds, int(render_config.num_samples_per_ray), sampler, processor, accumulator)
