"""The HIP render procedure -- drop-in for the reference's ``render_sh_voxel_grid``
(thre3d_atom/thre3d_reprs/renderers.py:48-102) behind the reference's own plug-in type

    RenderProcedure = Callable[[Module, Rays, RenderConfig, Optional[int]], RenderOut]
    (renderers.py:22-25; stored and invoked by VolumetricModel, modules/volumetric_model.py:112-114)

``SHVoxGridRenderConfig`` has the reference's fields and defaults (renderers.py:28-45).  The fused kernels implement the
reference defaults of the two callable fields (``density2occupancy_pb``, ``torch.sigmoid``), no density noise, and the four
density-activation pairs of voxels.resolve_density_mode; any other configuration is rendered by the COMPOSED path
(composable.py: the reference's sampler -> point processor -> accumulator composition with the HIP interpolation in the middle and
the callables applied by torch, on the device).  The fused training step (trainers.TrainStepper(fused=True)) takes the fused
configurations only and raises ``ValueError`` otherwise.
"""
import dataclasses
from typing import Any, Callable, Optional

import numpy as np
import torch
from torch import Tensor
from torch.nn import Module

from .camera import CameraBounds
from .constants import EXTRA_ACCUMULATED_WEIGHTS, EXTRA_DISPARITY
from .ops import KeyedJitter, draw_jitter_key, relu_field_render
from .render_interface import Rays, RenderOut
from .voxels import VoxelGrid, as_kernel_grid

RenderConfig = Any
RenderProcedure = Callable[[Module, Rays, RenderConfig, Optional[int]], RenderOut]


def density2occupancy_pb(densities: Tensor, deltas: Tensor) -> Tensor:
    """alpha = 1 - exp(-sigma * delta) (reference rendering/volumetric/accumulate.py:24-28).  Kept as the
    identity token of the config field; the kernel evaluates the same expression per sample."""
    return 1.0 - torch.exp(-(densities * deltas))


@dataclasses.dataclass
class SHVoxGridRenderConfig:
    num_samples_per_ray: int
    camera_bounds: CameraBounds
    perturb_sampled_points: bool = True
    optimized_sampling: bool = False
    density2occupancy: Callable[[Tensor, Tensor], Tensor] = density2occupancy_pb
    radiance_hdr_tone_map: Callable[[Tensor], Tensor] = torch.sigmoid
    stochastic_density_noise_std: float = 0.0
    white_bkgd: bool = False
    render_diffuse: bool = False
    render_num_samples_per_ray: int = 1024
    parallel_rays_chunk_size: int = 32768
    # --- extensions of this build (not in the reference) ---
    # skip cells that provably contribute nothing (exact for the ReLU field); the grid's occupancy mask
    # is rebuilt by the caller whenever densities change (VoxelGrid.build_occupancy)
    use_occupancy_mask: bool = False
    # the reference draws torch.randn for the density noise even when its std is 0 (accumulate.py:59-62),
    # which advances the RNG stream; set True to consume the same numbers and stay stream-compatible
    consume_reference_rng: bool = False
    # where the jitter of perturb_sampled_points comes from when no t_rand is supplied: "keyed" = a counter-based generator
    # evaluated inside the kernels (keyed from torch's CPU generator; no [N, S] tensor), "torch" = torch.rand(N, S) on the
    # rays' device exactly like the reference's sample.py:63
    jitter: str = "keyed"


def _check_supported(cfg) -> None:
    """``cfg``: this package's SHVoxGridRenderConfig or the reference's (same field names; the extension fields of this build
    default to off when absent)."""
    d2o = cfg.density2occupancy
    if not (d2o is density2occupancy_pb or getattr(d2o, "__name__", "") == "density2occupancy_pb"):
        raise ValueError("render_sh_voxel_grid (HIP): only density2occupancy_pb is supported")
    if cfg.radiance_hdr_tone_map is not torch.sigmoid:
        raise ValueError("render_sh_voxel_grid (HIP): only torch.sigmoid is supported as radiance_hdr_tone_map")
    if cfg.stochastic_density_noise_std != 0.0:
        # (the reference adds the noise to the ACTIVATED density of every sample, the last one included, whose interval is 1e10 |d|
        # (accumulate.py:49-62): about half of all rays come out with alpha = -inf, a non-finite colour and loss -- pinned against
        # the reference itself by tests/test_oracle_golden.py::test_g11_...; its CLI never sets the field.  Not reproduced.)
        raise ValueError("render_sh_voxel_grid (HIP): stochastic_density_noise_std must be 0.0 (the reference's noise path yields non-finite renders)")


def fused_kernels_apply(voxel_grid, render_config) -> bool:
    """True when the fused kernels implement this (grid, config) pair; else ``render_sh_voxel_grid`` composes the render."""
    try:
        _check_supported(render_config)
    except ValueError:
        return False
    fn = getattr(voxel_grid, "fused_kernels_apply", None)
    return True if fn is None else bool(fn())


def render_sh_voxel_grid(
    voxel_grid: VoxelGrid,
    rays: Rays,
    render_config: SHVoxGridRenderConfig,
    parallel_points_chunk_size: Optional[int] = None,
    t_rand=None,
    first_ray: int = 0,
) -> RenderOut:
    """Render flat rays [N, 3] through an SH voxel grid with the fused HIP kernels.

    ``parallel_points_chunk_size`` is accepted for signature compatibility and ignored: the fused
    kernel never materialises per-point tensors, so there is nothing to chunk.
    ``t_rand`` optionally supplies the stratified-sampling jitter: a [N, S] tensor, or a ``KeyedJitter``.  Otherwise, when
    ``perturb_sampled_points`` is set, it comes from ``render_config.jitter``: "keyed" = a counter-based generator inside the
    kernel (one 64-bit key drawn from torch's CPU generator per call; ``first_ray`` = position of ray 0 in that stream),
    "torch" = ``torch.rand(N, S)`` on the rays' device like sample.py:63."""
    # duck typing, like the reference's plug-in contract (renderers.py:22-25 takes "a Module"): this package's VoxelGrid, or any
    # module with the reference VoxelGrid's attribute names -- e.g. the reference's own grid object (TypeError otherwise)
    if not fused_kernels_apply(voxel_grid, render_config):
        # non-default plug-ins (density2occupancy, tone map, density noise, activation callables): the composed path
        from .composable import render_sh_voxel_grid_composed

        if isinstance(t_rand, KeyedJitter):
            raise ValueError("the composed render path takes a [N, S] t_rand tensor (or draws torch.rand like the reference), not a KeyedJitter")
        return render_sh_voxel_grid_composed(voxel_grid, rays, render_config, parallel_points_chunk_size, t_rand=t_rand)
    voxel_grid = as_kernel_grid(voxel_grid)
    _check_supported(render_config)
    origins, directions = rays.origins, rays.directions
    assert origins.dim() == directions.dim() == 2, "the render interface only works with FLAT rays"
    num_samples = int(render_config.num_samples_per_ray)
    n = origins.shape[0]
    if render_config.perturb_sampled_points:
        if t_rand is None:
            jitter = getattr(render_config, "jitter", "keyed")  # (the reference's config class has no such field)
            if jitter == "torch":
                t_rand = torch.rand(n, num_samples, dtype=torch.float32, device=origins.device)
            elif jitter == "keyed":
                t_rand = KeyedJitter(draw_jitter_key(), int(first_ray))
            else:
                raise ValueError("SHVoxGridRenderConfig.jitter must be 'keyed' or 'torch'")
    else:
        t_rand = None
    if getattr(render_config, "consume_reference_rng", False):
        torch.randn(n, num_samples, dtype=torch.float32, device=origins.device)
    bounds = render_config.camera_bounds
    colour, depth, acc, disparity = relu_field_render(
        voxel_grid,
        origins,
        directions,
        num_samples=num_samples,
        near=float(np.float32(bounds.near)),
        far=float(np.float32(bounds.far)),
        t_rand=t_rand,
        white_bkgd=bool(render_config.white_bkgd),
        render_diffuse=bool(render_config.render_diffuse),
        optimized_sampling=bool(render_config.optimized_sampling),
        use_occupancy=bool(getattr(render_config, "use_occupancy_mask", False)),
    )
    return RenderOut(colour=colour, depth=depth, extra={EXTRA_DISPARITY: disparity, EXTRA_ACCUMULATED_WEIGHTS: acc})


def render_sh_voxel_grid_pair(voxel_grid: VoxelGrid, rays: Rays, render_config: SHVoxGridRenderConfig, parallel_points_chunk_size: Optional[int] = None, t_rands=(None, None)):
    """The two renders of a training iteration on the same flat rays -- ``render_sh_voxel_grid(grid, rays, cfg)`` and the same call with
    ``render_diffuse=True``, each with its own jitter draw, in the reference's order (modules/trainers.py:306, 323-325) -- as
    (specular RenderOut, diffuse RenderOut).  Where the fused kernels apply and a gradient will be asked for, the pair is ONE
    autograd node: one forward launch, and a backward of two launches instead of four (ops.relu_field_render_pair); the outputs and
    the RNG consumption are those of the two single calls.  ``render_config.render_diffuse`` is ignored (the pair is both)."""
    import copy

    if not fused_kernels_apply(voxel_grid, render_config):
        cfgs = []
        for diffuse in (False, True):
            cfg = copy.copy(render_config)
            cfg.render_diffuse = diffuse
            cfgs.append(cfg)
        return tuple(render_sh_voxel_grid(voxel_grid, rays, cfg, parallel_points_chunk_size, t_rand=t) for cfg, t in zip(cfgs, t_rands))
    voxel_grid = as_kernel_grid(voxel_grid)
    origins, directions = rays.origins, rays.directions
    assert origins.dim() == directions.dim() == 2, "the render interface only works with FLAT rays"
    num_samples, n = int(render_config.num_samples_per_ray), origins.shape[0]
    jitters = []
    for given in t_rands:  # (the draws of the two single calls, in their order)
        t = given
        if not render_config.perturb_sampled_points:
            t = None
        elif t is None:
            kind = getattr(render_config, "jitter", "keyed")
            if kind == "torch":
                t = torch.rand(n, num_samples, dtype=torch.float32, device=origins.device)
            elif kind == "keyed":
                t = KeyedJitter(draw_jitter_key(), 0)
            else:
                raise ValueError("SHVoxGridRenderConfig.jitter must be 'keyed' or 'torch'")
        if getattr(render_config, "consume_reference_rng", False):
            torch.randn(n, num_samples, dtype=torch.float32, device=origins.device)
        jitters.append(t)
    from .ops import relu_field_render_pair

    bounds = render_config.camera_bounds
    outs = relu_field_render_pair(
        voxel_grid, origins, directions, num_samples, float(np.float32(bounds.near)), float(np.float32(bounds.far)), t_rands=tuple(jitters),
        white_bkgd=bool(render_config.white_bkgd), optimized_sampling=bool(render_config.optimized_sampling),
        use_occupancy=bool(getattr(render_config, "use_occupancy_mask", False)),
    )
    return tuple(RenderOut(colour=c, depth=d, extra={EXTRA_DISPARITY: q, EXTRA_ACCUMULATED_WEIGHTS: a}) for c, d, a, q in outs)


def render_sh_voxel_grid_frame(
    voxel_grid: VoxelGrid,
    camera_intrinsics,
    camera_pose,
    render_config: SHVoxGridRenderConfig,
    first_ray: int = 0,
    num_rays: Optional[int] = None,
) -> RenderOut:
    """The pixels [first_ray, first_ray + num_rays) of a whole posed-camera frame (row-major; default all of it) in ONE kernel
    launch: what ``VolumetricModel.render`` does per chunk -- cast_rays, slice, ``torch.rand``, render, concatenate
    (reference modules/volumetric_model.py:143-172) -- with the rays and the stratified jitter generated inside the kernel.
    Inference only (no autograd); results do not depend on how a frame is split into calls.  Configurations that ask for torch's own
    random streams -- ``consume_reference_rng``, or ``perturb_sampled_points`` with ``jitter="torch"`` (a torch.rand table per chunk) --
    are served the way ``VolumetricModel.render`` serves them: cast_rays, then ``render_sh_voxel_grid`` on chunks of
    ``parallel_rays_chunk_size`` rays of the range, concatenated (the reference's loop, volumetric_model.py:152-172)."""
    from .ops import render_flags, render_frame_raw

    voxel_grid = as_kernel_grid(voxel_grid)
    _check_supported(render_config)
    jitter = None
    # (the reference's own config class has neither ``jitter`` nor ``use_occupancy_mask`` nor ``consume_reference_rng``)
    use_occupancy = bool(getattr(render_config, "use_occupancy_mask", False))
    jitter_kind = getattr(render_config, "jitter", "keyed")
    if jitter_kind not in ("keyed", "torch"):
        raise ValueError("SHVoxGridRenderConfig.jitter must be 'keyed' or 'torch'")
    if getattr(render_config, "consume_reference_rng", False) or (render_config.perturb_sampled_points and jitter_kind != "keyed"):
        return _render_frame_in_chunks(voxel_grid, camera_intrinsics, camera_pose, render_config, first_ray, num_rays)
    if render_config.perturb_sampled_points:
        jitter = KeyedJitter(draw_jitter_key(), 0)
    if use_occupancy and not voxel_grid.occupancy_current():
        voxel_grid.build_occupancy()
    height, width, focal = camera_intrinsics
    bounds = render_config.camera_bounds
    flags = render_flags(render_config.white_bkgd, render_config.render_diffuse, render_config.optimized_sampling, use_occupancy)
    colour, depth, acc, disparity = render_frame_raw(
        voxel_grid, int(height), int(width), float(focal), camera_pose.rotation, camera_pose.translation, int(render_config.num_samples_per_ray),
        float(np.float32(bounds.near)), float(np.float32(bounds.far)), flags, jitter, first_ray=first_ray, num_rays=num_rays,
    )
    return RenderOut(colour=colour, depth=depth, extra={EXTRA_DISPARITY: disparity, EXTRA_ACCUMULATED_WEIGHTS: acc})


def _render_frame_in_chunks(voxel_grid, camera_intrinsics, camera_pose, render_config, first_ray: int, num_rays: Optional[int]) -> RenderOut:
    """``render_sh_voxel_grid_frame`` for the configurations that draw from torch's generators per chunk (above): the reference's
    chunk loop (modules/volumetric_model.py:152-172) over the pixel range, under no_grad."""
    from .ops import cast_rays_hip
    from .render_interface import collate_rendered_output

    height, width, focal = camera_intrinsics
    total = int(height) * int(width)
    count = total - int(first_ray) if num_rays is None else int(num_rays)
    if first_ray < 0 or count < 0 or first_ray + count > total:
        raise RuntimeError(f"render_sh_voxel_grid_frame: bad shape (pixels [{first_ray}, {first_ray + count}) of a {height} x {width} frame)")
    device = voxel_grid.kernel_tensors()[0].device
    o, d = cast_rays_hip(int(height), int(width), float(focal), camera_pose.rotation, camera_pose.translation, device)
    flat = Rays(o.reshape(-1, 3)[first_ray : first_ray + count], d.reshape(-1, 3)[first_ray : first_ray + count])
    chunk = max(1, int(getattr(render_config, "parallel_rays_chunk_size", 32768) or count or 1))
    with torch.no_grad():
        chunks = [render_sh_voxel_grid(voxel_grid, flat[s : s + chunk], render_config, first_ray=first_ray + s) for s in range(0, max(count, 1), chunk)]
    return collate_rendered_output(chunks)
