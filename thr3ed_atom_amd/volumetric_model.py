"""VolumetricModel -- holder of (representation, render procedure, render config); mirrors the
reference's thre3d_atom/modules/volumetric_model.py:30-197 (same methods, kwargs and checkpoint
dictionary layout).  ``render`` generates its rays with the HIP ray caster.
"""
import copy
import dataclasses
from pathlib import Path
from typing import Any, Callable, Dict, Optional, Tuple

import torch
from torch.nn import Module

from .camera import CameraIntrinsics, CameraPose
from .constants import (
    CONFIG_DICT,
    EXTRA_INFO,
    RENDER_CONFIG,
    RENDER_CONFIG_TYPE,
    RENDER_PROCEDURE,
    STATE_DICT,
    THRE3D_REPR,
)
from . import distributed as rfdist
from .ops import cast_rays_hip
from .render_interface import Rays, RenderOut, collate_rendered_output, flatten_rays, reshape_rendered_output
from .renderers import RenderConfig, RenderProcedure, render_sh_voxel_grid, render_sh_voxel_grid_frame, render_sh_voxel_grid_pair, fused_kernels_apply


def cast_rays(camera_intrinsics: CameraIntrinsics, pose: CameraPose, device=None) -> Rays:
    """Rays [H, W, 3] of a posed pinhole camera on a HIP device
    (reference rendering/volumetric/utils/misc.py:12-50)."""
    device = torch.device("cuda" if device is None else device)
    height, width, focal = camera_intrinsics
    o, d = cast_rays_hip(int(height), int(width), float(focal), pose.rotation, pose.translation, device)
    return Rays(o, d)


class VolumetricModel:
    def __init__(
        self,
        thre3d_repr: Module,
        render_procedure: RenderProcedure,
        render_config: RenderConfig,
        device: torch.device = torch.device("cuda" if torch.cuda.is_available() else "cpu"),
    ) -> None:
        self._thre3d_repr = thre3d_repr.to(device)
        self._render_procedure = render_procedure
        self._render_config = render_config
        self._device = torch.device(device)

    @property
    def thre3d_repr(self) -> Module:
        return self._thre3d_repr

    @thre3d_repr.setter
    def thre3d_repr(self, thre3d_repr: Module) -> None:
        self._thre3d_repr = thre3d_repr

    @property
    def render_procedure(self) -> RenderProcedure:
        return self._render_procedure

    @property
    def render_config(self) -> RenderConfig:
        return self._render_config

    @property
    def device(self) -> torch.device:
        return self._device

    @staticmethod
    def _update_render_config(render_config: RenderConfig, update_dict: Dict[str, Any]) -> RenderConfig:
        """Per-call overrides on a copy; unknown names raise ValueError (reference :67-81)."""
        updated = copy.deepcopy(render_config)
        for field, value in update_dict.items():
            if not hasattr(updated, field):
                raise ValueError(f"Unknown render configuration field {field} requested for overriding :(")
            setattr(updated, field, value)
        return updated

    def get_save_info(self, extra_info: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
        info = {
            THRE3D_REPR: {
                STATE_DICT: self._thre3d_repr.state_dict(),
                CONFIG_DICT: self._thre3d_repr.get_save_config_dict(),
            },
            RENDER_PROCEDURE: self._render_procedure,
            RENDER_CONFIG_TYPE: type(self._render_config),
            RENDER_CONFIG: dataclasses.asdict(self._render_config),
        }
        if extra_info is not None:
            info[EXTRA_INFO] = extra_info
        return info

    def render_rays(self, rays: Rays, parallel_points_chunk_size: Optional[int] = None, **kwargs) -> RenderOut:
        """Differentiable render of flat rays; kwargs override render-config fields for this call."""
        cfg = self._update_render_config(self._render_config, kwargs)
        return self._render_procedure(self._thre3d_repr, rays, cfg, parallel_points_chunk_size)

    def render_rays_pair(self, rays: Rays, parallel_points_chunk_size: Optional[int] = None, **kwargs) -> Tuple[RenderOut, RenderOut]:
        """``(render_rays(rays, **kwargs), render_rays(rays, render_diffuse=True, **kwargs))`` -- the two renders of a training
        iteration (reference modules/trainers.py:306, 323-325; an extension of this build) -- with the HIP procedure as ONE autograd
        node (renderers.render_sh_voxel_grid_pair); any other procedure is simply called twice."""
        if self._render_procedure is render_sh_voxel_grid:
            cfg = self._update_render_config(self._render_config, kwargs)
            return render_sh_voxel_grid_pair(self._thre3d_repr, rays, cfg, parallel_points_chunk_size)
        return self.render_rays(rays, parallel_points_chunk_size, **kwargs), self.render_rays(rays, parallel_points_chunk_size, **dict(kwargs, render_diffuse=True))

    def render(
        self,
        camera_pose: CameraPose,
        camera_intrinsics: CameraIntrinsics,
        parallel_rays_chunk_size: Optional[int] = 32768,
        parallel_points_chunk_size: Optional[int] = None,
        gpu_render: bool = True,
        verbose: bool = False,
        data_parallel: bool = False,
        **kwargs,
    ) -> RenderOut:
        """Full-image render under no_grad: cast rays, render them in chunks of
        ``parallel_rays_chunk_size`` (None = one chunk), concatenate and reshape to [H, W, .]
        (reference :116-174).  The fused kernel does not need chunking for memory; the argument is
        honoured so that outputs and RNG consumption follow the reference chunk by chunk.
        ``data_parallel=True`` (extension): under torch.distributed the frame's rays are split over the ranks and
        the results gathered; every rank returns the full image."""
        cfg = self._update_render_config(self._render_config, kwargs)
        height, width, _ = camera_intrinsics
        total_rays = int(height) * int(width)
        if (
            self._render_procedure is render_sh_voxel_grid
            and fused_kernels_apply(self._thre3d_repr, cfg)  # (else: chunk by chunk through the composed path)
            and gpu_render
            and not verbose
            and (not getattr(cfg, "perturb_sampled_points", False) or getattr(cfg, "jitter", "keyed") == "keyed")
            and not getattr(cfg, "consume_reference_rng", False)  # (that flag asks for the reference's per-chunk torch.randn draws)
        ):
            # the HIP procedure renders a frame (or this rank's share of it) in ONE launch: rays and jitter are generated inside
            # the kernel, so there is nothing to chunk -- ``parallel_rays_chunk_size`` does not change a single bit of the result
            lo, hi = (0, total_rays)
            dp = data_parallel and rfdist.world_size() > 1
            if dp:
                lo, hi = rfdist.shard_range(total_rays)
            with torch.no_grad():
                out = render_sh_voxel_grid_frame(self._thre3d_repr, camera_intrinsics, camera_pose, cfg, first_ray=lo, num_rays=hi - lo)
            if dp:
                out = self._gather_frame_shards(out, total_rays)
            return reshape_rendered_output(out, camera_intrinsics)
        flat = flatten_rays(cast_rays(camera_intrinsics, camera_pose, self._device))
        dp = data_parallel and rfdist.world_size() > 1
        if dp:
            # rays are independent: every rank renders one contiguous range of the frame against its own replica of
            # the grid; the only exchange is the gather of the [n, 6] per-ray results at the end
            lo, hi = rfdist.shard_range(total_rays)
            flat = flat[lo:hi]
        chunk = len(flat) if parallel_rays_chunk_size is None else int(parallel_rays_chunk_size)
        starts = range(0, len(flat), chunk)
        if verbose:
            from tqdm import tqdm

            starts = tqdm(starts)
        chunks = []
        with torch.no_grad():
            for start in starts:
                out = self.render_rays(flat[start : start + chunk], parallel_points_chunk_size, **kwargs)
                if not gpu_render and not dp:
                    out = out.to(torch.device("cpu"))
                chunks.append(out)
        out = collate_rendered_output(chunks)
        if dp:
            out = self._gather_frame_shards(out, total_rays)  # (on the device, whatever ``gpu_render`` says: RCCL moves device memory)
            if not gpu_render:
                out = out.to(torch.device("cpu"))
        return reshape_rendered_output(out, camera_intrinsics)

    @staticmethod
    def _gather_frame_shards(out: RenderOut, total_rays: int) -> RenderOut:
        """every rank's rows of a frame (``shard_range`` of its flat rays) -> the whole frame on every rank: ONE all-gather of the
        packed [n, 3 + 1 + extras] per-ray results; the row counts follow from the shard rule, nothing else is exchanged"""
        world = rfdist.world_size()
        sizes = [hi - lo for lo, hi in (rfdist.shard_range(total_rays, r, world) for r in range(world))]
        keys = sorted(out.extra.keys())
        packed = torch.cat([out.colour, out.depth] + [out.extra[k] for k in keys], dim=-1)
        packed = rfdist.all_gather_rows(packed, sizes)
        assert packed.shape[0] == total_rays
        return RenderOut(packed[:, :3], packed[:, 3:4], {k: packed[:, 4 + i : 5 + i] for i, k in enumerate(keys)})


def create_volumetric_model_from_saved_model(
    model_path: Path,
    thre3d_repr_creator: Callable[[Dict[str, Any]], Module],
    device: torch.device = torch.device("cpu"),
) -> Tuple[VolumetricModel, Dict[str, Any]]:
    """Load a checkpoint written by ``torch.save(vol_mod.get_save_info(...))`` (reference :177-197) -- by this package OR by
    the reference itself: the names a reference-written file pickles (``thre3d_atom...render_sh_voxel_grid``,
    ``SHVoxGridRenderConfig``, ``VoxelSize`` ...) are resolved to their counterparts here (_compat_pickle.py).
    The checkpoint pickles a function object and a class, hence weights_only=False."""
    from . import _compat_pickle

    data = torch.load(model_path, map_location="cpu", weights_only=False, pickle_module=_compat_pickle)
    repr_ = thre3d_repr_creator(data)
    known = {f.name for f in dataclasses.fields(data[RENDER_CONFIG_TYPE])}
    cfg = data[RENDER_CONFIG_TYPE](**{k: v for k, v in data[RENDER_CONFIG].items() if k in known})
    model = VolumetricModel(repr_, data[RENDER_PROCEDURE], cfg, device=device)
    return model, data.get(EXTRA_INFO, {})
