"""Ray / render-output value types of the render path.

Same field names, shapes and helper methods as the reference's
thre3d_atom/rendering/volumetric/render_interface.py:13-44 (Rays) and :47-83 (RenderOut), so
code written against the reference's types works unchanged with the HIP render procedure.
"""
import dataclasses
from typing import Any, Dict, Optional, Sequence

import torch
from torch import Tensor

from .camera import CameraIntrinsics
from .constants import NUM_COLOUR_CHANNELS, NUM_COORD_DIMENSIONS


@dataclasses.dataclass
class Rays:
    origins: Tensor  # [..., 3]
    directions: Tensor  # [..., 3], NOT normalised (see cast_rays)

    def __post_init__(self):
        if self.origins.shape != self.directions.shape:
            raise AssertionError("ray origins and directions must have the same shape")
        if self.origins.shape[-1] != NUM_COORD_DIMENSIONS:
            raise AssertionError("rays live in a 3-dimensional coordinate space")

    def __getitem__(self, item) -> "Rays":
        return Rays(self.origins[item, :], self.directions[item, :])

    def __len__(self) -> int:
        return len(self.origins)

    def to(self, device) -> "Rays":
        return Rays(self.origins.to(device), self.directions.to(device))


@dataclasses.dataclass
class RenderOut:
    colour: Tensor  # [..., 3]
    depth: Tensor  # [..., 1]
    extra: Optional[Dict[str, Any]] = None

    def __post_init__(self):
        if self.colour.shape[:-1] != self.depth.shape[:-1]:
            raise AssertionError("colour and depth maps are shape-incompatible")
        if self.colour.shape[-1] != NUM_COLOUR_CHANNELS:
            raise AssertionError("only RGB colour maps are supported")
        if self.depth.shape[-1] != 1:
            raise AssertionError("depth maps carry exactly one channel")
        if self.extra is None:
            self.extra = {}

    def detach(self) -> "RenderOut":
        return RenderOut(
            self.colour.detach(), self.depth.detach(), {k: v.detach() for k, v in self.extra.items()}
        )

    def to(self, device) -> "RenderOut":
        return RenderOut(
            self.colour.to(device), self.depth.to(device), {k: v.to(device) for k, v in self.extra.items()}
        )


def flatten_rays(rays: Rays) -> Rays:
    """[..., 3] -> [N, 3], row-major: ray index = i * W + j (reference utils/misc.py:53-57)."""
    return Rays(
        rays.origins.reshape(-1, NUM_COORD_DIMENSIONS), rays.directions.reshape(-1, NUM_COORD_DIMENSIONS)
    )


def collate_rays(rays_list: Sequence[Rays]) -> Rays:
    """reference utils/misc.py:60-65"""
    return Rays(
        torch.cat([r.origins for r in rays_list], dim=0), torch.cat([r.directions for r in rays_list], dim=0)
    )


def collate_rendered_output(chunks: Sequence[RenderOut]) -> RenderOut:
    """Concatenate per-chunk outputs along the ray axis (reference utils/misc.py:132-151)."""
    keys = list(chunks[0].extra.keys()) if chunks else []
    return RenderOut(
        colour=torch.cat([c.colour for c in chunks], dim=0),
        depth=torch.cat([c.depth for c in chunks], dim=0),
        extra={k: torch.cat([c.extra[k] for c in chunks], dim=0) for k in keys},
    )


def reshape_rendered_output(out: RenderOut, camera_intrinsics: CameraIntrinsics) -> RenderOut:
    """[H*W, c] -> [H, W, c] for colour, depth and every extra (reference utils/misc.py:154-163)."""
    shape = (camera_intrinsics.height, camera_intrinsics.width, -1)
    return RenderOut(
        colour=out.colour.reshape(*shape),
        depth=out.depth.reshape(*shape),
        extra={k: v.reshape(*shape) for k, v in out.extra.items()},
    )
