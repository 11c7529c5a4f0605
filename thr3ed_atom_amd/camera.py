"""Camera value types, pose builders and image metrics used around the render path.

Mirrors (same names, field order and meaning) the reference's
thre3d_atom/utils/imaging_utils.py:17-30 (CameraIntrinsics / CameraPose / CameraBounds),
:58-63 (the ``slack`` range mapping used to normalise points), :141-191 (pose_spherical)
and thre3d_atom/utils/metric_utils.py:10-21 (mse2psnr).
"""
import math
from typing import NamedTuple, Sequence, Tuple, Union

import numpy as np
import torch
from torch import Tensor

from .constants import INFINITY


class CameraIntrinsics(NamedTuple):
    height: int
    width: int
    focal: float


class CameraPose(NamedTuple):
    rotation: Union[np.ndarray, Tensor]  # [3, 3] camera-to-world
    translation: Union[np.ndarray, Tensor]  # [3, 1]


class CameraBounds(NamedTuple):
    near: float
    far: float


def slack_range_map(
    drange_in: Tuple[float, float], drange_out: Tuple[float, float] = (-1.0, 1.0)
) -> Tuple[np.float32, np.float32]:
    """(scale, bias) in float32 such that ``x * scale + bias`` maps drange_in -> drange_out.

    The arithmetic is done in numpy float32 exactly as the reference does it
    (imaging_utils.py:58-63) because the two constants enter every sample position."""
    lo_in, hi_in = np.float32(drange_in[0]), np.float32(drange_in[1])
    lo_out, hi_out = np.float32(drange_out[0]), np.float32(drange_out[1])
    scale = (hi_out - lo_out) / (hi_in - lo_in)
    bias = lo_out - lo_in * scale
    return np.float32(scale), np.float32(bias)


def scale_camera_intrinsics(intr: CameraIntrinsics, factor: float = 1.0) -> CameraIntrinsics:
    """imaging_utils.py:126-134: integer sizes are ceil-ed, focal scaled."""
    return CameraIntrinsics(
        height=int(np.ceil(intr.height * factor)),
        width=int(np.ceil(intr.width * factor)),
        focal=intr.focal * factor,
    )


def _homogeneous(rows: Sequence[Sequence[float]], device) -> Tensor:
    return torch.tensor(rows, dtype=torch.float32, device=device)


def pose_spherical(yaw: float, pitch: float, radius: float, device=torch.device("cpu")) -> CameraPose:
    """Camera on a sphere looking at the origin; angles in degrees (imaging_utils.py:185-191).

    c2w = Rz(yaw) . Rx(pitch) . Tz(radius), all built as float32 4x4 matrices."""
    p, y = pitch / 180.0 * np.pi, yaw / 180.0 * np.pi
    lift = _homogeneous(
        [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]], device
    )
    tilt = _homogeneous(
        [[1, 0, 0, 0], [0, np.cos(p), -np.sin(p), 0], [0, np.sin(p), np.cos(p), 0], [0, 0, 0, 1]],
        device,
    )
    spin = _homogeneous(
        [[np.cos(y), -np.sin(y), 0, 0], [np.sin(y), np.cos(y), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]],
        device,
    )
    c2w = spin @ (tilt @ lift)
    return CameraPose(rotation=c2w[:3, :3], translation=c2w[:3, 3:])


def get_thre360_animation_poses(hemispherical_radius: float, camera_pitch: float, num_poses: int):
    """Turn-table path (imaging_utils.py:199-209): num_poses-1 yaws in [0, 360)."""
    yaws = np.linspace(0, 360, num_poses)[:-1]
    return [pose_spherical(yaw, camera_pitch, hemispherical_radius) for yaw in yaws]


def get_thre360_spiral_animation_poses(horizontal_radius_range, vertical_camera_height: float, num_rounds: int, num_poses: int):
    """Spiral path (imaging_utils.py:211-234): the last pose is dropped so that a looped video is smooth."""
    horizontal_radii = np.linspace(*horizontal_radius_range, num_poses)[:-1]
    radii = [np.sqrt((h**2) + (vertical_camera_height**2)) for h in horizontal_radii]
    yaws = np.linspace(0, 360 * num_rounds, num_poses)[:-1]
    pitches = [math.atan(h / vertical_camera_height) * 180 / math.pi for h in horizontal_radii]
    return [pose_spherical(yaw, pitch, radius) for yaw, pitch, radius in zip(yaws, pitches, radii)]


def mse2psnr(x):
    """PSNR (dB) of a mean squared error on [0, 1] signals; inf for an exact match."""
    if isinstance(x, Tensor):
        if x == 0.0:
            return torch.tensor([INFINITY], dtype=x.dtype, device=x.device)
        ten = torch.tensor([10.0], dtype=x.dtype, device=x.device)
        return -10.0 * torch.log(x) / torch.log(ten)
    return -10.0 * math.log(x) / math.log(10.0) if x != 0.0 else math.inf


def compute_expected_density_scale_for_relu_field_grid(grid_world_size) -> float:
    """rho = (sqrt(27) * 100 / |diagonal|) / 3 (reference rendering/volumetric/utils/misc.py:68-78).
    33.333... for the default 3x3x3 world."""
    diagonal = float(np.sqrt(np.sum([float(e) ** 2 for e in grid_world_size])))
    return ((float(np.sqrt(3.0**3)) * 100.0) / diagonal) / 3


def compute_thre3d_grid_sizes(final_required_resolution, num_stages: int, scale_factor: float):
    """Stage-wise grid sizes, coarse to fine (reference utils/misc.py:38-50)."""
    dims = tuple(int(v) for v in final_required_resolution)
    sizes = [dims]
    for _ in range(num_stages - 1):
        dims = tuple(int(np.ceil((1 / scale_factor) * v)) for v in dims)
        sizes.insert(0, dims)
    return sizes
