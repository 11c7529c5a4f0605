"""Reference-WRITTEN checkpoint fixture -- runs ONLY in the build container, where /root/reference exists.

Builds a small VolumetricModel out of the reference's own classes (VoxelGrid, render_sh_voxel_grid, SHVoxGridRenderConfig),
saves it exactly the way the reference's trainer does (modules/trainers.py:439-455: torch.save(vol_mod.get_save_info(extra_info))),
and renders one view with the reference renderer:

    tests/golden/reference_checkpoint.pth   the pickle a reference user has on disk: tensors + pickled-BY-NAME references to
                                            thre3d_atom.thre3d_reprs.renderers.render_sh_voxel_grid / SHVoxGridRenderConfig,
                                            thre3d_atom.thre3d_reprs.voxels.VoxelSize / VoxelGridLocation,
                                            thre3d_atom.utils.imaging_utils.CameraBounds / CameraIntrinsics,
                                            thre3d_atom.rendering.volumetric.accumulate.density2occupancy_pb (data, no source text)
    tests/golden/reference_checkpoint_render.npz  what the reference renders from it (colour / depth / acc of a 12 x 14 view)

    python oracle/gen_golden_checkpoint.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
_easydict = types.ModuleType("easydict")
_easydict.EasyDict = dict
sys.modules.setdefault("easydict", _easydict)

from tests.helpers import GOLDEN_DIR, hash_uniform  # noqa: E402

from thre3d_atom.modules.volumetric_model import VolumetricModel, create_volumetric_model_from_saved_model  # noqa: E402
from thre3d_atom.thre3d_reprs.renderers import SHVoxGridRenderConfig, render_sh_voxel_grid  # noqa: E402
from thre3d_atom.thre3d_reprs.voxels import VoxelGrid, VoxelGridLocation, VoxelSize, create_voxel_grid_from_saved_info_dict  # noqa: E402
from thre3d_atom.utils.constants import CAMERA_BOUNDS, CAMERA_INTRINSICS, EXTRA_ACCUMULATED_WEIGHTS, HEMISPHERICAL_RADIUS  # noqa: E402
from thre3d_atom.utils.imaging_utils import CameraBounds, CameraIntrinsics, pose_spherical  # noqa: E402

DIMS, F = (6, 5, 7), 12  # SH degree 1
dens = torch.from_numpy(hash_uniform((*DIMS, 1), 4242))
feat = torch.from_numpy(hash_uniform((*DIMS, F), 4243))
grid = VoxelGrid(
    densities=dens, features=feat, voxel_size=VoxelSize(0.5, 0.6, 0.43), grid_location=VoxelGridLocation(0.05, -0.1, 0.0),
    density_preactivation=torch.nn.Identity(), density_postactivation=torch.nn.ReLU(), expected_density_scale=20.0, tunable=True,
)
bounds = CameraBounds(1.8, 6.6)
intr = CameraIntrinsics(12, 14, 17.5)
cfg = SHVoxGridRenderConfig(num_samples_per_ray=40, camera_bounds=bounds, perturb_sampled_points=False, white_bkgd=True, render_num_samples_per_ray=64)
vol_mod = VolumetricModel(thre3d_repr=grid, render_procedure=render_sh_voxel_grid, render_config=cfg, device=torch.device("cpu"))
extra = {CAMERA_BOUNDS: bounds, CAMERA_INTRINSICS: intr, HEMISPHERICAL_RADIUS: 4.0311}
path = os.path.join(GOLDEN_DIR, "reference_checkpoint.pth")
torch.save(vol_mod.get_save_info(extra_info=extra), path)

# what a reference user gets back from it, rendered by the reference
_load = torch.load
torch.load = lambda p, *a, **k: _load(p, *a, weights_only=False, **k)  # the reference predates the weights_only default
loaded, loaded_extra = create_volumetric_model_from_saved_model(path, create_voxel_grid_from_saved_info_dict, device=torch.device("cpu"))
torch.load = _load
pose = pose_spherical(35.0, -25.0, 4.0311)
out = loaded.render(pose, intr, gpu_render=False)
np.savez_compressed(
    os.path.join(GOLDEN_DIR, "reference_checkpoint_render.npz"),
    meta=np.array([f"torch={torch.__version__}", "reference=akanimax/thr3ed_atom@v1"]),
    colour=out.colour.numpy(), depth=out.depth.numpy(), acc=out.extra[EXTRA_ACCUMULATED_WEIGHTS].numpy(),
    rotation=pose.rotation.numpy(), translation=pose.translation.numpy(), intrinsics=np.array([intr.height, intr.width, intr.focal]),
    densities=dens.numpy(), features=feat.numpy(), radius=np.float64(loaded_extra[HEMISPHERICAL_RADIUS]),
)
print("wrote", path, os.path.getsize(path), "bytes; render", out.colour.shape, "mean colour", float(out.colour.mean()))
