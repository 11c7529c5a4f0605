"""thr3ed_atom_amd -- MI355X-native ReLU-Fields volume rendering hot path.

Host-side mirror of the reference's render interface (Rays / RenderOut / VoxelGrid /
SHVoxGridRenderConfig / render_sh_voxel_grid / VolumetricModel) over hand-written gfx950 HIP kernels
reached through the C ABI in include/relu_field.h.  Importing the package does not need a GPU; calling
the render path does, and it raises instead of falling back when the HIP library or a GPU is missing.
"""
from .camera import (  # noqa: F401
    CameraBounds,
    CameraIntrinsics,
    CameraPose,
    compute_expected_density_scale_for_relu_field_grid,
    compute_thre3d_grid_sizes,
    get_thre360_animation_poses,
    get_thre360_spiral_animation_poses,
    mse2psnr,
    pose_spherical,
    scale_camera_intrinsics,
)
from .render_interface import (  # noqa: F401
    Rays,
    RenderOut,
    collate_rays,
    collate_rendered_output,
    flatten_rays,
    reshape_rendered_output,
)
from .voxels import (  # noqa: F401
    AxisAlignedBoundingBox,
    VoxelGrid,
    VoxelGridLocation,
    VoxelSize,
    create_voxel_grid_from_saved_info_dict,
    scale_voxel_grid_with_required_output_size,
)
from .renderers import SHVoxGridRenderConfig, density2occupancy_pb, render_sh_voxel_grid, render_sh_voxel_grid_frame, render_sh_voxel_grid_pair  # noqa: F401
from .volumetric_model import VolumetricModel, cast_rays, create_volumetric_model_from_saved_model  # noqa: F401
from .composable import (  # noqa: F401  (the path at the granularity of the reference's plug-in points)
    SampledPointsOnRays,
    accumulate_radiance_density_on_rays,
    process_points_with_sh_voxel_grid,
    render,
    sample_aabb_bound_uniform_points_on_rays,
    sample_uniform_points_on_rays,
)

__version__ = "0.1.0"
