"""A ``pickle_module`` for ``torch.load`` that resolves the names a REFERENCE-written checkpoint pickles by qualified name.

``VolumetricModel.get_save_info`` of the reference (thre3d_atom/modules/volumetric_model.py:83-97) stores the render procedure
(a function object), the render-config class and -- inside the config dictionaries -- NamedTuples and a function of the
reference package.  pickle records those as ``module.qualname`` strings; on a machine that has this package instead of
``thre3d_atom`` they are mapped to the equivalents here (same fields, same behaviour on this path).  Everything else goes through
the standard Unpickler unchanged."""
import pickle
from pickle import *  # noqa: F401,F403  (torch.load expects the pickle module's surface)

REFERENCE_NAMES = {
    ("thre3d_atom.thre3d_reprs.renderers", "render_sh_voxel_grid"): ("thr3ed_atom_amd.renderers", "render_sh_voxel_grid"),
    ("thre3d_atom.thre3d_reprs.renderers", "SHVoxGridRenderConfig"): ("thr3ed_atom_amd.renderers", "SHVoxGridRenderConfig"),
    ("thre3d_atom.thre3d_reprs.voxels", "VoxelGrid"): ("thr3ed_atom_amd.voxels", "VoxelGrid"),
    ("thre3d_atom.thre3d_reprs.voxels", "VoxelSize"): ("thr3ed_atom_amd.voxels", "VoxelSize"),
    ("thre3d_atom.thre3d_reprs.voxels", "VoxelGridLocation"): ("thr3ed_atom_amd.voxels", "VoxelGridLocation"),
    ("thre3d_atom.thre3d_reprs.voxels", "AxisAlignedBoundingBox"): ("thr3ed_atom_amd.voxels", "AxisAlignedBoundingBox"),
    ("thre3d_atom.utils.imaging_utils", "CameraBounds"): ("thr3ed_atom_amd.camera", "CameraBounds"),
    ("thre3d_atom.utils.imaging_utils", "CameraIntrinsics"): ("thr3ed_atom_amd.camera", "CameraIntrinsics"),
    ("thre3d_atom.utils.imaging_utils", "CameraPose"): ("thr3ed_atom_amd.camera", "CameraPose"),
    ("thre3d_atom.rendering.volumetric.accumulate", "density2occupancy_pb"): ("thr3ed_atom_amd.renderers", "density2occupancy_pb"),
    ("thre3d_atom.rendering.volumetric.render_interface", "Rays"): ("thr3ed_atom_amd.render_interface", "Rays"),
    ("thre3d_atom.rendering.volumetric.render_interface", "RenderOut"): ("thr3ed_atom_amd.render_interface", "RenderOut"),
}


class Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        module, name = REFERENCE_NAMES.get((module, name), (module, name))
        if module.startswith("thre3d_atom"):
            raise pickle.UnpicklingError(
                f"the checkpoint refers to {module}.{name} of the reference package, which has no counterpart on the render path of "
                f"this build (mapped names: {sorted(n for _, n in REFERENCE_NAMES)})"
            )
        return super().find_class(module, name)


def load(file, **kwargs):
    return Unpickler(file, **kwargs).load()


def loads(data, **kwargs):
    import io

    return Unpickler(io.BytesIO(data), **kwargs).load()
