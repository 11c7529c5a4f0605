"""The reference-side binding of INTEGRATION.md (integration/renderers_hip.py): the file a maintainer of the reference would add.

CPU part (here): its struct mirrors have the sizes the library reports, and -- when the reference checkout is present (the build
container only) -- it imports against the REAL reference package and describes a real reference VoxelGrid.
GPU part: the binding and the package's own duck-typed procedure render a stand-in module that has exactly the reference
VoxelGrid's attributes, against the golden vectors G7 (outputs of the reference renderer) including gradients."""
import ctypes as C
import importlib.util
import os
import sys
import types

import numpy as np
import pytest
import torch

from tests.helpers import REPO_ROOT, hotdog_like_camera, load_golden, procedural_grid
from thr3ed_atom_amd import _lib

BINDING = os.path.join(REPO_ROOT, "integration", "renderers_hip.py")
REFERENCE = "/root/reference"


def _load_binding(monkeypatch, with_reference: bool):
    """import integration/renderers_hip.py either against the real reference package or against stand-ins for the three
    reference names it imports (this package's Rays / RenderOut have the same fields)."""
    monkeypatch.setenv("RELU_FIELD_HIP_LIB", _lib.LIB_PATH)
    if with_reference:
        monkeypatch.syspath_prepend(REFERENCE)
        if "easydict" not in sys.modules:  # imported by the reference for a type annotation only
            ed = types.ModuleType("easydict")
            ed.EasyDict = dict
            monkeypatch.setitem(sys.modules, "easydict", ed)
    else:
        import thr3ed_atom_amd as rf
        from thr3ed_atom_amd import constants

        names = ["thre3d_atom", "thre3d_atom.rendering", "thre3d_atom.rendering.volumetric", "thre3d_atom.rendering.volumetric.render_interface",
                 "thre3d_atom.utils", "thre3d_atom.utils.constants"]
        mods = {n: types.ModuleType(n) for n in names}
        mods["thre3d_atom.rendering.volumetric.render_interface"].Rays = rf.Rays
        mods["thre3d_atom.rendering.volumetric.render_interface"].RenderOut = rf.RenderOut
        mods["thre3d_atom.utils.constants"].EXTRA_DISPARITY = constants.EXTRA_DISPARITY
        mods["thre3d_atom.utils.constants"].EXTRA_ACCUMULATED_WEIGHTS = constants.EXTRA_ACCUMULATED_WEIGHTS
        for n, m in mods.items():
            monkeypatch.setitem(sys.modules, n, m)
    spec = importlib.util.spec_from_file_location("renderers_hip_under_test", BINDING)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_binding_struct_mirrors_match_the_library(monkeypatch):
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    rh = _load_binding(monkeypatch, with_reference=False)
    lib = rh._library()
    lib.rf_abi_struct_size.argtypes = [C.c_int]
    for which, mirror in ((0, rh.RFGrid), (1, rh.RFRayBatch), (2, rh.RFRenderOut), (3, rh.RFRenderGrads), (4, rh.RFBrickList), (6, rh.RFCamera), (8, rh.RFPassScratch)):
        assert lib.rf_abi_struct_size(which) == C.sizeof(mirror) == C.sizeof(_lib.ABI_STRUCTS[which])
    assert lib.rf_abi_struct_size(99) == -1
    for which, mirror in enumerate(_lib.ABI_STRUCTS):  # the package's own mirrors (also checked at load time)
        assert lib.rf_abi_struct_size(which) == C.sizeof(mirror)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "thre3d_atom")), reason="reference checkout not present (GPU box)")
def test_binding_imports_against_the_real_reference_and_describes_its_grid(monkeypatch):
    rh = _load_binding(monkeypatch, with_reference=True)
    from thre3d_atom.thre3d_reprs.voxels import VoxelGrid as RefGrid, VoxelGridLocation as RefLoc, VoxelSize as RefSize

    g = RefGrid(torch.rand(4, 5, 6, 1), torch.rand(4, 5, 6, 27), RefSize(0.1, 0.2, 0.3), RefLoc(0.5, 0.0, -0.25),
                density_preactivation=torch.nn.Identity(), density_postactivation=torch.nn.ReLU(), expected_density_scale=7.0, tunable=True)
    d = rh._describe_grid(g, g.densities, g.features)
    assert list(d.dims) == [4, 5, 6] and d.num_features == 27 and d.layout == 0 and d.density_mode == 0
    assert abs(d.aabb_min[0] - 0.3) < 1e-6 and abs(d.aabb_max[2] - 0.65) < 1e-6
    # the package's own duck-typed view describes the same object identically
    from thr3ed_atom_amd.voxels import as_kernel_grid

    view = as_kernel_grid(g)
    assert view.grid_dims == (4, 5, 6) and view.density_mode == "relu" and view.sh_degree == 2 and view.storage == "reference"
    from thr3ed_atom_amd.camera import slack_range_map

    for a in range(3):
        scale, bias = slack_range_map(tuple(g.aabb[a]))
        assert d.norm_scale[a] == float(scale) and d.norm_bias[a] == float(bias)
    with pytest.raises(RuntimeError, match="HIP device"):
        from thre3d_atom.rendering.volumetric.render_interface import Rays as RefRays
        from thre3d_atom.thre3d_reprs.renderers import SHVoxGridRenderConfig as RefCfg
        from thre3d_atom.utils.imaging_utils import CameraBounds as RefBounds

        rh.render_sh_voxel_grid_hip(g, RefRays(torch.zeros(3, 3), torch.ones(3, 3)), RefCfg(8, RefBounds(0.5, 4.0)))


class _ReferenceLikeGrid(torch.nn.Module):
    """A module with exactly the attributes the reference's VoxelGrid has (thre3d_reprs/voxels.py:93-124) and none of this
    package's: what `VolumetricModel(thre3d_repr=<reference grid>, ...)` hands to a render procedure."""

    def __init__(self, densities, features, voxel_size, location, pre, post, rho):
        super().__init__()
        self._densities = torch.nn.Parameter(densities)
        self._features = torch.nn.Parameter(features)
        self._density_preactivation, self._density_postactivation = pre, post
        self._feature_preactivation = self._feature_postactivation = torch.nn.Identity()
        self._radiance_transfer_function = None
        self._grid_location, self._voxel_size, self._expected_density_scale, self._tunable = location, voxel_size, rho, True
        self.width_x, self.depth_y, self.height_z = densities.shape[:3]
        half = [n * v / 2 for n, v in zip(densities.shape[:3], voxel_size)]
        self._aabb = tuple((c - h, c + h) for c, h in zip(location, half))

    densities = property(lambda self: self._densities)
    features = property(lambda self: self._features)
    aabb = property(lambda self: self._aabb)


@pytest.mark.gpu
@pytest.mark.parametrize("via", ["binding", "binding-binned", "package"])
@pytest.mark.parametrize("variant", ["relu_spec", "relu_diffuse", "relu_opt", "abs_spec", "softplus_spec"])
def test_reference_like_module_renders_like_the_reference(hip_device, monkeypatch, via, variant):
    """golden G7 (16^3, SH degree 2): colour / depth / acc and both gradients of L1(colour, target), through
    (a) integration/renderers_hip.py and (b) thr3ed_atom_amd.render_sh_voxel_grid's duck typing."""
    import thr3ed_atom_amd as rf

    g7 = load_golden("g7_grid16_render.npz")
    cam = hotdog_like_camera()
    dens, feat = procedural_grid((16, 16, 16), 27, 81)
    acts = {"abs_spec": (torch.abs, torch.nn.Identity()), "softplus_spec": (torch.nn.Identity(), torch.nn.Softplus())}.get(variant, (torch.nn.Identity(), torch.nn.ReLU()))
    rho = 1.0 if variant == "abs_spec" else float(g7["rho"])
    grid = _ReferenceLikeGrid(dens.to(hip_device), feat.to(hip_device), (3.0 / 16,) * 3, (0.0, 0.0, 0.0), acts[0], acts[1], rho).to(hip_device)
    cfg = rf.SHVoxGridRenderConfig(48, rf.CameraBounds(cam["near"], cam["far"]), perturb_sampled_points=False, white_bkgd=True,
                                   render_diffuse=variant == "relu_diffuse", optimized_sampling=variant == "relu_opt")
    rays = rf.Rays(torch.from_numpy(g7["origins"]).to(hip_device), torch.from_numpy(g7["directions"]).to(hip_device))
    if via.startswith("binding"):
        # ("binding-binned": the atomic-free adjoint the binding picks for training-size renders, forced here on the small one)
        monkeypatch.setenv("RELU_FIELD_HIP_BACKWARD", "binned" if via == "binding-binned" else "atomic")
        rh = _load_binding(monkeypatch, with_reference=False)
        out = rh.render_sh_voxel_grid_hip(grid, rays, cfg)
    else:
        out = rf.render_sh_voxel_grid(grid, rays, cfg)
    torch.nn.functional.l1_loss(out.colour, torch.from_numpy(g7["target"]).to(hip_device)).backward()
    np.testing.assert_allclose(out.colour.detach().cpu().numpy(), g7[f"{variant}_colour"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(out.extra["accumulated_weight"].detach().cpu().numpy(), g7[f"{variant}_acc"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(out.depth.detach().cpu().numpy(), g7[f"{variant}_depth"], rtol=0, atol=1e-5)
    for ours, key in ((grid.densities.grad, f"{variant}_gd"), (grid.features.grad, f"{variant}_gf")):
        ref = g7[key]
        np.testing.assert_allclose(ours.cpu().numpy(), ref, rtol=2e-4, atol=2e-6 * np.abs(ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("tiles", ["1", "0"])
def test_binding_frame_entry_equals_the_ray_list_procedure(hip_device, monkeypatch, tiles):
    """integration/renderers_hip.py::render_frame_hip -- VolumetricModel.render's frame loop (modules/volumetric_model.py:143-172) as
    one library call on the reference's own grid type: rays generated in-kernel, the grid gathered from a split-layout copy that
    rf_convert_grid keeps in step with the module's tensors, ray packets ($RF_FRAME_TILES=1) or the per-ray kernel (=0) -- against the
    binding's own ray-list procedure on cast_rays' rays (summation order), whole frame and a pixel range, and after an in-place edit
    of the grid (the copy must follow)."""
    import thr3ed_atom_amd as rf

    cam = hotdog_like_camera()
    dens, feat = procedural_grid((16, 16, 16), 27, 81)
    grid = _ReferenceLikeGrid(dens.to(hip_device), feat.to(hip_device), (3.0 / 16,) * 3, (0.0, 0.0, 0.0), torch.nn.Identity(), torch.nn.ReLU(), 100.0 / 3.0).to(hip_device)
    cfg = rf.SHVoxGridRenderConfig(48, rf.CameraBounds(cam["near"], cam["far"]), perturb_sampled_points=False, white_bkgd=True)
    intr = rf.CameraIntrinsics(37, 45, 70.0)
    pose = rf.pose_spherical(25.0, -35.0, cam["radius"])
    rays = rf.flatten_rays(rf.cast_rays(intr, pose, hip_device))
    monkeypatch.setenv("RF_FRAME_TILES", tiles)
    rh = _load_binding(monkeypatch, with_reference=False)
    for edit in (False, True):
        if edit:
            with torch.no_grad():
                grid._densities.mul_(0.5)  # (in place: the version counter moves, the split copy has to be refreshed)
        with torch.no_grad():
            ref = rh.render_sh_voxel_grid_hip(grid, rays, cfg)
            frame = rh.render_frame_hip(grid, tuple(intr), (pose.rotation, pose.translation), cfg)
            part = rh.render_frame_hip(grid, tuple(intr), (pose.rotation, pose.translation), cfg, first_ray=300, num_rays=500)
        assert frame.colour.shape == (37, 45, 3) and frame.depth.shape == (37, 45, 1)
        assert float((frame.colour.reshape(-1, 3) - ref.colour).abs().max()) <= 2e-6
        assert float((frame.depth.reshape(-1, 1) - ref.depth).abs().max()) <= 2e-5
        assert float((frame.extra["accumulated_weight"].reshape(-1, 1) - ref.extra["accumulated_weight"]).abs().max()) <= 2e-6
        assert torch.equal(part.colour, frame.colour.reshape(-1, 3)[300:800]) and torch.equal(part.depth, frame.depth.reshape(-1, 1)[300:800])
    assert float(frame.colour.min()) < 0.95


@pytest.mark.gpu
def test_binding_pair_procedure_equals_two_calls_and_follows_golden_g7(hip_device, monkeypatch):
    """integration/renderers_hip.py::render_sh_voxel_grid_pair_hip -- the two renders of modules/trainers.py:306, 323-325 as ONE autograd node
    (rf_render_forward_pair; backward: rf_bin_offsets_pair + rf_render_backward_emit_direct_pair + one rf_brick_accumulate over both lists)
    -- on the reference's own grid type: the outputs of the two single calls bit for bit (the same torch.rand draws in the same order), the
    gradient of L1 + L1 like theirs and, jitter off, like the reference's own (golden G7: specular + diffuse gradients added)."""
    import thr3ed_atom_amd as rf

    monkeypatch.setenv("RELU_FIELD_HIP_BACKWARD", "binned")
    rh = _load_binding(monkeypatch, with_reference=False)
    g7 = load_golden("g7_grid16_render.npz")
    cam = hotdog_like_camera()
    dens, feat = procedural_grid((16, 16, 16), 27, 81)
    rays = rf.Rays(torch.from_numpy(g7["origins"]).to(hip_device), torch.from_numpy(g7["directions"]).to(hip_device))
    target = torch.from_numpy(g7["target"]).to(hip_device)

    def run(paired, perturb):
        grid = _ReferenceLikeGrid(dens.to(hip_device), feat.to(hip_device), (3.0 / 16,) * 3, (0.0, 0.0, 0.0), torch.nn.Identity(), torch.nn.ReLU(), float(g7["rho"])).to(hip_device)
        cfg = rf.SHVoxGridRenderConfig(48, rf.CameraBounds(cam["near"], cam["far"]), perturb_sampled_points=perturb, white_bkgd=True)
        torch.manual_seed(9)
        if paired:
            spec, diff = rh.render_sh_voxel_grid_pair_hip(grid, rays, cfg)
            assert spec.colour.grad_fn is diff.colour.grad_fn
        else:
            import dataclasses

            spec = rh.render_sh_voxel_grid_hip(grid, rays, cfg)
            diff = rh.render_sh_voxel_grid_hip(grid, rays, dataclasses.replace(cfg, render_diffuse=True))
        (torch.nn.functional.l1_loss(spec.colour, target) + torch.nn.functional.l1_loss(diff.colour, target)).backward()
        return spec, diff, grid.densities.grad.cpu().numpy(), grid.features.grad.cpu().numpy()

    for perturb in (True, False):
        a, b = run(True, perturb), run(False, perturb)
        for x, y in ((a[0], b[0]), (a[1], b[1])):
            assert torch.equal(x.colour, y.colour) and torch.equal(x.depth, y.depth) and torch.equal(x.extra["accumulated_weight"], y.extra["accumulated_weight"])
        for x, y in ((a[2], b[2]), (a[3], b[3])):
            np.testing.assert_allclose(x, y, rtol=3e-4, atol=3e-6 * np.abs(y).max())
    np.testing.assert_allclose(a[0].colour.detach().cpu().numpy(), g7["relu_spec_colour"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(a[1].colour.detach().cpu().numpy(), g7["relu_diffuse_colour"], rtol=0, atol=1e-5)
    for ours, ref in ((a[2], g7["relu_spec_gd"] + g7["relu_diffuse_gd"]), (a[3], g7["relu_spec_gf"] + g7["relu_diffuse_gf"])):
        np.testing.assert_allclose(ours, ref, rtol=2e-4, atol=2e-6 * np.abs(ref).max())


@pytest.mark.gpu
def test_binding_split_shadow_hooks(hip_device, monkeypatch):
    """The split-layout copy that forward passes of the binding gather from follows the module's tensors through data pointers and version
    counters; a write that bumps neither (``p.data.mul_``) needs ``invalidate_split_shadow``; ``release_split_shadow`` frees the copy."""
    import thr3ed_atom_amd as rf

    rh = _load_binding(monkeypatch, with_reference=False)
    g7 = load_golden("g7_grid16_render.npz")
    cam = hotdog_like_camera()
    dens, feat = procedural_grid((16, 16, 16), 27, 81)
    grid = _ReferenceLikeGrid(dens.to(hip_device), feat.to(hip_device), (3.0 / 16,) * 3, (0.0, 0.0, 0.0), torch.nn.Identity(), torch.nn.ReLU(), float(g7["rho"])).to(hip_device)
    cfg = rf.SHVoxGridRenderConfig(48, rf.CameraBounds(cam["near"], cam["far"]), perturb_sampled_points=False, white_bkgd=True)
    rays = rf.Rays(torch.from_numpy(g7["origins"]).to(hip_device), torch.from_numpy(g7["directions"]).to(hip_device))
    with torch.no_grad():
        first = rh.render_sh_voxel_grid_hip(grid, rays, cfg).colour.clone()
        grid._features.data.mul_(-1.0)  # (no version bump, same storage)
        stale = rh.render_sh_voxel_grid_hip(grid, rays, cfg).colour.clone()
        assert torch.equal(stale, first)  # the documented constraint ...
        rh.invalidate_split_shadow(grid)
        fresh = rh.render_sh_voxel_grid_hip(grid, rays, cfg).colour.clone()
        assert not torch.equal(fresh, first)  # ... and its remedy
        assert grid in rh._FRAME_SHADOWS
        rh.release_split_shadow(grid)
        assert grid not in rh._FRAME_SHADOWS
        assert torch.equal(rh.render_sh_voxel_grid_hip(grid, rays, cfg).colour, fresh)
