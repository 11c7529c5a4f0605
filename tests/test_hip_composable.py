"""The path at the granularity of the reference's plug-in points (thr3ed_atom_amd/composable.py): samplers, point processor,
accumulator and their composition, against the golden outputs of the reference's own functions -- G2 (sample.py), G4
(spherical_harmonics.py), G5/G6 (process.py, accumulate.py incl. its debug outputs), G11 (stochastic density noise), G12 (non-default
density2occupancy / tone map / activation callables, with gradients).  HIP interpolation in the middle, torch around it, on the device."""
import numpy as np
import pytest
import torch

import thr3ed_atom_amd as rf
from thr3ed_atom_amd import composable as cp
from tests.helpers import load_golden, procedural_grid

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def make_grid(dev, dens, feat, G, rho, tunable=False, storage="reference", **acts):
    kw = dict(density_preactivation=torch.nn.Identity(), density_postactivation=torch.nn.ReLU())
    kw.update(acts)
    return rf.VoxelGrid(dens.clone().to(dev), feat.clone().to(dev), rf.VoxelSize(3.0 / G, 3.0 / G, 3.0 / G), expected_density_scale=rho, tunable=tunable, storage=storage, **kw)


def test_samplers_against_the_references(hip_device):
    g = load_golden("g2_sampling.npz")
    rays = rf.Rays(T(g["origins"]).to(hip_device), T(g["directions"]).to(hip_device))
    bounds = rf.CameraBounds(float(g["near"]), float(g["far"]))
    plain = cp.sample_uniform_points_on_rays(rays, bounds, 32, perturb=False)
    assert torch.equal(plain.depths.cpu(), T(g["z_plain"])) and torch.equal(plain.points.cpu(), T(g["pts_plain"]))
    jit = cp.sample_uniform_points_on_rays(rays, bounds, 32, perturb=True, t_rand=T(g["t_rand"]).to(hip_device))
    np.testing.assert_allclose(jit.depths.cpu().numpy(), g["z_jitter"], rtol=0, atol=5e-7)
    np.testing.assert_allclose(jit.points.cpu().numpy(), g["pts_jitter"], rtol=0, atol=2e-6)
    drawn = cp.sample_uniform_points_on_rays(rays, bounds, 32, perturb=True)  # its own torch.rand: inside the strata, increasing
    assert bool((drawn.depths[:, 1:] >= drawn.depths[:, :-1]).all()) and float((drawn.depths - plain.depths).abs().max()) < (bounds.far - bounds.near) / 31
    # the box-bounded sampler: uniform between the reference's per-ray entry / exit parameters
    box = rf.Rays(T(g["aabb_origins"]).to(hip_device), T(g["aabb_directions"]).to(hip_device))
    aabb = tuple(tuple(float(v) for v in r) for r in g["aabb"])
    got = cp.sample_aabb_bound_uniform_points_on_rays(box, bounds, 16, aabb, perturb=False)
    t = torch.linspace(0.0, 1.0, 16)
    want = T(g["aabb_bounds"])[:, :1] * (1.0 - t) + T(g["aabb_bounds"])[:, 1:] * t
    np.testing.assert_allclose(got.depths.cpu().numpy(), want.numpy(), rtol=0, atol=1e-6)
    disp = cp.sample_uniform_points_on_rays(rays, bounds, 8, perturb=False, linear_disparity_sampling=True)
    assert abs(float(disp.depths[0, 0]) - bounds.near) < 1e-5 and abs(float(disp.depths[0, -1]) - bounds.far) < 1e-5
    with pytest.raises(RuntimeError):
        cp.sample_uniform_points_on_rays(rf.Rays(T(g["origins"]), T(g["directions"])), bounds, 8)


def test_sh_basis_against_the_references(hip_device):
    g = load_golden("g4_sh.npz")
    v = T(g["viewdirs"]).to(hip_device)
    for deg in range(4):
        rad = (T(g[f"coeffs{deg}"]).to(hip_device) * cp.sh_basis(deg, v)[:, None, :]).sum(-1)
        np.testing.assert_allclose(rad.cpu().numpy(), g[f"radiance{deg}"], rtol=0, atol=3e-6)


def test_point_processor_and_accumulator_against_the_references(hip_device):
    g = load_golden("g5_g6_process_accumulate.npz")
    dens, feat = procedural_grid((8, 8, 8), 27, 61)
    rays = rf.Rays(T(g["origins"]).to(hip_device), T(g["directions"]).to(hip_device))
    z = T(g["z"]).to(hip_device)
    pts = cp.SampledPointsOnRays(rays.origins[:, None, :] + rays.directions[:, None, :] * z[:, :, None], z)
    for storage in ("reference", "split"):
        grid = make_grid(hip_device, dens, feat, 8, float(g["rho"]), storage=storage)
        for tag, diffuse in (("specular", False), ("diffuse", True)):
            proc = cp.process_points_with_sh_voxel_grid(pts, rays, grid, render_diffuse=diffuse, parallel_points_chunk_size=700 if diffuse else None)
            ref = g[f"processed_{tag}"]
            assert np.array_equal(proc.points.cpu().numpy()[..., 3], ref[..., 3])  # densities: the bit-exact interpolation
            np.testing.assert_allclose(proc.points.cpu().numpy()[..., :3], ref[..., :3], rtol=0, atol=5e-6 * np.abs(ref[np.abs(ref) < 1e9]).max())
            for white in (False, True):
                out = cp.accumulate_radiance_density_on_rays(cp.ProcessedPointsOnRays(T(ref).to(hip_device), z), rays, stochastic_density_noise_std=0.0, white_bkgd=white,
                                                             extra_debug_info=True)
                wtag = f"{tag}_{'white' if white else 'black'}"
                np.testing.assert_allclose(out.colour.cpu().numpy(), g[f"colour_{wtag}"], rtol=0, atol=2e-6)
                np.testing.assert_allclose(out.depth.cpu().numpy(), g[f"depth_{wtag}"], rtol=0, atol=5e-6)
                np.testing.assert_allclose(out.extra["accumulated_weight"].cpu().numpy(), g[f"acc_{wtag}"], rtol=0, atol=2e-6)
                np.testing.assert_allclose(out.extra["disparity"].cpu().numpy(), g[f"disparity_{wtag}"], rtol=2e-5, atol=1e-6, equal_nan=True)
                if tag == "specular" and not white:  # the debug outputs (accumulate.py:96-107)
                    np.testing.assert_allclose(out.extra["point_occupancies"].cpu().numpy(), g["alpha"], rtol=0, atol=1e-6)
                    np.testing.assert_allclose(out.extra["point_weights"].cpu().numpy(), g["weights"], rtol=0, atol=1e-6)
                    np.testing.assert_allclose(out.extra["deltas"].cpu().numpy(), g["deltas"], rtol=1e-6)
                    assert sorted(out.extra) == ["accumulated_weight", "deltas", "disparity", "point_densities", "point_depths", "point_occupancies", "point_weights"]


def tone_map(x):
    return torch.sigmoid(2.0 * x) * 0.9 + 0.05


def saturating_occupancy(densities, deltas):
    x = densities * deltas
    return x / (1.0 + x)


@pytest.mark.parametrize("storage", ["reference", "split"])
@pytest.mark.parametrize("variant", ["tone_d2o", "feature_acts", "density_acts"])
def test_g12_non_default_plugins_render_like_the_reference(hip_device, variant, storage):
    """render_sh_voxel_grid routes what the fused kernels do not implement through the composed path: colour / depth / acc /
    disparity and the gradients of L1(colour, target) w.r.t. both grid tensors against the reference's."""
    g = load_golden("g12_plugins.npz")
    dens, feat = procedural_grid((16, 16, 16), 27, 81)
    acts, cfg_kw = {
        "tone_d2o": (dict(), dict(density2occupancy=saturating_occupancy, radiance_hdr_tone_map=tone_map)),
        "feature_acts": (dict(feature_preactivation=torch.tanh, feature_postactivation=lambda x: 1.5 * x), dict()),
        "density_acts": (dict(density_preactivation=torch.tanh, density_postactivation=torch.nn.Softplus(beta=2.0)), dict(render_diffuse=True)),
    }[variant]
    grid = make_grid(hip_device, dens, feat, 16, float(g["rho"]), tunable=True, storage=storage, **acts)
    cfg = rf.SHVoxGridRenderConfig(40, rf.CameraBounds(float(g["near"]), float(g["far"])), perturb_sampled_points=False, white_bkgd=True, **cfg_kw)
    from thr3ed_atom_amd.renderers import fused_kernels_apply

    assert not fused_kernels_apply(grid, cfg)
    model = rf.VolumetricModel(grid, rf.render_sh_voxel_grid, cfg, device=hip_device)
    out = model.render_rays(rf.Rays(T(g["origins"]).to(hip_device), T(g["directions"]).to(hip_device)))
    loss = torch.nn.functional.l1_loss(out.colour, T(g["target"]).to(hip_device))
    loss.backward()
    np.testing.assert_allclose(out.colour.detach().cpu().numpy(), g[f"{variant}_colour"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(out.depth.detach().cpu().numpy(), g[f"{variant}_depth"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(out.extra["accumulated_weight"].detach().cpu().numpy(), g[f"{variant}_acc"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(float(loss.detach()), float(g[f"{variant}_loss"]), rtol=1e-5)
    gd, gf = grid.reference_gradients()
    for ours, key in ((gd, f"{variant}_gd"), (gf, f"{variant}_gf")):
        ref = g[key]
        np.testing.assert_allclose(ours.cpu().numpy(), ref, rtol=2e-4, atol=2e-6 * np.abs(ref).max())
    # a whole frame of such a configuration goes through the chunked path, and the fused training step refuses it
    frame = model.render(rf.pose_spherical(30.0, -30.0, 4.0311), rf.CameraIntrinsics(12, 10, 14.0), parallel_rays_chunk_size=50)
    assert frame.colour.shape == (12, 10, 3) and bool(torch.isfinite(frame.colour).all())
    from thr3ed_atom_amd.trainers import TrainStepper

    with pytest.raises(ValueError):
        TrainStepper(model, 96, 0.03, data_parallel=False).step_on(rf.Rays(T(g["origins"]).to(hip_device), T(g["directions"]).to(hip_device)), T(g["target"]).to(hip_device))


def test_g12_accumulator_with_noise_and_debug_outputs(hip_device):
    g = load_golden("g12_plugins.npz")
    rays = rf.Rays(T(g["origins"]).to(hip_device), T(g["directions"]).to(hip_device))
    pts = cp.ProcessedPointsOnRays(T(g["acc_processed"]).to(hip_device), T(g["acc_z"]).to(hip_device))
    out = cp.accumulate_radiance_density_on_rays(pts, rays, stochastic_density_noise_std=0.3, density2occupancy=saturating_occupancy, radiance_hdr_tone_map=tone_map,
                                                 white_bkgd=False, extra_debug_info=True, density_noise=T(g["acc_noise"]).to(hip_device))
    np.testing.assert_allclose(out.colour.cpu().numpy(), g["acc_colour"], rtol=2e-5, atol=2e-6, equal_nan=True)
    np.testing.assert_allclose(out.depth.cpu().numpy(), g["acc_depth"], rtol=2e-5, atol=5e-6, equal_nan=True)
    for key in ("accumulated_weight", "point_densities", "point_occupancies", "point_weights", "point_depths", "deltas"):
        np.testing.assert_allclose(out.extra[key].cpu().numpy(), g[f"acc_extra_{key}"], rtol=2e-5, atol=2e-6, equal_nan=True)


@pytest.mark.parametrize("tag,mode,over", [("relu", "relu", {}), ("relu_diffuse_black", "relu", {"render_diffuse": True, "white_bkgd": False}), ("softplus", "softplus", {})])
def test_g11_density_noise_like_the_reference(hip_device, tag, mode, over):
    """stochastic_density_noise_std != 0 (accumulate.py:58-62), fed the noise table the reference drew: the reference's output value
    for value -- including the non-finite rays (sigma + noise < 0 on the last sample, whose interval is 1e10 |d|: alpha = -inf)."""
    g = load_golden("g11_density_noise.npz")
    dens, feat = procedural_grid((16, 16, 16), 27, 81)
    post = torch.nn.ReLU() if mode == "relu" else torch.nn.Softplus()
    grid = make_grid(hip_device, dens, feat, 16, float(g["rho"]), density_postactivation=post)
    kw = dict(perturb_sampled_points=False, white_bkgd=True, stochastic_density_noise_std=float(g[f"{tag}_std"]))
    kw.update(over)
    cfg = rf.SHVoxGridRenderConfig(40, rf.CameraBounds(float(g["near"]), float(g["far"])), **kw)
    rays = rf.Rays(T(g["origins"]).to(hip_device), T(g["directions"]).to(hip_device))
    with torch.no_grad():
        out = cp.render_sh_voxel_grid_composed(grid, rays, cfg, density_noise=T(g[f"{tag}_noise"]).to(hip_device))
        drawn = rf.render_sh_voxel_grid(grid, rays, cfg)  # (its own torch.randn draw: same shapes, same share of non-finite rays)
    for ours, key in ((out.colour, "colour"), (out.depth, "depth"), (out.extra["accumulated_weight"], "acc")):
        ref = g[f"{tag}_{key}"]
        ours = ours.cpu().numpy()
        assert np.array_equal(np.isfinite(ours), np.isfinite(ref)) and np.array_equal(np.isnan(ours), np.isnan(ref))
        ok = np.isfinite(ref)
        np.testing.assert_allclose(ours[ok], ref[ok], rtol=1e-5, atol=2e-5)
    assert 0.2 < float((~torch.isfinite(drawn.extra["accumulated_weight"])).float().mean()) < 0.8


def test_render_pair_on_a_composed_configuration_is_the_two_single_renders(hip_device):
    """``render_sh_voxel_grid_pair`` / ``VolumetricModel.render_rays_pair`` (the two renders of an iteration, modules/trainers.py:306,
    323-325) on a configuration the fused kernels do not implement (a non-default tone map): the composed path, twice, with the same
    jitter draws in the same order -- outputs and gradients of the two single calls."""
    dens, feat = procedural_grid((8, 8, 8), 27, 61)
    rays = rf.Rays(T(np.tile(np.array([[0.2, 0.1, 4.0]], dtype=np.float32), (50, 1))).to(hip_device),
                   torch.nn.functional.normalize(T(np.random.RandomState(3).randn(50, 3).astype(np.float32) * 0.15 + np.array([0.0, 0.0, -1.0], dtype=np.float32)), dim=-1).to(hip_device))
    cfg = rf.SHVoxGridRenderConfig(24, rf.CameraBounds(2.0, 6.0), perturb_sampled_points=True, white_bkgd=True, radiance_hdr_tone_map=torch.tanh, jitter="torch")
    results = []
    for paired in (True, False):
        grid = make_grid(hip_device, dens, feat, 8, 10.0, tunable=True)
        model = rf.VolumetricModel(grid, rf.render_sh_voxel_grid, cfg, device=hip_device)
        torch.manual_seed(17)
        if paired:
            spec, diff = model.render_rays_pair(rays)
        else:
            spec, diff = model.render_rays(rays), model.render_rays(rays, render_diffuse=True)
        (spec.colour.square().mean() + diff.colour.mean()).backward()
        results.append((spec.colour.detach(), diff.colour.detach(), grid.densities.grad.clone(), grid.features.grad.clone()))
    for k, (a, b) in enumerate(zip(*results)):
        if k < 2:
            assert torch.equal(a, b)
        else:  # (the point query's adjoint scatters with float atomics: equal to summation order)
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-4, atol=2e-6 * float(b.abs().max()))
    assert float(results[0][2].abs().max()) > 0 and not torch.equal(results[0][0], results[0][1])
