"""SURVEY 8(f) rows on the GPU: stage transition (grid up-scaling) against the reference's golden output, the two CLI entry
points end to end (train a few iterations on the synthetic scene, checkpoint, render a camera path from it), and the test-set
PSNR loop against the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import thr3ed_atom_amd as rf
from thr3ed_atom_amd.trainers import PosedImagesInMemory, test_sh_vox_grid_vol_mod_with_posed_images
from oracle import relu_field_oracle as orc
from tests.helpers import REPO_ROOT, hotdog_like_camera, load_golden, procedural_grid

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("storage", ["reference", "split", "bricked"])
def test_grid_upscaling_matches_the_reference_golden(hip_device, storage):
    """scale_voxel_grid_with_required_output_size (reference voxels.py:334-373) on the GPU == the reference's own output for
    the anisotropic 5x6x7 -> 10x12x14 case (golden G3), BIT FOR BIT: rf_upsample_grid follows ATen's CPU arithmetic."""
    g = load_golden("g3_voxel_grid.npz")
    d5, f5 = procedural_grid((5, 6, 7), 27, 31)
    voxel = tuple(float(v) for v in g["aniso_voxel"])
    grid = rf.VoxelGrid(d5.to(hip_device), f5.to(hip_device), rf.VoxelSize(*voxel), rf.VoxelGridLocation(*[float(v) for v in g["aniso_loc"]]),
                        density_preactivation=torch.nn.Identity(), density_postactivation=torch.nn.ReLU(), expected_density_scale=float(g["aniso_rho"]),
                        tunable=True, storage=storage)
    up = rf.scale_voxel_grid_with_required_output_size(grid, (10, 12, 14))
    assert up.grid_dims == (10, 12, 14) and up.storage == storage and up.density_mode == "relu"
    assert np.array_equal(up.features.detach().cpu().numpy(), g["aniso_up_features"])
    assert np.array_equal(up.densities.detach().cpu().numpy(), g["aniso_up_densities"])
    np.testing.assert_allclose(np.array(up.voxel_size), g["aniso_up_voxel"], rtol=1e-12)
    # the world extent is unchanged
    for a, b in zip(up.aabb, grid.aabb):
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("src_storage,dst_dims,deg", [("split", (13, 9, 17), 2), ("bricked", (16, 16, 16), 1), ("reference", (7, 20, 5), 0),
                                                       ("bricked", (4, 3, 2), 2), ("split", (6, 7, 8), 3)])
def test_grid_resampling_equals_torch_cpu_interpolate(hip_device, src_storage, dst_dims, deg):
    """Non-integer ratios, down-sampling, every SH degree and storage: rf_upsample_grid == F.interpolate(mode='trilinear',
    align_corners=False) of the unified volume on the CPU (what the reference's function computes), bit for bit."""
    F = 3 * (deg + 1) ** 2
    d, f = procedural_grid((6, 7, 8), F, 5 + deg)
    grid = rf.VoxelGrid(d.to(hip_device), f.to(hip_device), rf.VoxelSize(0.5, 0.4, 0.3), density_preactivation=torch.nn.Identity(),
                        density_postactivation=torch.nn.ReLU(), expected_density_scale=3.0, tunable=False, storage=src_storage)
    up = rf.scale_voxel_grid_with_required_output_size(grid, dst_dims)
    unified = torch.cat([f, d], dim=-1)
    # the oracle's numpy restatement of ATen's channels-last CPU kernel with its 8-wide (AVX2) vector body -- itself pinned to
    # F.interpolate and to golden G3 by the CPU suite; a host whose torch dispatches 16-wide kernels sums channels 16..23 of a
    # 28-channel volume in the other order (1 ulp), so the live CPU result is only required to agree to that
    ref = torch.from_numpy(orc.trilinear_upsample_recipe(unified.numpy(), dst_dims, vector_width=8))
    assert torch.equal(up.features.detach().cpu(), ref[..., :-1])
    assert torch.equal(up.densities.detach().cpu(), ref[..., -1:])
    live = torch.nn.functional.interpolate(unified.permute(3, 0, 1, 2)[None], size=dst_dims, mode="trilinear", align_corners=False)[0].permute(1, 2, 3, 0)
    torch.testing.assert_close(ref, live, rtol=0, atol=5e-7)
    assert up.grid_dims == tuple(dst_dims) and up.storage == src_storage


def _run(args, timeout=600):
    env = dict(os.environ, PYTHONPATH=REPO_ROOT)
    return subprocess.run([sys.executable] + args, cwd=REPO_ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_cli_train_then_render_round_trip(hip_device, tmp_path):
    """scripts/train_sh_based_voxel_grid.py (the reference's option names) on the synthetic scene for two short stages, then
    scripts/render_sh_based_voxel_grid.py on its final checkpoint -- and on the reference-written checkpoint fixture."""
    out = tmp_path / "run"
    r = _run(["scripts/train_sh_based_voxel_grid.py", "-o", str(out), "--synthetic", "True", "--synthetic_size", "48", "--grid_dims", "32", "32", "32",
              "--sh_degree", "2", "--ray_batch_size", "2048", "--train_num_samples_per_ray", "64", "--render_num_samples_per_ray", "64",
              "--num_stages", "2", "--num_iterations_per_stage", "30", "--save_frequency", "1000", "--test_frequency", "30", "--summary_frequency", "10",
              "--num_workers", "2", "--feedback_frequency", "7", "--fast_debug_mode", "False"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert "training stage: 2" in r.stdout and "TEST SET PSNR" in r.stdout
    losses = [float(line.split("specular_loss: ")[1].split()[0]) for line in r.stdout.splitlines() if "specular_loss: " in line]
    assert losses[-1] < losses[0]
    ckpt = out / "saved_models" / "model_final.pth"
    assert ckpt.exists()
    model, extra = rf.create_volumetric_model_from_saved_model(ckpt, rf.create_voxel_grid_from_saved_info_dict, device=hip_device)
    assert model.thre3d_repr.grid_dims == (32, 32, 32) and model.thre3d_repr.storage == "reference"
    assert sorted(extra) == ["camera_bounds", "camera_intrinsics", "hemispherical_radius"]
    for tag, path, extra_args in (("own", ckpt, ["--camera_path", "thre360", "--num_frames", "4"]),
                                  ("reference", os.path.join(REPO_ROOT, "tests", "golden", "reference_checkpoint.pth"),
                                   ["--camera_path", "spiral", "--num_frames", "4", "--num_spiral_rounds", "1"])):
        frames = tmp_path / f"frames_{tag}"
        r = _run(["scripts/render_sh_based_voxel_grid.py", "-i", str(path), "-o", str(frames), "--overridden_num_samples_per_ray", "48",
                  "--render_scale_factor", "1.5", "--fps", "30"] + extra_args)
        assert r.returncode == 0, r.stderr[-2000:]
        files = sorted(os.listdir(frames))
        assert len(files) == 3 and files[0].startswith("frame_0000")  # num_frames - 1 poses (the loop-closing pose is dropped)
    r = _run(["scripts/train_sh_based_voxel_grid.py", "-o", str(out), "--normalize_scene_scale", "True"])
    assert r.returncode != 0 and "disk dataset loader" in r.stderr


def test_test_set_psnr_loop_against_the_oracle(hip_device):
    """test_sh_vox_grid_vol_mod_with_posed_images (reference modules/testers.py:17-71, PSNR part): mean PSNR over held-out
    views rendered with render_num_samples_per_ray samples, against the oracle rendering the same views on the CPU."""
    cam = hotdog_like_camera()
    G, hw = 16, 20
    dens, feat = procedural_grid((G, G, G), 12, 91)
    grid = rf.VoxelGrid(dens.to(hip_device), feat.to(hip_device), rf.VoxelSize(3.0 / G, 3.0 / G, 3.0 / G), density_preactivation=torch.nn.Identity(),
                        density_postactivation=torch.nn.ReLU(), expected_density_scale=100.0 / 3.0, storage="split")
    bounds = rf.CameraBounds(cam["near"], cam["far"])
    cfg = rf.SHVoxGridRenderConfig(24, bounds, perturb_sampled_points=False, white_bkgd=True, render_num_samples_per_ray=56)
    model = rf.VolumetricModel(grid, rf.render_sh_voxel_grid, cfg, device=hip_device)
    intr = rf.CameraIntrinsics(hw, hw, 27.0)
    poses = [rf.pose_spherical(70.0 * k, -35.0, cam["radius"]) for k in range(3)]
    gen = torch.Generator().manual_seed(4)
    images = torch.rand((3, 3, hw, hw), generator=gen)
    pose_mat = torch.stack([torch.cat([p.rotation, p.translation], dim=1) for p in poses])
    data = PosedImagesInMemory(images.to(hip_device), pose_mat.to(hip_device), intr, bounds)
    ours = test_sh_vox_grid_vol_mod_with_posed_images(model, data)
    aabb = orc.make_aabb((G,) * 3, (3.0 / G,) * 3)
    psnrs = []
    for k, pose in enumerate(poses):
        o, d = orc.cast_rays(hw, hw, intr.focal, pose.rotation, pose.translation)
        col = orc.render(dens, feat, o.reshape(-1, 3), d.reshape(-1, 3), aabb, cam["near"], cam["far"], 56, 100.0 / 3.0, "relu", white_bkgd=True)["colour"]
        mse = torch.nn.functional.mse_loss(col.reshape(hw, hw, 3), images[k].permute(1, 2, 0))
        psnrs.append(float(-10.0 * torch.log10(mse)))
    assert abs(ours - float(np.mean(psnrs))) < 1e-3, (ours, psnrs)


def test_bench_line_contract(hip_device):
    """bench.py the way the driver runs it (fewer steps): ONE JSON line with the contract's keys, the roofline object priced on the
    launch's algorithmic bytes (a fraction in (0, 1]; the counter bytes beside it), the CPU baseline measured by the oracle, nothing
    above the HBM peak anywhere, and the line consistent with itself: no launch longer than the step it is part of, no kernel longer
    than the synchronised frame call that contains it."""
    import json

    r = _run([os.path.join(REPO_ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--cpu-rays", "256", "--cpu-fwd-rays", "512",
              "--render-frames", "1", "--highres-frames", "0", "--dropin-steps", "2", "--second-point-rays", "20000"], timeout=900)
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 2 and line["higher_is_better"] is True
    assert line["unit"] == "ray-samples/s" and line["dtype"] == "f32" and line["data"] == "synthetic" and line["vs_baseline"] is None
    assert "workload" in line["config"] and "model" not in line["config"]
    np.testing.assert_allclose(line["value"], 2 * 16384 * 256 / (line["ms_per_step"] * 1e-3), rtol=1e-6)
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert 0.0 < roof["frac"] <= 1.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert roof["frac"] == roof["frac_processed"]  # run-only figure first; the counter-based one is `frac_fabric_counters`
    assert roof["frac_fabric_counters"] is None or 0.0 < roof["frac_fabric_counters"] <= 1.0
    for name, rec in line["kernels"].items():
        assert rec["avg_ms"] <= line["ms_per_step"], (name, rec, line["ms_per_step"])
    assert sum(rec["avg_ms"] for rec in line["kernels"].values()) <= 1.05 * line["ms_per_step"] + 0.05
    for leg in ("init_field", "traversal"):
        fr = line["fwd_render"][leg]
        assert fr["kernel_ms_per_frame"] <= 1.02 * fr["ms_per_frame"], (leg, fr["kernel_ms_per_frame"], fr["ms_per_frame"])
    sp = line["second_weak_scaling_point"]
    assert sp["rays_per_gpu_per_step"] == 20000 and sp["ms_per_step"] > 0
    np.testing.assert_allclose(sp["value"], 2 * 20000 * 256 / (sp["ms_per_step"] * 1e-3), rtol=1e-6)
    win = line["ms_per_step_windows"]
    assert win["windows"] == 9 and win["steps_per_window"] == 3 and 0.0 < win["min"] <= win["median"] <= win["max"]
    # the counter table is tied to the kernel source by hash: a stale table is never used (the fraction is then priced on the algorithmic bytes)
    assert (roof["traffic"] is None and roof["traffic_stale"]) or (roof["traffic"] > 0 and not roof["traffic_stale"])
    for rec in roof["by_kernel"].values():
        assert rec.get("frac_hbm") is None or 0.0 < rec["frac_hbm"] <= 1.0
    assert line["roofline_model_errors"] == []
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["unit"] == "ray-samples/s" and cpu["value"] > 0 and cpu["cores"] >= 1 and "sample" in cpu
    assert cpu["cfg1_full_frame"]["max_abs_colour_difference_gpu_vs_cpu"] <= 1e-5
    assert line["fwd_render"]["init_field"]["render_launches_per_frame"] == 1
