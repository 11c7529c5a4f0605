"""Golden-vector generator -- runs ONLY in the build container, where /root/reference exists.

It imports the reference (akanimax/thr3ed_atom) as a Python package, runs the reference's own
functions on deterministic inputs (tests/helpers.py) and stores inputs + outputs as small .npz
fixtures under tests/golden/.  Only arrays are stored; no reference source travels.

    python oracle/gen_golden.py            # regenerates every fixture

The reference imports ``easydict`` for one type annotation (thre3d_atom/utils/misc.py:6); the
package is not installed here, so a three-line stand-in is registered in THIS process only.
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

from tests.helpers import GOLDEN_DIR, hash_uniform, hotdog_like_camera, procedural_grid  # noqa: E402

from thre3d_atom.rendering.volumetric.accumulate import accumulate_radiance_density_on_rays  # noqa: E402
from thre3d_atom.rendering.volumetric.process import process_points_with_sh_voxel_grid  # noqa: E402
from thre3d_atom.rendering.volumetric.render_interface import Rays, SampledPointsOnRays  # noqa: E402
from thre3d_atom.rendering.volumetric.sample import (  # noqa: E402
    _ray_aabb_intersection,
    sample_uniform_points_on_rays,
)
from thre3d_atom.rendering.volumetric.utils.misc import cast_rays, flatten_rays  # noqa: E402
from thre3d_atom.rendering.volumetric.utils.spherical_harmonics import evaluate_spherical_harmonics  # noqa: E402
from thre3d_atom.modules.volumetric_model import VolumetricModel  # noqa: E402
from thre3d_atom.thre3d_reprs.renderers import SHVoxGridRenderConfig, render_sh_voxel_grid  # noqa: E402
from thre3d_atom.thre3d_reprs.voxels import (  # noqa: E402
    VoxelGrid,
    VoxelGridLocation,
    VoxelSize,
    scale_voxel_grid_with_required_output_size,
)
from thre3d_atom.utils.imaging_utils import CameraBounds, CameraIntrinsics, pose_spherical  # noqa: E402

META = np.array(
    [f"torch={torch.__version__}", f"numpy={np.__version__}", "reference=akanimax/thr3ed_atom@v1"]
)
CPU = torch.device("cpu")


def save(name, **arrays):
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(GOLDEN_DIR, name), meta=META, **out)
    size = os.path.getsize(os.path.join(GOLDEN_DIR, name))
    print(f"  wrote {name}: {size/1024:.1f} KiB, keys={sorted(out)}")


def density_kwargs(mode, rho):
    if mode == "relu":
        return dict(
            density_preactivation=torch.nn.Identity(),
            density_postactivation=torch.nn.ReLU(),
            expected_density_scale=rho,
        )
    if mode == "softplus":
        return dict(
            density_preactivation=torch.nn.Identity(),
            density_postactivation=torch.nn.Softplus(),
            expected_density_scale=rho,
        )
    return dict(
        density_preactivation=torch.abs,
        density_postactivation=torch.nn.Identity(),
        expected_density_scale=rho,
    )


def make_grid(dens, feat, voxel, loc=(0.0, 0.0, 0.0), mode="relu", rho=1.0, tunable=False):
    return VoxelGrid(
        densities=dens.clone(),
        features=feat.clone(),
        voxel_size=VoxelSize(*voxel),
        grid_location=VoxelGridLocation(*loc),
        tunable=tunable,
        **density_kwargs(mode, rho),
    )


# ------------------------------------------------------------------------------------------
def g1_cast_rays():
    cam = hotdog_like_camera()
    pose = pose_spherical(30.0, -30.0, cam["radius"])
    small = cast_rays(CameraIntrinsics(64, 64, 88.9), pose, CPU)
    H = W = 800
    big = flatten_rays(cast_rays(CameraIntrinsics(H, W, 1111.111), pose, CPU))
    rng = np.random.RandomState(7)
    idx = np.unique(
        np.concatenate([[0, W - 1, (H - 1) * W, H * W - 1, (H // 2) * W + W // 2], rng.randint(0, H * W, 512)])
    )
    pose2 = pose_spherical(135.0, -10.0, 4.5)
    other = cast_rays(CameraIntrinsics(5, 7, 3.3), pose2, CPU)
    save(
        "g1_cast_rays.npz",
        rotation=pose.rotation,
        translation=pose.translation,
        small_origins=small.origins,
        small_directions=small.directions,
        big_index=idx,
        big_origins=big.origins[idx],
        big_directions=big.directions[idx],
        pose2_rotation=pose2.rotation,
        pose2_translation=pose2.translation,
        other_origins=other.origins,
        other_directions=other.directions,
    )


def random_rays(n, seed, spread=0.6):
    """rays from points on a radius-4 shell aimed near the origin (some miss the unit-ish box)"""
    o = hash_uniform((n, 3), seed)
    o = o / np.linalg.norm(o, axis=-1, keepdims=True) * 4.0
    target = hash_uniform((n, 3), seed + 1) * spread
    d = target - o
    d = d / np.linalg.norm(d, axis=-1, keepdims=True) * (1.0 + 0.1 * hash_uniform((n, 1), seed + 2))
    return torch.from_numpy(o.astype(np.float32)), torch.from_numpy(d.astype(np.float32))


def g2_sampling():
    cam = hotdog_like_camera()
    bounds = CameraBounds(np.float32(2.0) * 0.9, np.float32(6.0) * 1.1)
    o, d = random_rays(16, 11)
    rays = Rays(o, d)
    plain = sample_uniform_points_on_rays(rays, bounds, 32, perturb=False)
    torch.manual_seed(123)
    jit = sample_uniform_points_on_rays(rays, bounds, 32, perturb=True)
    torch.manual_seed(123)
    t_rand = torch.rand(16, 32)
    # AABB-bounded sampling on a wider set of rays (hits and misses)
    o2, d2 = random_rays(256, 21, spread=2.5)
    grid = make_grid(*procedural_grid((4, 5, 6), 3, 5), voxel=(0.5, 0.4, 0.3), loc=(0.1, -0.2, 0.05))
    ab, hit = _ray_aabb_intersection(Rays(o2, d2), bounds, grid.aabb)
    save(
        "g2_sampling.npz",
        near=np.float64(cam["near"]),
        far=np.float64(cam["far"]),
        origins=o,
        directions=d,
        z_plain=plain.depths,
        pts_plain=plain.points,
        t_rand=t_rand,
        z_jitter=jit.depths,
        pts_jitter=jit.points,
        aabb=np.array(grid.aabb, dtype=np.float64),
        aabb_origins=o2,
        aabb_directions=d2,
        aabb_bounds=ab,
        aabb_hit=hit,
    )


def boundary_points(aabb, n, seed):
    """points inside, outside and within half a voxel of the faces of the box"""
    lo = np.array([r[0] for r in aabb], dtype=np.float32)
    hi = np.array([r[1] for r in aabb], dtype=np.float32)
    u = hash_uniform((n, 3), seed, 0.0, 1.0)
    p = lo + (hi - lo) * (u * 1.3 - 0.15)  # 15 % margin outside on each side
    # snap a quarter of them close to a face
    k = n // 4
    face = (hash_uniform((k, 3), seed + 1, 0.0, 1.0) > 0.5).astype(np.float32)
    eps = hash_uniform((k, 3), seed + 2) * 1e-3
    p[:k] = lo + (hi - lo) * face + eps
    return torch.from_numpy(p.astype(np.float32))


def g3_voxel_grid():
    out = {}
    dens, feat = procedural_grid((5, 6, 7), 27, 31)
    voxel, loc, rho = (0.3, 0.25, 0.2), (0.1, -0.05, 0.2), 7.0
    for mode in ("relu", "softplus", "abs"):
        grid = make_grid(dens, feat, voxel, loc, mode, rho)
        pts = boundary_points(grid.aabb, 2048, 41)
        out[f"aniso_{mode}"] = grid(pts)
        out["aniso_points"] = pts
        out["aniso_aabb"] = np.array(grid.aabb, dtype=np.float64)
        out["aniso_inside"] = grid.test_inside_volume(pts)
    dens2, feat2 = procedural_grid((16, 16, 16), 3, 32)
    grid2 = make_grid(dens2, feat2, (3.0 / 16,) * 3, mode="relu", rho=100.0 / 3.0)
    pts2 = boundary_points(grid2.aabb, 2048, 43)
    out["cube16_relu"] = grid2(pts2)
    out["cube16_points"] = pts2
    out["cube16_aabb"] = np.array(grid2.aabb, dtype=np.float64)
    # grid up-scaling (next row 2)
    up = scale_voxel_grid_with_required_output_size(make_grid(dens, feat, voxel, loc, "relu", rho), (10, 12, 14))
    out["aniso_up_densities"] = up.densities
    out["aniso_up_features"] = up.features
    out["aniso_up_voxel"] = np.array(tuple(up.voxel_size), dtype=np.float64)
    save("g3_voxel_grid.npz", aniso_voxel=np.array(voxel), aniso_loc=np.array(loc), aniso_rho=np.float64(rho), **out)


def g4_sh():
    v = hash_uniform((512, 3), 51)
    v = v / np.linalg.norm(v, axis=-1, keepdims=True)
    v = torch.from_numpy(v.astype(np.float32))
    out = {"viewdirs": v}
    for deg in range(4):
        K = (deg + 1) ** 2
        c = torch.from_numpy(hash_uniform((512, 3, K), 52 + deg))
        out[f"coeffs{deg}"] = c
        out[f"radiance{deg}"] = evaluate_spherical_harmonics(deg, c, v)
    save("g4_sh.npz", **out)


def g5_g6_process_accumulate():
    cam = hotdog_like_camera()
    bounds = CameraBounds(np.float32(2.0) * 0.9, np.float32(6.0) * 1.1)
    dens, feat = procedural_grid((8, 8, 8), 27, 61)
    rho = 100.0 / 3.0
    grid = make_grid(dens, feat, (3.0 / 8,) * 3, mode="relu", rho=rho)
    o, d = random_rays(48, 62, spread=1.8)
    rays = Rays(o, d)
    sampled = sample_uniform_points_on_rays(rays, bounds, 40, perturb=False)
    out = {"origins": o, "directions": d, "z": sampled.depths}
    for diffuse in (False, True):
        proc = process_points_with_sh_voxel_grid(sampled, rays, grid, render_diffuse=diffuse)
        tag = "diffuse" if diffuse else "specular"
        out[f"processed_{tag}"] = proc.points
        for white in (False, True):
            acc = accumulate_radiance_density_on_rays(
                proc, rays, stochastic_density_noise_std=0.0, white_bkgd=white, extra_debug_info=True
            )
            w = "white" if white else "black"
            out[f"colour_{tag}_{w}"] = acc.colour
            out[f"depth_{tag}_{w}"] = acc.depth
            out[f"acc_{tag}_{w}"] = acc.extra["accumulated_weight"]
            out[f"disparity_{tag}_{w}"] = acc.extra["disparity"]
            if white and not diffuse:
                out["alpha"] = acc.extra["point_occupancies"]
                out["weights"] = acc.extra["point_weights"]
                out["deltas"] = acc.extra["deltas"]
    save("g5_g6_process_accumulate.npz", near=np.float64(cam["near"]), far=np.float64(cam["far"]), rho=np.float64(rho), **out)


def run_render(grid, rays, cfg, target, dtype=torch.float32):
    """reference render + grads of L1(colour, target) wrt both grid tensors"""
    grid.zero_grad()
    out = render_sh_voxel_grid(grid, Rays(rays.origins.to(dtype), rays.directions.to(dtype)), cfg)
    loss = torch.nn.functional.l1_loss(out.colour, target.to(dtype))
    loss.backward()
    return out, loss, grid.densities.grad.clone(), grid.features.grad.clone()


def g7_g8_end_to_end():
    cam = hotdog_like_camera()
    bounds = CameraBounds(np.float32(2.0) * 0.9, np.float32(6.0) * 1.1)
    rho = 100.0 / 3.0
    pose = pose_spherical(30.0, -30.0, cam["radius"])

    # ---- cfg1: 64^3, deg 0, 64x64, S 32 (BASELINE.json configs[0]) ---------------------
    dens, feat = procedural_grid((64, 64, 64), 3, 71)
    grid = make_grid(dens, feat, (3.0 / 64,) * 3, mode="relu", rho=rho, tunable=True)
    rays = flatten_rays(cast_rays(CameraIntrinsics(64, 64, 88.9), pose, CPU))
    target = torch.from_numpy(hash_uniform((len(rays), 3), 72, 0.0, 1.0))
    out = {"cfg1_rotation": pose.rotation, "cfg1_translation": pose.translation, "cfg1_target": target}
    sub = torch.from_numpy(np.sort(np.random.RandomState(3).choice(64**3, 8192, replace=False)))
    out["cfg1_grad_index"] = sub
    for white in (True, False):
        for diffuse in (False, True):
            cfg = SHVoxGridRenderConfig(
                num_samples_per_ray=32,
                camera_bounds=bounds,
                perturb_sampled_points=False,
                white_bkgd=white,
                render_diffuse=diffuse,
            )
            tag = f"cfg1_{'white' if white else 'black'}_{'diffuse' if diffuse else 'specular'}"
            res, loss, gd, gf = run_render(grid, rays, cfg, target)
            out[f"{tag}_colour"] = res.colour
            out[f"{tag}_depth"] = res.depth
            out[f"{tag}_acc"] = res.extra["accumulated_weight"]
            out[f"{tag}_disparity"] = res.extra["disparity"]
            out[f"{tag}_loss"] = loss
            out[f"{tag}_gd_sub"] = gd.reshape(-1)[sub]
            out[f"{tag}_gf_sub"] = gf.reshape(-1, 3)[sub]
            out[f"{tag}_gd_sum"] = gd.double().sum()
            out[f"{tag}_gd_abs"] = gd.double().abs().sum()
            out[f"{tag}_gf_sum"] = gf.double().sum()
            out[f"{tag}_gf_abs"] = gf.double().abs().sum()
    # fp64 evaluation of the white/specular case (tolerance rule H1)
    grid64 = make_grid(dens.double(), feat.double(), (3.0 / 64,) * 3, mode="relu", rho=rho, tunable=True)
    cfg = SHVoxGridRenderConfig(32, bounds, perturb_sampled_points=False, white_bkgd=True)
    res, loss, gd, gf = run_render(grid64, rays, cfg, target, torch.float64)
    out["cfg1_f64_colour"] = res.colour
    out["cfg1_f64_depth"] = res.depth
    out["cfg1_f64_acc"] = res.extra["accumulated_weight"]
    save("g7_cfg1_render.npz", near=np.float64(cam["near"]), far=np.float64(cam["far"]), rho=np.float64(rho), **out)

    # ---- 16^3, deg 2 (F 27): full grads, jitter, optimized sampling, all density modes --------
    dens, feat = procedural_grid((16, 16, 16), 27, 81)
    o, d = random_rays(192, 82, spread=1.6)
    rays = Rays(o, d)
    target = torch.from_numpy(hash_uniform((192, 3), 83, 0.0, 1.0))
    torch.manual_seed(321)
    t_rand = torch.rand(192, 48)
    out = {"origins": o, "directions": d, "target": target, "t_rand": t_rand}
    cases = [
        ("relu_spec", "relu", dict()),
        ("relu_diffuse", "relu", dict(render_diffuse=True)),
        ("relu_black", "relu", dict(white_bkgd=False)),
        ("relu_opt", "relu", dict(optimized_sampling=True)),
        ("relu_jitter", "relu", dict(perturb_sampled_points=True)),
        ("relu_opt_jitter", "relu", dict(optimized_sampling=True, perturb_sampled_points=True)),
        ("softplus_spec", "softplus", dict()),
        ("abs_spec", "abs", dict()),
    ]
    for tag, mode, over in cases:
        scale = 1.0 if mode == "abs" else rho
        grid = make_grid(dens, feat, (3.0 / 16,) * 3, mode=mode, rho=scale, tunable=True)
        kw = dict(num_samples_per_ray=48, camera_bounds=bounds, perturb_sampled_points=False, white_bkgd=True)
        kw.update(over)
        cfg = SHVoxGridRenderConfig(**kw)
        torch.manual_seed(321)  # the jitter draw is the first RNG call of the render (sample.py:63)
        res, loss, gd, gf = run_render(grid, rays, cfg, target)
        out[f"{tag}_colour"] = res.colour
        out[f"{tag}_depth"] = res.depth
        out[f"{tag}_acc"] = res.extra["accumulated_weight"]
        out[f"{tag}_disparity"] = res.extra["disparity"]
        out[f"{tag}_loss"] = loss
        out[f"{tag}_gd"] = gd
        out[f"{tag}_gf"] = gf
        if tag in ("relu_spec", "relu_jitter"):
            grid64 = make_grid(dens.double(), feat.double(), (3.0 / 16,) * 3, mode=mode, rho=scale, tunable=True)
            torch.manual_seed(321)
            if tag == "relu_jitter":
                # the fp64 run must see the SAME jitter values: feed them through a patched rand
                real_rand = torch.rand
                torch.rand = lambda *a, **k: t_rand.double()
                try:
                    r64, _, gd64, gf64 = run_render(grid64, rays, cfg, target, torch.float64)
                finally:
                    torch.rand = real_rand
            else:
                r64, _, gd64, gf64 = run_render(grid64, rays, cfg, target, torch.float64)
            out[f"{tag}_f64_colour"] = r64.colour
            out[f"{tag}_f64_depth"] = r64.depth
            out[f"{tag}_f64_acc"] = r64.extra["accumulated_weight"]
            out[f"{tag}_f64_gd"] = gd64.float()  # stored rounded to float32 (compared with tolerance)
            out[f"{tag}_f64_gf"] = gf64.float()
    save("g7_grid16_render.npz", near=np.float64(cam["near"]), far=np.float64(cam["far"]), rho=np.float64(rho), **out)


def g13_last_sample_inside():
    """A far plane INSIDE the volume: the last sample of every ray carries the reference's 1e10 |d| interval (accumulate.py:49-52) and
    lies in the grid, so its density gradient is delta exp(-sigma delta) x activation slope -- with softplus densities around 1e-9
    (raw ~ -0.65 at rho 100/3) an ordinary-sized number built from one huge and two tiny factors.  Found by the randomised sweep
    (tests/parity_fuzz.py); pinned against the reference here, all three density modes."""
    dens, feat = procedural_grid((8, 8, 8), 12, 131)
    o = torch.from_numpy(hash_uniform((160, 3), 132))
    o = o / o.norm(dim=-1, keepdim=True) * 3.2
    d = torch.from_numpy(hash_uniform((160, 3), 133)) * 0.5 - o
    d = d / d.norm(dim=-1, keepdim=True)
    rays = Rays(o, d)
    target = torch.from_numpy(hash_uniform((160, 3), 134, 0.0, 1.0))
    bounds = CameraBounds(1.0, 3.4)
    out = {"origins": o, "directions": d, "target": target}
    for mode in ("softplus", "abs", "relu"):
        rho = 1.0 if mode == "abs" else 100.0 / 3.0
        grid = make_grid(dens, feat, (3.0 / 8,) * 3, mode=mode, rho=rho, tunable=True)
        cfg = SHVoxGridRenderConfig(num_samples_per_ray=24, camera_bounds=bounds, perturb_sampled_points=False, white_bkgd=True)
        res, loss, gd, gf = run_render(grid, rays, cfg, target)
        out[f"{mode}_colour"] = res.colour
        out[f"{mode}_depth"] = res.depth
        out[f"{mode}_acc"] = res.extra["accumulated_weight"]
        out[f"{mode}_loss"] = loss
        out[f"{mode}_gd"] = gd
        out[f"{mode}_gf"] = gf
        grid64 = make_grid(dens.double(), feat.double(), (3.0 / 8,) * 3, mode=mode, rho=rho, tunable=True)
        _, _, gd64, _ = run_render(grid64, rays, cfg, target, torch.float64)
        out[f"{mode}_f64_gd"] = gd64.float()
    save("g13_last_sample_inside.npz", near=np.float64(1.0), far=np.float64(3.4), **out)


def g11_density_noise():
    """stochastic_density_noise_std != 0 (accumulate.py:58-62): with perturb_sampled_points off the render's only RNG draw is
    torch.randn(N, S), so the noise table is reproducible from the seed and stored beside the outputs."""
    cam = hotdog_like_camera()
    bounds = CameraBounds(cam["near"], cam["far"])
    rho = 100.0 / 3.0
    dens, feat = procedural_grid((16, 16, 16), 27, 81)
    o, d = random_rays(160, 92, spread=1.6)
    rays = Rays(o, d)
    target = torch.from_numpy(hash_uniform((160, 3), 93, 0.0, 1.0))
    out = {"origins": o, "directions": d, "target": target}
    for tag, mode, std, over in (("relu", "relu", 0.7, dict()), ("relu_diffuse_black", "relu", 0.25, dict(render_diffuse=True, white_bkgd=False)),
                                 ("softplus", "softplus", 1.5, dict())):
        grid = make_grid(dens, feat, (3.0 / 16,) * 3, mode=mode, rho=rho, tunable=True)
        kw = dict(num_samples_per_ray=40, camera_bounds=bounds, perturb_sampled_points=False, white_bkgd=True, stochastic_density_noise_std=std)
        kw.update(over)
        cfg = SHVoxGridRenderConfig(**kw)
        torch.manual_seed(777)
        out[f"{tag}_noise"] = torch.randn(160, 40) * std
        torch.manual_seed(777)
        res, loss, gd, gf = run_render(grid, rays, cfg, target)
        out[f"{tag}_std"] = np.float64(std)
        out[f"{tag}_colour"] = res.colour
        out[f"{tag}_depth"] = res.depth
        out[f"{tag}_acc"] = res.extra["accumulated_weight"]
        out[f"{tag}_loss"] = loss
        out[f"{tag}_gd"] = gd
        out[f"{tag}_gf"] = gf
    save("g11_density_noise.npz", near=np.float64(cam["near"]), far=np.float64(cam["far"]), rho=np.float64(rho), **out)


def plugin_tone_map(x):
    """a non-default radiance_hdr_tone_map (renderers.py:38)"""
    return torch.sigmoid(2.0 * x) * 0.9 + 0.05


def plugin_density2occupancy(densities, deltas):
    """a non-default density2occupancy (renderers.py:37): saturating, not the physically based exponential"""
    x = densities * deltas
    return x / (1.0 + x)


def plugin_feature_post(x):
    return 1.5 * x


def g12_plugins():
    """The plug-in points of the path with NON-default callables (renderers.py:37-38 density2occupancy / radiance_hdr_tone_map,
    voxels.py:312-322 feature pre- / post-activations, voxels.py:292-309 a density activation pair outside the trainer's three) and the
    per-sample debug outputs of the accumulator (accumulate.py:96-107, extra_debug_info): outputs and gradients of the reference."""
    from thre3d_atom.rendering.volumetric.accumulate import accumulate_radiance_density_on_rays
    from thre3d_atom.rendering.volumetric.render_interface import SampledPointsOnRays

    cam = hotdog_like_camera()
    bounds = CameraBounds(cam["near"], cam["far"])
    rho = 100.0 / 3.0
    dens, feat = procedural_grid((16, 16, 16), 27, 81)
    o, d = random_rays(96, 94, spread=1.6)
    rays = Rays(o, d)
    target = torch.from_numpy(hash_uniform((96, 3), 95, 0.0, 1.0))
    out = {"origins": o, "directions": d, "target": target}
    variants = {
        "tone_d2o": (dict(), dict(density2occupancy=plugin_density2occupancy, radiance_hdr_tone_map=plugin_tone_map)),
        "feature_acts": (dict(feature_preactivation=torch.tanh, feature_postactivation=plugin_feature_post), dict()),
        "density_acts": (dict(density_preactivation=torch.tanh, density_postactivation=torch.nn.Softplus(beta=2.0)), dict(render_diffuse=True)),
    }
    for tag, (grid_kw, cfg_kw) in variants.items():
        kw = dict(density_preactivation=torch.nn.Identity(), density_postactivation=torch.nn.ReLU(), expected_density_scale=rho)
        kw.update(grid_kw)
        grid = VoxelGrid(densities=dens.clone(), features=feat.clone(), voxel_size=VoxelSize(3.0 / 16, 3.0 / 16, 3.0 / 16), tunable=True, **kw)
        ckw = dict(num_samples_per_ray=40, camera_bounds=bounds, perturb_sampled_points=False, white_bkgd=True)
        ckw.update(cfg_kw)
        res, loss, gd, gf = run_render(grid, rays, SHVoxGridRenderConfig(**ckw), target)
        out.update({f"{tag}_colour": res.colour, f"{tag}_depth": res.depth, f"{tag}_acc": res.extra["accumulated_weight"],
                    f"{tag}_disparity": res.extra["disparity"], f"{tag}_loss": loss, f"{tag}_gd": gd, f"{tag}_gf": gf})
    # the accumulator alone, with its debug outputs and a recorded noise table (the plug-in type AccumulatorFunction, render_interface.py:100)
    z = torch.from_numpy(hash_uniform((96, 24), 96, 0.0, 1.0)).cumsum(-1) * 0.2 + 2.0
    processed = torch.from_numpy(hash_uniform((96, 24, 4), 97, -2.0, 2.0))
    processed[..., -1] = processed[..., -1].clamp(min=0.0) * 8.0
    torch.manual_seed(778)
    noise = torch.randn(96, 24) * 0.3
    torch.manual_seed(778)
    acc = accumulate_radiance_density_on_rays(SampledPointsOnRays(processed, z), rays, stochastic_density_noise_std=0.3, density2occupancy=plugin_density2occupancy,
                                              radiance_hdr_tone_map=plugin_tone_map, white_bkgd=False, extra_debug_info=True)
    out.update({"acc_z": z, "acc_processed": processed, "acc_noise": noise, "acc_colour": acc.colour, "acc_depth": acc.depth})
    out.update({f"acc_extra_{k}": v for k, v in acc.extra.items()})
    save("g12_plugins.npz", near=np.float64(cam["near"]), far=np.float64(cam["far"]), rho=np.float64(rho), **out)


def g10_single_cube():
    """The reference's only render test scene (thre3d_reprs/tests/test_voxels.py:88-134): a 2x2x2
    grid with +-10 RGB logits on the corners, viewed from the 6 axis directions -- here at
    32x32 pixels x 64 samples, through VolumetricModel.render (row 11 of SURVEY 8a)."""
    feats = torch.tensor(
        [[-10, -10, -10], [-10, -10, 10], [-10, 10, -10], [-10, 10, 10], [10, -10, -10], [10, -10, 10], [10, 10, -10], [10, 10, 10]],
        dtype=torch.float32,
    ).reshape(2, 2, 2, 3)
    dens = torch.full((2, 2, 2, 1), 100.0, dtype=torch.float32)
    grid = VoxelGrid(
        densities=dens,
        features=feats,
        voxel_size=VoxelSize(1.0, 1.0, 1.0),
        density_preactivation=torch.nn.Identity(),
        density_postactivation=torch.nn.ReLU(),
        tunable=False,
    )
    cfg = SHVoxGridRenderConfig(64, CameraBounds(0.5, 8.0), perturb_sampled_points=False, white_bkgd=True)
    model = VolumetricModel(grid, render_sh_voxel_grid, cfg, device=CPU)
    intr = CameraIntrinsics(32, 32, 32.0)
    out = {"densities": dens, "features": feats}
    for i, (yaw, pitch) in enumerate([(0, 0), (90, 0), (180, 0), (270, 0), (0, -89.9), (0, 89.9)]):
        pose = pose_spherical(float(yaw), float(pitch), 3.0)
        res = model.render(pose, intr, parallel_rays_chunk_size=300)
        out[f"rotation{i}"] = pose.rotation
        out[f"translation{i}"] = pose.translation
        out[f"colour{i}"] = res.colour
        out[f"depth{i}"] = res.depth
        out[f"acc{i}"] = res.extra["accumulated_weight"]
        out[f"disparity{i}"] = res.extra["disparity"]
    save("g10_single_cube.npz", **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    wanted = sys.argv[1:]
    jobs = {
        "g1": g1_cast_rays,
        "g2": g2_sampling,
        "g3": g3_voxel_grid,
        "g4": g4_sh,
        "g56": g5_g6_process_accumulate,
        "g78": g7_g8_end_to_end,
        "g10": g10_single_cube,
        "g11": g11_density_noise,
        "g12": g12_plugins,
        "g13": g13_last_sample_inside,
    }
    for name, fn in jobs.items():
        if not wanted or name in wanted:
            print(name)
            fn()
