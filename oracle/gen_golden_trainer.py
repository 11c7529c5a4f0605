"""Golden TRAINING trajectory (fixture G9) -- build container only.

Runs the reference's REAL trainer ``train_sh_vox_grid_vol_mod_with_posed_images``
(thre3d_atom/modules/trainers.py:49) for one stage of 5 iterations on a tiny synthetic on-disk dataset and
records, per iteration, the ray/pixel batch the trainer selected, the two losses, and the parameters after
the Adam step.  Only arrays are written (tests/golden/g9_trainer_trajectory.npz).

The reference imports packages this image lacks for things OFF the render path (logging/IO only).  They
are replaced, in THIS process only, by inert stand-ins: easydict (type annotation), imageio (PNG writer),
lpips (test metric, unused in fast_debug_mode), torch.utils.tensorboard (scalar logger), torchvision
(ToTensor / Resize for the image pyramid of the stage schedule).  The reference's sources are not modified.

``python oracle/gen_golden_trainer.py``            -> G9  (one stage, 5 iterations; tests/golden/g9_trainer_trajectory.npz)
``python oracle/gen_golden_trainer.py --stages``   -> G9b (the full stage schedule of trainers.py:125-152,227-250,462-470: two stages,
    12^3 -> 24^3 at SH degree 2, per-stage Adam + ExponentialLR, stratified jitter ON, images of a procedural scene rendered by the
    reference renderer, a held-out view rendered by the reference at the end; tests/golden/g9b_trainer_stages.npz)
"""
import json
import os
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")


def _module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Writer:
    def __init__(self, *a, **k):
        self.scalars = []

    def add_scalar(self, name, value, global_step=None):
        self.scalars.append((name, float(value), global_step))


class _ToTensor:
    def __call__(self, img):
        arr = np.asarray(img, dtype=np.float32) / 255.0
        return torch.from_numpy(arr).permute(2, 0, 1).contiguous()


class _Identity:
    def __init__(self, *a, **k):
        pass

    def __call__(self, x):
        return x


class _Resize:
    """torchvision.transforms.Resize on a [C,H,W] tensor: identity at the native size (G9), antialiased bilinear otherwise (G9b's
    coarse stage; the pixels the trainer then selects are recorded in the fixture, so the filter itself is not part of any parity claim)"""

    def __init__(self, size, *a, **k):
        self.size = tuple(int(v) for v in size)

    def __call__(self, x):
        if tuple(x.shape[-2:]) == self.size:
            return x
        return torch.nn.functional.interpolate(x[None], size=self.size, mode="bilinear", align_corners=False, antialias=True)[0].clamp(0.0, 1.0)


class _Compose:
    def __init__(self, ts):
        self.ts = ts

    def __call__(self, x):
        for t in self.ts:
            x = t(x)
        return x


_module("easydict", EasyDict=dict)
_module("imageio", imwrite=lambda *a, **k: None, mimwrite=lambda *a, **k: None)
_module("lpips", LPIPS=lambda *a, **k: None)
_module("torch.utils.tensorboard", SummaryWriter=_Writer)
_tv = _module("torchvision")
_tv.transforms = _module(
    "torchvision.transforms", Compose=_Compose, RandomHorizontalFlip=_Identity, Resize=_Resize, ToTensor=_ToTensor
)

from PIL import Image  # noqa: E402

from tests.helpers import GOLDEN_DIR, hash_uniform  # noqa: E402
from thre3d_atom.data.datasets import PosedImagesDataset  # noqa: E402
from thre3d_atom.modules import trainers as ref_trainers  # noqa: E402
from thre3d_atom.modules.volumetric_model import VolumetricModel  # noqa: E402
from thre3d_atom.thre3d_reprs.renderers import SHVoxGridRenderConfig, render_sh_voxel_grid  # noqa: E402
from thre3d_atom.thre3d_reprs.voxels import VoxelGrid, VoxelSize  # noqa: E402
from thre3d_atom.utils.imaging_utils import pose_spherical  # noqa: E402

G, SH_DEG, HW, N_IMG, RAYS, STEPS, SAMPLES = 16, 1, 32, 12, 512, 5, 48
F = 3 * (SH_DEG + 1) ** 2
LR = 0.03


def procedural_init(t: torch.Tensor) -> torch.Tensor:
    """deterministic stand-in for uniform_(-1, 1): seed depends on the trailing dimension"""
    with torch.no_grad():
        t.copy_(torch.from_numpy(hash_uniform(tuple(t.shape), 900 + t.shape[-1])))
    return t


def main():
    torch.manual_seed(42)
    np.random.seed(42)
    tmp = Path(tempfile.mkdtemp(prefix="g9_"))
    img_dir = tmp / "images"
    img_dir.mkdir()
    params = {}
    for i in range(N_IMG):
        img = (hash_uniform((HW, HW, 3), 700 + i, 0.0, 1.0) * 255).astype(np.uint8)
        # a soft blob so that the images are not pure noise
        yy, xx = np.mgrid[0:HW, 0:HW]
        blob = np.exp(-(((xx - 16) ** 2 + (yy - 16) ** 2) / 60.0))[..., None]
        img = (img * 0.3 + 255 * 0.7 * (1 - blob) + blob * np.array([200, 60, 30]) * 0.7).clip(0, 255).astype(np.uint8)
        name = f"img_{i:02d}.png"
        Image.fromarray(img).save(img_dir / name)
        pose = pose_spherical(30.0 * i, -30.0, 4.0311)
        params[name] = {
            "extrinsic": {"rotation": pose.rotation.numpy().tolist(), "translation": pose.translation.numpy().tolist()},
            "intrinsic": {"height": HW, "width": HW, "focal": 44.4, "bounds": [2.0, 6.0]},
        }
    with open(tmp / "camera_params.json", "w") as fh:
        json.dump(params, fh)

    dataset = PosedImagesDataset(img_dir, tmp / "camera_params.json")
    grid = VoxelGrid(
        densities=torch.zeros(G, G, G, 1),
        features=torch.zeros(G, G, G, F),
        voxel_size=VoxelSize(3.0 / G, 3.0 / G, 3.0 / G),
        density_preactivation=torch.nn.Identity(),
        density_postactivation=torch.nn.ReLU(),
        expected_density_scale=100.0 / 3.0,
        tunable=True,
    )
    cfg = SHVoxGridRenderConfig(
        num_samples_per_ray=SAMPLES, camera_bounds=dataset.camera_bounds, perturb_sampled_points=False, white_bkgd=True
    )
    model = VolumetricModel(grid, render_sh_voxel_grid, cfg, device=torch.device("cpu"))

    rec = {"origins": [], "directions": [], "pixels": [], "losses": [], "dens": [], "feat": []}

    real_select = ref_trainers.sample_random_rays_and_pixels_synchronously

    def recording_select(rays, pixels, sample_size):
        r, p = real_select(rays, pixels, sample_size)
        rec["origins"].append(r.origins.clone())
        rec["directions"].append(r.directions.clone())
        rec["pixels"].append(p.clone())
        return r, p

    real_l1 = ref_trainers.l1_loss

    def recording_l1(a, b):
        out = real_l1(a, b)
        rec["losses"].append(float(out))
        return out

    real_step = torch.optim.Adam.step

    def recording_step(self, *a, **k):
        out = real_step(self, *a, **k)
        ps = [p for g in self.param_groups for p in g["params"]]
        rec["dens"].append(ps[0].detach().clone())
        rec["feat"].append(ps[1].detach().clone())
        return out

    ref_trainers.sample_random_rays_and_pixels_synchronously = recording_select
    ref_trainers.l1_loss = recording_l1
    ref_trainers.visualize_sh_vox_grid_vol_mod_rendered_feedback = lambda **k: None
    torch.optim.Adam.step = recording_step
    try:
        ref_trainers.train_sh_vox_grid_vol_mod_with_posed_images(
            vol_mod=model,
            train_dataset=dataset,
            output_dir=tmp / "out",
            random_initializer=procedural_init,
            image_batch_cache_size=8,
            ray_batch_size=RAYS,
            num_stages=1,
            num_iterations_per_stage=STEPS,
            learning_rate=LR,
            lr_decay_steps_per_stage=1000,
            save_freq=10**6,
            test_freq=10**6,
            feedback_freq=10**6,
            summary_freq=1,
            fast_debug_mode=True,
            verbose_rendering=False,
        )
    finally:
        torch.optim.Adam.step = real_step
        ref_trainers.sample_random_rays_and_pixels_synchronously = real_select
        ref_trainers.l1_loss = real_l1

    assert len(rec["origins"]) == STEPS and len(rec["losses"]) == 2 * STEPS and len(rec["dens"]) == STEPS
    assert rec["dens"][0].shape == (G, G, G, 1) and rec["feat"][0].shape == (G, G, G, F)
    out = {
        "origins": torch.stack(rec["origins"]).numpy(),
        "directions": torch.stack(rec["directions"]).numpy(),
        "pixels": torch.stack(rec["pixels"]).numpy(),
        "specular_loss": np.array(rec["losses"][0::2]),
        "diffuse_loss": np.array(rec["losses"][1::2]),
        "dens_after_step1": rec["dens"][0].numpy(),
        "feat_after_step1": rec["feat"][0].numpy(),
        "dens_final": rec["dens"][-1].numpy(),
        "feat_final": rec["feat"][-1].numpy(),
        "near": np.float64(dataset.camera_bounds.near),
        "far": np.float64(dataset.camera_bounds.far),
        "config": np.array([G, SH_DEG, HW, N_IMG, RAYS, STEPS, SAMPLES]),
        "lr": np.float64(LR),
        "meta": np.array([f"torch={torch.__version__}", "reference trainer: modules/trainers.py:49, num_stages=1, perturb off"]),
    }
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, "g9_trainer_trajectory.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB; losses {out['specular_loss']} {out['diffuse_loss']}")


# ------------------------------------------------------------------------------------------------------------------------------
# G9b: the stage schedule (modules/trainers.py:125-152 sizes + re-init, :227-250 per-stage Adam + ExponentialLR, :462-470 x2 up-scaling)
# ------------------------------------------------------------------------------------------------------------------------------
B_G, B_DEG, B_HW, B_RAYS, B_SAMPLES, B_ITERS, B_STAGES = 24, 2, 32, 512, 48, 300, 2
B_F = 3 * (B_DEG + 1) ** 2
B_LR, B_GAMMA, B_DECAY_STEPS, B_STAGE_GAMMA = 0.03, 0.5, 100, 0.9
B_FOCAL, B_RADIUS = 44.4, 4.0311
B_TRAIN_POSES = [(18.0 * i + 5.0, (-20.0, -40.0, -62.0)[i % 3]) for i in range(20)]   # (yaw, pitch) of pose_spherical
B_HELDOUT_POSE = (63.0, -33.0)
B_EVAL_SAMPLES = 96
JITTER_SEED0 = 50_000  # t_rand of the k-th torch.rand call of the run = hash_uniform((N, S), JITTER_SEED0 + k, 0, 1)


def ground_truth_scene(G=32):
    """a smooth procedural scene: one soft ball with a position- and view-dependent colour (reference layout tensors)"""
    ax = ((np.arange(G, dtype=np.float32) + 0.5) / G * 3.0 - 1.5)
    x, y, z = np.meshgrid(ax, ax, ax, indexing="ij")
    r = np.sqrt((x / 0.9) ** 2 + (y / 0.75) ** 2 + (z / 0.8) ** 2)
    dens = (0.9 - r).astype(np.float32)[..., None] * 0.6  # raw density, positive inside the ellipsoid
    K = (B_DEG + 1) ** 2
    feat = np.zeros((G, G, G, 3, K), dtype=np.float32)
    feat[..., 0, 0] = 2.0 * np.sin(2.1 * x) + 0.8
    feat[..., 1, 0] = 2.0 * np.cos(1.7 * y) - 0.4
    feat[..., 2, 0] = 2.0 * np.sin(1.3 * z + 0.5)
    feat[..., :, 1:4] = 0.6 * hash_uniform((G, G, G, 3, 3), 4242)  # mild view dependence
    return torch.from_numpy(dens), torch.from_numpy(feat.reshape(G, G, G, 3 * K))


def _perturbed_init(seed: int):
    """procedural_init with a pseudo-random half of the entries moved by ONE float32 ulp: a second, equally valid "run of the
    reference" (G9b's noise band: Adam turns rounding-level differences of near-zero gradients into full-size steps, so two
    float32 evaluations of the same schedule -- CPU vs GPU, one summation order vs another -- drift apart chaotically)."""

    def init(t: torch.Tensor) -> torch.Tensor:
        procedural_init(t)
        with torch.no_grad():
            flip = torch.from_numpy(hash_uniform(tuple(t.shape), 7000 + 31 * seed + t.shape[-1], 0.0, 1.0) < 0.5)
            moved = torch.nextafter(t, torch.full_like(t, 2.0))
            t.copy_(torch.where(flip, moved, t))
        return t

    return init


def run_reference_trainer_stages(tmp: Path, dataset, init_fn, held_pose, intr, full: bool):
    """One run of the REAL reference trainer over the two-stage schedule; returns what was recorded."""
    grid = VoxelGrid(densities=torch.zeros(B_G, B_G, B_G, 1), features=torch.zeros(B_G, B_G, B_G, B_F), voxel_size=VoxelSize(3.0 / B_G, 3.0 / B_G, 3.0 / B_G),
                     density_preactivation=torch.nn.Identity(), density_postactivation=torch.nn.ReLU(), expected_density_scale=100.0 / 3.0, tunable=True)
    cfg = SHVoxGridRenderConfig(num_samples_per_ray=B_SAMPLES, camera_bounds=dataset.camera_bounds, perturb_sampled_points=True, white_bkgd=True)
    model = VolumetricModel(grid, render_sh_voxel_grid, cfg, device=torch.device("cpu"))
    torch.manual_seed(4242)  # every run draws the same image batches and the same randperm prefixes
    np.random.seed(4242)

    rec = {"image_ids": [], "sel": [], "losses": [], "snap": {}, "rand_calls": 0, "last_perm": None, "steps": 0, "datasets": {}, "first_rand_call_of_step": [], "checkpoint_renders": []}

    # -- the trainer's random sources, recorded (randperm) or replaced by a procedural table (the stratified jitter: sample.py:63) --
    real_getitem = PosedImagesDataset.__getitem__

    def recording_getitem(self, index):
        rec["datasets"].setdefault(self.camera_intrinsics.height, self)
        rec["image_ids"].append(int(self._image_file_paths[index].name[4:6]))
        return real_getitem(self, index)

    real_randperm = torch.randperm

    def recording_randperm(n, *a, **k):
        out = real_randperm(n, *a, **k)
        rec["last_perm"] = out
        return out

    real_rand = torch.rand

    def procedural_rand(*size, **k):
        shape = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
        t = torch.from_numpy(hash_uniform(shape, JITTER_SEED0 + rec["rand_calls"], 0.0, 1.0))
        rec["rand_calls"] += 1
        return t.to(k.get("device", "cpu"))

    real_select = ref_trainers.sample_random_rays_and_pixels_synchronously

    def recording_select(rays, pixels, sample_size):
        r, p = real_select(rays, pixels, sample_size)
        perm = rec["last_perm"]
        assert perm is not None and perm.numel() == len(rays)
        sel = perm[:sample_size]
        assert torch.equal(r.origins, rays.origins[sel]) and torch.equal(p, pixels[sel])
        rec["sel"].append(sel.to(torch.int16).clone())
        rec["first_rand_call_of_step"].append(rec["rand_calls"])
        return r, p

    real_l1 = ref_trainers.l1_loss

    def recording_l1(a, b):
        out = real_l1(a, b)
        rec["losses"].append(float(out.detach()))
        return out

    real_step = torch.optim.Adam.step

    def recording_step(self, *a, **k):
        out = real_step(self, *a, **k)
        rec["steps"] += 1
        ps = [p for g in self.param_groups for p in g["params"]]
        if full and rec["steps"] in (1, B_ITERS, B_ITERS + 1, B_STAGES * B_ITERS):
            rec["snap"][rec["steps"]] = (ps[0].detach().clone(), ps[1].detach().clone(), float(self.param_groups[0]["lr"]))
        if rec["steps"] in B_CHECKPOINTS:  # (jitter off: no random number is drawn)
            rec["checkpoint_renders"].append(model.render(held_pose, intr, perturb_sampled_points=False, num_samples_per_ray=B_EVAL_SAMPLES).colour.clone())
        return out

    real_scale = ref_trainers.scale_voxel_grid_with_required_output_size
    rec["scaled"] = []

    def recording_scale(g, output_size, mode="trilinear"):
        out = real_scale(g, output_size=output_size, mode=mode)
        rec["scaled"].append((out.densities.detach().clone(), out.features.detach().clone()))
        return out

    PosedImagesDataset.__getitem__ = recording_getitem
    torch.randperm = recording_randperm
    torch.rand = procedural_rand
    ref_trainers.sample_random_rays_and_pixels_synchronously = recording_select
    ref_trainers.l1_loss = recording_l1
    ref_trainers.visualize_sh_vox_grid_vol_mod_rendered_feedback = lambda **k: None
    ref_trainers.scale_voxel_grid_with_required_output_size = recording_scale
    torch.optim.Adam.step = recording_step
    try:
        ref_trainers.train_sh_vox_grid_vol_mod_with_posed_images(
            vol_mod=model, train_dataset=dataset, output_dir=tmp / "out", random_initializer=init_fn, image_batch_cache_size=8,
            ray_batch_size=B_RAYS, num_stages=B_STAGES, num_iterations_per_stage=B_ITERS, scale_factor=2.0, learning_rate=B_LR,
            lr_decay_gamma_per_stage=B_GAMMA, lr_decay_steps_per_stage=B_DECAY_STEPS, stagewise_lr_decay_gamma=B_STAGE_GAMMA,
            save_freq=10**6, test_freq=10**6, feedback_freq=10**6, summary_freq=100, fast_debug_mode=True, verbose_rendering=False,
        )
        # the trained field, photographed by the reference from a view it never saw (jitter off)
        rec["held"] = model.render(held_pose, intr, perturb_sampled_points=False, num_samples_per_ray=B_EVAL_SAMPLES)
        rec["train0"] = model.render(pose_spherical(*B_TRAIN_POSES[0], B_RADIUS), intr, perturb_sampled_points=False, num_samples_per_ray=B_EVAL_SAMPLES)
    finally:
        torch.optim.Adam.step = real_step
        torch.randperm = real_randperm
        torch.rand = real_rand
        PosedImagesDataset.__getitem__ = real_getitem
        ref_trainers.sample_random_rays_and_pixels_synchronously = real_select
        ref_trainers.l1_loss = real_l1
        ref_trainers.scale_voxel_grid_with_required_output_size = real_scale
    rec["real_getitem"] = real_getitem
    return rec


N_RERUNS = 12
B_CHECKPOINTS = (25, 50, 100, 200, 300)  # global iterations after which the held-out view is photographed as well (all in stage 1 or at its end)


def main_stages():
    from thre3d_atom.rendering.volumetric.utils.misc import cast_rays, flatten_rays
    from thre3d_atom.utils.imaging_utils import CameraBounds, CameraIntrinsics
    from thre3d_atom.utils.metric_utils import mse2psnr

    tmp = Path(tempfile.mkdtemp(prefix="g9b_"))
    img_dir = tmp / "images"
    img_dir.mkdir()

    # ---- the scene, photographed by the REFERENCE renderer (8-bit PNGs: what a dataset on disk holds) ----
    gt_dens, gt_feat = ground_truth_scene()
    near, far = np.float32(2.0) * 0.9, np.float32(6.0) * 1.1
    gt_cfg = SHVoxGridRenderConfig(num_samples_per_ray=128, camera_bounds=CameraBounds(near, far), perturb_sampled_points=False, white_bkgd=True)
    gt_grid = VoxelGrid(densities=gt_dens, features=gt_feat, voxel_size=VoxelSize(3.0 / 32, 3.0 / 32, 3.0 / 32), density_preactivation=torch.nn.Identity(),
                        density_postactivation=torch.nn.ReLU(), expected_density_scale=100.0 / 3.0, tunable=False)
    gt_model = VolumetricModel(gt_grid, render_sh_voxel_grid, gt_cfg, device=torch.device("cpu"))
    intr = CameraIntrinsics(B_HW, B_HW, B_FOCAL)
    params = {}
    for i, (yaw, pitch) in enumerate(B_TRAIN_POSES):
        pose = pose_spherical(yaw, pitch, B_RADIUS)
        img = gt_model.render(pose, intr).colour.clamp(0, 1).numpy()
        name = f"img_{i:02d}.png"
        Image.fromarray((img * 255.0 + 0.5).astype(np.uint8)).save(img_dir / name)
        params[name] = {
            "extrinsic": {"rotation": pose.rotation.numpy().tolist(), "translation": pose.translation.numpy().tolist()},
            "intrinsic": {"height": B_HW, "width": B_HW, "focal": B_FOCAL, "bounds": [2.0, 6.0]},
        }
    with open(tmp / "camera_params.json", "w") as fh:
        json.dump(params, fh)
    held_pose = pose_spherical(*B_HELDOUT_POSE, B_RADIUS)
    held_truth = gt_model.render(held_pose, intr).colour.clamp(0, 1)
    dataset = PosedImagesDataset(img_dir, tmp / "camera_params.json")

    rec = run_reference_trainer_stages(tmp, dataset, procedural_init, held_pose, intr, full=True)
    real_getitem, scaled, held, train0 = rec["real_getitem"], rec["scaled"], rec["held"], rec["train0"]

    total = B_STAGES * B_ITERS
    # (the first three dataset[0] look-ups are the trainer's feedback-pose set-up, trainers.py:157-162; then 8 per iteration)
    ids = np.array(rec["image_ids"][3:], dtype=np.int16)
    assert len(rec["sel"]) == total and len(rec["losses"]) == 2 * total and ids.size == 8 * total, (len(rec["sel"]), ids.size)
    assert rec["rand_calls"] == 2 * total + 2 - 2 and rec["first_rand_call_of_step"] == list(range(0, 2 * total, 2))
    assert len(scaled) == 2  # the initial down-scaling (re-initialised afterwards) and the stage transition
    psnr = lambda a, b: float(mse2psnr(torch.nn.functional.mse_loss(a, b)))
    train0_truth = dataset[int(np.where(np.array([int(p.name[4:6]) for p in dataset._image_file_paths]) == 0)[0][0])][0].permute(1, 2, 0)

    # ---- the reference against ITSELF: the same schedule, same batches, same jitter, initial parameters moved by one ulp ----
    reruns = []
    for k in range(N_RERUNS):
        r = run_reference_trainer_stages(tmp, dataset, _perturbed_init(k + 1), held_pose, intr, full=False)
        assert all(torch.equal(a, b) for a, b in zip(r["sel"], rec["sel"])) and r["image_ids"] == rec["image_ids"]
        reruns.append({"ckpt": [psnr(c, held_truth) for c in r["checkpoint_renders"]], "held": psnr(r["held"].colour, held_truth), "train0": psnr(r["train0"].colour, train0_truth), "spec": np.array(r["losses"][0::2]),
                       "diff": np.array(r["losses"][1::2]), "held_img": r["held"].colour.numpy()})
        print(f"re-run {k + 1} (init moved by one ulp): held-out {reruns[-1]['held']:.3f} dB, training view {reruns[-1]['train0']:.3f} dB")

    # per-stage tables: the images the stage trained on and every ray of every image (cast by the reference)
    out = {}
    for stage, h in ((1, B_HW // 2), (2, B_HW)):
        ds = rec["datasets"][h]
        order = np.argsort([int(p.name[4:6]) for p in ds._image_file_paths])
        imgs, dirs, orgs = [], [], []
        for j in order:
            image, pose = real_getitem(ds, int(j))
            rays = flatten_rays(cast_rays(ds.camera_intrinsics, ref_trainers.CameraPose(rotation=pose[:, :3], translation=pose[:, 3:]), device=torch.device("cpu")))
            imgs.append(image.permute(1, 2, 0).reshape(-1, 3))
            dirs.append(rays.directions)
            orgs.append(rays.origins[0])
        out[f"pixels_stage{stage}"] = torch.stack(imgs).numpy()        # [20, h*w, 3]
        out[f"directions_stage{stage}"] = torch.stack(dirs).numpy()    # [20, h*w, 3]
        out["camera_origins"] = torch.stack(orgs).numpy()              # [20, 3]
        out[f"intrinsics_stage{stage}"] = np.array([ds.camera_intrinsics.height, ds.camera_intrinsics.width, ds.camera_intrinsics.focal], dtype=np.float64)
    sub = (np.add.outer(np.add.outer(np.arange(B_G), np.arange(B_G)), np.arange(B_G)) % 4) == 0  # nodes of the up-scaled grid kept in the fixture
    out.update({
        "image_ids": ids.reshape(total, 8),
        "selection": torch.stack(rec["sel"]).numpy(),                  # [total, 512] int16 into the step's 8 * h * w concatenated rays
        "specular_loss": np.array(rec["losses"][0::2]), "diffuse_loss": np.array(rec["losses"][1::2]),
        "dens_after_step1": rec["snap"][1][0].numpy(), "feat_after_step1": rec["snap"][1][1].numpy(),
        "dens_stage1_end": rec["snap"][B_ITERS][0].numpy(), "feat_stage1_end": rec["snap"][B_ITERS][1].numpy(),
        "upscaled_nodes_kept": sub, "dens_upscaled_kept": scaled[1][0].numpy()[sub], "feat_upscaled_kept": scaled[1][1].numpy()[sub],
        "dens_final": rec["snap"][total][0].numpy(), "feat_final": rec["snap"][total][1].numpy(),
        "lr_at": np.array([rec["snap"][k][2] for k in (1, B_ITERS, B_ITERS + 1, total)]),
        "heldout_rotation": held_pose.rotation.numpy(), "heldout_translation": held_pose.translation.numpy(),
        "heldout_truth": held_truth.numpy(), "heldout_render": held.colour.numpy(), "heldout_depth": held.depth.numpy(),
        "train0_render": train0.colour.numpy(), "train0_truth": train0_truth.numpy(),
        "heldout_psnr": np.float64(psnr(held.colour, held_truth)), "train0_psnr": np.float64(psnr(train0.colour, train0_truth)),
        "checkpoints": np.array(B_CHECKPOINTS), "checkpoint_heldout_render": torch.stack(rec["checkpoint_renders"]).numpy(),
        "checkpoint_heldout_psnr": np.array([psnr(c, held_truth) for c in rec["checkpoint_renders"]]),
        "rerun_checkpoint_heldout_psnr": np.array([r["ckpt"] for r in reruns]),
        # the reference's own spread (N_RERUNS runs whose initial parameters differ from the base run by one ulp)
        "rerun_heldout_psnr": np.array([r["held"] for r in reruns]), "rerun_train0_psnr": np.array([r["train0"] for r in reruns]),
        "rerun_specular_loss": np.stack([r["spec"] for r in reruns]), "rerun_diffuse_loss": np.stack([r["diff"] for r in reruns]),
        "rerun_heldout_max_abs_image_difference": np.array([np.abs(r["held_img"] - held.colour.numpy()).max() for r in reruns]),
        "near": np.float64(dataset.camera_bounds.near), "far": np.float64(dataset.camera_bounds.far),
        "config": np.array([B_G, B_DEG, B_HW, len(B_TRAIN_POSES), B_RAYS, B_ITERS, B_SAMPLES, B_STAGES, B_EVAL_SAMPLES, JITTER_SEED0]),
        "schedule": np.array([B_LR, B_GAMMA, B_DECAY_STEPS, B_STAGE_GAMMA]),
        "meta": np.array([f"torch={torch.__version__}", "reference trainer: modules/trainers.py:49, num_stages=2, perturb on (procedural torch.rand)"]),
    })
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, "g9b_trainer_stages.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB; held-out PSNR {out['heldout_psnr']:.3f} dB, training view {out['train0_psnr']:.3f} dB; "
          f"losses first/last {out['specular_loss'][0]:.4f} / {out['specular_loss'][-1]:.4f}; lr {out['lr_at']}")
    print("held-out PSNR at the checkpoints", B_CHECKPOINTS, ":", out["checkpoint_heldout_psnr"], "re-runs minus base:", out["rerun_checkpoint_heldout_psnr"] - out["checkpoint_heldout_psnr"][None])
    rel = np.abs(out["rerun_specular_loss"] / out["specular_loss"][None] - 1.0)
    print("specular loss relative difference of the re-runs, median per window of 50 iterations:", [float(np.median(rel[:, i:i + 50])) for i in range(0, rel.shape[1], 50)])
    print(f"reference vs its one-ulp re-runs: held-out {out['rerun_heldout_psnr']}, training view {out['rerun_train0_psnr']}; specular loss relative "
          f"difference median {np.median(rel):.4f}, 90 % {np.percentile(rel, 90):.4f}, max {rel.max():.4f}; image max-abs {out['rerun_heldout_max_abs_image_difference']}")


if __name__ == "__main__":
    if "--stages" in sys.argv[1:]:
        main_stages()
    else:
        main()
