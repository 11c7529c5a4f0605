"""Golden TRAINING trajectory (fixture G9) -- build container only.

Runs the reference's REAL trainer ``train_sh_vox_grid_vol_mod_with_posed_images``
(thre3d_atom/modules/trainers.py:49) for one stage of 5 iterations on a tiny synthetic on-disk dataset and
records, per iteration, the ray/pixel batch the trainer selected, the two losses, and the parameters after
the Adam step.  Only arrays are written (tests/golden/g9_trainer_trajectory.npz).

The reference imports packages this image lacks for things OFF the render path (logging/IO only).  They
are replaced, in THIS process only, by inert stand-ins: easydict (type annotation), imageio (PNG writer),
lpips (test metric, unused in fast_debug_mode), torch.utils.tensorboard (scalar logger), torchvision
(ToTensor / identity Resize for PIL images).  The reference's sources are not modified.
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
    "torchvision.transforms", Compose=_Compose, RandomHorizontalFlip=_Identity, Resize=_Identity, ToTensor=_ToTensor
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


if __name__ == "__main__":
    main()
