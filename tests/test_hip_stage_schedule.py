"""Trained-field parity THROUGH the stage schedule, on the GPU, against the REAL reference trainer (golden G9b:
oracle/gen_golden_trainer.py --stages; reference modules/trainers.py:125-152 stage sizes + re-initialisation, :227-250 per-stage
Adam + ExponentialLR, :462-470 x2 trilinear up-scaling).  The HIP path is fed the batches and jitter tables the reference trainer
drew, follows it for 2 x 300 iterations across the 12^3 -> 24^3 transition and is compared with it on a view neither ever saw --
to 0.01 dB while float32 trajectories can coincide (100 iterations, 30 dB), against the reference's own run-to-run spread where
they cannot (the fixture holds 12 re-runs of the reference from initial parameters moved by one ulp).  All through the C ABI."""
import numpy as np
import pytest
import torch

import thr3ed_atom_amd as rf
from thr3ed_atom_amd.optim import ExponentialLR
from thr3ed_atom_amd.trainers import TrainStepper
from oracle import relu_field_oracle as orc
from tests.helpers import g9b_batch, g9b_learning_rate, hash_uniform, load_golden

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def relu_grid(dev, dens, feat, storage, tunable=True):
    G = dens.shape[0]
    return rf.VoxelGrid(dens.clone().to(dev), feat.clone().to(dev), rf.VoxelSize(3.0 / G, 3.0 / G, 3.0 / G), density_preactivation=torch.nn.Identity(),
                        density_postactivation=torch.nn.ReLU(), expected_density_scale=100.0 / 3.0, tunable=tunable, storage=storage)


def psnr(a, b) -> float:
    return float(-10.0 * np.log10(np.mean((np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) ** 2)))


def oracle_gradient_of_step1(g):
    """autograd gradient of L1(specular) + L1(diffuse) at the initial parameters on the first batch (CPU oracle: the checker)"""
    G, deg, hw, n_img, n_rays, iters, S, stages, eval_S, seed0 = (int(v) for v in g["config"])
    g0, F = int(np.ceil(G / 2)), 3 * (deg + 1) ** 2
    dens = T(hash_uniform((g0, g0, g0, 1), 901)).requires_grad_(True)
    feat = T(hash_uniform((g0, g0, g0, F), 900 + F)).requires_grad_(True)
    o, d, px, t_spec, t_diff = (T(a) for a in g9b_batch(g, 0))
    kw = dict(origins=o, directions=d, aabb=orc.make_aabb((g0,) * 3, (3.0 / g0,) * 3), near=float(g["near"]), far=float(g["far"]), num_samples=S,
              density_scale=100.0 / 3.0, white_bkgd=True)
    loss = torch.nn.functional.l1_loss(orc.render(dens, feat, t_rand=t_spec, **kw)["colour"], px)
    loss = loss + torch.nn.functional.l1_loss(orc.render(dens, feat, render_diffuse=True, t_rand=t_diff, **kw)["colour"], px)
    loss.backward()
    return dens.grad.numpy(), feat.grad.numpy()


class _Follower:
    """One implementation of the training iteration under test."""

    def __init__(self, kind, dev, g):
        self.kind, self.dev, self.g = kind, dev, g
        self.gamma = float(g["schedule"][1])

    def start_stage(self, grid, lr, S, near, far):
        cfg = rf.SHVoxGridRenderConfig(S, rf.CameraBounds(near, far), perturb_sampled_points=True, white_bkgd=True)
        self.grid = grid
        self.model = rf.VolumetricModel(grid, rf.render_sh_voxel_grid, cfg, device=self.dev)
        if self.kind == "torch_optim":  # the reference user's loop: the autograd op + torch.optim.Adam + torch's ExponentialLR on the grid's own Parameters
            self.opt = torch.optim.Adam([{"params": list(grid.parameters()), "lr": lr}], betas=(0.9, 0.999))
            self.sched = torch.optim.lr_scheduler.ExponentialLR(self.opt, gamma=self.gamma)
        else:
            self.stepper = TrainStepper(self.model, int(self.g["config"][4]), learning_rate=lr, fused=not self.kind.startswith("autograd"), data_parallel=False,
                                        deterministic=getattr(self, "deterministic", False), backward="binned" if getattr(self, "deterministic", False) else "auto")
            self.sched = ExponentialLR(self.stepper.optimizer, self.gamma)

    def lr(self):
        return self.opt.param_groups[0]["lr"] if self.kind == "torch_optim" else self.stepper.optimizer.lr

    def step(self, rays, pixels, t_spec, t_diff):
        if self.kind == "fused":
            st = self.stepper.step_on(rays, pixels, t_rand=(t_spec, t_diff))
            return float(st.specular_loss), float(st.diffuse_loss)
        cfg = self.model.render_config
        if self.kind == "autograd_pair":  # both renders as ONE autograd node, both loss lines as one (what TrainStepper(fused=False) runs)
            from thr3ed_atom_amd import ops

            spec_out, diff_out = rf.render_sh_voxel_grid_pair(self.grid, rays, cfg, t_rands=(t_spec, t_diff))
            total, spec, _, diff, _ = ops.l1_loss_pair_with_mse(spec_out.colour, diff_out.colour, pixels)
            self.stepper.optimizer.zero_grad()
            total.backward()
            self.stepper.optimizer.step()
            return float(spec), float(diff)
        spec = torch.nn.functional.l1_loss(rf.render_sh_voxel_grid(self.grid, rays, cfg, t_rand=t_spec).colour, pixels)
        import dataclasses

        diff = torch.nn.functional.l1_loss(rf.render_sh_voxel_grid(self.grid, rays, dataclasses.replace(cfg, render_diffuse=True), t_rand=t_diff).colour, pixels)
        opt = self.opt if self.kind == "torch_optim" else self.stepper.optimizer
        opt.zero_grad()
        (spec + diff).backward()
        opt.step()
        return float(spec.detach()), float(diff.detach())

    def end_stage(self):
        if self.kind != "torch_optim":
            self.stepper.flat.detach()


def _heldout(run, g, dev):
    hw, eval_S = int(g["config"][2]), int(g["config"][8])
    pose = rf.CameraPose(T(g["heldout_rotation"]).to(dev), T(g["heldout_translation"]).to(dev))
    intr = rf.CameraIntrinsics(hw, hw, float(g["intrinsics_stage2"][2]))
    return run.model.render(pose, intr, perturb_sampled_points=False, num_samples_per_ray=eval_S).colour.cpu().numpy()


def _follow(kind, storage, dev, g, dens0, feat0, checks=None):
    """Follow the reference trainer's schedule from the given initial parameters; returns (follower, final grid, per-step relative loss
    differences).  ``checks``: callbacks (step, grid, run) evaluated after every iteration."""
    G, deg, hw, n_img, n_rays, iters, S, stages, eval_S, seed0 = (int(v) for v in g["config"])
    near, far = float(g["near"]), float(g["far"])
    grid = relu_grid(dev, dens0, feat0, storage)
    run = _Follower(kind, dev, g)
    rel = []
    for stage in range(stages):
        run.start_stage(grid, g9b_learning_rate(g, stage * iters), S, near, far)
        for it in range(iters):
            step = stage * iters + it
            assert abs(run.lr() - g9b_learning_rate(g, step)) <= 1e-9 * run.lr()
            o, d, px, t_spec, t_diff = (T(a).to(dev) for a in g9b_batch(g, step))
            ls, ld = run.step(rf.Rays(o, d), px, t_spec, t_diff)
            rel.append(max(abs(ls / g["specular_loss"][step] - 1.0), abs(ld / g["diffuse_loss"][step] - 1.0)))
            if (it + 1) % int(g["schedule"][2]) == 0:
                run.sched.step()
            if checks is not None:
                checks(step, grid, run)
        run.end_stage()
        if stage == 0:
            if checks is not None:
                checks("transition", grid, run)
            with torch.no_grad():
                grid = rf.scale_voxel_grid_with_required_output_size(grid, (G, G, G)).to(dev)
            assert grid.storage == storage and grid.grid_dims == (G, G, G)
    return run, grid, np.array(rel)


# ``policy``: which adjoint / optimizer machinery backward="auto" and optim.FlatGrid(deferred) pick per stage.  tests/conftest.py sets
# RF_AUTO_BINNED_MIN_BRICKS=0 (binned records + brick pass + Adam in the flush on EVERY grid); "production" removes the override -- the
# shipped default: atomic adjoint + rf_adam_step below 256 bricks, i.e. on both grids of this schedule (8 and 27 bricks of 8^3 nodes),
# what a CLI user gets on the 16^3 / 32^3 stages of the reference's schedule (modules/trainers.py:125-152); "switch" puts the
# threshold BETWEEN the two stages (atomic at 12^3, binned with Adam in the flush at 24^3): the change of machinery at a stage
# boundary that the reference's 16^3 -> 128^3 schedule makes at its third stage.
_POLICY_MIN_BRICKS = {"binned-on-every-grid": "0", "production": None, "switch": "20"}


@pytest.mark.parametrize("kind,storage,policy", [
    ("fused", "split", "binned-on-every-grid"), ("fused", "bricked", "binned-on-every-grid"), ("autograd", "reference", "binned-on-every-grid"),
    ("torch_optim", "reference", "binned-on-every-grid"),
    ("fused", "split", "production"), ("autograd", "reference", "production"), ("torch_optim", "reference", "production"),
    ("fused", "split", "switch"), ("autograd", "reference", "switch"),
    ("autograd_pair", "reference", "binned-on-every-grid"), ("autograd_pair", "reference", "switch"), ("autograd_pair", "reference", "production")])
def test_g9b_trained_field_through_the_stage_transition(hip_device, monkeypatch, kind, storage, policy):
    """What float32 allows to be asserted, and what it does not.  Adam turns rounding-level differences of near-zero gradients into
    full-size steps and the L1 loss flips the sign of a pixel's gradient at |error| ~ 1e-7, so two float32 evaluations of this
    schedule drift apart chaotically: G9b holds 12 re-runs of the REFERENCE ITSELF from initial parameters moved by one ulp --
    identical to 1e-4 dB for the first 100 iterations, 0.03-0.07 dB apart at iterations 200-300, and 40.2 ... 43.9 dB (base run:
    43.2) on the held-out view after 2 x 300 iterations.  So:
      * iterations 1-100 (held-out PSNR 14.9 -> 30.1 dB): PSNR within 0.01 dB of the reference's, pictures equal to 1e-4;
      * iterations 200, 300 (32.3 dB, end of stage 1): within 0.05 dB + the reference's own spread at that checkpoint;
      * the transition: rf_upsample_grid on the reference's stage-1 parameters == the reference's up-scaled grid, bit for bit;
      * after stage 2: a trained field (> 22 dB; here ~42 dB) inside the range of the reference's own runs (the distribution is
        compared in test_g9b_final_psnr_distribution_is_the_references)."""
    if _POLICY_MIN_BRICKS[policy] is None:
        monkeypatch.delenv("RF_AUTO_BINNED_MIN_BRICKS", raising=False)
    else:
        monkeypatch.setenv("RF_AUTO_BINNED_MIN_BRICKS", _POLICY_MIN_BRICKS[policy])
    g = load_golden("g9b_trainer_stages.npz")
    G, deg, hw, n_img, n_rays, iters, S, stages, eval_S, seed0 = (int(v) for v in g["config"])
    F = 3 * (deg + 1) ** 2
    g0 = int(np.ceil(G / 2))
    dev = hip_device
    gd_ref, gf_ref = oracle_gradient_of_step1(g)
    checkpoints = [int(c) for c in g["checkpoints"]]
    seen = {}
    machinery = []  # per stage: what the policy picked

    def checks(step, grid, run):
        if step in (0, iters) and kind != "torch_optim":
            st = run.stepper
            machinery.append(("binned" if (st.backward == "binned" or st.flat.deferred) else "atomic", bool(st.fuse_optimizer or st.flat.deferred)))
        if step == 0:
            # Adam's first update is lr * sign(g) wherever |g| >> eps: every parameter with a gradient that is not summation noise
            # (|g| > 1e-6; the L1 gradients of this batch are ~1e-4) must agree with the reference's to 2e-5 -- ALL of them
            for ours, ref, grad in ((grid.densities, g["dens_after_step1"], gd_ref), (grid.features, g["feat_after_step1"], gf_ref)):
                err = np.abs(ours.detach().cpu().numpy() - ref)
                firm = np.abs(grad) > 1e-6
                assert firm.mean() > 0.2 and err[firm].max() <= 2e-5, (firm.mean(), err[firm].max())
                assert err.max() <= 2 * g9b_learning_rate(g, 0) + 1e-6  # the others: at most one update in the other direction
        elif step == "transition":
            dd = np.abs(grid.densities.detach().cpu().numpy() - g["dens_stage1_end"])
            df = np.abs(grid.features.detach().cpu().numpy() - g["feat_stage1_end"])
            assert np.mean(dd < 5e-3) > 0.85 and np.mean(df < 5e-3) > 0.85, (np.mean(dd < 5e-3), np.mean(df < 5e-3))
            # the transition itself, on the REFERENCE's stage-1 parameters: rf_upsample_grid == the reference's up-scaled grid, bit for bit
            ref_small = relu_grid(dev, T(g["dens_stage1_end"]), T(g["feat_stage1_end"]), storage, tunable=False)
            up = rf.scale_voxel_grid_with_required_output_size(ref_small, (G, G, G))
            keep = g["upscaled_nodes_kept"]
            assert np.array_equal(up.densities.cpu().numpy()[keep], g["dens_upscaled_kept"])
            assert np.array_equal(up.features.cpu().numpy()[keep], g["feat_upscaled_kept"])
        elif step + 1 in checkpoints:
            k = checkpoints.index(step + 1)
            img = _heldout(run, g, dev)
            ours, ref = psnr(img, g["heldout_truth"]), float(g["checkpoint_heldout_psnr"][k])
            band = float(np.abs(g["rerun_checkpoint_heldout_psnr"][:, k] - ref).max())  # the reference against its own one-ulp re-runs
            seen[step + 1] = (ours, ref, band)
            if step + 1 <= 100:
                assert band < 1e-3 and abs(ours - ref) <= 0.01, (step + 1, ours, ref, band)
                assert np.abs(img - g["checkpoint_heldout_render"][k]).max() <= 1e-4
            else:
                assert abs(ours - ref) <= 0.05 + band, (step + 1, ours, ref, band)

    run, grid, rel = _follow(kind, storage, dev, g, T(hash_uniform((g0, g0, g0, 1), 901)), T(hash_uniform((g0, g0, g0, F), 900 + F)), checks)
    assert sorted(seen) == checkpoints and seen[100][1] > 30.0  # (PSNR within 0.01 dB after 100 equal steps, at 30 dB)
    if kind != "torch_optim":  # the machinery the policy is meant to pick, per stage (binned adjoint?, Adam inside the brick flush?)
        want = {"binned-on-every-grid": [("binned", True)] * 2, "production": [("atomic", False)] * 2, "switch": [("atomic", False), ("binned", True)]}[policy]
        assert machinery == want, (machinery, want)
    # per-step losses: tight while the trajectories coincide, inside (a multiple of) the reference's own spread afterwards
    ref_rel = np.abs(g["rerun_specular_loss"] / g["specular_loss"][None] - 1.0)
    assert rel[:3].max() <= 2e-5 and rel[:100].max() <= 2e-3, (rel[:3].max(), rel[:100].max())
    assert np.median(rel[iters:]) <= 4.0 * np.median(ref_rel[:, iters:]) and rel.max() < 1.0, (np.median(rel[iters:]), np.median(ref_rel[:, iters:]), rel.max())
    # ---- the trained field ----
    ours = psnr(_heldout(run, g, dev), g["heldout_truth"])
    ensemble = np.concatenate([[float(g["heldout_psnr"])], g["rerun_heldout_psnr"]])
    assert ours >= 22.0 and ensemble.min() - 0.75 <= ours <= ensemble.max() + 0.75, (ours, ensemble)
    intr = rf.CameraIntrinsics(hw, hw, float(g["intrinsics_stage2"][2]))
    train0 = run.model.render(rf.pose_spherical(5.0, -20.0, 4.0311), intr, perturb_sampled_points=False, num_samples_per_ray=eval_S)
    t0 = psnr(train0.colour.cpu().numpy(), g["train0_truth"])
    ensemble0 = np.concatenate([[float(g["train0_psnr"])], g["rerun_train0_psnr"]])
    assert ensemble0.min() - 0.75 <= t0 <= ensemble0.max() + 0.75, (t0, ensemble0)


def _one_ulp_init(g, seed):
    """the initial parameters of oracle/gen_golden_trainer.py's re-run ``seed`` (0 = the base run)"""
    G, deg = int(g["config"][0]), int(g["config"][1])
    g0, F = int(np.ceil(G / 2)), 3 * (deg + 1) ** 2
    d, f = hash_uniform((g0, g0, g0, 1), 901), hash_uniform((g0, g0, g0, F), 900 + F)
    if seed:
        for arr in (d, f):
            flip = hash_uniform(arr.shape, 7000 + 31 * seed + arr.shape[-1], 0.0, 1.0) < 0.5
            arr[...] = np.where(flip, np.nextafter(arr, np.float32(2.0)), arr)
    return T(d), T(f)


def test_g9b_final_psnr_distribution_is_the_references(hip_device):
    """PSNR after equal training steps, where single runs cannot be compared (see above): the default HIP step from the same 13
    initialisations as the reference's base run + 12 one-ulp re-runs (batches and jitter identical).  The two samples of final
    held-out PSNR must have the same mean within 3 standard errors (measured: 42.2 vs 42.3 dB, sigma 0.9) and a comparable spread."""
    g = load_golden("g9b_trainer_stages.npz")
    ref = np.concatenate([[float(g["heldout_psnr"])], g["rerun_heldout_psnr"]])
    ours = []
    for seed in range(len(ref)):
        d0, f0 = _one_ulp_init(g, seed)
        run, grid, rel = _follow("fused", "split", hip_device, g, d0, f0)
        ours.append(psnr(_heldout(run, g, hip_device), g["heldout_truth"]))
    ours = np.array(ours)
    se = np.sqrt(ref.var(ddof=1) / len(ref) + ours.var(ddof=1) / len(ours))
    assert ours.min() >= 22.0 and abs(ours.mean() - ref.mean()) <= max(3.0 * se, 0.05), (ours, ref, se)
    assert 0.3 * ref.std(ddof=1) <= ours.std(ddof=1) <= 3.0 * ref.std(ddof=1), (ours.std(ddof=1), ref.std(ddof=1))
