"""Posed-image training of an SH voxel grid -- the build's counterpart of the reference's
``train_sh_vox_grid_vol_mod_with_posed_images`` (thre3d_atom/modules/trainers.py:49-514).

What is reproduced: the per-iteration core (:278-341) -- a batch of
``image_batch_cache_size`` images, one synchronous random subset of ``ray_batch_size`` rays/pixels out of
ALL their pixels (``torch.randperm``, rendering/volumetric/utils/misc.py:117-129), a specular and a
diffuse render of the same rays with independent jitter, ``L1 + L1``, Adam(0.9, 0.999) -- plus the stage
schedule (:125-152, :227-250, :462-470): grids of ceil(G / 2^k), U(-1,1) re-initialisation, per-stage
learning rate and ExponentialLR, x2 trilinear up-scaling between stages, and checkpoints in the
reference's dictionary layout.  TensorBoard, LPIPS and feedback-image writing are out of scope.

What is done the MI355X way instead of translated:
  * rays are generated only for the selected pixels (rf_cast_selected_rays) instead of casting
    8 x H x W rays and discarding all but 16384 of them every iteration;
  * gradients accumulate in one flat bucket and Adam is one fused kernel (optim.py); the specular backward of the
    fused step bins per-sample gradient records by 8^3-node brick and sums every brick on chip (MFMA accumulators) without atomics;
  * with WORLD_SIZE > 1 every rank draws its own ray batch and the ranks exchange gradient RECORDS by owner over RCCL (all-gather of
    the offset tables -> all-to-all of record slices -> every rank sums its own x-slabs of bricks and applies Adam there -> in-place
    all-gather of the parameters, pipelined in two interleaved halves: TrainStepper._owner_step); grids the
    owner step does not cover exchange one flat gradient bucket (reduce-scatter, Adam on 1/N of the grid per rank, all-gather).
"""
import ctypes as C
import dataclasses
import os
import time
from pathlib import Path
from typing import Callable, Dict, Iterator, List, Optional

import numpy as np
import torch
from torch import Tensor
from torch.nn.functional import mse_loss

from . import _lib, ops
from . import distributed as rfdist
from .camera import CameraBounds, CameraIntrinsics, compute_thre3d_grid_sizes, mse2psnr, scale_camera_intrinsics
from .constants import CAMERA_BOUNDS, CAMERA_INTRINSICS, HEMISPHERICAL_RADIUS
from .ops import (
    BRICK_4X8X8,
    brick_accumulate_adam_raw,
    brick_accumulate_raw,
    brick_counts,
    cast_selected_rays_hip,
    render_backward_emit_raw,
    render_backward_emit_direct_raw,
    bin_offsets,
    sort_records_by_brick,
    expanded_record_floats,
    l1_loss_grad_hip,
    render_backward_raw,
    render_flags,
    render_forward_raw,
    select_rays_and_pixels_hip,
)
from .optim import ExponentialLR, FlatGrid, FusedAdam
from .render_interface import Rays
from .renderers import _check_supported, render_sh_voxel_grid
from .volumetric_model import VolumetricModel
from .voxels import VoxelGrid, scale_voxel_grid_with_required_output_size


class PosedImagesInMemory:
    """Images [M, 3, H, W] in [0, 1] + camera-to-world poses [M, 3, 4] resident on the device: the
    *outputs* of the reference's PosedImagesDataset (data/datasets.py:31; cached mode), without its disk
    loader.  Attribute names follow the reference where the trainer reads them."""

    def __init__(
        self,
        images: Tensor,
        poses: Tensor,
        camera_intrinsics: CameraIntrinsics,
        camera_bounds: CameraBounds,
        downsample_factor: float = 1.0,
    ):
        assert images.dim() == 4 and images.shape[1] == 3 and poses.shape[1:] == (3, 4)
        assert images.shape[0] == poses.shape[0]
        self.images = images.to(torch.float32).contiguous()
        self.poses = poses.to(torch.float32).contiguous()
        self.camera_intrinsics = camera_intrinsics
        self.camera_bounds = camera_bounds
        self.downsample_factor = downsample_factor
        self.cached_data_mode = True
        # [M * H * W, 3] pixel table in ray order (i * W + j), built once
        self.pixels = self.images.permute(0, 2, 3, 1).reshape(-1, 3).contiguous()

    def __len__(self) -> int:
        return self.images.shape[0]

    def __getitem__(self, i):
        return self.images[i], self.poses[i]

    def get_hemispherical_radius_estimate(self) -> float:
        return float(self.poses[:, :, 3].norm(dim=-1).mean().item())

    def downsampled(self, factor: float) -> "PosedImagesInMemory":
        """Images resized by 1/factor (bilinear, antialias off like torchvision Resize on tensors is not
        reproduced bit-for-bit; the multi-stage schedule only needs the coarse images)."""
        if factor == 1.0:
            return self
        intr = scale_camera_intrinsics(self.camera_intrinsics, 1.0 / factor)
        imgs = torch.nn.functional.interpolate(
            self.images, size=(intr.height, intr.width), mode="bilinear", align_corners=False, antialias=True
        ).clamp(0.0, 1.0)
        return PosedImagesInMemory(imgs, self.poses, intr, self.camera_bounds, self.downsample_factor * factor)

    def image_batches(self, batch_size: int, generator: Optional[torch.Generator] = None, prefetch_epochs: int = 256) -> Iterator[Tensor]:
        """Endless stream of image-index batches: shuffled epochs, drop_last (the reference's
        DataLoader(shuffle=True, drop_last=True) wrapped in infinite_dataloader, trainers.py:164-166).
        The shuffles come from a generator of their own (seeded from torch's global CPU generator when none is given:
        ``torch.manual_seed`` keeps runs reproducible) and ``prefetch_epochs`` of them are drawn and copied to the device at a time;
        the per-iteration batches are device-side slices.  A host->device copy per iteration is what this avoids: it comes out of
        pageable memory, so the host waits for the stream to drain and the GPU then idles for a launch latency (~55 us of a
        0.65 ms iteration, rocprofv3 kernel trace) -- and with 8 images and 8 images per batch every iteration is an epoch."""
        m = len(self)
        batch_size = min(batch_size, m)
        if generator is None:
            generator = torch.Generator()
            generator.manual_seed(int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item()))
        while True:
            orders = torch.stack([torch.randperm(m, generator=generator) for _ in range(int(prefetch_epochs))]).to(self.images.device)
            for order in orders:
                for s in range(0, m - batch_size + 1, batch_size):
                    yield order[s : s + batch_size]


class StepStats:
    """Losses of one iteration as 0-d device tensors: ``specular_loss`` / ``diffuse_loss`` (mean L1) and ``specular_mse`` /
    ``diffuse_mse`` (for the PSNR the reference logs, trainers.py:315-317, 334-336).  The fused step hands over the raw sums its loss
    kernel wrote -- (sum |d|, sum d^2) per render, in a slot of a ring of ``LOSS_RING`` iterations -- and the division by 3 N happens
    when a value is READ: an iteration enqueues no extra launch for numbers that are looked at every ``summary_freq`` steps.  The ring
    slot is re-used LOSS_RING iterations later: a StepStats read after that RAISES instead of returning another iteration's sums
    (``ring`` = (the executor's step counter, the step that produced this object)); ``materialize()`` copies the four sums out of
    the ring (one small launch) for code that keeps a history of StepStats objects."""

    def __init__(self, specular_loss=None, diffuse_loss=None, specular_mse=None, diffuse_mse=None, sums: Optional[Tensor] = None, count: float = 1.0,
                 has_diffuse: bool = True, ring=None):
        self._values = (specular_loss, diffuse_loss, specular_mse, diffuse_mse)
        self._sums, self._count, self._has_diffuse = sums, float(count), has_diffuse
        self._ring = ring  # (executor dict holding "serial", serial of the producing step) or None

    def _live_sums(self) -> Tensor:
        if self._ring is not None:
            ex, serial = self._ring
            if ex["serial"] - serial >= LOSS_RING:
                raise RuntimeError(f"this StepStats was produced {ex['serial'] - serial} iterations ago: its slot of the loss ring ({LOSS_RING} iterations) "
                                   "has been re-used -- read it earlier or keep StepStats.materialize() of it")
        return self._sums

    def materialize(self) -> "StepStats":
        """Detach from the ring: the four sums are copied now (in stream order), the object stays valid for ever."""
        if self._sums is not None and self._ring is not None:
            self._sums = self._live_sums().clone()
            self._ring = None
        return self

    def _get(self, i: int, j: int):
        if self._sums is None:
            return self._values[i]
        if i in (1, 3) and not self._has_diffuse:
            return None
        return self._live_sums()[j] / self._count

    specular_loss = property(lambda self: self._get(0, 0))
    specular_mse = property(lambda self: self._get(2, 1))
    diffuse_loss = property(lambda self: self._get(1, 2))
    diffuse_mse = property(lambda self: self._get(3, 3))

    def psnr(self) -> Dict[str, float]:
        out = {"specular_psnr": float(mse2psnr(self.specular_mse))}
        if self.diffuse_mse is not None:
            out["diffuse_psnr"] = float(mse2psnr(self.diffuse_mse))
        return out


LOSS_RING = 4096  # iterations whose loss sums stay readable (StepStats)


# owner-computes data parallelism: leave the all-gather of the `rest` parameters in flight across the iteration boundary (RF_OWNER_OVERLAP_PARAMETERS=0
# makes every iteration wait for it at its end instead)
OWNER_OVERLAP_PARAMETERS = os.environ.get("RF_OWNER_OVERLAP_PARAMETERS", "1") != "0"
# The PIPELINED form of the owner-computes step (interleaved halves, all-gathers in flight across the iteration boundary, several
# workgroups per owned brick, ProcessGroup entry points without wrappers) has run under RCCL with ONE rank only (no multi-GPU box in
# any round; every multi-rank test goes through gloo, where collectives complete on return).  Until a multi-GPU run has validated it,
# a multi-rank RCCL group gets the CONSERVATIVE form by default -- one contiguous ownership range per rank, every all-gather waited
# for at the end of its iteration, one workgroup per brick, torch.distributed's public collectives -- and $RF_OWNER_PIPELINED=1 opts
# in (bench.py's first, supervised and self-validating attempt does).  Other backends (gloo: the tests) keep the pipelined form.
OWNER_PIPELINED = os.environ.get("RF_OWNER_PIPELINED")


def owner_pipelined() -> bool:
    if OWNER_PIPELINED is not None:
        return OWNER_PIPELINED != "0"
    multi_rank_rccl = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl" and torch.distributed.get_world_size() > 1
    return not multi_rank_rccl
# owner-computes data parallelism: the x-slabs of bricks are owned in this many interleaved "halves" (1 = one contiguous range per rank):
# the brick pass of half h + 1 runs while the parameters of half h are all-gathered (TrainStepper._owner_state)
OWNER_HALVES = int(os.environ.get("RF_OWNER_HALVES", "2"))
# ... and how many workgroups share one owned brick in the owner's brick pass (rf_brick_accumulate_adam_split; 0 = by world size: 2 from
# four ranks on).  With N ranks an owned brick receives N times the records while the launch covers 1 / N of the bricks: one workgroup per
# brick leaves the machine idle behind the heaviest bricks (tools/owner_brick_emulation.py, N = 8: the piece through the middle of the
# volume 0.35 ms with one workgroup per brick, 0.18 ms with two; 4 and 8 are no faster: the partial images then cost what the split saves)
OWNER_BRICK_PARTS = int(os.environ.get("RF_OWNER_BRICK_PARTS", "0"))
# ... and on how many of a run's first iterations the replicas' parameters are compared (a float64 checksum all-gathered, one device
# synchronisation each; bench.py validates a multi-GPU configuration on them before it times anything)
OWNER_CHECK_STEPS = int(os.environ.get("RF_OWNER_CHECK_STEPS", "3"))


class _ParameterWait:
    """Makes the CURRENT stream wait for an all-gather that is still in flight on RCCL's stream (its work handle; None: the backend
    completed the call on return).  ``tag``: which parameter tensor it fills ("base" / "rest")."""

    def __init__(self, work, tag="rest"):
        self.work, self.tag = work, tag

    def __call__(self) -> None:
        if self.work is not None:
            self.work.wait()
            self.work = None


class TrainStepper:
    """One optimisation step of modules/trainers.py:278-341 on a grid held in a FlatGrid bucket."""

    def __init__(
        self,
        vol_mod: VolumetricModel,
        ray_batch_size: int,
        learning_rate: float,
        apply_diffuse_render_regularization: bool = True,
        data_parallel: bool = True,
        ray_selection: str = "keyed",
        fused: bool = True,
        backward: str = "auto",
        deterministic: bool = False,
        shard_optimizer: bool = True,
        global_batch: bool = False,
        merge_bricks: Optional[bool] = None,
        fuse_optimizer: Optional[bool] = None,
        exchange: str = "auto",
        brick_size: Optional[int] = None,
    ):
        """``brick_size``: 8 or 4 (cubic bricks of the binned backward) or ops.BRICK_4X8X8; None = $RF_BRICK_SIZE, else 4 x 8 x 8 for the
        single-process step with Adam in the brick flush and 8 otherwise.

        ``ray_selection``: "randperm" draws torch.randperm over all B*H*W pixels exactly like the reference
        (utils/misc.py:123) and keeps the first ``ray_batch_size``; "keyed" draws the same kind of sample
        (distinct, uniformly random pixels) with one fused kernel (rf_select_rays_and_pixels) keyed from torch's
        CPU generator -- no 5-million-key sort per iteration; "randperm_blocks" is for users who want torch.randperm's own draws
        without paying one per iteration (0.42 ms for 5.12 M pixels, more than half a step): ONE torch.randperm over all B*H*W pixels
        per block of floor(B*H*W / ray_batch_size) iterations, consumed in consecutive slices -- every iteration's batch is still
        ``ray_batch_size`` distinct uniformly random pixels; batches of one block do not repeat a pixel (sampling without replacement
        across the block, where the reference re-draws independently per iteration)."""
        if ray_selection not in ("keyed", "randperm", "randperm_blocks"):
            raise ValueError("ray_selection must be 'keyed', 'randperm' or 'randperm_blocks'")
        self.ray_selection = ray_selection
        # fused=True runs the iteration as a fixed sequence of launches (forward, loss+gradient, backward per
        # render, then Adam which also clears the gradient bucket) without building an autograd graph;
        # fused=False goes through torch.autograd like a user of render_rays would.  Same arithmetic either way.
        self.fused = bool(fused)
        self._grad_clean = True  # FlatGrid starts zero-filled
        # backward="binned" (fused steps only): both passes write per-sample gradient records, bin them by
        # (8^3-node brick, boundary flags) and sum each brick on chip without float atomics -- one
        # brick pass over both record lists on a single GPU (merge_bricks), one per render in the data-parallel order.
        # deterministic=True bins with a stable radix sort instead of the counting sort: no atomics anywhere, fixed
        # float32 summation order, run-to-run bit-identical gradients (slower).
        # "auto" = binned for SH degree 2 and 3 (fused step; measured faster at degree 2), else atomic.
        if backward not in ("auto", "atomic", "binned"):
            raise ValueError("backward must be 'auto', 'atomic' or 'binned'")
        self.deterministic = bool(deterministic)
        # data parallel: every rank runs Adam on 1/N of the grid only (ZeRO stage 1) -- see _fused_step_on
        self.shard_optimizer = bool(shard_optimizer)
        # data parallel, strong scaling: ``ray_batch_size`` is the GLOBAL batch; every rank draws the SAME random
        # permutation (identical CPU RNG state on all ranks is the caller's job: seed them equally) and takes its own
        # contiguous slice of it, so that N ranks reproduce the single-GPU iteration up to float summation order.
        # Default (False) = weak scaling: every rank draws its own ``ray_batch_size`` rays.
        self.global_batch = bool(global_batch)
        brick_size_given = brick_size is not None or "RF_BRICK_SIZE" in os.environ
        self.brick_size = int(brick_size) if brick_size is not None else int(os.environ.get("RF_BRICK_SIZE", "8"))
        self._bins = None
        self._exec = None
        self.step_events = None  # an ops.StepEvents: the next rf_train_step call records its per-launch HIP events there
        self.host_timing = None  # a list: the owner-computes step appends the host-side durations of its sections (seconds)
        grid = vol_mod.thre3d_repr
        if not isinstance(grid, VoxelGrid):
            raise AssertionError(f"cannot train a {type(grid)}; only a VoxelGrid can be used")
        if vol_mod.render_procedure is not render_sh_voxel_grid:
            raise AssertionError("only the SH voxel-grid render procedure can be used with this trainer")
        self.vol_mod = vol_mod
        self.ray_batch_size = int(ray_batch_size)
        self.diffuse = bool(apply_diffuse_render_regularization)
        self.data_parallel = data_parallel
        # autograd steps (fused=False) on a grid in the reference's own tensors: the backward passes leave record lists and the
        # optimizer sums them in one merged brick pass with Adam in its flush (optim.FlatGrid(deferred=True))
        self.flat = FlatGrid(grid, deferred=not self.fused and not (self.data_parallel and rfdist._collectives_on()))
        self.optimizer = FusedAdam(self.flat, lr=learning_rate, betas=(0.9, 0.999))
        if backward == "auto":
            # binned from 256 bricks of 8^3 nodes on: the brick pass has one workgroup per brick, and on the first grids of a progressive
            # schedule a handful of workgroups would sum two million records (16^3: 2.26 ms per iteration against 0.79 ms with the atomic
            # adjoint, whose targets then sit in L2; 32^3: 1.13 / 0.49; 64^3: 0.37 / 0.44; 128^3: 0.50 / 1.33 -- tools/small_grid_steps.py)
            # ($RF_AUTO_BINNED_MIN_BRICKS: the tests run the binned machinery on small grids -- tests/conftest.py sets 0)
            nb = brick_counts(grid, 8)
            bricks = nb[0] * nb[1] * nb[2]
            min_bricks = int(os.environ.get("RF_AUTO_BINNED_MIN_BRICKS", "256"))
            backward = "binned" if (self.fused and grid.sh_degree >= 2 and min_bricks <= bricks and bricks * 16 <= (1 << 21)) else "atomic"
        self.backward = backward
        # merge_bricks (binned, non-deterministic steps with the diffuse regulariser): BOTH renders emit their records first and
        # ONE brick pass sums them (the 4-channel diffuse records in the first channel columns of the same accumulators), so
        # the atomic diffuse scatter (0.28 ms at 0.13 of the HBM roofline in round 1) disappears.  Default: on, except under
        # data parallelism, where the order "specular bricks -> exchange of the `rest` gradients overlapped with the diffuse
        # pass" is kept.
        single = not (self.data_parallel and rfdist._collectives_on())
        can_merge = self.fused and backward == "binned" and not self.deterministic and self.diffuse
        self.merged_bricks = can_merge and (single if merge_bricks is None else bool(merge_bricks))
        if merge_bricks and not can_merge:
            raise ValueError("merge_bricks needs the fused, binned, non-deterministic step with the diffuse render")
        # fuse_optimizer (merged steps on one GPU, split/bricked storage, SH degree 0 or 2 = whole float4s per node): the brick
        # flush applies the Adam update itself (rf_brick_accumulate_adam) -- no gradient bucket in HBM, no separate optimizer pass
        # (the flush keeps element offsets in 31 bits: rf_brick_accumulate_adam refuses larger grids, e.g. 512^3 at degree 2 --
        # those keep the gradient bucket and the separate optimizer kernel)
        padded_nodes = 1
        for d in grid.grid_dims:
            padded_nodes *= (d + 7) // 8 * 8
        fits_flush = padded_nodes * max(4, grid.num_features - 3) < (1 << 31)
        can_fuse = self.merged_bricks and single and grid.storage != "reference" and (grid.num_features + 1) % 4 == 0 and fits_flush
        # exchange (data parallel; the reference has none: modules/trainers.py:338-341 is one device's backward + step):
        #   "owner": OWNER-COMPUTES.  Every rank owns an equal range of x-slabs of bricks.  The ranks exchange their gradient RECORDS
        #     (48 / 32 B each, already sorted by brick, so what an owner needs of a rank's list is ONE slice) instead of the dense
        #     235 MB gradient; each owner runs the merged brick pass with Adam in its flush on its own bricks (1/N of the optimizer
        #     traffic per rank), then the parameters are all-gathered.  ~280 MB per rank and step over xGMI at N = 8 instead of 411.
        #   "dense": reduce-scatter of the gradient bucket -> sharded Adam -> all-gather (_fused_step_on).
        #   "auto": owner where it applies (fused, merged, split/bricked storage, SH degree 0 or 2, X divisible by 8 N, N <= 8).
        if exchange not in ("auto", "owner", "dense"):
            raise ValueError("exchange must be 'auto', 'owner' or 'dense'")
        world = rfdist.world_size()
        can_owner = (not single and can_merge and merge_bricks is not False and fuse_optimizer is not False and grid.storage != "reference"
                     and (grid.num_features + 1) % 4 == 0 and fits_flush and world <= (8 if grid.num_features > 3 else 4) and self.brick_size == 8
                     and grid.grid_dims[0] % (8 * world) == 0)  # (a degree-0 grid's 2 W lists are all of ONE kind: at most 8 per brick pass)
        if exchange == "owner" and not can_owner:
            raise ValueError("exchange='owner' needs the fused, merged step on split or bricked storage, SH degree 0 or 2, at most 8 ranks (4 at degree 0) and X divisible by 8 x ranks")
        self.exchange = "owner" if (can_owner and exchange != "dense") else "dense"
        self.owner_records = []
        self.exchange_bytes = []  # bytes this rank SENT in each of the last steps (record slices + offset tables + parameter chunks)
        self.phase_events = None  # a list: the owner step appends a tuple of torch events around its phases (bench.py)
        self._owner = None
        if self.exchange == "owner":
            self.merged_bricks = True
            can_fuse = True
        self.fuse_optimizer = can_fuse if fuse_optimizer is None else bool(fuse_optimizer)
        if self.fuse_optimizer and not can_fuse:
            raise ValueError("fuse_optimizer needs a merged brick pass on a single process, split or bricked storage, SH degree 0 or 2 and fewer than 2^31 elements per grid tensor")
        # one process, Adam in the flush: bricks of 4 x 8 x 8 nodes -- four 256-thread workgroups per CU sum them instead of two
        # 512-thread ones (RF_BRICK_4X8X8; the flush needs its 32-bit byte offsets: every grid tensor below 2^30 elements)
        if (not brick_size_given and single and self.fuse_optimizer and self.exchange != "owner" and grid.num_features in (3, 27)
                and padded_nodes <= (1 << 24) and padded_nodes * max(4, grid.num_features - 3) < (1 << 30)):
            self.brick_size = BRICK_4X8X8

    def select(self, dataset: PosedImagesInMemory, image_ids: Tensor):
        """Synchronous random subset of rays and pixels of the given images
        (utils/misc.py:117-129): randperm over all B*H*W pixels, first ``ray_batch_size`` kept."""
        intr = dataset.camera_intrinsics
        hw = intr.height * intr.width
        dev = dataset.pixels.device
        total = min(self.ray_batch_size, image_ids.numel() * hw)
        lo, hi = self._global_slice(total)
        self._batch_first = lo  # position of this rank's rays in the (global) batch: offset into the keyed jitter streams
        if self.ray_selection == "keyed":
            key = int(torch.randint(-(2**63), 2**63 - 1, (1,), dtype=torch.int64).item())
            o, d, px = select_rays_and_pixels_hip(intr.height, intr.width, float(intr.focal), dataset.poses, image_ids, dataset.pixels, hi - lo, key, first_index=lo)
            return Rays(o, d), px
        image_ids = image_ids.to(dev)
        if self.ray_selection == "randperm_blocks" and not self.global_batch:
            block = getattr(self, "_perm_block", None)
            if block is None or block[0].numel() != image_ids.numel() * hw or block[1] + total > block[0].numel():
                block = self._perm_block = [torch.randperm(image_ids.numel() * hw, dtype=torch.long, device=dev), 0]
            perm = block[0][block[1] : block[1] + total]
            block[1] += total
        else:
            perm = torch.randperm(image_ids.numel() * hw, dtype=torch.long)[lo:hi].to(dev) if self.global_batch else torch.randperm(image_ids.numel() * hw, dtype=torch.long, device=dev)[:total]
        poses = dataset.poses[image_ids]
        origins, directions = cast_selected_rays_hip(intr.height, intr.width, float(intr.focal), poses, perm)
        b = torch.div(perm, hw, rounding_mode="floor")
        pixels = dataset.pixels[image_ids[b] * hw + (perm - b * hw)]
        return Rays(origins, directions), pixels

    def _unit_gradient(self, loss: Tensor) -> Tensor:
        """the seed of ``loss.backward()`` -- a one of the loss's shape -- kept between iterations (autograd would fill a fresh one per call)"""
        one = getattr(self, "_one", None)
        if one is None or one.device != loss.device or one.shape != loss.shape:
            one = self._one = torch.ones_like(loss)
        return one

    def _jitter_first(self) -> int:
        """Offset of this rank's rays in the keyed jitter streams: its position in the global batch (0 unless ``global_batch``)."""
        return int(getattr(self, "_batch_first", 0)) if (self.global_batch and self.data_parallel) else 0

    def _global_slice(self, total: int):
        """[lo, hi) of the permutation prefix this rank renders (everything unless ``global_batch`` under data parallelism)."""
        if not (self.global_batch and self.data_parallel):
            return 0, total
        # equal shares only: every rank scales its L1 gradient by 1 / (3 * own rays) and the ranks are averaged with equal
        # weight, which is the global-batch mean only when all ranks hold the same number of rays
        if total % rfdist.world_size() != 0:
            raise ValueError(f"global_batch: the ray batch ({total}) must be divisible by the world size ({rfdist.world_size()})")
        return rfdist.shard_range(total)

    def step_on(self, rays: Rays, pixels: Tensor, t_rand=None) -> StepStats:
        """One iteration on the given rays / target pixels.  ``t_rand`` = (jitter of the specular render, jitter of the diffuse
        render), each [N, S] in [0, 1), replaces the draws of ``perturb_sampled_points`` (parity tests)."""
        vol_mod = self.vol_mod
        cfg = vol_mod.render_config
        grid = vol_mod.thre3d_repr
        if cfg.use_occupancy_mask and not grid.occupancy_current():
            grid.build_occupancy()  # densities changed in the last optimizer step
        if self.fused:
            if not self.merged_bricks:
                stats = self._fused_step_on(rays, pixels, t_rand)
            elif ops.KERNEL_TIMER is not None and self.exchange != "owner":
                stats = self._merged_step_pieces(rays, pixels, t_rand)
            else:
                stats = self._merged_step_on(rays, pixels, t_rand)
            grid.invalidate_occupancy()
            return stats
        if t_rand is not None:
            raise ValueError("t_rand is only taken by the fused step (pass t_rand= to render_sh_voxel_grid on the autograd path)")
        self.optimizer.zero_grad()
        # the reference's loss lines (modules/trainers.py:311-317, 329-336: l1_loss for the gradient, mse_loss for the logged PSNR) as
        # ONE launch per render: ops.l1_loss_with_mse, a torch.autograd.Function like the render op itself.
        # (The diffuse render, its loss and its adjoint on a side stream -- so that the two forward kernels and the two adjoints share
        # the machine like the fused step's paired launches -- measured no gain: 0.822 against 0.812 ms per iteration; each kernel
        # fills the machine by itself.  docs/experiments.md D.)
        if self.diffuse and ops.PAIR_RENDERS:
            # both renders as ONE autograd node and both loss lines as one (VolumetricModel.render_rays_pair, ops.l1_loss_pair_with_mse):
            # the launches of the fused step's pairs -- one forward, one loss, [backward:] one offsets, one emit -- under autograd
            spec_out, diff_out = vol_mod.render_rays_pair(rays)
            total, spec_loss, spec_mse, diff_loss, diff_mse = ops.l1_loss_pair_with_mse(spec_out.colour, diff_out.colour, pixels)
        else:
            spec = vol_mod.render_rays(rays).colour
            total, spec_mse = ops.l1_loss_with_mse(spec, pixels)
            spec_loss, diff_loss, diff_mse = total.detach(), None, None
            if self.diffuse:
                diff = vol_mod.render_rays(rays, render_diffuse=True).colour
                dl, diff_mse = ops.l1_loss_with_mse(diff, pixels)
                total = total + dl
                diff_loss = dl.detach()
        total.backward(self._unit_gradient(total))
        if self.data_parallel:
            rfdist.all_reduce_mean_(self.flat.flat_grad)
        self.optimizer.step()
        if not self.flat.deferred:  # (the deferred step re-laid the Parameters out and marked shadow and mask state itself)
            grid.invalidate_occupancy()
        self._grad_clean = False
        return StepStats(spec_loss, diff_loss, spec_mse, diff_mse)

    def _draw_jitter(self, cfg, n: int, S: int, device, given, i: int, first_ray: int = 0):
        if given is not None:
            t = given[i].detach().to(device, torch.float32).contiguous()
            if tuple(t.shape) != (n, S):
                raise ValueError(f"t_rand[{i}] must be [{n}, {S}], got {tuple(t.shape)}")
            return t
        if not cfg.perturb_sampled_points:
            t = None
        elif cfg.jitter == "keyed":  # counter-based jitter inside the kernels, keyed from torch's CPU generator
            # ``first_ray``: this rank's offset in a GLOBAL batch, so that N ranks draw the jitter of rays lo..hi of the same
            # stream the single-GPU iteration draws for rays 0..n (global_batch data parallelism)
            t = ops.KeyedJitter(ops.draw_jitter_key(), int(first_ray))
        elif cfg.jitter == "torch":
            t = torch.rand(n, S, dtype=torch.float32, device=device)
        else:
            raise ValueError("SHVoxGridRenderConfig.jitter must be 'keyed' or 'torch'")
        if cfg.consume_reference_rng:
            torch.randn(n, S, dtype=torch.float32, device=device)
        return t

    def _merged_step_on(self, rays: Optional[Rays], pixels: Optional[Tensor], t_rand=None, selection=None) -> StepStats:
        """Both renders forward (each counts its records per key), both adjoints emitted as records at their final positions,
        ONE brick pass over the two lists; the Adam update inside its flush when ``fuse_optimizer``.  The whole iteration is
        enqueued by ONE call into the library (rf_train_step, include/relu_field.h): nine launches, one FFI crossing.
        ``selection`` = (dataset, image_ids, key, first_index, count) lets the call draw the batch itself."""
        vol_mod, grid = self.vol_mod, self.vol_mod.thre3d_repr
        cfg = vol_mod.render_config
        _check_supported(cfg)
        S = int(cfg.num_samples_per_ray)
        if selection is not None:
            dataset, image_ids, key, first, n = selection
            dev = dataset.pixels.device
        else:
            origins = rays.origins.detach().to(torch.float32).contiguous()
            directions = rays.directions.detach().to(torch.float32).contiguous()
            pixels = pixels.detach().to(torch.float32).contiguous()
            n, dev = origins.shape[0], origins.device
        ex = self._executor(n, S, dev)
        st = ex["step"]
        # the loss sums of this iteration: the next slot of the ring (cleared by the selection launch / a memset of the call)
        slot = ex["slot"]
        ex["slot"] = (slot + 1) % LOSS_RING
        ex["serial"] = serial = ex.get("serial", 0) + 1
        sums = ex["sums_ring"][slot]
        st.loss_sums_dev = ex["sums_ptr"] + 16 * slot
        if selection is not None:
            intr = dataset.camera_intrinsics
            ids = image_ids.detach().to(dev, torch.int64).contiguous()
            sel = ex["select"]
            sel.height, sel.width, sel.focal = int(intr.height), int(intr.width), float(np.float32(intr.focal))
            sel.poses_dev, sel.image_ids_dev, sel.num_batch_images = dataset.poses.data_ptr(), ids.data_ptr(), int(ids.numel())
            sel.pixel_table_dev, sel.key, sel.first_index = dataset.pixels.data_ptr(), int(key) & 0xFFFFFFFFFFFFFFFF, int(first)
            st.select = C.pointer(sel)
            st.origins_dev, st.directions_dev, st.pixels_dev = ex["origins"].data_ptr(), ex["directions"].data_ptr(), ex["pixels"].data_ptr()
            keep = (ids,)
        else:
            st.select = None
            st.origins_dev, st.directions_dev, st.pixels_dev = origins.data_ptr(), directions.data_ptr(), pixels.data_ptr()
            keep = (origins, directions, pixels)
        st.near, st.far = float(np.float32(cfg.camera_bounds.near)), float(np.float32(cfg.camera_bounds.far))
        flags = render_flags(cfg.white_bkgd, False, cfg.optimized_sampling, cfg.use_occupancy_mask)
        jitter_first = int(selection[3]) if selection is not None else self._jitter_first()
        jit = [self._draw_jitter(cfg, n, S, dev, t_rand, i, jitter_first) for i in range(2)]
        st.first_ray = jitter_first
        for i in range(2):
            if isinstance(jit[i], ops.KeyedJitter):
                flags |= _lib.FLAG_JITTER_KEYED
                st.pass_[i].jitter_key, st.pass_[i].t_rand_dev = jit[i].key, None
            else:
                st.pass_[i].t_rand_dev = None if jit[i] is None else jit[i].data_ptr()
        st.flags = flags
        rf_grid = grid.to_rf_grid(use_occupancy=cfg.use_occupancy_mask, wait_parameters=self.exchange != "owner")
        opt = self.optimizer
        if self.exchange == "owner":
            self._owner_step(ex, st, rf_grid, n, S, dev)
            del keep, jit
            self._grad_clean = True
            return StepStats(sums=sums, count=3 * n, ring=(ex, serial))
        st.phases, st.loss_scale = 0, 1.0
        if self.fuse_optimizer:
            opt.step_count += 1
            ad = ex["adam"]
            ad.lr, ad.step = float(opt.lr), int(opt.step_count)
            first, second = grid.kernel_tensors()
            ad.param_first_dev, ad.param_second_dev = first.data_ptr(), None if second is None else second.data_ptr()
            st.adam = C.pointer(ad)
        else:
            st.adam = None
            gd, gf = self.flat.views_for_accumulation()
            st.grad_first_dev, st.grad_second_dev = gd.data_ptr(), None if gf is None else gf.data_ptr()
        st.timing_events = self.step_events.array if self.step_events is not None else None
        with ops._span("train_step", dev):
            rc = _lib.load().rf_train_step(C.byref(rf_grid), C.byref(st), torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "rf_train_step")
        del keep, jit
        if self.fuse_optimizer:
            self._grad_clean = True  # the bucket is not used at all
        else:
            if self.data_parallel and rfdist._collectives_on():
                rfdist.all_reduce_mean_(self.flat.flat_grad)
            opt.step()
            self._grad_clean = False
        return StepStats(sums=sums, count=3 * n, ring=(ex, serial))

    def _owner_state(self, ex, device):
        """Persistent state of the owner-computes exchange: who owns which x-slabs of bricks, the ranks' offset tables, where every
        piece's key range starts and ends in them, a pinned host copy of those bounds, the side stream that fetches it, the receive
        buffers."""
        ow = self._owner
        if ow is not None and ow["ex"] is ex:
            return ow
        grid = self.vol_mod.thre3d_repr
        W, me = rfdist.world_size(), rfdist.rank()
        nb = brick_counts(grid, self.brick_size)
        nbyz = nb[1] * nb[2]
        nkeys = nb[0] * nbyz * 8
        # OWNERSHIP, interleaved in H "halves" so that the step pipelines: the x-slabs of bricks are cut into H * W equal PIECES in slab
        # order; piece p = h W + r (half h, rank r) = the slabs [p q, (p + 1) q).  Half h of every parameter tensor is one contiguous
        # chunk made of the W ranks' pieces in rank order -- what an in-place all-gather wants -- so the parameters of half 0 travel
        # while the brick pass of half 1 still runs.  H = 1 is the plain contiguous split.
        pipelined = owner_pipelined()
        H = OWNER_HALVES if (pipelined and OWNER_HALVES > 1 and nb[0] % (OWNER_HALVES * W) == 0) else 1
        q = nb[0] // (H * W)
        assert q >= 1 and q * H * W == nb[0]
        # everything that touches the nodes of piece p is the key range [key(slab p q - 1, f_x = 1), key(slab (p + 1) q, f_x = 0))
        # (brick_key: ((2 bx + f_x) nby nbz + ...) << 2): ONE slice of a sorted list
        idx = []
        for p in range(H * W):
            lo = 0 if p == 0 else (2 * p * q - 1) * nbyz * 4
            idx += [lo, 2 * (p + 1) * q * nbyz * 4]
        assert idx[-1] == nkeys
        t = ex["tensors"]
        widths = [t[f"pass{k}"]["sorted"].shape[1] for k in range(2)]
        ev = torch.cuda.Event
        ow = {
            "ex": ex, "W": W, "me": me, "H": H, "q": q,
            "bricks": [((h * W + me) * q * nbyz, q * nbyz) for h in range(H)],
            "all_offsets": torch.empty((W, 2, nkeys + 1), dtype=torch.int64, device=device),
            "key_idx": torch.tensor(idx, dtype=torch.int64, device=device),
            "bounds_host": torch.empty((W, 2, 2 * H * W), dtype=torch.int64).pin_memory(),
            "side": torch.cuda.Stream(device), "forward_done": ev(), "ready": ev(),
            "recv": [[torch.empty((0, widths[k]), dtype=torch.float32, device=device) for _ in range(H)] for k in range(2)],
            "checked": 0,  # iterations whose replicas have been compared (the first OWNER_CHECK_STEPS of a run)
        }
        parts = OWNER_BRICK_PARTS if OWNER_BRICK_PARTS > 0 else (2 if (W >= 4 and pipelined) else 1)
        ow["pipelined"] = pipelined
        ow["parts"] = max(1, min(parts, W, 8))
        # (one scratch for all halves: their brick passes follow one another on the compute stream)
        ow["split"] = (ow["parts"], ops.brick_split_scratch(grid, q * nbyz, ow["parts"])) if ow["parts"] > 1 else None
        self._owner = ow
        return ow

    def _owner_step(self, ex, st, rf_grid, n: int, S: int, dev) -> None:
        """The data-parallel iteration, owner-computes and pipelined (see __init__ and _owner_state):

          [base of the previous step arrived]  selection, render_diffuse forward, its loss + offsets, its adjoint   (reads `base` only)
          [rest of the previous step arrived]  specular forward, its loss + offsets
          all-gather of the two offset tables  ||  specular adjoint; the host reads the N x N x H slice bounds (the ONE host round
                                                   trip of the step, behind the specular adjoint)
          per half h:  exchange of the record slices of half h (diffuse list first: its records exist since the first phase)
                       -> merged brick pass + Adam on the rank's own bricks of half h over all ranks' lists
                       -> all-gather of half h of `base`, then of `rest`, in flight while half h + 1 is summed -- and, for the last
                          half, across the iteration boundary.

        Streams: every kernel on the caller's stream; every collective is issued asynchronously from it -- RCCL orders it after the
        work enqueued so far and runs it on its own stream, ``work.wait()`` makes the compute stream wait where (and only where) it
        needs the result.  (gloo, the CPU-side test backend, completes every call on return.)"""
        ow = self._owner_state(ex, dev)
        W, me, H = ow["W"], ow["me"], ow["H"]
        ht = [time.perf_counter()] if self.host_timing is not None else None  # host-side time line of the step (development aid)
        tick = (lambda: ht.append(time.perf_counter())) if ht is not None else (lambda: None)
        # a 1-rank group (bench.py --dp-style-step, tests) has nothing to exchange: the record exchange and the parameter all-gather
        # are skipped unless RF_OWNER_FORCE_COLLECTIVES asks for the calls themselves to be exercised (tests do)
        collect = W > 1 or bool(os.environ.get("RF_OWNER_FORCE_COLLECTIVES"))
        fast = collect and rfdist.fast_path() and rfdist.FAST_COLLECTIVES and ow["pipelined"]
        lib, grid, opt = _lib.load(), self.vol_mod.thre3d_repr, self.optimizer
        main = torch.cuda.current_stream(dev)
        main_ptr = main.cuda_stream
        t = ex["tensors"]
        marks = [] if self.phase_events is not None else None

        def mark(name):  # (bench.py: timing events on the compute stream around the phases and around the waits for communication)
            if marks is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record(main)
                marks.append((name, e))

        def run(phases, what):
            st.phases = phases
            rc = lib.rf_train_step(rf_grid, st, main_ptr)
            if rc:
                _lib.check(rc, f"rf_train_step[{what}]")

        st.adam, st.loss_scale = None, 1.0 / W
        mark("start")
        waits = grid.take_parameter_waits()  # all-gathers of the previous iteration that are still arriving
        for w_ in waits:
            if w_.tag == "base":
                w_()
        mark("wait: base parameters of the previous iteration")
        run(_lib.STEP_DIFFUSE_CHAIN, "diffuse chain")
        mark("selection + diffuse forward + its loss/offsets + its adjoint")
        for w_ in waits:
            if w_.tag != "base":
                w_()
        mark("wait: rest parameters of the previous iteration")
        run(_lib.STEP_SPECULAR_FORWARD, "specular forward")
        mark("specular forward + its loss/offsets")
        if fast:
            work = rfdist.fast_all_gather_rows(ow["all_offsets"], t["offsets2"])
        else:
            work = rfdist.all_gather_rows_equal(ow["all_offsets"], t["offsets2"], async_op=True)
        ow["forward_done"].record(main)
        run(_lib.STEP_EMIT_SPECULAR, "emit specular")
        mark("specular adjoint")
        side = ow["side"]
        with torch.cuda.stream(side):
            side.wait_event(ow["forward_done"])
            if work is not None:
                work.wait()  # (makes the side stream wait for RCCL's stream)
            ow["bounds_host"].copy_(ow["all_offsets"].index_select(2, ow["key_idx"]), non_blocking=True)
            ow["ready"].record(side)
        if work is not None:
            work.wait()  # the brick passes on the compute stream read the tables too
        tick()  # [1] forward + emit phases issued
        ow["ready"].synchronize()
        tick()  # [2] slice bounds on the host
        b = ow["bounds_host"].numpy()  # [source, list, (lo, hi) per piece]
        lo_, hi_ = b[:, :, 0::2], b[:, :, 1::2]  # [source, list, piece]
        cnt = hi_ - lo_
        sent = (W - 1) * t["offsets2"].numel() * 8  # the all-gather of the offset tables
        nd = self.flat.flat_gradient_parts()[0].numel()
        has_second = self.flat.flat_gradient_parts()[1] is not None
        opt.step_count += 1
        consumed = [0, 0]
        new_waits, issue = [], []
        for h in range(H):
            piece = h * W + me
            pending, entries = [], [None, None]
            for k in (1, 0):  # the diffuse slices first: their records exist since the first phase of the step
                srt = t[f"pass{k}"]["sorted"]
                rec_bytes = srt.shape[1] * 4
                counts = cnt[:, k, piece].tolist()
                own = counts[me]
                counts[me] = 0
                total = sum(counts)
                consumed[k] += total + own
                if ow["recv"][k][h].shape[0] < total:  # grow-only receive buffer (sizes change slowly from step to step)
                    ow["recv"][k][h] = torch.empty((int(total * 1.25) + 1024, srt.shape[1]), dtype=torch.float32, device=dev)
                recv_buf = ow["recv"][k][h]
                if collect:
                    recv = list(recv_buf[:total].split(counts))
                    s_lo, s_hi = lo_[me, k, h * W : (h + 1) * W].tolist(), hi_[me, k, h * W : (h + 1) * W].tolist()
                    send = [srt[s_lo[d] : s_hi[d]] if d != me else srt[:0] for d in range(W)]
                    sent += (int(cnt[me, k, h * W : (h + 1) * W].sum()) - own) * rec_bytes
                    # (a 1-rank RCCL group goes through the same calls)
                    pending.append(rfdist.fast_exchange(recv, send) if fast else rfdist.exchange_slices(send, recv, async_op=True))
                # the lists of the brick pass: every source's slice positioned such that ptr + offsets_s[key] * record size is the first
                # record of `key` (own list: where it lies; received ones: inside the receive buffer)
                base_ptr, pos, entry = recv_buf.data_ptr(), 0, []
                s_first = lo_[:, k, piece].tolist()
                for s_ in range(W):
                    if s_ == me:
                        entry.append((srt.data_ptr(), t["offsets2"][k], k == 1))
                    else:
                        entry.append((base_ptr + (pos - s_first[s_]) * rec_bytes, ow["all_offsets"][s_, k], k == 1))
                        pos += counts[s_]
                entries[k] = entry
            issue.append((h, entries[0] + entries[1], pending))  # (the brick pass takes the full-width lists first)
        tick()  # [3] exchanges issued
        exp_avg = (opt.exp_avg[:nd], opt.exp_avg[nd:] if has_second else None)
        exp_avg_sq = (opt.exp_avg_sq[:nd], opt.exp_avg_sq[nd:] if has_second else None)
        parts = (("base", self.flat._flat_param[:nd]), ("rest", self.flat._flat_param[nd:] if has_second else None))
        for h, lists, pending in issue:
            for wk in pending:  # the brick pass of this half waits for its two exchanges
                if wk is not None:
                    wk.wait()
            mark(f"wait: record slices of half {h}")
            brick_accumulate_adam_raw(grid, self.brick_size, lists, exp_avg, exp_avg_sq, opt.lr, opt.betas[0], opt.betas[1], opt.eps, opt.step_count,
                                      brick_range=ow["bricks"][h], rf_grid=rf_grid, split=ow["split"])
            mark(f"brick pass + Adam, half {h}")
            if collect:
                for tag, part in parts:
                    if part is None:
                        continue
                    chunk = part.numel() // H
                    half = part[h * chunk : (h + 1) * chunk]
                    wk = rfdist.fast_all_gather_in_place(half) if fast else rfdist.all_gather_chunks_(half, async_op=True)
                    new_waits.append(_ParameterWait(wk, tag))
        tick()  # [4] brick passes + parameter all-gathers issued
        sent += (W - 1) * (self.flat._flat_param.numel() // W) * 4
        if collect:
            if OWNER_OVERLAP_PARAMETERS and ow["pipelined"]:
                # left in flight: the next reader of the grid waits for them (VoxelGrid.wait_for_parameters) -- the next iteration
                # piece by piece: `base` in front of its diffuse chain, `rest` only in front of its specular forward pass
                for w_ in new_waits:
                    grid.defer_parameter_wait(w_)
            else:
                for w_ in new_waits:
                    w_()
        mark("end")
        self.exchange_bytes = (self.exchange_bytes + [sent])[-64:]
        # records this rank's brick passes consumed (its own slices of its own lists + what it received), per list
        self.owner_records = (self.owner_records + [(consumed[0], consumed[1])])[-64:]
        if marks is not None:
            self.phase_events.append(marks)
        if ht is not None:
            tick()  # [5] end of the host side
            self.host_timing.append([b_ - a_ for a_, b_ in zip(ht[:-1], ht[1:])])
        if collect and ow["checked"] < OWNER_CHECK_STEPS:
            # The FIRST iterations of a run: the replicas must hold the same parameters (a wrong slice, a lost record or a torn
            # all-gather shows here, not as a silently diverging run).  Three of them, not one: the all-gathers left in flight across
            # the iteration boundary are first consumed -- `base` in front of the diffuse chain, `rest` in front of the specular
            # forward pass -- by the SECOND iteration, so an ordering bug of that overlap cannot show after the first.  One device
            # synchronisation each.
            ow["checked"] += 1
            grid.wait_for_parameters()
            rfdist.assert_replicas_identical(self.flat.flat_param, f"owner-computes exchange, iteration {ow['checked']} of the run")
            if ow["checked"] == 1 and os.environ.get("RF_OWNER_INJECT_FAILURE") == str(me):  # (test hook: exercises bench.py's fall-back to the dense exchange)
                raise RuntimeError("injected failure of the owner-computes step (RF_OWNER_INJECT_FAILURE)")
            if os.environ.get("RF_OWNER_INJECT_HANG") == f"{me}:{ow['checked']}":  # (test hook "rank:iteration": bench.py's watchdog)
                if os.environ.get("RF_OWNER_INJECT_DEATH"):
                    os._exit(41)
                time.sleep(3600.0)

    def _executor(self, n: int, S: int, device):
        """Persistent scratch + the ctypes description of one iteration (rebuilt when the batch shape changes)."""
        ex = self._exec
        if ex is not None and ex["shape"] == (n, S):
            return ex
        grid = self.vol_mod.thre3d_repr
        nb = brick_counts(grid, self.brick_size)
        nkeys = nb[0] * nb[1] * nb[2] * 8
        if nkeys > (1 << 21):
            raise ValueError("backward='binned' needs at most 2^18 bricks")
        f32 = dict(dtype=torch.float32, device=device)
        t = {
            "origins": torch.empty((n, 3), **f32), "directions": torch.empty((n, 3), **f32), "pixels": torch.empty((n, 3), **f32),
            "sums_ring": torch.zeros((LOSS_RING, 4), **f32), "t_vals": ops.t_vals_for(S, device),
        }
        t["offsets2"] = torch.empty((2, nkeys + 1), dtype=torch.int64, device=device)  # both lists' tables: ONE collective buffer
        step, sel, adam = _lib.RFTrainStep(), _lib.RFRaySelection(), _lib.RFAdamState()
        step.num_rays, step.num_samples, step.t_vals_dev, step.loss_sums_dev = n, S, t["t_vals"].data_ptr(), t["sums_ring"].data_ptr()
        for i, diffuse in enumerate((False, True)):
            p = {
                "colour": torch.empty((n, 3), **f32), "depth": torch.empty(n, **f32), "acc": torch.empty(n, **f32), "disparity": torch.empty(n, **f32),
                "cache": torch.empty((n, S, 4), **f32), "tcache": torch.empty((n, S), **f32), "stop": torch.empty(n, dtype=torch.int32, device=device),
                "cmask": torch.empty((n, (S + 63) // 64), dtype=torch.int64, device=device),
                "g_colour": torch.empty((n, 3), **f32),
                "hist": torch.zeros(nkeys, dtype=torch.int32, device=device), "cursor": torch.empty(nkeys, dtype=torch.int32, device=device),
                "offsets": t["offsets2"][i],
                "sorted": torch.empty((n * S, expanded_record_floats(grid, diffuse)), **f32),
            }
            t[f"pass{i}"] = p
            ps = step.pass_[i]
            o = ps.out
            o.colour_dev, o.depth_dev, o.acc_dev, o.disparity_dev = p["colour"].data_ptr(), p["depth"].data_ptr(), p["acc"].data_ptr(), p["disparity"].data_ptr()
            o.sample_cache_dev, o.trans_cache_dev, o.stop_cache_dev, o.chunk_mask_dev = p["cache"].data_ptr(), p["tcache"].data_ptr(), p["stop"].data_ptr(), p["cmask"].data_ptr()
            o.key_hist_dev, o.brick_size = p["hist"].data_ptr(), int(self.brick_size)
            ps.grad_colour_dev, ps.cursor_dev, ps.offsets_dev, ps.records_sorted_dev = p["g_colour"].data_ptr(), p["cursor"].data_ptr(), p["offsets"].data_ptr(), p["sorted"].data_ptr()
        opt = self.optimizer
        nd = self.flat.flat_gradient_parts()[0].numel()
        has_second = self.flat.flat_gradient_parts()[1] is not None
        adam.exp_avg_first_dev, adam.exp_avg_sq_first_dev = opt.exp_avg[:nd].data_ptr(), opt.exp_avg_sq[:nd].data_ptr()
        adam.exp_avg_second_dev = opt.exp_avg[nd:].data_ptr() if has_second else None
        adam.exp_avg_sq_second_dev = opt.exp_avg_sq[nd:].data_ptr() if has_second else None
        adam.beta1, adam.beta2, adam.eps = opt.betas[0], opt.betas[1], opt.eps
        self._exec = {"shape": (n, S), "tensors": t, "step": step, "select": sel, "adam": adam, "sums_ring": t["sums_ring"], "sums_ptr": t["sums_ring"].data_ptr(), "slot": 0,
                      "origins": t["origins"], "directions": t["directions"], "pixels": t["pixels"]}
        return self._exec

    def _merged_step_pieces(self, rays: Rays, pixels: Tensor, t_rand=None) -> StepStats:
        """The launches of rf_train_step issued one by one from Python (the same kernels in the same order): used when per-kernel
        HIP events are being recorded (ops.KERNEL_TIMER, bench.py's roofline leg) -- rf_train_step leaves no room for events
        between its launches."""
        vol_mod, grid = self.vol_mod, self.vol_mod.thre3d_repr
        cfg = vol_mod.render_config
        _check_supported(cfg)
        origins = rays.origins.detach().to(torch.float32).contiguous()
        directions = rays.directions.detach().to(torch.float32).contiguous()
        pixels = pixels.detach().to(torch.float32).contiguous()
        n, S = origins.shape[0], int(cfg.num_samples_per_ray)
        dev = origins.device
        near, far = float(np.float32(cfg.camera_bounds.near)), float(np.float32(cfg.camera_bounds.far))
        bins = self._bin_buffers(n, S, dev)
        sums = torch.zeros(4, dtype=torch.float32, device=dev)
        passes = []
        for i, diffuse in enumerate((False, True)):
            jit = self._draw_jitter(cfg, n, S, dev, t_rand, i, self._jitter_first())
            flags = render_flags(cfg.white_bkgd, diffuse, cfg.optimized_sampling, cfg.use_occupancy_mask)
            b = bins["diffuse"] if diffuse else bins
            colour, _, _, _, caches = render_forward_raw(grid, origins, directions, jit, S, near, far, flags, save=True, key_hist=b["hist"], brick_size=self.brick_size)
            g_colour = l1_loss_grad_hip(colour, pixels, sums[2 * i : 2 * i + 2])
            passes.append((diffuse, jit, flags, b, caches, g_colour))
        lists = []
        for diffuse, jit, flags, b, caches, g_colour in passes:
            offsets = bin_offsets(b["hist"], b["offsets"], b["cursor"])
            render_backward_emit_direct_raw(grid, origins, directions, jit, S, near, far, flags, caches, g_colour, None, None, self.brick_size,
                                            b["cursor"], b["sorted"], hist_clear=b["hist"])
            lists.append((b["sorted"], offsets, diffuse))
        opt = self.optimizer
        if self.fuse_optimizer:
            opt.step_count += 1
            nd = self.flat.flat_gradient_parts()[0].numel()
            has_second = self.flat.flat_gradient_parts()[1] is not None
            halves = lambda t: (t[:nd], t[nd:] if has_second else None)
            brick_accumulate_adam_raw(grid, self.brick_size, lists, halves(opt.exp_avg), halves(opt.exp_avg_sq), opt.lr, opt.betas[0], opt.betas[1],
                                      opt.eps, opt.step_count)
            self._grad_clean = True  # the bucket is not used at all
        else:
            gd, gf = self.flat.views_for_accumulation()
            brick_accumulate_raw(grid, self.brick_size, lists, gd, gf, accumulate=False)  # overwrites the whole bucket
            if self.data_parallel and rfdist._collectives_on():
                rfdist.all_reduce_mean_(self.flat.flat_grad)
            opt.step()
            self._grad_clean = False
        means = sums / float(3 * n)
        return StepStats(means[0], means[2], means[1], means[3])

    def _fused_step_on(self, rays: Rays, pixels: Tensor, t_rand_given=None) -> StepStats:
        vol_mod, grid = self.vol_mod, self.vol_mod.thre3d_repr
        cfg = vol_mod.render_config
        _check_supported(cfg)
        origins = rays.origins.detach().to(torch.float32).contiguous()
        directions = rays.directions.detach().to(torch.float32).contiguous()
        pixels = pixels.detach().to(torch.float32).contiguous()
        n, S = origins.shape[0], int(cfg.num_samples_per_ray)
        near, far = float(np.float32(cfg.camera_bounds.near)), float(np.float32(cfg.camera_bounds.far))
        binned = self.backward == "binned"
        if binned:
            bins = self._bin_buffers(n, S, origins.device)
        if not self._grad_clean and not binned:  # the binned pass overwrites the whole bucket
            self.flat.zero_grad()
        gd, gf = self.flat.views_for_accumulation()
        sums = torch.zeros(4, dtype=torch.float32, device=origins.device)
        # data parallel: with split storage the diffuse pass only touches `base`, so the all-reduce of the `rest`
        # gradients (201 of the 235 MB at degree 2) starts right after the specular backward and overlaps it
        dp = self.data_parallel and rfdist._collectives_on()
        overlap = dp and self.diffuse and grid.storage != "reference" and gf is not None
        # ... and with equal chunks the exchange is split around a sharded Adam (reduce-scatter | update 1/N | all-gather)
        sharded = overlap and self.shard_optimizer and rfdist.can_shard(gd.numel()) and rfdist.can_shard(gf.numel())
        reduce_async = rfdist.reduce_scatter_mean_async if sharded else rfdist.all_reduce_mean_async
        pending = []
        for i, diffuse in enumerate((False, True) if self.diffuse else (False,)):
            t_rand = self._draw_jitter(cfg, n, S, origins.device, t_rand_given, i, self._jitter_first())
            flags = render_flags(cfg.white_bkgd, diffuse, cfg.optimized_sampling, cfg.use_occupancy_mask)
            use_bricks = binned
            fused_binning = use_bricks and not self.deterministic  # the forward pass counts the records per key
            colour, _, _, _, caches = render_forward_raw(
                grid, origins, directions, t_rand, S, near, far, flags, save=True,
                key_hist=bins["hist"] if fused_binning else None, brick_size=self.brick_size,
            )
            g_colour = l1_loss_grad_hip(colour, pixels, sums[2 * i : 2 * i + 2])
            if use_bricks:
                # per-sample gradient records -> binned by (8^3-node brick, boundary flags) -> every brick summed on chip
                # without atomics and written with plain stores.  The specular pass overwrites the whole bucket (no
                # zero-fill); the diffuse pass carries the 4 base channels only, adds on top and reuses the same scratch
                # buffers (stream order).  Measured on the fresh field (bench.py --dp-style-step): emit 0.060 + offsets 0.011 +
                # bricks 0.146 = 0.22 ms against 0.30 ms for the atomic scatter of the same gradient (a 4-channel brick pass is
                # all per-brick fixed cost; on a trained, sparse field the two are about even).
                if fused_binning:
                    # counting sort whose counting ran inside the forward pass: offsets, then the backward pass writes
                    # the expanded records straight to their final positions (and clears the counters)
                    offsets = bin_offsets(bins["hist"], bins["offsets"], bins["cursor"])
                    render_backward_emit_direct_raw(
                        grid, origins, directions, t_rand, S, near, far, flags, caches, g_colour, None, None, self.brick_size,
                        bins["cursor"], bins["sorted"], hist_clear=bins["hist"],
                    )
                else:  # stable 16-bit radix sort of per-slot keys: fixed summation order
                    render_backward_emit_raw(
                        grid, origins, directions, t_rand, S, near, far, flags, caches, g_colour, None, None, self.brick_size,
                        bins["keys"], bins["records"], None,
                    )
                    offsets = sort_records_by_brick(grid, bins["keys"], bins["records"], diffuse, bins["sorted"], bins["offsets"], bins["boundaries"])
                brick_accumulate_raw(grid, self.brick_size, [(bins["sorted"], offsets, diffuse)], gd, gf, accumulate=diffuse)
            else:
                render_backward_raw(grid, origins, directions, t_rand, S, near, far, flags, caches, g_colour, None, None, gd, gf)
            if overlap and i == 0:
                pending.append(reduce_async(self.flat.flat_gradient_parts()[1]))
        self._grad_clean = False
        if overlap:
            pending.append(reduce_async(self.flat.flat_gradient_parts()[0]))
            for handle in pending:
                handle.wait()
        elif dp:
            rfdist.all_reduce_mean_(self.flat.flat_grad)
        # (clearing the bucket inside the Adam kernel measured slower than a separate memset: 0.37 vs 0.28 + 0.05 ms)
        if sharded:
            # ZeRO stage 1: this rank holds the averaged gradients of one chunk of `rest` and one of `base`; it updates
            # only those parameters, then the chunks are exchanged
            nd = self.flat.flat_gradient_parts()[0].numel()
            rest_h, base_h = pending
            self.optimizer.step(ranges=[(base_h.lo, base_h.hi, base_h.shard), (nd + rest_h.lo, nd + rest_h.hi, rest_h.shard)])
            rfdist.all_gather_chunks_(self.flat.flat_param[nd:])
            rfdist.all_gather_chunks_(self.flat.flat_param[:nd])
        else:
            self.optimizer.step()
        self._grad_clean = False
        means = sums / float(3 * n)
        return StepStats(means[0], means[2] if self.diffuse else None, means[1], means[3] if self.diffuse else None)

    def _bin_buffers(self, n: int, S: int, device):
        b = self._bins
        if b is None or b["shape"] != (n, S):
            grid = self.vol_mod.thre3d_repr
            nb = brick_counts(grid, self.brick_size)
            num_bricks = nb[0] * nb[1] * nb[2]
            if num_bricks > 4096 and self.deterministic:
                raise ValueError("deterministic binned backward needs at most 4096 bricks ((brick, flags) keys are 16-bit sort keys)")
            if num_bricks * 8 > (1 << 21):
                raise ValueError("backward='binned' needs at most 2^18 bricks")
            b = {
                "shape": (n, S),
                "num_bricks": num_bricks,
                # per-slot keys / records: only the deterministic (radix sort) path needs them
                "keys": torch.empty(n * S if self.deterministic else 0, dtype=torch.int16, device=device),
                "records": torch.empty((n * S if self.deterministic else 0, expanded_record_floats(grid)), dtype=torch.float32, device=device),
                "sorted": torch.empty((n * S, expanded_record_floats(grid)), dtype=torch.float32, device=device),
                "boundaries": torch.arange(num_bricks * 8, dtype=torch.int16, device=device) if self.deterministic else None,
                "offsets": torch.full((num_bricks * 8 + 1,), n * S, dtype=torch.int64, device=device),
                "hist": torch.zeros(num_bricks * 8, dtype=torch.int32, device=device),
                "cursor": torch.empty(num_bricks * 8, dtype=torch.int32, device=device),
            }
            if self.merged_bricks:  # the diffuse render's own list (8-float records: index quad + the 4 base channels)
                b["diffuse"] = {
                    "sorted": torch.empty((n * S, expanded_record_floats(grid, True)), dtype=torch.float32, device=device),
                    "offsets": torch.full((num_bricks * 8 + 1,), n * S, dtype=torch.int64, device=device),
                    "hist": torch.zeros(num_bricks * 8, dtype=torch.int32, device=device),
                    "cursor": torch.empty(num_bricks * 8, dtype=torch.int32, device=device),
                }
            self._bins = b
        return b

    def step(self, dataset: PosedImagesInMemory, image_ids: Tensor) -> StepStats:
        if self.fused and self.merged_bricks and self.ray_selection == "keyed" and (ops.KERNEL_TIMER is None or self.exchange == "owner"):
            # the library call draws the batch as well (rf_select_rays_and_pixels is its first launch)
            grid, cfg = self.vol_mod.thre3d_repr, self.vol_mod.render_config
            if cfg.use_occupancy_mask and not grid.occupancy_current():
                grid.build_occupancy()
            intr = dataset.camera_intrinsics
            total = min(self.ray_batch_size, image_ids.numel() * intr.height * intr.width)
            lo, hi = self._global_slice(total)
            key = int(torch.randint(-(2**63), 2**63 - 1, (1,), dtype=torch.int64).item())
            stats = self._merged_step_on(None, None, selection=(dataset, image_ids, key, lo, hi - lo))
            grid.invalidate_occupancy()
            return stats
        rays, pixels = self.select(dataset, image_ids)
        return self.step_on(rays, pixels)


def train_sh_vox_grid_vol_mod_with_posed_images(
    vol_mod: VolumetricModel,
    train_dataset: PosedImagesInMemory,
    output_dir: Optional[Path] = None,
    random_initializer: Callable[[Tensor], Tensor] = lambda t: torch.nn.init.uniform_(t, -1.0, 1.0),
    test_dataset: Optional[PosedImagesInMemory] = None,
    image_batch_cache_size: int = 8,
    ray_batch_size: int = 32768,
    num_stages: int = 4,
    num_iterations_per_stage: int = 2000,
    scale_factor: float = 2.0,
    learning_rate: float = 0.03,
    lr_decay_gamma_per_stage: float = 0.1,
    lr_decay_steps_per_stage: int = 1000,
    stagewise_lr_decay_gamma: float = 0.9,
    save_freq: int = 1000,
    test_freq: int = 1000,
    summary_freq: int = 10,
    apply_diffuse_render_regularization: bool = True,
    log: Callable[[str], None] = print,
    history: Optional[List[dict]] = None,
    storage: Optional[str] = "split",
    ray_selection: str = "keyed",
    global_batch: bool = False,
) -> VolumetricModel:
    """Same arguments (minus the feedback/visualisation ones) and same schedule as the reference's
    trainer.  Returns the trained model; ``history`` (if given) collects the logged scalars.
    ``storage`` selects the HBM layout the grid is trained in ("split" = MI355X-native, "reference", or
    None = keep the model's); checkpoints and ``.densities`` / ``.features`` stay in the reference layout and the
    returned model's grid is converted back to the storage it came in.
    ``global_batch`` (data parallel): ``ray_batch_size`` is split over the ranks instead of drawn per rank (seed all
    ranks equally)."""
    grid = vol_mod.thre3d_repr
    assert isinstance(grid, VoxelGrid), f"cannot use a {type(grid)} with this TrainProcedure"
    assert vol_mod.render_procedure is render_sh_voxel_grid, "non SH-based VoxelGrids cannot be used with this TrainProcedure"
    is_main = rfdist.rank() == 0
    incoming_storage = grid.storage

    stage_sizes = compute_thre3d_grid_sizes(grid.grid_dims, num_stages, scale_factor)
    stage_sets = [train_dataset]
    for stage in range(1, num_stages):
        stage_sets.insert(0, train_dataset.downsampled(scale_factor**stage))

    with torch.no_grad():
        small = scale_voxel_grid_with_required_output_size(grid, stage_sizes[0])
        if storage is not None:
            small = small.to_storage(storage)
        for tensor in small.kernel_tensors():  # densities + features, in whatever layout they are stored
            if tensor is not None:
                random_initializer(tensor)
                rfdist.broadcast_(tensor.data)  # every rank must start from the same parameters
        vol_mod.thre3d_repr = small.to(vol_mod.device)

    model_dir = None
    if output_dir is not None and is_main:
        model_dir = Path(output_dir) / "saved_models"
        model_dir.mkdir(parents=True, exist_ok=True)

    def save(name: str) -> None:
        if model_dir is None:
            return
        extra = {
            CAMERA_BOUNDS: train_dataset.camera_bounds,
            CAMERA_INTRINSICS: train_dataset.camera_intrinsics,
            HEMISPHERICAL_RADIUS: train_dataset.get_hemispherical_radius_estimate(),
        }
        torch.save(vol_mod.get_save_info(extra_info=extra), model_dir / name)

    trained_seconds = 0.0
    for stage in range(1, num_stages + 1):
        data = stage_sets[stage - 1]
        batches = data.image_batches(image_batch_cache_size)
        lr = learning_rate * (stagewise_lr_decay_gamma ** (stage - 1))
        stepper = TrainStepper(vol_mod, ray_batch_size, lr, apply_diffuse_render_regularization, ray_selection=ray_selection, global_batch=global_batch)
        scheduler = ExponentialLR(stepper.optimizer, lr_decay_gamma_per_stage)
        if is_main:
            log(
                f"training stage: {stage}   voxel grid resolution: {vol_mod.thre3d_repr.grid_dims} "
                f"training images resolution: [{data.camera_intrinsics.height} x {data.camera_intrinsics.width}] lr: {lr}"
            )
        last = time.perf_counter()
        for it in range(1, num_iterations_per_stage + 1):
            stats = stepper.step(data, next(batches))
            global_step = (stage - 1) * num_iterations_per_stage + it
            if global_step % summary_freq == 0 or it == 1 or it == num_iterations_per_stage:
                torch.cuda.synchronize()
                trained_seconds += time.perf_counter() - last
                row = {"stage": stage, "global_step": global_step, "specular_loss": float(stats.specular_loss)}
                if stats.diffuse_loss is not None:
                    row["diffuse_loss"] = float(stats.diffuse_loss)
                row.update(stats.psnr())
                if history is not None:
                    history.append(row)
                if is_main:
                    log(" ".join(f"{k}: {v:.3f}" if isinstance(v, float) else f"{k}: {v}" for k, v in row.items()))
                last = time.perf_counter()
            if it % lr_decay_steps_per_stage == 0:
                scheduler.step()
            if test_dataset is not None and (global_step % test_freq == 0 or it == num_iterations_per_stage):
                psnr = test_sh_vox_grid_vol_mod_with_posed_images(vol_mod, test_dataset)
                if history is not None:
                    history.append({"global_step": global_step, "test_psnr": psnr})
                if is_main:
                    log(f"TEST SET PSNR: {psnr:.3f}")
                last = time.perf_counter()
            if global_step % save_freq == 0 or it == 1 or it == num_iterations_per_stage:
                save(f"model_stage_{stage}_iter_{global_step}.pth")
                last = time.perf_counter()
        stepper.flat.detach()
        if stage != num_stages:
            with torch.no_grad():
                vol_mod.thre3d_repr = scale_voxel_grid_with_required_output_size(
                    vol_mod.thre3d_repr, stage_sizes[stage]
                ).to(vol_mod.device)
    save("model_final.pth")
    if vol_mod.thre3d_repr.storage != incoming_storage:
        # hand the grid back in the storage it came in (normally "reference"): there ``.densities`` / ``.features`` are
        # the Parameters themselves again, so reference-style code (in-place edits, ``.grad``, ``parameters()``) behaves
        with torch.no_grad():
            vol_mod.thre3d_repr = vol_mod.thre3d_repr.to_storage(incoming_storage).to(vol_mod.device)
    if is_main:
        log(f"Total actual training time: {trained_seconds:.1f} s")
    return vol_mod


def test_sh_vox_grid_vol_mod_with_posed_images(
    vol_mod: VolumetricModel, test_dataset: PosedImagesInMemory, parallel_rays_chunk_size: Optional[int] = 32768
) -> float:
    """Mean PSNR over held-out images (reference modules/testers.py:17-71, PSNR part; LPIPS is out of
    scope).  Rendered with render_num_samples_per_ray samples, optimized_sampling off."""
    from .camera import CameraPose

    cfg = vol_mod.render_config
    psnrs = []
    for i in range(len(test_dataset)):
        image, pose = test_dataset[i]
        out = vol_mod.render(
            CameraPose(pose[:, :3], pose[:, 3:]),
            test_dataset.camera_intrinsics,
            parallel_rays_chunk_size=parallel_rays_chunk_size,
            optimized_sampling=False,
            num_samples_per_ray=cfg.render_num_samples_per_ray,
        )
        psnrs.append(float(mse2psnr(mse_loss(out.colour, image.permute(1, 2, 0)))))
    return float(np.mean(psnrs))


test_sh_vox_grid_vol_mod_with_posed_images.__test__ = False  # not a pytest test
