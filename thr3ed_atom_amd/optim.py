"""Flat parameter / gradient bucket and fused Adam for the voxel grid.

The reference optimises two Parameters with ``torch.optim.Adam`` (modules/trainers.py:242-250) and lets
autograd allocate, zero-fill and sum full-size gradient tensors for each of the two renders of an
iteration (:306-341).  On MI355X the grid is 235 MB at 128^3 / SH degree 2, so those fills and adds cost
more HBM traffic than the render itself.  ``FlatGrid`` re-homes both tensors into ONE contiguous
float32 buffer (densities | features) with one matching gradient buffer:

  * the backward kernel accumulates straight into the gradient buffer (no per-render allocation, no
    AccumulateGrad add);
  * ``zero_grad`` is one memset, the data-parallel exchange is ONE collective pass over one bucket on
    RCCL/xGMI (distributed.py), and Adam is one fused kernel over the flat buffer (rf_adam_step).

The Parameters stay ordinary ``nn.Parameter`` objects (views of the flat storage), so ``state_dict`` and
checkpoints are unchanged.
"""
from typing import Optional, Tuple

import os

import torch
from torch import Tensor

from .ops import adam_step_hip
from .voxels import VoxelGrid


# deferred optimizer steps on a reference-storage grid: let the brick flush write the Parameters' own layout beside the split shadow
# ($RF_MIRROR_FLUSH=0: update the shadow only and re-lay it out into the Parameters with a launch of its own -- A/B runs)
MIRROR_FLUSH = os.environ.get("RF_MIRROR_FLUSH", "1") != "0"


class FlatGrid:
    def __init__(self, grid: VoxelGrid, deferred: bool = False):
        """``deferred`` (opt-in; reference storage, SH degree 0 or 2, single process): the autograd op does not sum its gradient
        into the bucket at all.  Every backward pass leaves its gradient as a sorted RECORD LIST (``pending``); ``FusedAdam.step``
        sums all lists of the iteration in ONE merged brick pass that applies Adam in its flush on the grid's split-layout shadow
        (voxels.KernelGridInterface.forward_rf_grid) and re-lays the result out into the Parameters -- the fused single-GPU
        step, driven by torch.autograd.  ``.grad`` / ``flat_grad`` are then NOT populated (``materialize()`` fills them from the
        pending lists for inspection)."""
        d, f = grid.kernel_tensors()  # (densities, features) or (base, rest); f is None for split storage at degree 0
        if not (isinstance(d, torch.nn.Parameter) and (f is None or isinstance(f, torch.nn.Parameter))):
            raise ValueError("FlatGrid needs a tunable VoxelGrid")
        self.grid = grid
        self._d, self._f = d, f
        nd, nf = d.numel(), (0 if f is None else f.numel())
        self._flat_param = torch.empty(nd + nf, dtype=torch.float32, device=d.device)
        self.flat_param[:nd].copy_(d.detach().reshape(-1))
        d.data = self.flat_param[:nd].view(d.shape)
        self.flat_grad = torch.zeros_like(self.flat_param)
        self._gd = self.flat_grad[:nd].view(d.shape)
        self._gf = None
        d.grad = self._gd
        if f is not None:
            self.flat_param[nd:].copy_(f.detach().reshape(-1))
            f.data = self.flat_param[nd:].view(f.shape)
            self._gf = self.flat_grad[nd:].view(f.shape)
            f.grad = self._gf
        grid._grad_bucket = self
        from . import distributed as rfdist
        from . import voxels

        # (the flush of the merged brick pass keeps element offsets in 31 bits and the binned adjoint addresses at most 2^18 bricks:
        # larger grids -- 512^3 at SH degree 2 -- keep the gradient bucket and the atomic adjoint they trained through before)
        padded_nodes, bricks = 1, 1
        for dim in grid.grid_dims:
            padded_nodes *= (dim + 7) // 8 * 8
            bricks *= (dim + 7) // 8
        # (and, like TrainStepper(backward="auto"), fewer than 256 bricks keep the atomic adjoint: a handful of brick workgroups would sum
        # the whole batch's records -- $RF_AUTO_BINNED_MIN_BRICKS, 0 in the tests)
        fits = (padded_nodes * max(4, grid.num_features - 3) < (1 << 31) and int(os.environ.get("RF_AUTO_BINNED_MIN_BRICKS", "256")) <= bricks <= (1 << 18)
                and voxels.shadow_allowed(grid))
        self.deferred = (bool(deferred) and grid.storage == "reference" and (grid.num_features + 1) % 4 == 0 and voxels.SPLIT_SHADOW
                         and not rfdist._collectives_on() and fits)
        self.pending = []  # [(records, offsets, render_diffuse)] of the backward passes since the last zero_grad / step
        # bricks of the record lists: 4 x 8 x 8 nodes (ops.BRICK_4X8X8: four 256-thread workgroups per CU in the optimizer's brick pass)
        # where its one-round flush applies -- SH degree 0 or 2, every shadow tensor below 2^30 elements --, else 8^3
        from .ops import AUTOGRAD_BRICK_SIZE, BRICK_4X8X8

        small = padded_nodes <= (1 << 24) and padded_nodes * max(4, grid.num_features - 3) < (1 << 30) and bricks * 2 * 8 <= (1 << 21)
        self.brick_size = BRICK_4X8X8 if (self.deferred and grid.num_features in (3, 27) and small and "RF_BRICK_SIZE" not in os.environ) else AUTOGRAD_BRICK_SIZE

    @property
    def flat_param(self) -> Tensor:
        """The flat parameter buffer (first tensor | second tensor).  Under the owner-computes data-parallel step the parameter
        all-gathers of an iteration stay in flight after ``step()`` returns: like every accessor of the VoxelGrid, this one makes the
        current stream wait for them first, so that a reader (a clone for a test, a checksum, a checkpoint) never sees torn
        parameters.  (The step itself works on ``_flat_param``.)"""
        self.grid.wait_for_parameters()
        return self._flat_param

    # ---- protocol used by ops._ReluFieldRender.backward -------------------------------------
    def matches(self, first: Tensor, second: Optional[Tensor]) -> bool:
        same = first.data_ptr() == self._d.data_ptr()
        if self._f is not None:
            same = same and second is not None and second.data_ptr() == self._f.data_ptr()
        return same

    def views_for_accumulation(self) -> Tuple[Tensor, Tensor]:
        return self._gd, self._gf

    def flat_gradient_parts(self) -> Tuple[Tensor, Optional[Tensor]]:
        """The gradient bucket as its two contiguous 1-D parts (first tensor | second tensor)."""
        nd = self._d.numel()
        return self.flat_grad[:nd], (self.flat_grad[nd:] if self._f is not None else None)

    @staticmethod
    def autograd_return() -> Tuple[None, None]:
        # the kernel has already added into .grad; returning None keeps autograd from adding again
        return None, None

    # ---------------------------------------------------------------------------------------
    def materialize(self) -> Tensor:
        """deferred mode: sum the pending record lists into ``flat_grad`` (reference layout; overwrites it) -- for inspection, the
        optimizer does not need it."""
        from .ops import brick_accumulate_raw

        if self.deferred and self.pending:
            lists = sorted(self.pending, key=lambda l: bool(l[2]))  # full-width lists first
            if all(l[2] for l in lists) and self.grid.num_features != 3:
                self.flat_grad.zero_()  # (render_diffuse lists only cover the base channels)
            brick_accumulate_raw(self.grid, self.brick_size, lists, self._gd, self._gf, accumulate=False)
        return self.flat_grad

    def zero_grad(self) -> None:
        self.pending = []
        if self.deferred:  # (no gradient tensor is in use)
            return
        self.flat_grad.zero_()
        if self._d.grad is not self._gd:
            self._d.grad = self._gd
        if self._f is not None and self._f.grad is not self._gf:
            self._f.grad = self._gf

    def detach(self) -> None:
        if getattr(self.grid, "_grad_bucket", None) is self:
            self.grid._grad_bucket = None


class FusedAdam:
    """torch.optim.Adam(betas, eps, no weight decay) over a FlatGrid, one HIP kernel per step."""

    def __init__(self, flat: FlatGrid, lr: float, betas=(0.9, 0.999), eps: float = 1e-8):
        self.flat = flat
        self.lr = float(lr)
        self.betas = (float(betas[0]), float(betas[1]))
        self.eps = float(eps)
        self._split_moments = None  # deferred gradients: the moments live in the split layout of the shadow the flush updates
        self.exp_avg = None if flat.deferred else torch.zeros_like(flat.flat_param)
        self.exp_avg_sq = None if flat.deferred else torch.zeros_like(flat.flat_param)
        self.step_count = 0

    @property
    def param_groups(self):
        return [{"lr": self.lr}]

    def zero_grad(self) -> None:
        self.flat.zero_grad()

    @torch.no_grad()
    def step(self, zero_grad: bool = False, ranges=None) -> None:
        """``zero_grad=True`` clears the gradient bucket in the same pass.  ``ranges`` = [(lo, hi, grad), ...] restricts
        the update to those slices of the flat buffer, with the (averaged) gradients of each slice given separately
        (sharded data-parallel optimizer: the moments of the other slices stay untouched on this rank, their
        parameters arrive by all-gather)."""
        if self.flat.deferred:
            self._deferred_step()
            return
        self.step_count += 1
        if ranges is None:
            ranges = [(0, self.flat.flat_param.numel(), self.flat.flat_grad)]
        for lo, hi, grad in ranges:
            if hi > lo:
                assert grad.numel() == hi - lo
                adam_step_hip(
                    self.flat.flat_param[lo:hi], grad, self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi],
                    self.lr, self.betas[0], self.betas[1], self.eps, self.step_count, zero_grad,
                )
        self.flat.grid.invalidate_occupancy()  # the kernel wrote the densities through raw pointers


    def _deferred_step(self) -> None:
        """All record lists of the iteration -> ONE merged brick pass with Adam in its flush on the split shadow -> Parameters."""
        from .ops import brick_accumulate_adam_raw, mirror_flush_applies

        flat, grid = self.flat, self.flat.grid
        if not flat.pending:
            return  # nothing was rendered since the last step
        lists = sorted(flat.pending, key=lambda l: bool(l[2]))  # full-width lists first
        if sum(1 for l in lists if not l[2]) > 8 or sum(1 for l in lists if l[2]) > 8:
            raise ValueError("deferred gradients: at most 8 specular and 8 render_diffuse backward passes per optimizer step")
        self.step_count += 1
        sh, rf_grid = grid._shadow(refresh=True)
        if self._split_moments is None:
            z = lambda t: None if t is None else torch.zeros_like(t)
            self._split_moments = ((z(sh["base"]), z(sh["rest"])), (z(sh["base"]), z(sh["rest"])))
        m, v = self._split_moments
        # the flush updates the shadow AND writes the Parameters' own (reference) layout where it can (rf_brick_accumulate_adam_mirror);
        # else one re-layout launch copies the shadow into the Parameters afterwards (81 us at 128^3 / SH degree 2)
        d, f = grid.kernel_tensors()
        mirror = (d.data, f.data) if (MIRROR_FLUSH and mirror_flush_applies(grid, flat.brick_size, d.data, f.data)) else None
        brick_accumulate_adam_raw(grid, flat.brick_size, lists, m, v, self.lr, self.betas[0], self.betas[1], self.eps, self.step_count,
                                  rf_grid=rf_grid, params=(sh["base"], sh["rest"]), mirror=mirror)
        grid.adopt_shadow(relayout=mirror is None)
        flat.pending = []


class ExponentialLR:
    """lr <- lr * gamma per ``step()`` (torch.optim.lr_scheduler.ExponentialLR for FusedAdam)."""

    def __init__(self, optimizer: FusedAdam, gamma: float):
        self.optimizer, self.gamma = optimizer, float(gamma)

    def step(self) -> None:
        self.optimizer.lr *= self.gamma
