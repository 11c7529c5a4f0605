"""torch-facing operators over the C ABI (include/relu_field.h).

``relu_field_render`` is the ``torch.autograd.Function`` that replaces the reference's whole
sample -> VoxelGrid.forward -> SH -> mask -> accumulate composition
(thre3d_atom/thre3d_reprs/renderers.py:48-102) and its autograd graph: forward enqueues
rf_render_forward, backward enqueues rf_render_backward on the current HIP stream.  Tensors cross
the boundary as raw device pointers; PyTorch is only the allocator and the stream owner.

Nothing here computes on the CPU and nothing falls back: CPU tensors or a missing library raise.
"""
import ctypes as C
import os
from typing import Dict, NamedTuple, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _lib
from .voxels import VoxelGrid, as_kernel_grid

_T_VALS_CACHE: Dict[Tuple[int, str], Tensor] = {}


class KernelTimer:
    """Optional HIP-event timing of the render launches (bench.py's roofline leg).  Events are recorded
    on the stream the kernels are enqueued on; reading them needs a device synchronise, which the caller
    does once after the timed region."""

    def __init__(self, preallocate: int = 0):
        self.records: Dict[str, list] = {}
        # creating HIP events inside a timed loop perturbs it (the runtime grows its signal pool in bursts); a
        # caller that knows how many spans it will record can have them created up front
        self._pool = [torch.cuda.Event(enable_timing=True) for _ in range(2 * preallocate)]

    def _event(self):
        return self._pool.pop() if self._pool else torch.cuda.Event(enable_timing=True)

    def span(self, name: str, device):
        return _TimedSpan(self, name, device)

    def summary(self) -> Dict[str, dict]:
        out = {}
        for name, spans in self.records.items():
            ms = [a.elapsed_time(b) for a, b in spans]
            out[name] = {"launches": len(ms), "avg_ms": sum(ms) / max(len(ms), 1), "total_ms": sum(ms)}
        return out

    def reset(self) -> None:
        self.records.clear()


class _TimedSpan:
    def __init__(self, timer, name, device):
        self.timer, self.name, self.device = timer, name, device

    def __enter__(self):
        self.a = self.timer._event()
        self.b = self.timer._event()
        self.a.record()

    def __exit__(self, *exc):
        self.b.record()
        self.timer.records.setdefault(self.name, []).append((self.a, self.b))
        return False


class _NoSpan:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


KERNEL_TIMER: Optional[KernelTimer] = None


class StepEvents:
    """The RF_TRAIN_STEP_EVENTS HIP events rf_train_step records around its launches (RFTrainStep.timing_events): raw
    hipEvent_t handles created through the HIP runtime (torch.cuda.Event creates its handle lazily and cannot be handed to C).
    ``elapsed_ms()`` needs the stream to have passed the last event (synchronise first)."""

    _hip = None

    def __init__(self):
        if StepEvents._hip is None:
            StepEvents._hip = C.CDLL("libamdhip64.so")
        hip = StepEvents._hip
        self.array = (C.c_void_p * _lib.TRAIN_STEP_EVENTS)()
        for i in range(_lib.TRAIN_STEP_EVENTS):
            ev = C.c_void_p()
            if hip.hipEventCreate(C.byref(ev)) != 0:
                raise RuntimeError("hipEventCreate failed")
            self.array[i] = ev.value

    def elapsed_ms(self) -> Dict[str, float]:
        hip = StepEvents._hip
        out = {}
        for i, name in enumerate(_lib.TRAIN_STEP_EVENT_NAMES):
            ms = C.c_float()
            if hip.hipEventElapsedTime(C.byref(ms), C.c_void_p(self.array[i]), C.c_void_p(self.array[i + 1])) != 0:
                raise RuntimeError("hipEventElapsedTime failed (events not recorded or not complete)")
            out[name] = float(ms.value)
        # paired launches (both renders / both adjoints of the iteration in one launch): one entry per launch
        fwd_pair, emit_pair = _lib.train_step_pairing()
        for paired, first, gap, second, merged in (
                (fwd_pair, "render_forward[spec,save]", "(no launch: loss slot of the first render)", "render_forward[diffuse,save]", "render_forward[spec+diffuse,save]"),
                (emit_pair, "render_backward_emit_direct[spec]", "(no launch: offsets slot of the second list)", "render_backward_emit_direct[diffuse]",
                 "render_backward_emit_direct[spec+diffuse]")):
            if paired:
                # (the launch sits in the pair's first slot; the two empty slots behind it hold nothing but the cost of recording an event)
                out = {(merged if k == first else k): v for k, v in out.items() if k not in (gap, second)}
        return out

    def __del__(self):
        hip = StepEvents._hip
        if hip is not None:
            for i in range(len(self.array)):
                if self.array[i]:
                    hip.hipEventDestroy(C.c_void_p(self.array[i]))


def _span(name: str, device):
    return KERNEL_TIMER.span(name, device) if KERNEL_TIMER is not None else _NoSpan()


def _variant(grid: "VoxelGrid", flags: int) -> str:
    return "diffuse" if (flags & _lib.FLAG_RENDER_DIFFUSE) else f"sh{grid.sh_degree}"


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _require_hip(t: Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on a HIP device (got {t.device}); the render path has no CPU fallback")


def t_vals_for(num_samples: int, device) -> Tensor:
    """linspace(0, 1, S) as the reference's CPU path computes it (sample.py:46), evaluated once on the
    host so that every device sees bit-identical sample parameters, then cached on the device."""
    key = (int(num_samples), str(device))
    t = _T_VALS_CACHE.get(key)
    if t is None:
        t = torch.linspace(0.0, 1.0, int(num_samples), dtype=torch.float32).to(device)
        _T_VALS_CACHE[key] = t
    return t


def _ptr(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


class KeyedJitter(NamedTuple):
    """Stands in for a ``t_rand`` tensor: the jitter of (ray r, sample s) is a counter-based hash of
    (key, first_ray + r, s) evaluated inside the kernels (RF_FLAG_JITTER_KEYED) -- no [N, S] tensor exists."""

    key: int
    first_ray: int = 0


def draw_jitter_key() -> int:
    """A 64-bit key from torch's CPU generator (so torch.manual_seed makes jittered renders reproducible)."""
    return int(torch.randint(-(2**63), 2**63 - 1, (1,), dtype=torch.int64).item()) & 0xFFFFFFFFFFFFFFFF


def _jitter_flags(flags: int, t_rand) -> int:
    return int(flags) | (_lib.FLAG_JITTER_KEYED if isinstance(t_rand, KeyedJitter) else 0)


def _ray_batch(origins: Tensor, directions: Tensor, num_samples: int, near: float, far: float, t_rand):
    """``t_rand``: None (no jitter), a [N, S] tensor, or a KeyedJitter."""
    rb = _lib.RFRayBatch()
    rb.origins_dev = origins.data_ptr()
    rb.directions_dev = directions.data_ptr()
    rb.num_rays = origins.shape[0]
    rb.num_samples = int(num_samples)
    rb.near = near  # c_float rounds to float32 exactly like np.float32 does
    rb.far = far
    tv = t_vals_for(num_samples, origins.device)
    rb.t_vals_dev = tv.data_ptr()
    if isinstance(t_rand, KeyedJitter):
        rb.t_rand_dev = None
        rb.jitter_key, rb.first_ray = int(t_rand.key) & 0xFFFFFFFFFFFFFFFF, int(t_rand.first_ray)
    else:
        rb.t_rand_dev = _ptr(t_rand)
    return rb, tv


def render_forward_raw(grid: VoxelGrid, origins: Tensor, directions: Tensor, t_rand: Optional[Tensor], num_samples: int,
                       near: float, far: float, flags: int, save: bool, key_hist: Optional[Tensor] = None, brick_size: int = 8):
    """Enqueue rf_render_forward.  Returns (colour [N,3], depth [N,1], acc [N,1], disparity [N,1], caches) where
    ``caches`` = (sample_cache [N,S,4], trans_cache [N,S], stop [N], chunk_mask [N, ceil(S/64)] int64) when ``save`` else None:
    only the samples that can carry gradient are cached, compacted per 64-sample chunk, the mask says which (relu_field.h).
    No autograd.  ``key_hist`` (int32 [8 * num_bricks], with ``save``): the pass also counts those samples per (brick, flags)
    key for ``render_backward_emit_direct_raw``."""
    lib = _lib.load()
    for name, t in (("ray origins", origins), ("ray directions", directions)):
        _require_hip(t, name)
    n = origins.shape[0]
    dev = origins.device
    rf_grid = grid.forward_rf_grid(use_occupancy=bool(flags & _lib.FLAG_OCCUPANCY_SKIP))  # (reference storage: its split shadow)
    rb, tv = _ray_batch(origins, directions, num_samples, near, far, t_rand)
    colour = torch.empty((n, 3), dtype=torch.float32, device=dev)
    depth = torch.empty((n, 1), dtype=torch.float32, device=dev)
    acc = torch.empty((n, 1), dtype=torch.float32, device=dev)
    disparity = torch.empty((n, 1), dtype=torch.float32, device=dev)
    out = _lib.RFRenderOut()
    out.colour_dev, out.depth_dev, out.acc_dev, out.disparity_dev = colour.data_ptr(), depth.data_ptr(), acc.data_ptr(), disparity.data_ptr()
    caches = None
    if save:
        cache = torch.empty((n, num_samples, 4), dtype=torch.float32, device=dev)
        tcache = torch.empty((n, num_samples), dtype=torch.float32, device=dev)
        stop = torch.empty((n,), dtype=torch.int32, device=dev)
        cmask = torch.empty((n, (num_samples + 63) // 64), dtype=torch.int64, device=dev)
        out.sample_cache_dev, out.trans_cache_dev, out.stop_cache_dev, out.chunk_mask_dev = cache.data_ptr(), tcache.data_ptr(), stop.data_ptr(), cmask.data_ptr()
        caches = (cache, tcache, stop, cmask)
        if key_hist is not None:
            out.key_hist_dev, out.brick_size = key_hist.data_ptr(), int(brick_size)
    with _span(f"render_forward[{_variant(grid, flags)}{',save' if save else ''}]", dev):
        rc = lib.rf_render_forward(C.byref(rf_grid), C.byref(rb), _jitter_flags(flags, t_rand), C.byref(out), _stream(dev))
    _lib.check(rc, "rf_render_forward")
    return colour, depth, acc, disparity, caches


def render_frame_raw(grid: VoxelGrid, height: int, width: int, focal: float, rotation, translation, num_samples: int, near: float, far: float,
                     flags: int, jitter: Optional[KeyedJitter], first_ray: int = 0, num_rays: Optional[int] = None):
    """Pixels [first_ray, first_ray + num_rays) (row-major; default: the whole frame) of a posed pinhole camera in ONE launch of
    rf_render_forward: the rays are generated inside the kernel (RFRayBatch.camera: cast_rays of utils/misc.py:12-50 fused) and
    the stratified jitter, if any, is the counter-based one keyed by (jitter.key, pixel index) -- no ray tensors, no [N, S]
    random tensor.  Returns (colour [n,3], depth [n,1], acc [n,1], disparity [n,1]).  Inference only (no cache)."""
    lib = _lib.load()
    first, _ = grid.kernel_tensors()
    dev = first.device
    _require_hip(first, "grid tensor")
    n = int(height) * int(width) - int(first_ray) if num_rays is None else int(num_rays)
    cam = _lib.RFCamera()
    cam.height, cam.width, cam.focal = int(height), int(width), float(np.float32(focal))
    rot = torch.as_tensor(rotation).detach().to("cpu", torch.float32).reshape(3, 3)
    trans = torch.as_tensor(translation).detach().to("cpu", torch.float32).reshape(3)
    for i in range(3):
        for j in range(3):
            cam.pose[4 * i + j] = float(rot[i, j])
        cam.pose[4 * i + 3] = float(trans[i])
    rb = _lib.RFRayBatch()
    rb.num_rays, rb.num_samples, rb.near, rb.far = n, int(num_samples), near, far
    tv = t_vals_for(num_samples, dev)
    rb.t_vals_dev = tv.data_ptr()
    rb.first_ray = int(first_ray)
    rb.camera = C.pointer(cam)
    if jitter is not None:
        rb.jitter_key = int(jitter.key) & 0xFFFFFFFFFFFFFFFF
        flags = int(flags) | _lib.FLAG_JITTER_KEYED
    rf_grid = grid.forward_rf_grid(use_occupancy=bool(flags & _lib.FLAG_OCCUPANCY_SKIP))
    colour = torch.empty((n, 3), dtype=torch.float32, device=dev)
    depth = torch.empty((n, 1), dtype=torch.float32, device=dev)
    acc = torch.empty((n, 1), dtype=torch.float32, device=dev)
    disparity = torch.empty((n, 1), dtype=torch.float32, device=dev)
    out = _lib.RFRenderOut()
    out.colour_dev, out.depth_dev, out.acc_dev, out.disparity_dev = colour.data_ptr(), depth.data_ptr(), acc.data_ptr(), disparity.data_ptr()
    with _span(f"render_forward[{_variant(grid, flags)},frame]", dev):
        rc = lib.rf_render_forward(C.byref(rf_grid), C.byref(rb), int(flags), C.byref(out), _stream(dev))
    _lib.check(rc, "rf_render_forward")
    return colour, depth, acc, disparity


def render_backward_raw(grid: VoxelGrid, origins: Tensor, directions: Tensor, t_rand: Optional[Tensor], num_samples: int,
                        near: float, far: float, flags: int, caches, g_colour: Optional[Tensor], g_depth: Optional[Tensor],
                        g_acc: Optional[Tensor], grad_first: Tensor, grad_second: Optional[Tensor]) -> None:
    """Enqueue rf_render_backward: ACCUMULATES dL/d(grid tensors) into ``grad_first`` / ``grad_second`` (storage
    order of grid.kernel_tensors())."""
    lib = _lib.load()
    dev = origins.device
    rf_grid = grid.to_rf_grid(use_occupancy=bool(flags & _lib.FLAG_OCCUPANCY_SKIP))
    rb, tv = _ray_batch(origins, directions, num_samples, near, far, t_rand)
    grads = _lib.RFRenderGrads()
    grads.grad_colour_dev, grads.grad_depth_dev, grads.grad_acc_dev = _ptr(g_colour), _ptr(g_depth), _ptr(g_acc)
    cache, tcache, stop, cmask = caches
    fwd = _lib.RFRenderOut()
    fwd.sample_cache_dev, fwd.trans_cache_dev, fwd.stop_cache_dev, fwd.chunk_mask_dev = cache.data_ptr(), tcache.data_ptr(), stop.data_ptr(), cmask.data_ptr()
    with _span(f"render_backward[{_variant(grid, flags)}]", dev):
        rc = lib.rf_render_backward(
            C.byref(rf_grid), C.byref(rb), _jitter_flags(flags, t_rand), C.byref(fwd), C.byref(grads), grad_first.data_ptr(), _ptr(grad_second), _stream(dev)
        )
    _lib.check(rc, "rf_render_backward")


NO_BRICK = 0x7FFF


BRICK_4X8X8 = 488  # RF_BRICK_4X8X8 of relu_field.h: bricks of 4 x 8 x 8 nodes (the single-GPU optimizer pass)


def brick_edges(brick_size: int) -> Tuple[int, int, int]:
    return (4, 8, 8) if int(brick_size) == BRICK_4X8X8 else (int(brick_size),) * 3


def brick_counts(grid: VoxelGrid, brick_size: int) -> Tuple[int, int, int]:
    return tuple((d + e - 1) // e for d, e in zip(grid.grid_dims, brick_edges(brick_size)))


def render_backward_emit_raw(grid: VoxelGrid, origins: Tensor, directions: Tensor, t_rand: Optional[Tensor], num_samples: int,
                             near: float, far: float, flags: int, caches, g_colour: Optional[Tensor], g_depth: Optional[Tensor],
                             g_acc: Optional[Tensor], brick_size: int, keys: Tensor, records: Tensor,
                             hist: Optional[Tensor] = None) -> None:
    """Enqueue rf_render_backward_emit: per-slot gradient records ([N*S, record_floats]) + brick keys instead of a scatter;
    ``hist`` (int32 [8 * num_bricks], optional) is incremented by the number of records per key (for ``bin_records_by_brick``)."""
    lib = _lib.load()
    dev = origins.device
    rf_grid = grid.to_rf_grid(use_occupancy=bool(flags & _lib.FLAG_OCCUPANCY_SKIP))
    rb, tv = _ray_batch(origins, directions, num_samples, near, far, t_rand)
    grads = _lib.RFRenderGrads()
    grads.grad_colour_dev, grads.grad_depth_dev, grads.grad_acc_dev = _ptr(g_colour), _ptr(g_depth), _ptr(g_acc)
    cache, tcache, stop, cmask = caches
    fwd = _lib.RFRenderOut()
    fwd.sample_cache_dev, fwd.trans_cache_dev, fwd.stop_cache_dev, fwd.chunk_mask_dev = cache.data_ptr(), tcache.data_ptr(), stop.data_ptr(), cmask.data_ptr()
    with _span(f"render_backward_emit[{_variant(grid, flags)}]", dev):
        rc = lib.rf_render_backward_emit(
            C.byref(rf_grid), C.byref(rb), _jitter_flags(flags, t_rand), C.byref(fwd), C.byref(grads), int(brick_size), keys.data_ptr(), records.data_ptr(),
            _ptr(hist), _stream(dev),
        )
    _lib.check(rc, "rf_render_backward_emit")


def render_backward_emit_direct_raw(grid: VoxelGrid, origins: Tensor, directions: Tensor, t_rand: Optional[Tensor], num_samples: int,
                                    near: float, far: float, flags: int, caches, g_colour: Optional[Tensor], g_depth: Optional[Tensor],
                                    g_acc: Optional[Tensor], brick_size: int, cursor: Tensor, records_sorted: Tensor,
                                    hist_clear: Optional[Tensor] = None) -> None:
    """Enqueue rf_render_backward_emit_direct: the samples counted by the forward pass (``render_forward_raw(key_hist=)``,
    then ``bin_offsets``) write their expanded records straight to their final positions in ``records_sorted``."""
    lib = _lib.load()
    dev = origins.device
    rf_grid = grid.to_rf_grid(use_occupancy=bool(flags & _lib.FLAG_OCCUPANCY_SKIP))
    rb, tv = _ray_batch(origins, directions, num_samples, near, far, t_rand)
    grads = _lib.RFRenderGrads()
    grads.grad_colour_dev, grads.grad_depth_dev, grads.grad_acc_dev = _ptr(g_colour), _ptr(g_depth), _ptr(g_acc)
    cache, tcache, stop, cmask = caches
    fwd = _lib.RFRenderOut()
    fwd.sample_cache_dev, fwd.trans_cache_dev, fwd.stop_cache_dev, fwd.chunk_mask_dev = cache.data_ptr(), tcache.data_ptr(), stop.data_ptr(), cmask.data_ptr()
    with _span(f"render_backward_emit_direct[{_variant(grid, flags)}]", dev):
        rc = lib.rf_render_backward_emit_direct(
            C.byref(rf_grid), C.byref(rb), _jitter_flags(flags, t_rand), C.byref(fwd), C.byref(grads), int(brick_size), cursor.data_ptr(),
            records_sorted.data_ptr(), _ptr(hist_clear), _stream(dev),
        )
    _lib.check(rc, "rf_render_backward_emit_direct")


def _render_out(n: int, num_samples: int, dev, save: bool, key_hist: Optional[Tensor], brick_size: int):
    """fresh output (and cache) tensors of one render + their RFRenderOut"""
    f32 = dict(dtype=torch.float32, device=dev)
    colour, depth, acc, disparity = torch.empty((n, 3), **f32), torch.empty((n, 1), **f32), torch.empty((n, 1), **f32), torch.empty((n, 1), **f32)
    out = _lib.RFRenderOut()
    out.colour_dev, out.depth_dev, out.acc_dev, out.disparity_dev = colour.data_ptr(), depth.data_ptr(), acc.data_ptr(), disparity.data_ptr()
    caches = None
    if save:
        caches = (torch.empty((n, num_samples, 4), **f32), torch.empty((n, num_samples), **f32), torch.empty((n,), dtype=torch.int32, device=dev),
                  torch.empty((n, (num_samples + 63) // 64), dtype=torch.int64, device=dev))
        out.sample_cache_dev, out.trans_cache_dev, out.stop_cache_dev, out.chunk_mask_dev = (t.data_ptr() for t in caches)
        if key_hist is not None:
            out.key_hist_dev, out.brick_size = key_hist.data_ptr(), int(brick_size)
    return (colour, depth, acc, disparity), caches, out


def render_forward_pair_raw(grid: VoxelGrid, origins: Tensor, directions: Tensor, t_rands, num_samples: int, near: float, far: float, flags,
                            key_hists, brick_size: int = 8):
    """Enqueue rf_render_forward_pair: BOTH saving forward renders of a training iteration over the same rays -- [0] specular,
    [1] render_diffuse, each with its own jitter (``t_rands[i]``: None, a [N, S] tensor or a KeyedJitter) and its own record counters
    ``key_hists[i]`` -- in ONE launch (modules/trainers.py:306, 323-325).  Returns ((colour, depth, acc, disparity), caches) per render.
    No autograd.  Raises like the single call; ``None`` when the library says the two renders do not pair up."""
    lib = _lib.load()
    for name, t in (("ray origins", origins), ("ray directions", directions)):
        _require_hip(t, name)
    n, dev = origins.shape[0], origins.device
    rf_grid = grid.forward_rf_grid(use_occupancy=bool(flags[0] & _lib.FLAG_OCCUPANCY_SKIP))
    rays, outs, fl = (_lib.RFRayBatch * 2)(), (_lib.RFRenderOut * 2)(), (C.c_uint32 * 2)()
    results, keep = [], []
    for i in range(2):
        rb, tv = _ray_batch(origins, directions, num_samples, near, far, t_rands[i])
        tensors, caches, out = _render_out(n, num_samples, dev, True, key_hists[i], brick_size)
        rays[i], outs[i], fl[i] = rb, out, _jitter_flags(flags[i], t_rands[i])
        results.append((tensors, caches))
        keep.append(tv)
    with _span(f"render_forward[{_variant(grid, flags[0])}+diffuse,save]", dev):
        rc = lib.rf_render_forward_pair(C.byref(rf_grid), rays, fl, outs, _stream(dev))
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "rf_render_forward_pair")
    return results


def render_backward_emit_direct_pair_raw(grid: VoxelGrid, origins: Tensor, directions: Tensor, t_rands, num_samples: int, near: float, far: float, flags,
                                         caches2, g_colours, brick_size: int, hists, offsets2: Tensor, cursor2: Tensor, records2) -> bool:
    """The adjoints of both renders as record lists: ONE launch for the offsets of both lists (rf_bin_offsets_pair) and ONE for both
    emits (rf_render_backward_emit_direct_pair); the forward pass's counters ``hists[i]`` are cleared for their next use.
    ``offsets2`` [2, nkeys + 1] int64, ``cursor2`` [2, nkeys] int32, ``records2[i]`` [N * S, expanded_record_floats].  Returns False
    (nothing launched but the offsets) when the library says the two adjoints do not pair up."""
    lib = _lib.load()
    dev = origins.device
    nkeys = int(hists[0].numel())
    vp2 = C.c_void_p * 2
    with _span("bin_offsets[both]", dev):
        rc = lib.rf_bin_offsets_pair(vp2(hists[0].data_ptr(), hists[1].data_ptr()), nkeys, vp2(offsets2[0].data_ptr(), offsets2[1].data_ptr()),
                                     vp2(cursor2[0].data_ptr(), cursor2[1].data_ptr()), _stream(dev))
    _lib.check(rc, "rf_bin_offsets_pair")
    rf_grid = grid.to_rf_grid(use_occupancy=bool(flags[0] & _lib.FLAG_OCCUPANCY_SKIP))
    rays, passes, fl = (_lib.RFRayBatch * 2)(), (_lib.RFPassScratch * 2)(), (C.c_uint32 * 2)()
    keep = []
    for i in range(2):
        rb, tv = _ray_batch(origins, directions, num_samples, near, far, t_rands[i])
        keep.append(tv)
        rays[i], fl[i] = rb, _jitter_flags(flags[i], t_rands[i])
        ps = passes[i]
        cache, tcache, stop, cmask = caches2[i]
        ps.out.sample_cache_dev, ps.out.trans_cache_dev, ps.out.stop_cache_dev, ps.out.chunk_mask_dev = cache.data_ptr(), tcache.data_ptr(), stop.data_ptr(), cmask.data_ptr()
        ps.out.key_hist_dev, ps.out.brick_size = hists[i].data_ptr(), int(brick_size)
        ps.grad_colour_dev, ps.cursor_dev, ps.offsets_dev, ps.records_sorted_dev = g_colours[i].data_ptr(), cursor2[i].data_ptr(), offsets2[i].data_ptr(), records2[i].data_ptr()
    with _span(f"render_backward_emit_direct[{_variant(grid, flags[0])}+diffuse]", dev):
        rc = lib.rf_render_backward_emit_direct_pair(C.byref(rf_grid), rays, fl, passes, _stream(dev))
    if rc == _lib.ERR_UNSUPPORTED:
        return False
    _lib.check(rc, "rf_render_backward_emit_direct_pair")
    return True


def bin_offsets(hist: Tensor, offsets: Tensor, cursor: Tensor) -> Tensor:
    """``hist`` (records per key) -> ``offsets`` [len + 1] (int64 exclusive prefix sums) and ``cursor`` (int32 copy)."""
    lib = _lib.load()
    dev = hist.device
    with _span("bin_offsets", dev):
        rc = lib.rf_bin_offsets(hist.data_ptr(), int(hist.numel()), offsets.data_ptr(), cursor.data_ptr(), _stream(dev))
    _lib.check(rc, "rf_bin_offsets")
    return offsets


def expanded_record_floats(grid: VoxelGrid, render_diffuse: bool = False) -> int:
    """floats per record in the sorted lists consumed by rf_brick_accumulate: 12 (compact: index, d density, d raw rgb, unit view
    direction -- the SH expansion happens inside the brick pass), or 8 for degree-0 grids and render_diffuse passes"""
    return int(_lib.load().rf_expanded_record_floats(3 if render_diffuse else int(grid.num_features)))


def sort_records_by_brick(grid: VoxelGrid, keys: Tensor, records: Tensor, render_diffuse: bool,
                          records_sorted: Tensor, offsets: Tensor, boundaries: Tensor) -> Tensor:
    """Sort the dense key array (16-bit radix sort; key = brick * 8 + boundary flags, -1 = slot without gradient) and
    write the records of keyed samples, expanded to per-channel values, in that order.  ``offsets``
    [8 * num_bricks + 1] (int64, last element preset to keys.numel()) receives the start of every (brick, flags)
    class inside ``records_sorted`` [keys.numel(), expanded_record_floats]; positions are absolute, the unkeyed slots
    occupy [0, offsets[0])."""
    lib = _lib.load()
    dev = keys.device
    with _span("sort_keys", dev):
        sorted_keys, perm = torch.sort(keys)
        torch.searchsorted(sorted_keys, boundaries, out=offsets[:-1])
    rf_grid = grid.to_rf_grid()
    with _span("expand_records", dev):
        rc = lib.rf_expand_records(C.byref(rf_grid), records.data_ptr(), perm.data_ptr(), offsets.data_ptr(), keys.numel(),
                                   int(bool(render_diffuse)), records_sorted.data_ptr(), _stream(dev))
    _lib.check(rc, "rf_expand_records")
    return offsets


def bin_records_by_brick(grid: VoxelGrid, keys: Tensor, records: Tensor, render_diffuse: bool,
                         hist: Tensor, cursor: Tensor, records_sorted: Tensor, offsets: Tensor) -> Tensor:
    """Counting sort instead of torch.sort: ``hist`` (filled by render_backward_emit_raw) -> ``offsets``
    [8 * num_bricks + 1] (int64; positions start at 0) and ``cursor`` (int32 scratch); every keyed slot then takes the
    next free position of its key class (atomic cursor) and its expanded record is written there.  ``hist`` is cleared."""
    lib = _lib.load()
    dev = keys.device
    with _span("bin_offsets", dev):
        rc = lib.rf_bin_offsets(hist.data_ptr(), int(hist.numel()), offsets.data_ptr(), cursor.data_ptr(), _stream(dev))
    _lib.check(rc, "rf_bin_offsets")
    rf_grid = grid.to_rf_grid()
    with _span(f"scatter_records[{'diffuse' if render_diffuse or grid.sh_degree == 0 else 'sh' + str(grid.sh_degree)}]", dev):
        rc = lib.rf_scatter_records(C.byref(rf_grid), keys.data_ptr(), records.data_ptr(), keys.numel(), cursor.data_ptr(),
                                    int(bool(render_diffuse)), records_sorted.data_ptr(), hist.data_ptr(), int(hist.numel()), _stream(dev))
    _lib.check(rc, "rf_scatter_records")
    return offsets


def brick_accumulate_raw(grid: VoxelGrid, brick_size: int, lists, grad_first: Tensor, grad_second: Optional[Tensor],
                         accumulate: bool) -> None:
    """``lists`` = [(records_sorted, offsets, render_diffuse), ...] (1 or 2 entries, specular first).  Enqueue
    rf_brick_accumulate."""
    lib = _lib.load()
    dev = grad_first.device
    arr = (_lib.RFBrickList * len(lists))()
    for i, (rec, off, diffuse) in enumerate(lists):
        arr[i].records_sorted_dev, arr[i].offsets_dev, arr[i].render_diffuse = rec.data_ptr(), off.data_ptr(), int(bool(diffuse))
    rf_grid = grid.to_rf_grid()
    with _span(f"brick_accumulate[{'diffuse' if lists[0][2] or grid.sh_degree == 0 else 'sh' + str(grid.sh_degree)}]", dev):
        rc = lib.rf_brick_accumulate(C.byref(rf_grid), int(brick_size), arr, len(lists), grad_first.data_ptr(), _ptr(grad_second), int(bool(accumulate)), _stream(dev))
    _lib.check(rc, "rf_brick_accumulate")


def brick_accumulate_adam_raw(grid: VoxelGrid, brick_size: int, lists, exp_avg, exp_avg_sq, lr: float, beta1: float, beta2: float,
                              eps: float, step: int, brick_range=None, rf_grid=None, params=None, split=None, mirror=None) -> None:
    """Enqueue rf_brick_accumulate_adam: the brick pass over ``lists`` (as in ``brick_accumulate_raw``; all renders of the
    iteration -- of all ranks, under data parallelism) with the Adam update of the grid's own tensors applied in the flush.
    ``exp_avg`` / ``exp_avg_sq`` are pairs of tensors shaped like ``grid.kernel_tensors()`` (second entry None when the grid has no
    second tensor).  ``brick_range`` = (first_brick, num_bricks) restricts the pass (and the update) to those bricks.  An entry of
    ``lists`` may give its records as an int (a raw device address) instead of a tensor.  ``rf_grid`` / ``params``: update these
    tensors (described by this RFGrid) instead of the grid's own -- the split-layout shadow of a reference-storage grid.
    ``split`` = (parts, scratch): rf_brick_accumulate_adam_split -- ``parts`` workgroups per brick, the lists of each kind dealt out to
    them, ``scratch`` a zero-initialised uint8 tensor of ``brick_split_scratch_bytes`` bytes (needs ``brick_range``).
    ``mirror`` = (densities [X,Y,Z,1], features [X,Y,Z,F]): rf_brick_accumulate_adam_mirror -- the flush also writes the updated
    parameters in the reference layout into these tensors (see ``mirror_flush_applies``)."""
    lib = _lib.load()
    # (``_tensors``: no wait for parameters a data-parallel step left in flight -- that step orders its launches against them itself)
    first, second = (grid._tensors() if rf_grid is not None else grid.kernel_tensors()) if params is None else params
    dev = first.device
    arr = (_lib.RFBrickList * len(lists))()
    for i, (rec, off, diffuse) in enumerate(lists):
        arr[i].records_sorted_dev, arr[i].offsets_dev, arr[i].render_diffuse = (rec if isinstance(rec, int) else rec.data_ptr()), off.data_ptr(), int(bool(diffuse))
    st = _lib.RFAdamState()
    st.param_first_dev, st.param_second_dev = first.data_ptr(), _ptr(second)
    st.exp_avg_first_dev, st.exp_avg_second_dev = exp_avg[0].data_ptr(), _ptr(exp_avg[1])
    st.exp_avg_sq_first_dev, st.exp_avg_sq_second_dev = exp_avg_sq[0].data_ptr(), _ptr(exp_avg_sq[1])
    st.lr, st.beta1, st.beta2, st.eps, st.step = float(lr), float(beta1), float(beta2), float(eps), int(step)
    if rf_grid is None:
        rf_grid = grid.to_rf_grid()
    with _span(f"brick_accumulate_adam[{'diffuse' if lists[0][2] or grid.sh_degree == 0 else 'sh' + str(grid.sh_degree)}]", dev):
        if mirror is not None:
            rc = lib.rf_brick_accumulate_adam_mirror(C.byref(rf_grid), int(brick_size), arr, len(lists), C.byref(st), mirror[0].data_ptr(), mirror[1].data_ptr(), _stream(dev))
        elif split is not None and int(split[0]) > 1:
            rc = lib.rf_brick_accumulate_adam_split(C.byref(rf_grid), int(brick_size), arr, len(lists), C.byref(st), int(brick_range[0]), int(brick_range[1]),
                                                    int(split[0]), split[1].data_ptr(), int(split[1].numel()), _stream(dev))
        elif brick_range is None:
            rc = lib.rf_brick_accumulate_adam(C.byref(rf_grid), int(brick_size), arr, len(lists), C.byref(st), _stream(dev))
        else:
            rc = lib.rf_brick_accumulate_adam_range(C.byref(rf_grid), int(brick_size), arr, len(lists), C.byref(st), int(brick_range[0]), int(brick_range[1]), _stream(dev))
    _lib.check(rc, "rf_brick_accumulate_adam")


def mirror_flush_applies(grid, brick_size: int, densities: Tensor, features: Tensor) -> bool:
    """True when rf_brick_accumulate_adam_mirror takes this grid: 4 x 8 x 8 bricks, dims multiples of the brick, the two reference-layout
    tensors contiguous float32 at 16-byte aligned addresses."""
    X, Y, Z = grid.grid_dims
    return (int(brick_size) == BRICK_4X8X8 and X % 4 == 0 and Y % 8 == 0 and Z % 8 == 0 and features is not None
            and all(t.is_contiguous() and t.dtype == torch.float32 and t.data_ptr() % 16 == 0 for t in (densities, features)))


def brick_split_scratch(grid: VoxelGrid, num_bricks: int, parts: int) -> Tensor:
    """zero-initialised scratch of rf_brick_accumulate_adam_split for launches over ``num_bricks`` bricks with ``parts`` workgroups each"""
    first, _ = grid._tensors() if hasattr(grid, "_tensors") else grid.kernel_tensors()
    n = int(_lib.load().rf_brick_split_scratch_bytes(C.byref(grid.to_rf_grid(wait_parameters=False)), int(num_bricks), int(parts)))
    if n < 0:
        raise ValueError("rf_brick_split_scratch_bytes: bad arguments")
    return torch.zeros(n, dtype=torch.uint8, device=first.device)


# How the autograd op computes its adjoint: "atomic" = rf_render_backward (float32 atomic scatter, any configuration), "binned" =
# counting in the forward pass -> rf_render_backward_emit_direct -> rf_brick_accumulate (no float atomics),
# "auto" = binned for renders (specular and render_diffuse) of SH-degree-2 / -3 grids with at least 2^20 samples (where the atomic
# scatter is pinned on the memory-side atomic unit: 0.97 vs 0.40 ms for the specular adjoint of 16384 x 256 samples at 128^3; the
# reference-storage iteration of bench.py's strict drop-in leg: 2.79 -> 1.99 ms with the diffuse adjoint binned as well), atomic
# otherwise.
AUTOGRAD_BACKWARD = "auto"
# the two renders of a training iteration as ONE autograd node where it applies (relu_field_render_pair); $RF_AUTOGRAD_PAIR=0 / False:
# always the two single nodes (A/B runs, tests)
PAIR_RENDERS = os.environ.get("RF_AUTOGRAD_PAIR", "1") != "0"
AUTOGRAD_BINNED_MAX_BYTES = 1 << 30
AUTOGRAD_BRICK_SIZE = 8


def _autograd_brick_size(grid) -> int:
    """Bricks of the binned adjoint's record lists: the deferred bucket's choice (optim.FlatGrid.brick_size) when it consumes them."""
    bucket = getattr(grid, "_grad_bucket", None)
    return int(getattr(bucket, "brick_size", AUTOGRAD_BRICK_SIZE)) if getattr(bucket, "deferred", False) else AUTOGRAD_BRICK_SIZE


def _autograd_uses_bricks(grid, flags: int, n: int, num_samples: int) -> bool:
    nb = brick_counts(grid, _autograd_brick_size(grid))
    deferred = getattr(getattr(grid, "_grad_bucket", None), "deferred", False)
    if nb[0] * nb[1] * nb[2] * 8 > (1 << 21):
        if deferred:
            raise ValueError("deferred gradients (optim.FlatGrid(deferred=True)) need at most 2^18 bricks")
        return False
    if deferred:  # the optimizer consumes record lists: every adjoint is binned
        return True
    if AUTOGRAD_BACKWARD == "atomic":
        return False
    if AUTOGRAD_BACKWARD == "binned":
        return True
    # the binned adjoint allocates one worst-case record list per backward (48 B per sample slot): above AUTOGRAD_BINNED_MAX_BYTES
    # the atomic kernel, which needs no scratch, is used instead (user code that fitted before keeps fitting)
    if n * num_samples * 4 * expanded_record_floats(grid) > AUTOGRAD_BINNED_MAX_BYTES:
        return False
    # (and, like TrainStepper(backward="auto") and optim.FlatGrid(deferred): below 256 bricks of 8^3 nodes a handful of brick workgroups
    # would sum the whole render's records -- 16^3: 2.26 ms per iteration binned against 0.79 ms atomic, tools/small_grid_steps.py)
    nb8 = brick_counts(grid, 8)
    if nb8[0] * nb8[1] * nb8[2] < int(os.environ.get("RF_AUTO_BINNED_MIN_BRICKS", "256")):
        return False
    return grid.sh_degree >= 2 and n * num_samples >= (1 << 20)


# zeroed per-key record counters of the autograd op's binned adjoint: a forward pass takes one, the adjoint that consumes it has its emit
# launch clear it again (stream order) and hands it back -- no torch.zeros launch per render.  (A forward pass whose adjoint never
# runs simply keeps its buffer; the pool allocates another.)
_HIST_POOL: Dict[Tuple[str, int], list] = {}


def _take_hist(device, nkeys: int) -> Tensor:
    pool = _HIST_POOL.setdefault((str(device), int(nkeys)), [])
    return pool.pop() if pool else torch.zeros(int(nkeys), dtype=torch.int32, device=device)


def _return_hist(hist: Tensor) -> None:
    pool = _HIST_POOL.setdefault((str(hist.device), int(hist.numel())), [])
    if len(pool) < 8:
        pool.append(hist)


class _ReluFieldRender(torch.autograd.Function):
    @staticmethod
    def forward(ctx, first, second, origins, directions, t_rand, grid: VoxelGrid, num_samples, near, far, flags, need_grad):
        for t in (first, second):
            if t is not None:
                _require_hip(t, "grid tensor")
        origins = origins.detach().to(torch.float32).contiguous()
        directions = directions.detach().to(torch.float32).contiguous()
        n = origins.shape[0]
        keyed = t_rand if isinstance(t_rand, KeyedJitter) else None
        if keyed is not None:
            t_rand = None
        if t_rand is not None:
            _require_hip(t_rand, "t_rand")
            t_rand = t_rand.detach().to(torch.float32).contiguous()
            if tuple(t_rand.shape) != (n, num_samples):
                raise ValueError(f"t_rand must be [{n}, {num_samples}], got {tuple(t_rand.shape)}")
        key_hist = None
        brick_size = _autograd_brick_size(grid)
        if need_grad and _autograd_uses_bricks(grid, int(flags), n, int(num_samples)):
            nb = brick_counts(grid, brick_size)
            key_hist = _take_hist(origins.device, nb[0] * nb[1] * nb[2] * 8)
        colour, depth, acc, disparity, caches = render_forward_raw(
            grid, origins, directions, keyed if keyed is not None else t_rand, int(num_samples), float(near), float(far), int(flags), bool(need_grad),
            key_hist=key_hist, brick_size=brick_size,
        )
        ctx.grid, ctx.flags, ctx.keyed, ctx.key_hist, ctx.brick_size = grid, int(flags), keyed, key_hist, brick_size
        ctx.num_samples, ctx.near, ctx.far = int(num_samples), float(near), float(far)
        ctx.has_rand = t_rand is not None
        ctx.has_second = second is not None
        ctx.need_grad = bool(need_grad)
        saved = [first] + ([second] if second is not None else []) + [origins, directions]
        if need_grad:
            saved += list(caches)
        if t_rand is not None:
            saved.append(t_rand)
        ctx.save_for_backward(*saved)
        ctx.mark_non_differentiable(disparity)
        # (outputs the loss does not use -- depth, accumulated weight in every reference use -- arrive as None in backward instead of
        # zero tensors autograd would fill per render: two [N, 1] fill launches each, ~20 us of GPU time per training iteration)
        ctx.set_materialize_grads(False)
        return colour, depth, acc, disparity

    @staticmethod
    def backward(ctx, g_colour, g_depth, g_acc, _g_disparity):
        if not ctx.need_grad:
            return (None,) * 11
        saved = list(ctx.saved_tensors)
        first = saved.pop(0)
        second = saved.pop(0) if ctx.has_second else None
        origins, directions, cache, tcache, stop, cmask = saved[:6]
        t_rand = saved[6] if ctx.has_rand else ctx.keyed
        grid: VoxelGrid = ctx.grid
        cur_first, cur_second = grid.kernel_tensors()
        if cur_first.data_ptr() != first.data_ptr() or (second is not None and cur_second.data_ptr() != second.data_ptr()):
            raise RuntimeError("the VoxelGrid's tensors were replaced between forward and backward")

        def prep(g):
            return None if g is None else g.detach().to(torch.float32).contiguous()

        # gradient buffers: a caller-provided flat bucket when there is one (optim.FlatGrid), else fresh tensors
        bucket = getattr(grid, "_grad_bucket", None)
        binned = ctx.key_hist is not None
        diffuse = bool(ctx.flags & _lib.FLAG_RENDER_DIFFUSE)
        overwrite = False
        if bucket is not None and bucket.matches(first, second):
            gd, gf = bucket.views_for_accumulation()
            ret_d, ret_f = bucket.autograd_return()
        else:
            # fresh tensors.  The brick pass can OVERWRITE them (every element a full-width list covers; the density and degree-0
            # elements for a render_diffuse list): no zero-fill and no read-add-write of 235 MB for a specular render
            overwrite = binned
            covers_all = binned and (not diffuse or grid.sh_degree == 0)  # (a render_diffuse list leaves the higher degrees alone)
            fresh = torch.empty_like if covers_all else torch.zeros_like
            gd = fresh(first)
            gf = None if second is None else fresh(second)
            ret_d, ret_f = gd, gf
        if ctx.key_hist is not None:
            # binned adjoint: the forward pass counted the records per (brick, flags) key; offsets -> records written at their
            # final positions -> one workgroup per brick sums them and ADDS the brick to the gradient tensors
            hist, dev = ctx.key_hist, origins.device
            offsets = torch.empty(hist.numel() + 1, dtype=torch.int64, device=dev)
            cursor = torch.empty(hist.numel(), dtype=torch.int32, device=dev)
            records = torch.empty((origins.shape[0] * ctx.num_samples, expanded_record_floats(grid, diffuse)), dtype=torch.float32, device=dev)
            bin_offsets(hist, offsets, cursor)
            render_backward_emit_direct_raw(
                grid, origins, directions, t_rand, ctx.num_samples, ctx.near, ctx.far, ctx.flags, (cache, tcache, stop, cmask),
                prep(g_colour), prep(g_depth), prep(g_acc), ctx.brick_size, cursor, records, hist_clear=hist,
            )
            _return_hist(hist)  # (zero again once the emit launch above has run: every later user is behind it on the stream)
            if bucket is not None and getattr(bucket, "deferred", False) and bucket.matches(first, second) and ctx.brick_size == bucket.brick_size:
                # deferred gradients: the sorted list IS the gradient of this render; the optimizer sums all lists of the iteration
                bucket.pending.append((records, offsets, diffuse))
            elif ctx.brick_size != AUTOGRAD_BRICK_SIZE:
                raise RuntimeError("the grid's deferred gradient bucket was replaced between forward and backward")
            else:
                brick_accumulate_raw(grid, AUTOGRAD_BRICK_SIZE, [(records, offsets, diffuse)], gd, gf, accumulate=not overwrite)
            ctx.key_hist = None  # (a second backward through the same graph would find the counters consumed)
        else:
            if bucket is not None and getattr(bucket, "deferred", False) and bucket.matches(first, second):
                # (the optimizer of a deferred bucket consumes record lists only: a gradient scattered into the bucket would be lost)
                raise RuntimeError("deferred gradients (optim.FlatGrid(deferred=True)): a render can be back-propagated once per optimizer step "
                                   "(its record counters were consumed by the first backward pass); use FlatGrid(deferred=False) for retained graphs")
            render_backward_raw(
                grid, origins, directions, t_rand, ctx.num_samples, ctx.near, ctx.far, ctx.flags, (cache, tcache, stop, cmask),
                prep(g_colour), prep(g_depth), prep(g_acc), gd, gf,
            )
        return ret_d, ret_f, None, None, None, None, None, None, None, None, None


class _ReluFieldRenderPair(torch.autograd.Function):
    """BOTH renders of a training iteration -- ``render_rays(rays)`` and ``render_rays(rays, render_diffuse=True)``
    (modules/trainers.py:306, 323-325), each with its own jitter -- as ONE autograd node: one forward launch
    (rf_render_forward_pair), and a backward of two launches (offsets of both record lists, both adjoints) whose record lists go to
    the deferred bucket's optimizer or through ONE brick pass into the gradient tensors.  Used where the single op would take the binned
    adjoint (``_autograd_uses_bricks``); outputs (colour, depth, acc, disparity) x 2."""

    @staticmethod
    def forward(ctx, first, second, origins, directions, t_rand0, t_rand1, grid: VoxelGrid, num_samples, near, far, flags0, flags1):
        for t in (first, second):
            if t is not None:
                _require_hip(t, "grid tensor")
        origins = origins.detach().to(torch.float32).contiguous()
        directions = directions.detach().to(torch.float32).contiguous()
        n, S = origins.shape[0], int(num_samples)
        t_rands, keyed = [], []
        for t in (t_rand0, t_rand1):
            k = t if isinstance(t, KeyedJitter) else None
            if k is None and t is not None:
                _require_hip(t, "t_rand")
                t = t.detach().to(torch.float32).contiguous()
                if tuple(t.shape) != (n, S):
                    raise ValueError(f"t_rand must be [{n}, {S}], got {tuple(t.shape)}")
            keyed.append(k)
            t_rands.append(None if k is not None else t)
        brick_size = _autograd_brick_size(grid)
        nb = brick_counts(grid, brick_size)
        hists = [_take_hist(origins.device, nb[0] * nb[1] * nb[2] * 8) for _ in range(2)]
        flags = (int(flags0), int(flags1))
        res = render_forward_pair_raw(grid, origins, directions, [keyed[i] if keyed[i] is not None else t_rands[i] for i in range(2)], S, float(near), float(far), flags,
                                      hists, brick_size)
        if res is None:
            raise RuntimeError("rf_render_forward_pair: the two renders do not pair up (flags must say specular, render_diffuse)")
        ctx.grid, ctx.flags, ctx.keyed, ctx.hists, ctx.brick_size = grid, flags, keyed, hists, brick_size
        ctx.num_samples, ctx.near, ctx.far = S, float(near), float(far)
        ctx.has_rand = [t is not None for t in t_rands]
        ctx.has_second = second is not None
        saved = [first] + ([second] if second is not None else []) + [origins, directions]
        for (_, caches), t in zip(res, t_rands):
            saved += list(caches)
            if t is not None:
                saved.append(t)
        ctx.save_for_backward(*saved)
        outs = tuple(res[0][0]) + tuple(res[1][0])
        ctx.mark_non_differentiable(outs[3], outs[7])
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    def backward(ctx, gc0, gd0, ga0, _gq0, gc1, gd1, ga1, _gq1):
        saved = list(ctx.saved_tensors)
        first = saved.pop(0)
        second = saved.pop(0) if ctx.has_second else None
        origins, directions = saved.pop(0), saved.pop(0)
        caches2, t_rands = [], []
        for i in range(2):
            caches2.append(tuple(saved[:4]))
            del saved[:4]
            t_rands.append(saved.pop(0) if ctx.has_rand[i] else ctx.keyed[i])
        grid: VoxelGrid = ctx.grid
        if ctx.hists is None:
            raise RuntimeError("a render pair can be back-propagated once (its record counters were consumed by the first backward pass)")
        cur_first, cur_second = grid.kernel_tensors()
        if cur_first.data_ptr() != first.data_ptr() or (second is not None and cur_second.data_ptr() != second.data_ptr()):
            raise RuntimeError("the VoxelGrid's tensors were replaced between forward and backward")

        def prep(g):
            return None if g is None else g.detach().to(torch.float32).contiguous()

        dev, n, S = origins.device, origins.shape[0], ctx.num_samples
        hists, nkeys = ctx.hists, int(ctx.hists[0].numel())
        offsets2 = torch.empty((2, nkeys + 1), dtype=torch.int64, device=dev)
        cursor2 = torch.empty((2, nkeys), dtype=torch.int32, device=dev)
        records2 = [torch.empty((n * S, expanded_record_floats(grid, bool(i))), dtype=torch.float32, device=dev) for i in range(2)]
        g_colours, g_depths, g_accs = [prep(gc0), prep(gc1)], [prep(gd0), prep(gd1)], [prep(ga0), prep(ga1)]
        colours_only = all(g is not None for g in g_colours) and all(g is None for g in g_depths + g_accs)
        paired = colours_only and render_backward_emit_direct_pair_raw(grid, origins, directions, t_rands, S, ctx.near, ctx.far, ctx.flags, caches2, g_colours,
                                                                       ctx.brick_size, hists, offsets2, cursor2, records2)
        if not paired:  # upstream gradients of depth / accumulated weight, or an unused render: one adjoint at a time (the general kernel)
            for i in range(2):
                if not colours_only:
                    bin_offsets(hists[i], offsets2[i], cursor2[i])
                if g_colours[i] is None and g_depths[i] is None and g_accs[i] is None:
                    g_colours[i] = torch.zeros((n, 3), dtype=torch.float32, device=dev)  # (an unused render: its records are zeros)
                render_backward_emit_direct_raw(grid, origins, directions, t_rands[i], S, ctx.near, ctx.far, ctx.flags[i], caches2[i], g_colours[i], g_depths[i], g_accs[i],
                                                ctx.brick_size, cursor2[i], records2[i], hist_clear=hists[i])
        for h in hists:
            _return_hist(h)
        ctx.hists = None
        lists = [(records2[0], offsets2[0], False), (records2[1], offsets2[1], True)]
        bucket = getattr(grid, "_grad_bucket", None)
        if bucket is not None and bucket.matches(first, second):
            if getattr(bucket, "deferred", False) and ctx.brick_size == bucket.brick_size:
                bucket.pending.extend(lists)  # the sorted lists ARE the gradient: the optimizer sums all lists of the iteration
                ret_d, ret_f = bucket.autograd_return()
                return (ret_d, ret_f) + (None,) * 10
            if ctx.brick_size != AUTOGRAD_BRICK_SIZE:
                raise RuntimeError("the grid's deferred gradient bucket was replaced between forward and backward")
            gd, gf = bucket.views_for_accumulation()
            ret_d, ret_f = bucket.autograd_return()
            brick_accumulate_raw(grid, AUTOGRAD_BRICK_SIZE, lists, gd, gf, accumulate=True)
            return (ret_d, ret_f) + (None,) * 10
        if ctx.brick_size != AUTOGRAD_BRICK_SIZE:
            raise RuntimeError("the grid's deferred gradient bucket was replaced between forward and backward")
        # fresh gradient tensors, OVERWRITTEN by the one brick pass over both lists (the specular list covers every element)
        gd = torch.empty_like(first)
        gf = None if second is None else torch.empty_like(second)
        brick_accumulate_raw(grid, AUTOGRAD_BRICK_SIZE, lists, gd, gf, accumulate=False)
        return (gd, gf) + (None,) * 10


class _GridQuery(torch.autograd.Function):
    @staticmethod
    def forward(ctx, first, second, points, grid: VoxelGrid):
        lib = _lib.load()
        _require_hip(points, "points")
        points = points.detach().to(torch.float32).contiguous()
        m = points.shape[0]
        out = torch.empty((m, grid.num_features + 1), dtype=torch.float32, device=points.device)
        rf_grid = grid.to_rf_grid()
        _lib.check(lib.rf_grid_query(C.byref(rf_grid), points.data_ptr(), m, out.data_ptr(), _stream(points.device)), "rf_grid_query")
        ctx.grid, ctx.has_second = grid, second is not None
        ctx.save_for_backward(*([first] + ([second] if second is not None else []) + [points]))
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = _lib.load()
        saved = list(ctx.saved_tensors)
        first = saved.pop(0)
        second = saved.pop(0) if ctx.has_second else None
        points = saved[0]
        g_out = g_out.detach().to(torch.float32).contiguous()
        gd = torch.zeros_like(first)
        gf = None if second is None else torch.zeros_like(second)
        rf_grid = ctx.grid.to_rf_grid()
        _lib.check(
            lib.rf_grid_query_backward(C.byref(rf_grid), points.data_ptr(), points.shape[0], g_out.data_ptr(), gd.data_ptr(), _ptr(gf), _stream(points.device)),
            "rf_grid_query_backward",
        )
        return gd, gf, None, None


class _InterpolateTensors(torch.autograd.Function):
    """Trilinear interpolation (the bit-exact grid_sample recipe of rf_grid_query) of two EXPLICIT reference-layout tensors
    [X,Y,Z,1] / [X,Y,Z,F] at points [M,3] of a grid's box: no density scale, no activation -- the building block of the composed
    path, where arbitrary callables run in torch before and after it.  Differentiable w.r.t. both tensors."""

    @staticmethod
    def forward(ctx, dens, feat, points, geometry):
        lib = _lib.load()
        dens = dens.detach().to(torch.float32).contiguous()
        feat = feat.detach().to(torch.float32).contiguous()
        points = points.detach().to(torch.float32).contiguous()
        _require_hip(points, "points")
        rf = _lib.RFGrid()
        C.memmove(C.byref(rf), C.byref(geometry), C.sizeof(_lib.RFGrid))
        rf.densities_dev, rf.features_dev = dens.data_ptr(), feat.data_ptr()
        rf.num_features, rf.density_stride, rf.feature_stride = int(feat.shape[-1]), 1, int(feat.shape[-1])
        rf.layout, rf.density_scale, rf.density_mode, rf.occupancy_dev = _lib.LAYOUTS["reference"], 1.0, _lib.DENSITY_MODES["identity"], None
        m = points.shape[0]
        out = torch.empty((m, feat.shape[-1] + 1), dtype=torch.float32, device=points.device)
        _lib.check(lib.rf_grid_query(C.byref(rf), points.data_ptr(), m, out.data_ptr(), _stream(points.device)), "rf_grid_query")
        ctx.rf = rf
        ctx.save_for_backward(dens, feat, points)
        return out

    @staticmethod
    def backward(ctx, g_out):
        dens, feat, points = ctx.saved_tensors
        g_out = g_out.detach().to(torch.float32).contiguous()
        gd, gf = torch.zeros_like(dens), torch.zeros_like(feat)
        _lib.check(_lib.load().rf_grid_query_backward(C.byref(ctx.rf), points.data_ptr(), points.shape[0], g_out.data_ptr(), gd.data_ptr(), gf.data_ptr(),
                                                      _stream(points.device)), "rf_grid_query_backward")
        return gd, gf, None, None


def interpolate_tensors(grid, dens: Tensor, feat: Tensor, points: Tensor) -> Tensor:
    """[M, F + 1] = (interp(feat), interp(dens)) at ``points`` inside ``grid``'s box (its dims, AABB and normalisation; its own
    tensors, density scale and activations are NOT used)."""
    if tuple(dens.shape[:3]) != tuple(grid.grid_dims) or tuple(feat.shape[:3]) != tuple(grid.grid_dims) or dens.shape[-1] != 1:
        raise AssertionError("interpolate_tensors: tensors must be [X,Y,Z,1] / [X,Y,Z,F] of the grid's dimensions")
    g = _lib.RFGrid()
    for a in range(3):
        g.dims[a] = grid.grid_dims[a]
        lo, hi = grid._aabb[a]
        g.aabb_min[a], g.aabb_max[a] = float(np.float32(lo)), float(np.float32(hi))
        from .camera import slack_range_map

        scale, bias = slack_range_map((lo, hi))
        g.norm_scale[a], g.norm_bias[a] = float(scale), float(bias)
    return _InterpolateTensors.apply(dens, feat, points, g)


def grid_query(grid: VoxelGrid, points: Tensor) -> Tensor:
    """[M, 3] points -> [M, F+1] = (interpolated features in the reference order, activated density); differentiable
    w.r.t. the grid (reference VoxelGrid.forward, thre3d_reprs/voxels.py:276-331)."""
    if points.dim() != 2 or points.shape[-1] != 3:
        raise AssertionError("points must be [M, 3]")
    first, second = grid.kernel_tensors()
    return _GridQuery.apply(first, second, points, grid)


def render_flags(white_bkgd: bool, render_diffuse: bool, optimized_sampling: bool, use_occupancy: bool) -> int:
    flags = 0
    flags |= _lib.FLAG_WHITE_BKGD if white_bkgd else 0
    flags |= _lib.FLAG_RENDER_DIFFUSE if render_diffuse else 0
    flags |= _lib.FLAG_AABB_SAMPLING if optimized_sampling else 0
    flags |= _lib.FLAG_OCCUPANCY_SKIP if use_occupancy else 0
    return flags


def l1_loss_grad_hip(colour: Tensor, target: Tensor, sums: Tensor, scale: float = 1.0) -> Tensor:
    """d(scale * mean|colour - target|)/d colour as one kernel; ``sums`` [2] (+= sum |d|, += sum d^2) for logging."""
    _require_hip(colour, "colour")
    lib = _lib.load()
    grad = torch.empty_like(colour)
    with _span("l1_loss_grad", colour.device):
        rc = lib.rf_l1_loss_grad(colour.data_ptr(), target.data_ptr(), colour.shape[0], float(scale), grad.data_ptr(), sums.data_ptr(), _stream(colour.device))
    _lib.check(rc, "rf_l1_loss_grad")
    return grad


# zeroed (sum |d|, sum d^2) slots for l1_loss_with_mse: handed out one after the other, the whole ring cleared once per turn -- no
# torch.zeros launch per loss (a tiny launch still costs ~5 us of GPU time between two 100 us kernels)
_L1_SUMS_RING: Dict[Tuple[str, int], list] = {}
_L1_RING_SLOTS = 1024


def _l1_sums_slot(device) -> Tensor:
    # (one ring per stream: a ring is zeroed by a launch on the stream that creates it)
    st = _L1_SUMS_RING.setdefault((str(device), torch.cuda.current_stream(device).cuda_stream), [None, _L1_RING_SLOTS])
    if st[1] >= _L1_RING_SLOTS:
        # (a fresh ring per turn: slots handed out earlier may still be referenced by losses somebody kept)
        st[0], st[1] = torch.zeros((_L1_RING_SLOTS, 2), dtype=torch.float32, device=device), 0
    st[1] += 1
    return st[0][st[1] - 1]


class _L1LossWithMSE(torch.autograd.Function):
    """mean |colour - target| (differentiable w.r.t. colour) and mean (colour - target)^2 (for the PSNR the reference logs,
    modules/trainers.py:311-317) in ONE launch of rf_l1_loss_grad, which also leaves d loss / d colour for the backward pass: what
    torch.nn.functional.l1_loss + mse_loss and their autograd graph do in ~14 launches."""

    @staticmethod
    def forward(ctx, colour, target):
        colour_c = colour.detach().to(torch.float32).contiguous()
        target_c = target.detach().to(colour_c.device, torch.float32).contiguous()
        if colour_c.shape != target_c.shape or colour_c.dim() != 2 or colour_c.shape[1] != 3:
            raise ValueError(f"l1_loss_with_mse takes [N, 3] colours and targets, got {tuple(colour.shape)} and {tuple(target.shape)}")
        sums = _l1_sums_slot(colour_c.device)
        grad = l1_loss_grad_hip(colour_c, target_c, sums)
        ctx.save_for_backward(grad)
        means = sums * (1.0 / float(colour_c.numel()))
        loss, mse = means[0], means[1]
        ctx.mark_non_differentiable(mse)
        ctx.set_materialize_grads(False)
        return loss, mse

    @staticmethod
    def backward(ctx, g_loss, _g_mse):
        if g_loss is None:
            return None, None
        (grad,) = ctx.saved_tensors
        return grad * g_loss, None


_L1_PAIR_WORKSPACE: Dict[Tuple[str, int], Tensor] = {}


class _L1LossPairWithMSE(torch.autograd.Function):
    """The loss lines of an iteration (modules/trainers.py:311-317, 329-336) for BOTH renders in ONE launch (rf_l1_loss_grad_pair):
    returns [5] = (L1(specular) + L1(diffuse), L1(specular), MSE(specular), L1(diffuse), MSE(diffuse)) as views of one tensor the
    kernel's last workgroup writes; differentiable through the FIRST (the sum, w.r.t. both colours)."""

    @staticmethod
    def forward(ctx, colour0, colour1, target):
        c0 = colour0.detach().to(torch.float32).contiguous()
        c1 = colour1.detach().to(c0.device, torch.float32).contiguous()
        tg = target.detach().to(c0.device, torch.float32).contiguous()
        if not (c0.shape == c1.shape == tg.shape) or c0.dim() != 2 or c0.shape[1] != 3 or c0.shape[0] < 1:
            raise ValueError(f"l1_loss_pair_with_mse takes [N, 3] colours and targets, got {tuple(colour0.shape)}, {tuple(colour1.shape)} and {tuple(target.shape)}")
        dev = c0.device
        key = (str(dev), torch.cuda.current_stream(dev).cuda_stream)  # (one self-cleaning workspace per stream: zero between calls)
        ws = _L1_PAIR_WORKSPACE.get(key)
        if ws is None:
            ws = _L1_PAIR_WORKSPACE[key] = torch.zeros(8, dtype=torch.float32, device=dev)
        grads = torch.empty((2,) + tuple(c0.shape), dtype=torch.float32, device=dev)
        out = torch.empty(5, dtype=torch.float32, device=dev)
        vp2 = C.c_void_p * 2
        with _span("l1_loss_grad[both]", dev):
            rc = _lib.load().rf_l1_loss_grad_pair(vp2(c0.data_ptr(), c1.data_ptr()), tg.data_ptr(), c0.shape[0], 1.0, vp2(grads[0].data_ptr(), grads[1].data_ptr()),
                                                  ws.data_ptr(), out.data_ptr(), _stream(dev))
        _lib.check(rc, "rf_l1_loss_grad_pair")
        ctx.save_for_backward(grads)
        total, l0, mse0, l1, mse1 = out.unbind(0)
        ctx.mark_non_differentiable(l0, mse0, l1, mse1)
        ctx.set_materialize_grads(False)
        return total, l0, mse0, l1, mse1

    @staticmethod
    def backward(ctx, g_total, *_unused):
        if g_total is None:
            return None, None, None
        (grads,) = ctx.saved_tensors
        g = grads * g_total  # (one launch for both renders' upstream gradients)
        return g[0], g[1], None


def l1_loss_pair_with_mse(colour_specular: Tensor, colour_diffuse: Tensor, target: Tensor):
    """(L1(specular) + L1(diffuse) -- differentiable --, L1(specular), MSE(specular), L1(diffuse), MSE(diffuse)): both renders of an
    iteration against the same target pixels, one launch."""
    _require_hip(colour_specular, "colour")
    _require_hip(colour_diffuse, "colour")
    return _L1LossPairWithMSE.apply(colour_specular, colour_diffuse, target)


def l1_loss_with_mse(colour: Tensor, target: Tensor) -> Tuple[Tensor, Tensor]:
    """(mean L1 loss -- differentiable w.r.t. ``colour`` --, mean squared error) of [N, 3] colours against their targets, one launch."""
    _require_hip(colour, "colour")
    return _L1LossWithMSE.apply(colour, target)


def relu_field_render(
    grid: VoxelGrid,
    origins: Tensor,
    directions: Tensor,
    num_samples: int,
    near: float,
    far: float,
    t_rand=None,
    white_bkgd: bool = False,
    render_diffuse: bool = False,
    optimized_sampling: bool = False,
    use_occupancy: bool = False,
) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """colour [N,3], depth [N,1], accumulated weight [N,1], disparity [N,1] for flat rays [N,3].
    Differentiable w.r.t. ``grid.densities`` and ``grid.features`` only (like every reference use).
    ``t_rand``: None (no jitter), a [N, S] tensor of jitter values, or a ``KeyedJitter``."""
    grid = as_kernel_grid(grid)
    if origins.dim() != 2 or origins.shape != directions.shape or origins.shape[-1] != 3:
        raise AssertionError("the render op works with FLAT rays [N, 3] only")
    if int(num_samples) < 1:
        raise ValueError("num_samples must be >= 1")
    if use_occupancy and not grid.occupancy_current():
        grid.build_occupancy()  # no mask yet, or the densities changed since it was built (optimizer step, in-place edit)
    flags = render_flags(white_bkgd, render_diffuse, optimized_sampling, use_occupancy)
    # the per-sample cache for the backward pass is only written when a gradient can be asked for
    # grid tensors in storage order: (densities, features) or, for split storage, (base, rest)
    ta, tb = grid.kernel_tensors()
    need_grad = torch.is_grad_enabled() and (ta.requires_grad or (tb is not None and tb.requires_grad))
    return _ReluFieldRender.apply(
        ta, tb, origins, directions, t_rand, grid, int(num_samples), float(near), float(far), flags, need_grad
    )


def relu_field_render_pair(grid: VoxelGrid, origins: Tensor, directions: Tensor, num_samples: int, near: float, far: float, t_rands=(None, None),
                           white_bkgd: bool = False, optimized_sampling: bool = False, use_occupancy: bool = False):
    """``relu_field_render(...)`` and ``relu_field_render(..., render_diffuse=True)`` on the same rays -- the two renders of a training
    iteration (modules/trainers.py:306, 323-325) -- returned as two (colour, depth, acc, disparity) tuples.  Where a gradient is asked
    for and the single op would take the binned adjoint, the pair is ONE autograd node (one forward launch, two backward launches);
    otherwise it is the two single ops, in the reference's order.  ``t_rands``: the two renders' jitter (each None, [N, S] or KeyedJitter)."""
    grid = as_kernel_grid(grid)
    if origins.dim() != 2 or origins.shape != directions.shape or origins.shape[-1] != 3:
        raise AssertionError("the render op works with FLAT rays [N, 3] only")
    if int(num_samples) < 1:
        raise ValueError("num_samples must be >= 1")
    if use_occupancy and not grid.occupancy_current():
        grid.build_occupancy()
    flags = [render_flags(white_bkgd, diffuse, optimized_sampling, use_occupancy) for diffuse in (False, True)]
    ta, tb = grid.kernel_tensors()
    need_grad = torch.is_grad_enabled() and (ta.requires_grad or (tb is not None and tb.requires_grad))
    n = origins.shape[0]
    if (PAIR_RENDERS and need_grad and n > 0 and _autograd_uses_bricks(grid, flags[0], n, int(num_samples)) and _autograd_uses_bricks(grid, flags[1], n, int(num_samples))):
        o = _ReluFieldRenderPair.apply(ta, tb, origins, directions, t_rands[0], t_rands[1], grid, int(num_samples), float(near), float(far), flags[0], flags[1])
        return o[:4], o[4:]
    return tuple(_ReluFieldRender.apply(ta, tb, origins, directions, t_rands[i], grid, int(num_samples), float(near), float(far), flags[i], need_grad) for i in range(2))


# --------------------------------------------------------------------------------------------
# ray generation
# --------------------------------------------------------------------------------------------
def _pose_to_host(rotation, translation):
    rot = torch.as_tensor(rotation).detach().to("cpu", torch.float32).reshape(9).tolist()
    trans = torch.as_tensor(translation).detach().to("cpu", torch.float32).reshape(3).tolist()
    return (C.c_float * 9)(*rot), (C.c_float * 3)(*trans)


def cast_rays_hip(height: int, width: int, focal: float, rotation, translation, device) -> Tuple[Tensor, Tensor]:
    """All pixel-centre rays of one camera -> (origins, directions) [H, W, 3] float32 on ``device``
    (reference rendering/volumetric/utils/misc.py:12-50)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError(f"cast_rays runs on a HIP device only (got {device})")
    lib = _lib.load()
    rot, trans = _pose_to_host(rotation, translation)
    o = torch.empty((height, width, 3), dtype=torch.float32, device=device)
    d = torch.empty((height, width, 3), dtype=torch.float32, device=device)
    _lib.check(
        lib.rf_cast_rays(int(height), int(width), float(np.float32(focal)), rot, trans, o.data_ptr(), d.data_ptr(), _stream(device)),
        "rf_cast_rays",
    )
    return o, d


def cast_selected_rays_hip(height: int, width: int, focal: float, poses: Tensor, pixel_index: Tensor) -> Tuple[Tensor, Tensor]:
    """Rays of selected pixels of a batch of cameras: ``poses`` [B, 3, 4] (rotation | translation) and
    ``pixel_index`` [R] int64 into the B*H*W concatenated pixels (the ray set the reference builds per
    training iteration at modules/trainers.py:281-303, without materialising the other rays)."""
    _require_hip(poses, "poses")
    _require_hip(pixel_index, "pixel_index")
    lib = _lib.load()
    poses = poses.detach().to(torch.float32).contiguous()
    pixel_index = pixel_index.detach().to(torch.int64).contiguous()
    n = pixel_index.shape[0]
    o = torch.empty((n, 3), dtype=torch.float32, device=poses.device)
    d = torch.empty((n, 3), dtype=torch.float32, device=poses.device)
    _lib.check(
        lib.rf_cast_selected_rays(
            int(height), int(width), float(np.float32(focal)), poses.data_ptr(), int(poses.shape[0]), pixel_index.data_ptr(), n, o.data_ptr(), d.data_ptr(), _stream(poses.device)
        ),
        "rf_cast_selected_rays",
    )
    return o, d


def select_rays_and_pixels_hip(height: int, width: int, focal: float, poses: Tensor, image_ids: Tensor, pixel_table: Tensor, num_rays: int, key: int, return_index: bool = False, first_index: int = 0):
    """Fused random batch selection: ``num_rays`` distinct pixels of the images ``image_ids`` chosen by a keyed
    pseudo-random permutation, with their rays and target colours -> (origins, directions, pixels[, index]).
    ``poses`` [M,3,4] and ``pixel_table`` [M*H*W,3] cover the whole dataset.  ``first_index`` > 0 draws elements
    [first_index, first_index + num_rays) of the same permutation (disjoint slices of one global batch)."""
    _require_hip(poses, "poses")
    _require_hip(pixel_table, "pixel_table")
    lib = _lib.load()
    dev = poses.device
    image_ids = image_ids.detach().to(dev, torch.int64).contiguous()
    n = int(num_rays)
    o = torch.empty((n, 3), dtype=torch.float32, device=dev)
    d = torch.empty((n, 3), dtype=torch.float32, device=dev)
    px = torch.empty((n, 3), dtype=torch.float32, device=dev)
    idx = torch.empty((n,), dtype=torch.int64, device=dev) if return_index else None
    with _span("select_rays_and_pixels", dev):
        rc = lib.rf_select_rays_and_pixels(
            int(height), int(width), float(np.float32(focal)), poses.data_ptr(), image_ids.data_ptr(), int(image_ids.numel()),
            pixel_table.data_ptr(), int(key) & 0xFFFFFFFFFFFFFFFF, int(first_index), n, o.data_ptr(), d.data_ptr(), px.data_ptr(), _ptr(idx), _stream(dev),
        )
    _lib.check(rc, "rf_select_rays_and_pixels")
    return (o, d, px, idx) if return_index else (o, d, px)


def ray_aabb_bounds_hip(origins: Tensor, directions: Tensor, near: float, far: float, aabb) -> Tuple[Tensor, Tensor]:
    """Per-ray [t_enter, t_exit] and hit flags (reference rendering/volumetric/sample.py:71-184)."""
    _require_hip(origins, "ray origins")
    lib = _lib.load()
    origins = origins.detach().to(torch.float32).contiguous()
    directions = directions.detach().to(torch.float32).contiguous()
    n = origins.shape[0]
    bounds = torch.empty((n, 2), dtype=torch.float32, device=origins.device)
    hit = torch.empty((n, 1), dtype=torch.float32, device=origins.device)
    lo = _lib.float3([np.float32(r[0]) for r in aabb])
    hi = _lib.float3([np.float32(r[1]) for r in aabb])
    _lib.check(
        lib.rf_ray_aabb_bounds(
            origins.data_ptr(), directions.data_ptr(), n, float(np.float32(near)), float(np.float32(far)), lo, hi, bounds.data_ptr(), hit.data_ptr(), _stream(origins.device)
        ),
        "rf_ray_aabb_bounds",
    )
    return bounds, hit


def adam_step_hip(param: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, lr: float, beta1: float, beta2: float, eps: float, step: int, zero_grad: bool = False) -> None:
    """In-place fused Adam update of a flat float32 buffer (torch.optim.Adam arithmetic); ``zero_grad`` also clears
    ``grad`` in the same pass (the next iteration's optimizer.zero_grad())."""
    for name, t in (("param", param), ("grad", grad), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
        _require_hip(t, name)
        if not (t.is_contiguous() and t.dtype == torch.float32 and t.numel() == param.numel()):
            raise ValueError(f"{name} must be a contiguous float32 tensor with {param.numel()} elements")
    lib = _lib.load()
    with _span("adam_step", param.device):
        rc = lib.rf_adam_step(
            param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), param.numel(), float(lr), float(beta1), float(beta2), float(eps), int(step), int(bool(zero_grad)), _stream(param.device)
        )
    _lib.check(rc, "rf_adam_step")
