"""thre3d_atom/thre3d_reprs/renderers_hip.py -- the file a maintainer of akanimax/thr3ed_atom adds to use the MI355X library.

It binds `librelu_field_hip.so` (C ABI: include/relu_field.h, RF_ABI_VERSION 4) with ctypes and exposes

    render_sh_voxel_grid_hip(voxel_grid, rays, render_config, parallel_points_chunk_size=None) -> RenderOut

a `RenderProcedure` (thre3d_atom/thre3d_reprs/renderers.py:22-25) that is a drop-in for `render_sh_voxel_grid` (:48-102):

    vol_mod = VolumetricModel(thre3d_repr=voxel_grid, render_procedure=render_sh_voxel_grid_hip, render_config=...)

`voxel_grid` is the REFERENCE's own VoxelGrid (thre3d_reprs/voxels.py) living on a HIP device; `rays`/`RenderOut` are the
reference's own types.  The render is differentiable w.r.t. `voxel_grid.densities` / `.features` through a
torch.autograd.Function: forward = rf_render_forward with the per-sample cache; backward = rf_render_backward (float atomics; any
configuration) or, for large renders of SH-degree >= 2 grids -- a training batch --, the atomic-free binned adjoint: the forward pass
counts the gradient records per (brick, flags) key, rf_bin_offsets turns the counts into positions, rf_render_backward_emit_direct
writes the records there and rf_brick_accumulate sums every 8^3-node brick on chip (0.97 -> 0.40 ms for 16384 x 256 samples at 128^3).
This file depends on torch, numpy, ctypes and the reference package only -- NOT on the thr3ed_atom_amd Python package.

The one other edit the reference needs: the identity assert of the trainer (modules/trainers.py:116-122) must accept the new
procedure, e.g. `vol_mod.render_procedure in (render_sh_voxel_grid, render_sh_voxel_grid_hip)`.

Optional second edit, for the trainer's iteration (modules/trainers.py:306-330 renders the same rays twice: specular, then
`render_diffuse=True`): `render_sh_voxel_grid_pair_hip(voxel_grid, rays, render_config)` returns both RenderOuts from ONE autograd
node -- one forward launch (rf_render_forward_pair), and a backward of three (rf_bin_offsets_pair, rf_render_backward_emit_direct_pair,
one rf_brick_accumulate over both record lists) instead of six.

Forward passes of a training-size render and frames gather from a split-layout COPY of the grid (see _split_shadow).  The copy follows
the module's tensors through their data pointers and autograd version counters; a write that bumps neither (`p.data.copy_(...)`,
`p.data.clamp_()`, a raw-pointer kernel) must be followed by `invalidate_split_shadow(voxel_grid)`.  `release_split_shadow(voxel_grid)`
frees the copy (235 MB at 128^3 / SH-2, 1.9 GB at 256^3) when training ends; RELU_FIELD_HIP_SPLIT_SHADOW=0 never makes one.
"""
import ctypes as C
import os

import numpy as np
import torch

from thre3d_atom.rendering.volumetric.render_interface import Rays, RenderOut
from thre3d_atom.utils.constants import EXTRA_ACCUMULATED_WEIGHTS, EXTRA_DISPARITY

RF_ABI_VERSION = 4
_LIB_PATH = os.environ.get("RELU_FIELD_HIP_LIB", "librelu_field_hip.so")


# ---- struct mirrors of include/relu_field.h, field for field ------------------------------------------------------------
class RFGrid(C.Structure):
    _fields_ = [("densities_dev", C.c_void_p), ("features_dev", C.c_void_p), ("dims", C.c_int32 * 3), ("num_features", C.c_int32),
                ("density_stride", C.c_int64), ("feature_stride", C.c_int64), ("aabb_min", C.c_float * 3), ("aabb_max", C.c_float * 3),
                ("norm_scale", C.c_float * 3), ("norm_bias", C.c_float * 3), ("density_scale", C.c_float), ("density_mode", C.c_int32),
                ("layout", C.c_int32), ("occupancy_dev", C.c_void_p)]


class RFRayBatch(C.Structure):
    _fields_ = [("origins_dev", C.c_void_p), ("directions_dev", C.c_void_p), ("num_rays", C.c_int64), ("num_samples", C.c_int32),
                ("near", C.c_float), ("far", C.c_float), ("t_vals_dev", C.c_void_p), ("t_rand_dev", C.c_void_p),
                ("jitter_key", C.c_uint64), ("first_ray", C.c_int64), ("camera", C.c_void_p)]


class RFRenderOut(C.Structure):
    _fields_ = [("colour_dev", C.c_void_p), ("depth_dev", C.c_void_p), ("acc_dev", C.c_void_p), ("disparity_dev", C.c_void_p),
                ("sample_cache_dev", C.c_void_p), ("trans_cache_dev", C.c_void_p), ("stop_cache_dev", C.c_void_p),
                ("chunk_mask_dev", C.c_void_p), ("key_hist_dev", C.c_void_p), ("brick_size", C.c_int32)]


class RFRenderGrads(C.Structure):
    _fields_ = [("grad_colour_dev", C.c_void_p), ("grad_depth_dev", C.c_void_p), ("grad_acc_dev", C.c_void_p)]


class RFCamera(C.Structure):
    _fields_ = [("height", C.c_int32), ("width", C.c_int32), ("focal", C.c_float), ("pose", C.c_float * 12)]


class RFBrickList(C.Structure):
    _fields_ = [("records_sorted_dev", C.c_void_p), ("offsets_dev", C.c_void_p), ("render_diffuse", C.c_int32)]


class RFPassScratch(C.Structure):
    _fields_ = [("out", RFRenderOut), ("grad_colour_dev", C.c_void_p), ("cursor_dev", C.c_void_p), ("offsets_dev", C.c_void_p),
                ("records_sorted_dev", C.c_void_p), ("t_rand_dev", C.c_void_p), ("jitter_key", C.c_uint64)]


# backward: "auto" = the binned adjoint for SH degree >= 2, at least 2^20 samples and at least RELU_FIELD_HIP_MIN_BRICKS (256) bricks of
# 8^3 nodes -- the measured crossover, see _use_binned_adjoint --, else float atomics; "atomic" / "binned" force one
BACKWARD = os.environ.get("RELU_FIELD_HIP_BACKWARD", "auto")
MIN_BRICKS = int(os.environ.get("RELU_FIELD_HIP_MIN_BRICKS", "256"))
BRICK = 8


RF_FLAG_WHITE_BKGD, RF_FLAG_RENDER_DIFFUSE, RF_FLAG_AABB_SAMPLING = 1, 2, 4
RF_DENSITY_RELU, RF_DENSITY_SOFTPLUS, RF_DENSITY_ABS, RF_DENSITY_IDENTITY = 0, 1, 2, 3
RF_LAYOUT_REFERENCE, RF_LAYOUT_SPLIT = 0, 1
RF_FLAG_JITTER_KEYED = 16

_lib = None


def _library():
    global _lib
    if _lib is None:
        lib = C.CDLL(_LIB_PATH)
        lib.rf_abi_version.restype = C.c_int
        lib.rf_error_string.restype = C.c_char_p
        lib.rf_error_string.argtypes = [C.c_int]
        lib.rf_render_forward.restype = C.c_int
        lib.rf_render_forward.argtypes = [C.POINTER(RFGrid), C.POINTER(RFRayBatch), C.c_uint32, C.POINTER(RFRenderOut), C.c_void_p]
        lib.rf_render_backward.restype = C.c_int
        lib.rf_render_backward.argtypes = [C.POINTER(RFGrid), C.POINTER(RFRayBatch), C.c_uint32, C.POINTER(RFRenderOut), C.POINTER(RFRenderGrads),
                                           C.c_void_p, C.c_void_p, C.c_void_p]
        lib.rf_bin_offsets.restype = lib.rf_render_backward_emit_direct.restype = lib.rf_brick_accumulate.restype = C.c_int
        lib.rf_expanded_record_floats.restype = C.c_int32
        lib.rf_expanded_record_floats.argtypes = [C.c_int32]
        lib.rf_bin_offsets.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.rf_render_backward_emit_direct.argtypes = [C.POINTER(RFGrid), C.POINTER(RFRayBatch), C.c_uint32, C.POINTER(RFRenderOut), C.POINTER(RFRenderGrads),
                                                       C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.rf_brick_accumulate.argtypes = [C.POINTER(RFGrid), C.c_int32, C.POINTER(RFBrickList), C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        # the paired entry points (every array argument holds two entries: [0] specular, [1] render_diffuse)
        vp2 = C.POINTER(C.c_void_p)
        lib.rf_render_forward_pair.restype = lib.rf_bin_offsets_pair.restype = lib.rf_render_backward_emit_direct_pair.restype = C.c_int
        lib.rf_render_forward_pair.argtypes = [C.POINTER(RFGrid), C.POINTER(RFRayBatch), C.POINTER(C.c_uint32), C.POINTER(RFRenderOut), C.c_void_p]
        lib.rf_bin_offsets_pair.argtypes = [vp2, C.c_int32, vp2, vp2, C.c_void_p]
        lib.rf_render_backward_emit_direct_pair.argtypes = [C.POINTER(RFGrid), C.POINTER(RFRayBatch), C.POINTER(C.c_uint32), C.POINTER(RFPassScratch), C.c_void_p]
        if lib.rf_abi_version() != RF_ABI_VERSION:
            raise RuntimeError(f"{_LIB_PATH}: ABI version {lib.rf_abi_version()}, this binding was written for {RF_ABI_VERSION}")
        _lib = lib
    return _lib


def _check(code, what):
    if code != 0:
        raise RuntimeError(f"{what} failed: {_library().rf_error_string(code).decode()} (code {code})")


def _is_identity(fn):
    return fn is None or isinstance(fn, torch.nn.Identity)


def _density_mode(voxel_grid):
    pre, post = voxel_grid._density_preactivation, voxel_grid._density_postactivation
    if _is_identity(pre) and isinstance(post, torch.nn.ReLU):
        return RF_DENSITY_RELU
    if _is_identity(pre) and isinstance(post, torch.nn.Softplus) and post.beta == 1 and post.threshold == 20:
        return RF_DENSITY_SOFTPLUS
    if pre is torch.abs and _is_identity(post):
        return RF_DENSITY_ABS
    if _is_identity(pre) and _is_identity(post):
        return RF_DENSITY_IDENTITY
    raise ValueError(f"unsupported density activations for the HIP renderer: pre={pre}, post={post}")


def _describe_grid(voxel_grid, densities, features):
    """RFGrid of the reference's VoxelGrid: its own two tensors (RF_LAYOUT_REFERENCE), its AABB (voxels.py:187-212) and the
    float32 (scale, bias) of adjust_dynamic_range(slack=True) (utils/imaging_utils.py:58-63) that _normalize_points applies."""
    if not (_is_identity(voxel_grid._feature_preactivation) and _is_identity(voxel_grid._feature_postactivation)):
        raise ValueError("the HIP renderer supports identity feature activations only")
    if voxel_grid._radiance_transfer_function is not None:
        raise ValueError("the HIP renderer does not take a radiance transfer function")
    g = RFGrid()
    g.densities_dev, g.features_dev = densities.data_ptr(), features.data_ptr()
    for a, (lo, hi) in enumerate(voxel_grid.aabb):
        g.dims[a] = int(densities.shape[a])
        g.aabb_min[a], g.aabb_max[a] = float(np.float32(lo)), float(np.float32(hi))
        scale = (np.float32(1.0) - np.float32(-1.0)) / (np.float32(hi) - np.float32(lo))
        g.norm_scale[a] = float(scale)
        g.norm_bias[a] = float(np.float32(-1.0) - np.float32(lo) * scale)
    g.num_features, g.density_stride, g.feature_stride = int(features.shape[-1]), 1, int(features.shape[-1])
    g.density_scale = float(voxel_grid._expected_density_scale)
    g.density_mode = _density_mode(voxel_grid)
    g.layout = RF_LAYOUT_REFERENCE
    g.occupancy_dev = None
    return g


def _use_binned_adjoint(features, n, num_samples):
    nkeys = 8
    for d in features.shape[:3]:
        nkeys *= (int(d) + BRICK - 1) // BRICK
    if nkeys > (1 << 21) or BACKWARD == "atomic":
        return 0
    record_bytes = 4 * int(_library().rf_expanded_record_floats(int(features.shape[-1])))
    # "auto": the binned adjoint from 2^20 samples on a degree >= 2 grid of at least 256 bricks (measured crossover on one MI355X,
    # fwd+bwd of 16384 x 256 samples: 128^3 0.50 ms binned / 1.33 atomic, 64^3 0.37 / 0.44, 32^3 1.13 / 0.49, 16^3 2.26 / 0.79 --
    # on a small grid a handful of brick workgroups would sum every record, while the atomic adjoint's targets sit in L2)
    big = (int(features.shape[-1]) >= 27 and n * num_samples >= (1 << 20) and n * num_samples * record_bytes <= (1 << 30)
           and nkeys // 8 >= MIN_BRICKS)
    return nkeys if (BACKWARD == "binned" or big) else 0


class _RenderFunction(torch.autograd.Function):
    """forward: rf_render_forward (+ the per-sample cache when a gradient can be asked for); backward: rf_render_backward
    into zero-filled gradient tensors, or the binned adjoint (see the module docstring) -- what autograd does for the reference
    through ~120 ATen ops."""

    @staticmethod
    def forward(ctx, densities, features, origins, directions, t_vals, t_rand, voxel_grid, num_samples, near, far, flags, need_grad):
        lib = _library()
        dev = origins.device
        n = origins.shape[0]
        grid = _describe_grid(voxel_grid, densities, features)
        rb = RFRayBatch(origins.data_ptr(), directions.data_ptr(), n, num_samples, near, far, t_vals.data_ptr(),
                        None if t_rand is None else t_rand.data_ptr(), 0, 0, None)
        colour = torch.empty((n, 3), dtype=torch.float32, device=dev)
        depth, acc, disparity = (torch.empty((n, 1), dtype=torch.float32, device=dev) for _ in range(3))
        out = RFRenderOut(colour.data_ptr(), depth.data_ptr(), acc.data_ptr(), disparity.data_ptr(), None, None, None, None, None, 0)
        caches = ()
        if need_grad:  # per-sample cache of the samples that carry gradient + which ones they are (relu_field.h: RFRenderOut)
            caches = (torch.empty((n, num_samples, 4), dtype=torch.float32, device=dev), torch.empty((n, num_samples), dtype=torch.float32, device=dev),
                      torch.empty((n,), dtype=torch.int32, device=dev), torch.empty((n, (num_samples + 63) // 64), dtype=torch.int64, device=dev))
            out.sample_cache_dev, out.trans_cache_dev, out.stop_cache_dev, out.chunk_mask_dev = (c.data_ptr() for c in caches)
            nkeys = _use_binned_adjoint(features, n, num_samples)
            if nkeys:  # the forward pass counts the records of the binned adjoint per (brick, flags) key
                caches = caches + (torch.zeros(nkeys, dtype=torch.int32, device=dev),)
                out.key_hist_dev, out.brick_size = caches[-1].data_ptr(), BRICK
        ctx.binned = len(caches) == 5
        stream = torch.cuda.current_stream(dev).cuda_stream
        # forward passes gather from the split-layout copy of the grid (an aligned 16-byte base record per corner instead of 108
        # unaligned feature bytes + 4 bytes in another tensor: the diffuse forward of a training batch 0.29 -> 0.08 ms), refreshed by
        # one rf_convert_grid launch (0.12 ms) when an optimizer step has changed the tensors; adjoints keep the module's own layout
        fgrid, keep = _split_shadow(voxel_grid, densities, features, grid) if int(features.shape[-1]) in (3, 27) else (grid, None)
        _check(lib.rf_render_forward(C.byref(fgrid), C.byref(rb), flags, C.byref(out), stream), "rf_render_forward")
        del keep
        ctx.voxel_grid, ctx.args, ctx.need_grad, ctx.has_rand = voxel_grid, (num_samples, near, far, flags), need_grad, t_rand is not None
        ctx.save_for_backward(densities, features, origins, directions, t_vals, *caches, *(() if t_rand is None else (t_rand,)))
        ctx.mark_non_differentiable(disparity)
        ctx.set_materialize_grads(False)  # (unused outputs' gradients arrive as None, not as zero tensors filled per render)
        return colour, depth, acc, disparity

    @staticmethod
    def backward(ctx, g_colour, g_depth, g_acc, _g_disparity):
        if not ctx.need_grad:
            return (None,) * 12
        lib = _library()
        saved = ctx.saved_tensors
        densities, features, origins, directions, t_vals, cache, tcache, stop, cmask = saved[:9]
        hist = saved[9] if ctx.binned else None
        t_rand = saved[-1] if ctx.has_rand else None
        num_samples, near, far, flags = ctx.args
        dev = origins.device
        grid = _describe_grid(ctx.voxel_grid, densities, features)
        rb = RFRayBatch(origins.data_ptr(), directions.data_ptr(), origins.shape[0], num_samples, near, far, t_vals.data_ptr(),
                        None if t_rand is None else t_rand.data_ptr(), 0, 0, None)
        fwd = RFRenderOut(None, None, None, None, cache.data_ptr(), tcache.data_ptr(), stop.data_ptr(), cmask.data_ptr(), None, 0)
        keep = [None if g is None else g.detach().to(torch.float32).contiguous() for g in (g_colour, g_depth, g_acc)]
        grads = RFRenderGrads(*[None if g is None else g.data_ptr() for g in keep])
        stream = torch.cuda.current_stream(dev).cuda_stream
        if hist is not None:
            diffuse = bool(flags & RF_FLAG_RENDER_DIFFUSE)
            width = int(lib.rf_expanded_record_floats(3 if diffuse else int(features.shape[-1])))
            offsets = torch.empty(hist.numel() + 1, dtype=torch.int64, device=dev)
            cursor = torch.empty(hist.numel(), dtype=torch.int32, device=dev)
            records = torch.empty((origins.shape[0] * num_samples, width), dtype=torch.float32, device=dev)
            _check(lib.rf_bin_offsets(hist.data_ptr(), hist.numel(), offsets.data_ptr(), cursor.data_ptr(), stream), "rf_bin_offsets")
            _check(lib.rf_render_backward_emit_direct(C.byref(grid), C.byref(rb), flags, C.byref(fwd), C.byref(grads), BRICK, cursor.data_ptr(), records.data_ptr(),
                                                      None, stream), "rf_render_backward_emit_direct")
            # the brick pass OVERWRITES what its list covers: everything for a specular render; density + degree-0 coefficients for a
            # render_diffuse pass (the other coefficients get no gradient from it: zero-filled)
            grad_d = torch.empty_like(densities)
            grad_f = torch.zeros_like(features) if diffuse and features.shape[-1] > 3 else torch.empty_like(features)
            lst = RFBrickList(records.data_ptr(), offsets.data_ptr(), int(diffuse))
            _check(lib.rf_brick_accumulate(C.byref(grid), BRICK, C.byref(lst), 1, grad_d.data_ptr(), grad_f.data_ptr(), 0, stream), "rf_brick_accumulate")
            return grad_d, grad_f, None, None, None, None, None, None, None, None, None, None
        grad_d, grad_f = torch.zeros_like(densities), torch.zeros_like(features)  # the library accumulates (+=)
        _check(lib.rf_render_backward(C.byref(grid), C.byref(rb), flags, C.byref(fwd), C.byref(grads), grad_d.data_ptr(), grad_f.data_ptr(), stream),
               "rf_render_backward")
        return grad_d, grad_f, None, None, None, None, None, None, None, None, None, None


def render_sh_voxel_grid_hip(voxel_grid, rays: Rays, render_config, parallel_points_chunk_size=None) -> RenderOut:
    """RenderProcedure: same arguments and results as thre3d_reprs/renderers.py:48-102 (parallel_points_chunk_size is accepted
    and ignored: the fused kernel materialises nothing per point).  Configurations the kernels do not implement raise."""
    if render_config.density2occupancy.__name__ != "density2occupancy_pb" or render_config.radiance_hdr_tone_map is not torch.sigmoid:
        raise ValueError("render_sh_voxel_grid_hip: only density2occupancy_pb / torch.sigmoid are implemented")
    if render_config.stochastic_density_noise_std != 0.0:
        raise ValueError("render_sh_voxel_grid_hip: stochastic_density_noise_std must be 0.0")
    origins = rays.origins.detach().to(torch.float32).contiguous()
    directions = rays.directions.detach().to(torch.float32).contiguous()
    assert origins.dim() == 2 and directions.shape == origins.shape, "the render interface only works with FLAT rays"
    if not origins.is_cuda:
        raise RuntimeError("render_sh_voxel_grid_hip runs on a HIP device only")
    densities, features = voxel_grid.densities, voxel_grid.features
    if not (densities.is_cuda and densities.is_contiguous() and features.is_contiguous() and densities.dtype == features.dtype == torch.float32):
        raise RuntimeError("the VoxelGrid's tensors must be contiguous float32 on the HIP device")
    n, s = origins.shape[0], int(render_config.num_samples_per_ray)
    t_vals = torch.linspace(0.0, 1.0, s, dtype=torch.float32).to(origins.device)  # computed on the host like sample.py:46 on CPU
    t_rand = torch.rand(n, s, dtype=torch.float32, device=origins.device) if render_config.perturb_sampled_points else None  # sample.py:63
    flags = (RF_FLAG_WHITE_BKGD if render_config.white_bkgd else 0) | (RF_FLAG_RENDER_DIFFUSE if render_config.render_diffuse else 0) | (
        RF_FLAG_AABB_SAMPLING if render_config.optimized_sampling else 0)
    near, far = float(np.float32(render_config.camera_bounds.near)), float(np.float32(render_config.camera_bounds.far))
    # (grad mode is off inside Function.forward: whether the per-sample cache is needed is decided here)
    need_grad = torch.is_grad_enabled() and (densities.requires_grad or features.requires_grad)
    colour, depth, acc, disparity = _RenderFunction.apply(densities, features, origins, directions, t_vals, t_rand, voxel_grid, s, near, far, flags, need_grad)
    return RenderOut(colour=colour, depth=depth, extra={EXTRA_DISPARITY: disparity, EXTRA_ACCUMULATED_WEIGHTS: acc})


# ---- the two renders of a training iteration (modules/trainers.py:306, 323-325) as ONE autograd node ---------------------------------
class _RenderPairFunction(torch.autograd.Function):
    """forward: rf_render_forward_pair (both saving renders of the same rays in one launch, each with its own jitter table and its own
    per-key record counters); backward: rf_bin_offsets_pair + rf_render_backward_emit_direct_pair + ONE rf_brick_accumulate over both
    record lists, which OVERWRITES the gradient tensors (the specular list covers every element).  Upstream gradients of the two
    colours only -- every reference use; anything else raises (use two calls of render_sh_voxel_grid_hip)."""

    @staticmethod
    def forward(ctx, densities, features, origins, directions, t_vals, t_rand0, t_rand1, voxel_grid, num_samples, near, far, flags, nkeys):
        lib, dev, n = _library(), origins.device, origins.shape[0]
        grid = _describe_grid(voxel_grid, densities, features)
        rays, outs, fl = (RFRayBatch * 2)(), (RFRenderOut * 2)(), (C.c_uint32 * 2)(flags & ~RF_FLAG_RENDER_DIFFUSE, flags | RF_FLAG_RENDER_DIFFUSE)
        results, saved = [], []
        for i, t_rand in enumerate((t_rand0, t_rand1)):
            rays[i] = RFRayBatch(origins.data_ptr(), directions.data_ptr(), n, num_samples, near, far, t_vals.data_ptr(), None if t_rand is None else t_rand.data_ptr(), 0, 0, None)
            colour = torch.empty((n, 3), dtype=torch.float32, device=dev)
            depth, acc, disparity = (torch.empty((n, 1), dtype=torch.float32, device=dev) for _ in range(3))
            caches = (torch.empty((n, num_samples, 4), dtype=torch.float32, device=dev), torch.empty((n, num_samples), dtype=torch.float32, device=dev),
                      torch.empty((n,), dtype=torch.int32, device=dev), torch.empty((n, (num_samples + 63) // 64), dtype=torch.int64, device=dev),
                      torch.zeros(nkeys, dtype=torch.int32, device=dev))
            outs[i] = RFRenderOut(colour.data_ptr(), depth.data_ptr(), acc.data_ptr(), disparity.data_ptr(), caches[0].data_ptr(), caches[1].data_ptr(), caches[2].data_ptr(),
                                  caches[3].data_ptr(), caches[4].data_ptr(), BRICK)
            results += [colour, depth, acc, disparity]
            saved += list(caches)
        fgrid, keep = _split_shadow(voxel_grid, densities, features, grid) if int(features.shape[-1]) in (3, 27) else (grid, None)
        _check(lib.rf_render_forward_pair(C.byref(fgrid), rays, fl, outs, torch.cuda.current_stream(dev).cuda_stream), "rf_render_forward_pair")
        del keep
        ctx.voxel_grid, ctx.args, ctx.has_rand = voxel_grid, (num_samples, near, far, int(fl[0]), int(fl[1])), (t_rand0 is not None, t_rand1 is not None)
        ctx.save_for_backward(densities, features, origins, directions, t_vals, *saved, *[t for t in (t_rand0, t_rand1) if t is not None])
        ctx.mark_non_differentiable(results[3], results[7])
        ctx.set_materialize_grads(False)
        return tuple(results)

    @staticmethod
    def backward(ctx, gc0, gd0, ga0, _gq0, gc1, gd1, ga1, _gq1):
        if gc0 is None or gc1 is None or any(g is not None for g in (gd0, ga0, gd1, ga1)):
            raise RuntimeError("render_sh_voxel_grid_pair_hip back-propagates the two colours only (the trainer's use); use two calls of render_sh_voxel_grid_hip")
        lib = _library()
        saved = list(ctx.saved_tensors)
        densities, features, origins, directions, t_vals = saved[:5]
        caches = [saved[5:10], saved[10:15]]
        rands = saved[15:]
        num_samples, near, far, fl0, fl1 = ctx.args
        dev, n = origins.device, origins.shape[0]
        grid = _describe_grid(ctx.voxel_grid, densities, features)
        nkeys = caches[0][4].numel()
        offsets = torch.empty((2, nkeys + 1), dtype=torch.int64, device=dev)
        cursor = torch.empty((2, nkeys), dtype=torch.int32, device=dev)
        records = [torch.empty((n * num_samples, int(lib.rf_expanded_record_floats(f))), dtype=torch.float32, device=dev) for f in (int(features.shape[-1]), 3)]
        g_colours = [g.detach().to(torch.float32).contiguous() for g in (gc0, gc1)]
        stream = torch.cuda.current_stream(dev).cuda_stream
        vp2 = C.c_void_p * 2
        _check(lib.rf_bin_offsets_pair(vp2(caches[0][4].data_ptr(), caches[1][4].data_ptr()), nkeys, vp2(offsets[0].data_ptr(), offsets[1].data_ptr()),
                                       vp2(cursor[0].data_ptr(), cursor[1].data_ptr()), stream), "rf_bin_offsets_pair")
        rays, passes, fl = (RFRayBatch * 2)(), (RFPassScratch * 2)(), (C.c_uint32 * 2)(fl0, fl1)
        for i in range(2):
            t_rand = rands.pop(0) if ctx.has_rand[i] else None
            rays[i] = RFRayBatch(origins.data_ptr(), directions.data_ptr(), n, num_samples, near, far, t_vals.data_ptr(), None if t_rand is None else t_rand.data_ptr(), 0, 0, None)
            c = caches[i]
            passes[i].out = RFRenderOut(None, None, None, None, c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr(), c[3].data_ptr(), c[4].data_ptr(), BRICK)
            passes[i].grad_colour_dev, passes[i].cursor_dev, passes[i].offsets_dev, passes[i].records_sorted_dev = (
                g_colours[i].data_ptr(), cursor[i].data_ptr(), offsets[i].data_ptr(), records[i].data_ptr())
        _check(lib.rf_render_backward_emit_direct_pair(C.byref(grid), rays, fl, passes, stream), "rf_render_backward_emit_direct_pair")
        grad_d, grad_f = torch.empty_like(densities), torch.empty_like(features)
        lists = (RFBrickList * 2)(RFBrickList(records[0].data_ptr(), offsets[0].data_ptr(), 0), RFBrickList(records[1].data_ptr(), offsets[1].data_ptr(), 1))
        _check(lib.rf_brick_accumulate(C.byref(grid), BRICK, lists, 2, grad_d.data_ptr(), grad_f.data_ptr(), 0, stream), "rf_brick_accumulate")
        return (grad_d, grad_f) + (None,) * 11


def render_sh_voxel_grid_pair_hip(voxel_grid, rays: Rays, render_config, parallel_points_chunk_size=None):
    """(render_sh_voxel_grid_hip(grid, rays, cfg), render_sh_voxel_grid_hip(grid, rays, cfg with render_diffuse=True)) -- the two renders
    of modules/trainers.py:306, 323-325, with the same torch.rand draws in the same order -- as ONE autograd node where the binned adjoint
    applies (a training batch on an SH degree >= 2 grid, a gradient will be asked for); otherwise simply the two calls."""
    import copy

    densities, features = voxel_grid.densities, voxel_grid.features
    n, s = int(rays.origins.shape[0]), int(render_config.num_samples_per_ray)
    need_grad = torch.is_grad_enabled() and (densities.requires_grad or features.requires_grad)
    nkeys = _use_binned_adjoint(features, n, s) if (need_grad and rays.origins.is_cuda) else 0
    if not nkeys:
        diffuse = copy.copy(render_config)
        diffuse.render_diffuse = True
        plain = copy.copy(render_config)
        plain.render_diffuse = False
        return render_sh_voxel_grid_hip(voxel_grid, rays, plain), render_sh_voxel_grid_hip(voxel_grid, rays, diffuse)
    if render_config.density2occupancy.__name__ != "density2occupancy_pb" or render_config.radiance_hdr_tone_map is not torch.sigmoid:
        raise ValueError("render_sh_voxel_grid_pair_hip: only density2occupancy_pb / torch.sigmoid are implemented")
    if render_config.stochastic_density_noise_std != 0.0:
        raise ValueError("render_sh_voxel_grid_pair_hip: stochastic_density_noise_std must be 0.0")
    origins = rays.origins.detach().to(torch.float32).contiguous()
    directions = rays.directions.detach().to(torch.float32).contiguous()
    assert origins.dim() == 2 and directions.shape == origins.shape, "the render interface only works with FLAT rays"
    if not (densities.is_cuda and densities.is_contiguous() and features.is_contiguous() and densities.dtype == features.dtype == torch.float32):
        raise RuntimeError("the VoxelGrid's tensors must be contiguous float32 on the HIP device")
    t_vals = torch.linspace(0.0, 1.0, s, dtype=torch.float32).to(origins.device)
    t_rands = [torch.rand(n, s, dtype=torch.float32, device=origins.device) if render_config.perturb_sampled_points else None for _ in range(2)]  # sample.py:63, twice
    flags = (RF_FLAG_WHITE_BKGD if render_config.white_bkgd else 0) | (RF_FLAG_AABB_SAMPLING if render_config.optimized_sampling else 0)
    near, far = float(np.float32(render_config.camera_bounds.near)), float(np.float32(render_config.camera_bounds.far))
    o = _RenderPairFunction.apply(densities, features, origins, directions, t_vals, t_rands[0], t_rands[1], voxel_grid, s, near, far, flags, nkeys)
    return tuple(RenderOut(colour=o[4 * i], depth=o[4 * i + 1], extra={EXTRA_DISPARITY: o[4 * i + 3], EXTRA_ACCUMULATED_WEIGHTS: o[4 * i + 2]}) for i in range(2))


# ---- whole frames (modules/volumetric_model.py:143-172) in ONE launch --------------------------------------------------------------
# The frame loop of VolumetricModel.render -- cast_rays, slices of parallel_rays_chunk_size, one procedure call per chunk, concatenate
# -- as one library call: rays and stratified jitter are generated inside the kernel (RFRayBatch.camera), and on frames whose 8 x 8
# pixel tiles stay within ~2 voxels the library renders RAY PACKETS (rf_frame_render_kernel says which kernel a frame gets).  The
# packet kernel gathers from a split-layout copy of the grid (base [X,Y,Z,4] = density + degree-0 coefficients, rest [X,Y,Z,F-3]):
# rf_convert_grid makes it, once per change of the grid's tensors (data pointers / in-place version counters).
import weakref

_FRAME_SHADOWS = weakref.WeakKeyDictionary()


SPLIT_SHADOW = os.environ.get("RELU_FIELD_HIP_SPLIT_SHADOW", "1") != "0"


def invalidate_split_shadow(voxel_grid) -> None:
    """Call after a write to the grid's tensors that bumps neither their data pointer nor their autograd version counter
    (``p.data.copy_(...)``, ``p.data.clamp_()``, a raw-pointer kernel): the next forward pass re-makes the split copy."""
    sh = _FRAME_SHADOWS.get(voxel_grid)
    if sh is not None:
        sh["stamp"] = None


def release_split_shadow(voxel_grid) -> None:
    """Free the split copy of the grid (as large as the grid itself); the next forward pass that wants one makes it again."""
    _FRAME_SHADOWS.pop(voxel_grid, None)


def _split_shadow(voxel_grid, densities, features, reference_grid):
    if not SPLIT_SHADOW:
        return reference_grid, None
    f = int(features.shape[-1])
    stamp = (densities.data_ptr(), features.data_ptr(), densities._version, features._version, tuple(features.shape))
    sh = _FRAME_SHADOWS.get(voxel_grid)
    if sh is None or sh["shape"] != tuple(features.shape) or sh["base"].device != features.device:
        dims = tuple(features.shape[:3])
        sh = {"shape": tuple(features.shape), "stamp": None, "base": torch.empty(dims + (4,), dtype=torch.float32, device=features.device),
              "rest": torch.empty(dims + (f - 3,), dtype=torch.float32, device=features.device) if f > 3 else None}
        _FRAME_SHADOWS[voxel_grid] = sh
    g = RFGrid()
    C.memmove(C.byref(g), C.byref(reference_grid), C.sizeof(RFGrid))
    g.densities_dev, g.features_dev = sh["base"].data_ptr(), None if sh["rest"] is None else sh["rest"].data_ptr()
    g.density_stride, g.feature_stride, g.layout = 4, f - 3, RF_LAYOUT_SPLIT
    if sh["stamp"] != stamp:
        lib = _library()
        lib.rf_convert_grid.argtypes = [C.POINTER(RFGrid), C.POINTER(RFGrid), C.c_void_p]
        _check(lib.rf_convert_grid(C.byref(reference_grid), C.byref(g), torch.cuda.current_stream(features.device).cuda_stream), "rf_convert_grid")
        sh["stamp"] = stamp
    return g, sh


def render_frame_hip(voxel_grid, camera_intrinsics, camera_pose, render_config, first_ray=0, num_rays=None) -> RenderOut:
    """The pixels [first_ray, first_ray + num_rays) (row-major; default: the whole frame) of a posed camera in one launch: what
    VolumetricModel.render (modules/volumetric_model.py:143-172) computes chunk by chunk.  ``camera_intrinsics`` = (height, width,
    focal), ``camera_pose`` = (rotation [3,3], translation [3,1]) like utils/imaging_utils.py:17-30.  Inference only.  With
    ``perturb_sampled_points`` the stratified jitter is drawn inside the kernel from a per-call key taken from torch's CPU generator
    (the law of sample.py:63's torch.rand(N, S), no tensor).  A whole frame comes back as [H, W, .] tensors, a pixel range flat."""
    if render_config.density2occupancy.__name__ != "density2occupancy_pb" or render_config.radiance_hdr_tone_map is not torch.sigmoid:
        raise ValueError("render_frame_hip: only density2occupancy_pb / torch.sigmoid are implemented")
    if render_config.stochastic_density_noise_std != 0.0:
        raise ValueError("render_frame_hip: stochastic_density_noise_std must be 0.0")
    densities, features = voxel_grid.densities.detach(), voxel_grid.features.detach()
    if not (densities.is_cuda and densities.is_contiguous() and features.is_contiguous() and densities.dtype == features.dtype == torch.float32):
        raise RuntimeError("the VoxelGrid's tensors must be contiguous float32 on the HIP device")
    lib, dev = _library(), features.device
    height, width, focal = (int(camera_intrinsics[0]), int(camera_intrinsics[1]), float(np.float32(camera_intrinsics[2])))
    n = height * width - int(first_ray) if num_rays is None else int(num_rays)
    cam = RFCamera()
    cam.height, cam.width, cam.focal = height, width, focal
    rot = torch.as_tensor(camera_pose[0]).detach().to("cpu", torch.float32).reshape(3, 3)
    trans = torch.as_tensor(camera_pose[1]).detach().to("cpu", torch.float32).reshape(3)
    for i in range(3):
        for j in range(3):
            cam.pose[4 * i + j] = float(rot[i, j])
        cam.pose[4 * i + 3] = float(trans[i])
    grid = _describe_grid(voxel_grid, densities, features)
    keep = None
    if int(features.shape[-1]) in (3, 12, 27, 48):  # (the layout the packet kernel gathers from: SH degree 0 .. 3)
        grid, keep = _split_shadow(voxel_grid, densities, features, grid)
    s = int(render_config.num_samples_per_ray)
    t_vals = torch.linspace(0.0, 1.0, s, dtype=torch.float32).to(dev)
    flags = (RF_FLAG_WHITE_BKGD if render_config.white_bkgd else 0) | (RF_FLAG_RENDER_DIFFUSE if render_config.render_diffuse else 0) | (
        RF_FLAG_AABB_SAMPLING if render_config.optimized_sampling else 0)
    rb = RFRayBatch(None, None, n, s, float(np.float32(render_config.camera_bounds.near)), float(np.float32(render_config.camera_bounds.far)),
                    t_vals.data_ptr(), None, 0, int(first_ray), C.cast(C.pointer(cam), C.c_void_p))
    if render_config.perturb_sampled_points:
        rb.jitter_key = int(torch.randint(-(2**63), 2**63 - 1, (1,), dtype=torch.int64).item()) & 0xFFFFFFFFFFFFFFFF
        flags |= RF_FLAG_JITTER_KEYED
    colour = torch.empty((n, 3), dtype=torch.float32, device=dev)
    depth, acc, disparity = (torch.empty((n, 1), dtype=torch.float32, device=dev) for _ in range(3))
    out = RFRenderOut(colour.data_ptr(), depth.data_ptr(), acc.data_ptr(), disparity.data_ptr(), None, None, None, None, None, 0)
    _check(lib.rf_render_forward(C.byref(grid), C.byref(rb), flags, C.byref(out), torch.cuda.current_stream(dev).cuda_stream), "rf_render_forward")
    del keep
    if int(first_ray) == 0 and n == height * width:
        colour, depth, acc, disparity = (t.reshape(height, width, -1) for t in (colour, depth, acc, disparity))
    return RenderOut(colour=colour, depth=depth, extra={EXTRA_DISPARITY: disparity, EXTRA_ACCUMULATED_WEIGHTS: acc})
