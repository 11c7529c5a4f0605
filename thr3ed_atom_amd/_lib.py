"""ctypes binding of the C ABI in include/relu_field.h (thr3ed_atom_amd/csrc/librelu_field_hip.so).

The library is the product: there is NO fallback.  ``load()`` raises when the shared object is
missing and every wrapper raises ``RuntimeError`` on a non-zero return code.
"""
import ctypes as C
import os
import subprocess
from typing import Optional

CSRC_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
INCLUDE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
LIB_NAME = "librelu_field_hip.so"
LIB_PATH = os.environ.get("RF_LIB_PATH") or os.path.join(CSRC_DIR, LIB_NAME)  # RF_LIB_PATH: development builds (tools/)
SOURCES = ["relu_field_kernels.hip"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]

ABI_VERSION = 4  # RF_ABI_VERSION of include/relu_field.h (4: compact records + compacted sample cache, x-slab-major keys, brick ranges)

# enums of relu_field.h
DENSITY_MODES = {"relu": 0, "softplus": 1, "abs": 2, "identity": 3}
LAYOUTS = {"reference": 0, "split": 1, "bricked": 2}
FLAG_WHITE_BKGD = 1
FLAG_RENDER_DIFFUSE = 2
FLAG_AABB_SAMPLING = 4
FLAG_OCCUPANCY_SKIP = 8
FLAG_JITTER_KEYED = 16
ERR_UNSUPPORTED = -3  # RF_ERR_UNSUPPORTED
STEP_FORWARD = 1
STEP_EMIT = 2
STEP_BRICKS = 4
STEP_EMIT_SPECULAR = 8
STEP_EMIT_DIFFUSE = 16
STEP_SELECT_AND_DIFFUSE_FORWARD = 32
STEP_SPECULAR_FORWARD_AND_LOSSES = 64
STEP_DIFFUSE_CHAIN = 128  # selection + render_diffuse forward + its loss/offsets + its adjoint (reads the base tensor only)
STEP_SPECULAR_FORWARD = 256  # specular forward + its loss/offsets

EXPORTED_SYMBOLS = [
    "rf_abi_version",
    "rf_abi_struct_size",
    "rf_error_string",
    "rf_cast_rays",
    "rf_cast_selected_rays",
    "rf_select_rays_and_pixels",
    "rf_ray_aabb_bounds",
    "rf_render_forward",
    "rf_frame_render_kernel",
    "rf_render_backward",
    "rf_render_backward_emit",
    "rf_expand_records",
    "rf_expanded_record_floats",
    "rf_bin_offsets",
    "rf_render_backward_emit_direct",
    "rf_scatter_records",
    "rf_brick_accumulate",
    "rf_brick_accumulate_adam",
    "rf_brick_accumulate_adam_mirror",
    "rf_brick_accumulate_adam_range",
    "rf_brick_accumulate_adam_split",
    "rf_brick_split_scratch_bytes",
    "rf_grid_query",
    "rf_grid_query_backward",
    "rf_build_occupancy",
    "rf_upsample_grid",
    "rf_convert_grid",
    "rf_l1_loss_grad",
    "rf_adam_step",
    "rf_train_step",
    "rf_render_forward_pair",
    "rf_l1_loss_grad_pair",
    "rf_bin_offsets_pair",
    "rf_render_backward_emit_direct_pair",
]


class RFGrid(C.Structure):
    _fields_ = [
        ("densities_dev", C.c_void_p),
        ("features_dev", C.c_void_p),
        ("dims", C.c_int32 * 3),
        ("num_features", C.c_int32),
        ("density_stride", C.c_int64),
        ("feature_stride", C.c_int64),
        ("aabb_min", C.c_float * 3),
        ("aabb_max", C.c_float * 3),
        ("norm_scale", C.c_float * 3),
        ("norm_bias", C.c_float * 3),
        ("density_scale", C.c_float),
        ("density_mode", C.c_int32),
        ("layout", C.c_int32),
        ("occupancy_dev", C.c_void_p),
    ]


class RFCamera(C.Structure):
    _fields_ = [("height", C.c_int32), ("width", C.c_int32), ("focal", C.c_float), ("pose", C.c_float * 12)]


class RFRayBatch(C.Structure):
    _fields_ = [
        ("origins_dev", C.c_void_p),
        ("directions_dev", C.c_void_p),
        ("num_rays", C.c_int64),
        ("num_samples", C.c_int32),
        ("near", C.c_float),
        ("far", C.c_float),
        ("t_vals_dev", C.c_void_p),
        ("t_rand_dev", C.c_void_p),
        ("jitter_key", C.c_uint64),
        ("first_ray", C.c_int64),
        ("camera", C.POINTER(RFCamera)),
    ]


class RFRenderOut(C.Structure):
    _fields_ = [
        ("colour_dev", C.c_void_p),
        ("depth_dev", C.c_void_p),
        ("acc_dev", C.c_void_p),
        ("disparity_dev", C.c_void_p),
        ("sample_cache_dev", C.c_void_p),
        ("trans_cache_dev", C.c_void_p),
        ("stop_cache_dev", C.c_void_p),
        ("chunk_mask_dev", C.c_void_p),
        ("key_hist_dev", C.c_void_p),
        ("brick_size", C.c_int32),
    ]


class RFRenderGrads(C.Structure):
    _fields_ = [
        ("grad_colour_dev", C.c_void_p),
        ("grad_depth_dev", C.c_void_p),
        ("grad_acc_dev", C.c_void_p),
    ]


class RFBrickList(C.Structure):
    _fields_ = [
        ("records_sorted_dev", C.c_void_p),
        ("offsets_dev", C.c_void_p),
        ("render_diffuse", C.c_int32),
    ]


class RFAdamState(C.Structure):
    _fields_ = [
        ("param_first_dev", C.c_void_p),
        ("param_second_dev", C.c_void_p),
        ("exp_avg_first_dev", C.c_void_p),
        ("exp_avg_second_dev", C.c_void_p),
        ("exp_avg_sq_first_dev", C.c_void_p),
        ("exp_avg_sq_second_dev", C.c_void_p),
        ("lr", C.c_float),
        ("beta1", C.c_float),
        ("beta2", C.c_float),
        ("eps", C.c_float),
        ("step", C.c_int32),
    ]


class RFRaySelection(C.Structure):
    _fields_ = [
        ("height", C.c_int32),
        ("width", C.c_int32),
        ("focal", C.c_float),
        ("poses_dev", C.c_void_p),
        ("image_ids_dev", C.c_void_p),
        ("num_batch_images", C.c_int32),
        ("pixel_table_dev", C.c_void_p),
        ("key", C.c_uint64),
        ("first_index", C.c_int64),
    ]


class RFPassScratch(C.Structure):
    _fields_ = [
        ("out", RFRenderOut),
        ("grad_colour_dev", C.c_void_p),
        ("cursor_dev", C.c_void_p),
        ("offsets_dev", C.c_void_p),
        ("records_sorted_dev", C.c_void_p),
        ("t_rand_dev", C.c_void_p),
        ("jitter_key", C.c_uint64),
    ]


class RFTrainStep(C.Structure):
    _fields_ = [
        ("select", C.POINTER(RFRaySelection)),
        ("origins_dev", C.c_void_p),
        ("directions_dev", C.c_void_p),
        ("pixels_dev", C.c_void_p),
        ("num_rays", C.c_int64),
        ("num_samples", C.c_int32),
        ("near", C.c_float),
        ("far", C.c_float),
        ("t_vals_dev", C.c_void_p),
        ("flags", C.c_uint32),
        ("pass_", RFPassScratch * 2),
        ("loss_sums_dev", C.c_void_p),
        ("adam", C.POINTER(RFAdamState)),
        ("grad_first_dev", C.c_void_p),
        ("grad_second_dev", C.c_void_p),
        ("first_ray", C.c_int64),
        ("phases", C.c_uint32),
        ("loss_scale", C.c_float),
        ("timing_events", C.POINTER(C.c_void_p)),
    ]


TRAIN_STEP_EVENTS = 11
def train_step_pairing():
    """(forward renders paired, emit launches paired): rf_train_step runs both renders of an iteration -- and both adjoints -- in ONE
    launch each unless $RF_FWD_PAIR / $RF_EMIT_PAIR say 0 (the library reads the same variables); the first slot of a pair then holds
    the launch and the second one is empty."""
    return os.environ.get("RF_FWD_PAIR", "1") != "0", os.environ.get("RF_EMIT_PAIR", "1") != "0"


TRAIN_STEP_EVENT_NAMES = ["select_rays_and_pixels", "render_forward[spec,save]", "(no launch: loss slot of the first render)", "render_forward[diffuse,save]",
                          "l1_loss_grad+bin_offsets[both]", "(no launch: offsets slot)", "render_backward_emit_direct[spec]",
                          "(no launch: offsets slot of the second list)", "render_backward_emit_direct[diffuse]", "brick_accumulate"]


# order of rf_abi_struct_size(which)
ABI_STRUCTS = [RFGrid, RFRayBatch, RFRenderOut, RFRenderGrads, RFBrickList, RFAdamState, RFCamera, RFRaySelection, RFPassScratch, RFTrainStep]


def build(force: bool = False, verbose: bool = False) -> str:
    """Cross-compile the HIP sources for gfx950 into csrc/librelu_field_hip.so (hipcc needs no GPU)."""
    srcs = [os.path.join(CSRC_DIR, s) for s in SOURCES]
    deps = srcs + [os.path.join(INCLUDE_DIR, "relu_field.h")]
    if not force and os.path.exists(LIB_PATH):
        if os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(d) for d in deps):
            return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + HIPCC_FLAGS + ["-I", INCLUDE_DIR] + srcs + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


_LIB: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen the HIP library; raises (never falls back) when it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP render library has not been built. "
            f"Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            f"There is no CPU fallback for the render path."
        )
    lib = C.CDLL(LIB_PATH)
    lib.rf_abi_version.restype = C.c_int
    lib.rf_error_string.restype = C.c_char_p
    lib.rf_error_string.argtypes = [C.c_int]
    vp, i32, i64, f32, u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint32
    fp = C.POINTER(C.c_float)
    lib.rf_cast_rays.argtypes = [i32, i32, f32, fp, fp, vp, vp, vp]
    lib.rf_cast_selected_rays.argtypes = [i32, i32, f32, vp, i32, vp, i64, vp, vp, vp]
    lib.rf_select_rays_and_pixels.argtypes = [i32, i32, f32, vp, vp, i32, vp, C.c_uint64, i64, i64, vp, vp, vp, vp, vp]
    lib.rf_ray_aabb_bounds.argtypes = [vp, vp, i64, f32, f32, fp, fp, vp, vp, vp]
    lib.rf_render_forward.argtypes = [C.POINTER(RFGrid), C.POINTER(RFRayBatch), u32, C.POINTER(RFRenderOut), vp]
    lib.rf_frame_render_kernel.argtypes = [C.POINTER(RFGrid), C.POINTER(RFCamera), u32]
    lib.rf_render_backward.argtypes = [
        C.POINTER(RFGrid),
        C.POINTER(RFRayBatch),
        u32,
        C.POINTER(RFRenderOut),
        C.POINTER(RFRenderGrads),
        vp,
        vp,
        vp,
    ]
    lib.rf_render_backward_emit.argtypes = [
        C.POINTER(RFGrid), C.POINTER(RFRayBatch), u32, C.POINTER(RFRenderOut), C.POINTER(RFRenderGrads), i32, vp, vp, vp, vp,
    ]
    lib.rf_expand_records.argtypes = [C.POINTER(RFGrid), vp, vp, vp, i64, i32, vp, vp]
    lib.rf_expanded_record_floats.argtypes = [i32]
    lib.rf_expanded_record_floats.restype = i32
    lib.rf_bin_offsets.argtypes = [vp, i32, vp, vp, vp]
    lib.rf_render_backward_emit_direct.argtypes = [
        C.POINTER(RFGrid), C.POINTER(RFRayBatch), u32, C.POINTER(RFRenderOut), C.POINTER(RFRenderGrads), i32, vp, vp, vp, vp,
    ]
    lib.rf_scatter_records.argtypes = [C.POINTER(RFGrid), vp, vp, i64, vp, i32, vp, vp, i32, vp]
    lib.rf_brick_accumulate.argtypes = [C.POINTER(RFGrid), i32, C.POINTER(RFBrickList), i32, vp, vp, i32, vp]
    lib.rf_brick_accumulate_adam.argtypes = [C.POINTER(RFGrid), i32, C.POINTER(RFBrickList), i32, C.POINTER(RFAdamState), vp]
    lib.rf_brick_accumulate_adam_range.argtypes = [C.POINTER(RFGrid), i32, C.POINTER(RFBrickList), i32, C.POINTER(RFAdamState), i32, i32, vp]
    lib.rf_brick_accumulate_adam_mirror.argtypes = [C.POINTER(RFGrid), i32, C.POINTER(RFBrickList), i32, C.POINTER(RFAdamState), vp, vp, vp]
    lib.rf_brick_accumulate_adam_split.argtypes = [C.POINTER(RFGrid), i32, C.POINTER(RFBrickList), i32, C.POINTER(RFAdamState), i32, i32, i32, vp, i64, vp]
    lib.rf_brick_split_scratch_bytes.argtypes = [C.POINTER(RFGrid), i32, i32]
    lib.rf_train_step.argtypes = [C.POINTER(RFGrid), C.POINTER(RFTrainStep), vp]
    lib.rf_grid_query.argtypes = [C.POINTER(RFGrid), vp, i64, vp, vp]
    lib.rf_grid_query_backward.argtypes = [C.POINTER(RFGrid), vp, i64, vp, vp, vp, vp]
    lib.rf_upsample_grid.argtypes = [C.POINTER(RFGrid), C.POINTER(RFGrid), vp]
    lib.rf_convert_grid.argtypes = [C.POINTER(RFGrid), C.POINTER(RFGrid), vp]
    lib.rf_build_occupancy.argtypes = [C.POINTER(RFGrid), f32, vp, vp]
    lib.rf_l1_loss_grad.argtypes = [vp, vp, i64, f32, vp, vp, vp]
    lib.rf_adam_step.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32, i32, i32, vp]
    # the paired entry points: every array argument holds two entries ([0] specular, [1] render_diffuse)
    lib.rf_render_forward_pair.argtypes = [C.POINTER(RFGrid), C.POINTER(RFRayBatch), C.POINTER(u32), C.POINTER(RFRenderOut), vp]
    lib.rf_l1_loss_grad_pair.argtypes = [C.POINTER(vp), vp, i64, f32, C.POINTER(vp), vp, vp, vp]
    lib.rf_bin_offsets_pair.argtypes = [C.POINTER(vp), i32, C.POINTER(vp), C.POINTER(vp), vp]
    lib.rf_render_backward_emit_direct_pair.argtypes = [C.POINTER(RFGrid), C.POINTER(RFRayBatch), C.POINTER(u32), C.POINTER(RFPassScratch), vp]
    for name in EXPORTED_SYMBOLS:
        if name not in ("rf_error_string",):
            getattr(lib, name).restype = C.c_int64 if name == "rf_brick_split_scratch_bytes" else C.c_int
    if lib.rf_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH}: ABI version {lib.rf_abi_version()} != {ABI_VERSION} (stale build? run __graft_entry__.build())")
    lib.rf_abi_struct_size.argtypes = [C.c_int]
    for which, mirror in enumerate(ABI_STRUCTS):
        if lib.rf_abi_struct_size(which) != C.sizeof(mirror):
            raise RuntimeError(f"{LIB_PATH}: sizeof({mirror.__name__}) is {lib.rf_abi_struct_size(which)} in the library, {C.sizeof(mirror)} in the binding")
    _LIB = lib
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().rf_error_string(code).decode()
        raise RuntimeError(f"{what} failed: {msg} (code {code})")


def float3(values):
    return (C.c_float * 3)(*[float(v) for v in values])
