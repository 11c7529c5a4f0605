"""Dense SH + density voxel grid (the ReLU Field) -- host-side mirror of the reference's
thre3d_atom/thre3d_reprs/voxels.py (VoxelGrid :46-331, scale_voxel_grid_with_required_output_size
:334-373, create_voxel_grid_from_saved_info_dict :376-383).

Same constructor arguments, properties, state_dict keys (``_densities`` / ``_features``) and config
dictionaries as the reference, so trainers and checkpoints written against it keep working.  The
arithmetic itself (normalise -> trilinear gather -> ReLU) lives in the HIP kernels
(csrc/relu_field_kernels.hip); this class only owns the tensors and describes them to the C ABI.
"""
from typing import Any, Callable, Dict, NamedTuple, Optional, Tuple

import ctypes as C
import weakref

import numpy as np
import torch
from torch import Tensor
from torch.nn import Module

from . import _lib
from .camera import slack_range_map
from .constants import CONFIG_DICT, STATE_DICT, THRE3D_REPR, u_DENSITIES, u_FEATURES


class VoxelSize(NamedTuple):
    x_size: float = 1.0
    y_size: float = 1.0
    z_size: float = 1.0


class VoxelGridLocation(NamedTuple):
    x_coord: float = 0.0
    y_coord: float = 0.0
    z_coord: float = 0.0


class AxisAlignedBoundingBox(NamedTuple):
    x_range: Tuple[float, float]
    y_range: Tuple[float, float]
    z_range: Tuple[float, float]


STORAGES = ("reference", "split", "bricked")
BRICK = 8  # edge of a storage brick in nodes (RF_LAYOUT_BRICKED)


def pack_split(densities: Tensor, features: Tensor):
    """reference tensors -> (base [..., 4] = (density, sh0 r, g, b), rest [..., 3(K-1)] or None)."""
    K = features.shape[-1] // 3
    f = features.unflatten(-1, (3, K))
    base = torch.cat([densities, f[..., 0]], dim=-1).contiguous()
    rest = f[..., 1:].reshape(*features.shape[:-1], 3 * (K - 1)).contiguous() if K > 1 else None
    return base, rest


def unpack_split(base: Tensor, rest: Optional[Tensor]):
    """(base, rest) -> reference tensors (densities [..., 1], features [..., 3K], index = colour*K + k)."""
    densities = base[..., :1].contiguous()
    sh0 = base[..., 1:4]
    if rest is None:
        return densities, sh0.contiguous()
    K = rest.shape[-1] // 3 + 1
    f = torch.cat([sh0[..., None], rest.unflatten(-1, (3, K - 1))], dim=-1)
    return densities, f.flatten(-2).contiguous()


def brick_nodes(t: Tensor) -> Tensor:
    """[X,Y,Z,C] -> brick-major [NBX,NBY,NBZ,8,8,8,C] (dims padded with zeros to multiples of 8)."""
    X, Y, Z, C = t.shape
    px, py, pz = (-X) % BRICK, (-Y) % BRICK, (-Z) % BRICK
    if px or py or pz:
        t = torch.nn.functional.pad(t, (0, 0, 0, pz, 0, py, 0, px))
    nbx, nby, nbz = (X + px) // BRICK, (Y + py) // BRICK, (Z + pz) // BRICK
    return t.reshape(nbx, BRICK, nby, BRICK, nbz, BRICK, C).permute(0, 2, 4, 1, 3, 5, 6).contiguous()


def unbrick_nodes(t: Tensor, dims) -> Tensor:
    """brick-major [NBX,NBY,NBZ,8,8,8,C] -> [X,Y,Z,C] (padding dropped)."""
    nbx, nby, nbz, _, _, _, C = t.shape
    full = t.permute(0, 3, 1, 4, 2, 5, 6).reshape(nbx * BRICK, nby * BRICK, nbz * BRICK, C)
    return full[: dims[0], : dims[1], : dims[2]].contiguous()


def pack_storage(densities: Tensor, features: Tensor, storage: str):
    """reference tensors -> the two tensors of ``storage`` ("split" or "bricked")."""
    base, rest = pack_split(densities, features)
    if storage == "bricked":
        base, rest = brick_nodes(base), (None if rest is None else brick_nodes(rest))
    return base, rest


def unpack_storage(first: Tensor, second: Optional[Tensor], storage: str, dims):
    """the two tensors of ``storage`` (parameters or gradients) -> reference tensors (densities, features)."""
    if storage == "reference":
        return first, second
    if storage == "bricked":
        first, second = unbrick_nodes(first, dims), (None if second is None else unbrick_nodes(second, dims))
    return unpack_split(first, second)


def _is_identity(fn) -> bool:
    return fn is None or isinstance(fn, torch.nn.Identity)


def resolve_density_mode(pre: Callable, post: Callable) -> str:
    """Map the reference's (pre-activation, post-activation) callables onto the kernel's density mode.
    Supported pairs are the three the reference's trainer script builds
    (train_sh_based_voxel_grid_with_posed_images.py:169-192) plus plain identity."""
    pre_abs = pre is torch.abs or getattr(pre, "__name__", "") in ("abs", "absolute")
    if _is_identity(pre):
        if isinstance(post, torch.nn.ReLU) or post is torch.relu or post is torch.nn.functional.relu:
            return "relu"
        if isinstance(post, torch.nn.Softplus):
            if post.beta != 1 or post.threshold != 20:
                raise ValueError("only Softplus(beta=1, threshold=20) is supported by the HIP kernels")
            return "softplus"
        if _is_identity(post):
            return "identity"
    elif pre_abs and _is_identity(post):
        return "abs"
    raise ValueError(
        f"unsupported density activation pair (pre={pre}, post={post}); supported: "
        f"Identity/ReLU, Identity/Softplus, torch.abs/Identity, Identity/Identity"
    )


# per-object caches of the operators (ctypes descriptors, foreign-grid views), weakly keyed by the grid object so that nothing
# un-picklable ever lives in a module's __dict__ (copy.deepcopy(module) / torch.save(module) keep working after a HIP render)
_RF_GRID_CACHE: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()
_RF_VIEW_CACHE: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()
_RF_SHADOW_CACHE: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()
# data-parallel training: parameter all-gathers still in flight on the communication stream (trainers.TrainStepper._owner_step), a list
# of callables per grid that make the CURRENT stream wait for them.  Kept outside the module like the caches above (an RCCL work
# handle / a HIP event in a module's __dict__ would break copy.deepcopy and torch.save of it).
_RF_PENDING_PARAMETERS: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()
# forward passes of reference-storage grids gather from a split-layout shadow (KernelGridInterface.forward_rf_grid): a second copy
# of the grid, so only up to SHADOW_MAX_BYTES of parameters (512^3 at SH degree 2 would be 15 GB more); $RF_SPLIT_SHADOW=0 turns it off
import os as _os

SPLIT_SHADOW = _os.environ.get("RF_SPLIT_SHADOW", "1") != "0"
SHADOW_MAX_BYTES = int(_os.environ.get("RF_SHADOW_MAX_BYTES", str(2 << 30)))


def shadow_allowed(grid) -> bool:
    nodes = 1
    for dim in grid.grid_dims:
        nodes *= int(dim)
    return SPLIT_SHADOW and nodes * (int(grid._num_features) + 1) * 4 <= SHADOW_MAX_BYTES


class KernelGridInterface:
    """What the render operators need from a grid object -- described to the C ABI (RFGrid) from a handful of attributes:
    ``kernel_tensors()``, ``grid_dims``, ``_aabb``, ``_expected_density_scale``, ``density_mode``, ``storage``, ``_num_features``.
    ``VoxelGrid`` (below) provides them from its own state; ``ForeignVoxelGridView`` reads them live from ANY module with the
    reference VoxelGrid's attribute names (thre3d_atom/thre3d_reprs/voxels.py:47-124,187-212), which is how the reference's own
    grid object renders through the HIP procedure."""

    _occupancy: Optional[Tensor] = None

    @property
    def sh_degree(self) -> int:
        return int(np.sqrt(self._num_features // 3)) - 1

    def wait_for_parameters(self) -> None:
        """Data-parallel training leaves all-gathers of the updated parameters in flight across the iteration boundary
        (trainers.TrainStepper._owner_step): whoever reads the grid next -- an operator's descriptor, ``kernel_tensors()``,
        ``.densities`` / ``.features``, ``state_dict()``, ``parameters()``, ``Module.to()`` -- makes its stream wait for them here.
        A no-op otherwise."""
        if _RF_PENDING_PARAMETERS:
            pending = _RF_PENDING_PARAMETERS.pop(self, None)
            if pending:
                for wait in pending:
                    wait()

    def defer_parameter_wait(self, wait) -> None:
        """(trainer) ``wait()`` makes the calling stream wait for parameters that are still arriving"""
        _RF_PENDING_PARAMETERS.setdefault(self, []).append(wait)

    def take_parameter_waits(self) -> list:
        """(trainer) the pending waits, removed: the caller orders its own launches against them piece by piece"""
        return _RF_PENDING_PARAMETERS.pop(self, None) or []

    def to_rf_grid(self, use_occupancy: bool = False, wait_parameters: bool = True) -> "_lib.RFGrid":
        if wait_parameters:
            self.wait_for_parameters()
        d, f = self._tensors()
        for t in (d, f):
            if t is None:
                continue
            if not t.is_cuda:
                raise RuntimeError(
                    "the ReLU-field render path runs on the GPU only: move the VoxelGrid to a HIP device "
                    "(there is no CPU fallback)"
                )
            if not (t.is_contiguous() and t.dtype == torch.float32):
                raise RuntimeError("grid tensors must be contiguous float32")
        # the descriptor is rebuilt only when something it describes changed (it is on the per-launch host path)
        occ_ptr = self._occupancy.data_ptr() if (use_occupancy and self._occupancy is not None) else None
        mode = self.density_mode
        if mode is None or not (_is_identity(getattr(self, "_feature_preactivation", None)) and _is_identity(getattr(self, "_feature_postactivation", None))):
            raise ValueError("the fused HIP kernels implement Identity/ReLU, Identity/Softplus, torch.abs/Identity and Identity/Identity density activations "
                             "with identity feature activations; this grid's callables need the composed path (thr3ed_atom_amd.composable)")
        key = (d.data_ptr(), None if f is None else f.data_ptr(), occ_ptr, self._aabb, self._expected_density_scale, mode, tuple(d.shape))
        # (kept OUTSIDE the object: a ctypes struct with pointers in a module's __dict__ breaks copy.deepcopy / torch.save of it)
        cached = _RF_GRID_CACHE.get(self)
        if cached is not None and cached[0] == key:
            return cached[1]
        g = _lib.RFGrid()
        g.densities_dev = d.data_ptr()
        g.features_dev = None if f is None else f.data_ptr()
        for a in range(3):
            g.dims[a] = self.grid_dims[a]
            lo, hi = self._aabb[a]
            g.aabb_min[a] = float(np.float32(lo))
            g.aabb_max[a] = float(np.float32(hi))
            scale, bias = slack_range_map((lo, hi))
            g.norm_scale[a] = float(scale)
            g.norm_bias[a] = float(bias)
        g.num_features = self._num_features
        g.density_stride = int(d.shape[-1])
        g.feature_stride = 0 if f is None else int(f.shape[-1])
        g.layout = _lib.LAYOUTS[self.storage]
        g.density_scale = float(self._expected_density_scale)
        g.density_mode = _lib.DENSITY_MODES[mode]
        g.occupancy_dev = occ_ptr
        _RF_GRID_CACHE[self] = (key, g)
        return g

    def forward_rf_grid(self, use_occupancy: bool = False) -> "_lib.RFGrid":
        """The descriptor the FORWARD passes should gather from.  A grid held in the reference's two tensors (one corner = 108
        unaligned feature bytes + 4 bytes in another tensor) is rendered from a split-layout SHADOW instead -- base [X,Y,Z,4] +
        rest [X,Y,Z,F-3], refreshed by one re-layout launch (rf_convert_grid) whenever the Parameters changed: their data pointers,
        their in-place version counters, or invalidate_occupancy() (what the fused optimizer calls after writing through raw
        pointers; edits through ``.data`` bypass the version counter: call invalidate_occupancy() after them).  Measured on the
        128^3 / SH-2 training step: diffuse forward 0.288 -> 0.084 ms for a 0.13 ms refresh per optimizer step.  Adjoints are
        unaffected: they produce gradients in the layout of the Parameters."""
        if self.storage != "reference" or not shadow_allowed(self):
            return self.to_rf_grid(use_occupancy)
        return self._shadow(use_occupancy, refresh=True)[1]

    def release_shadow(self) -> None:
        """Free the split-layout shadow of a reference-storage grid (it is rebuilt by the next render that wants it)."""
        _RF_SHADOW_CACHE.pop(self, None)

    def _shadow_stamp(self):
        d, f = self.kernel_tensors()
        return (d.data_ptr(), f.data_ptr(), d._version, f._version, self.__dict__.get("_shadow_epoch", 0), tuple(d.shape), tuple(f.shape))

    def _shadow(self, use_occupancy: bool = False, refresh: bool = True):
        """(shadow tensors dict, RFGrid describing them); ``refresh``: re-layout from the Parameters when they changed."""
        d, f = self.kernel_tensors()
        real = self.to_rf_grid(use_occupancy)
        sh = _RF_SHADOW_CACHE.get(self)
        if sh is None or sh["shape"] != (tuple(d.shape), tuple(f.shape)) or sh["device"] != d.device:
            X, Y, Z = d.shape[:3]
            F = f.shape[-1]
            sh = {"shape": (tuple(d.shape), tuple(f.shape)), "device": d.device, "stamp": None,
                  "base": torch.empty((X, Y, Z, 4), dtype=torch.float32, device=d.device),
                  "rest": torch.empty((X, Y, Z, F - 3), dtype=torch.float32, device=d.device) if F > 3 else None}
            _RF_SHADOW_CACHE[self] = sh
        g = _lib.RFGrid()
        C.memmove(C.byref(g), C.byref(real), C.sizeof(_lib.RFGrid))
        g.densities_dev = sh["base"].data_ptr()
        g.features_dev = None if sh["rest"] is None else sh["rest"].data_ptr()
        g.density_stride, g.feature_stride = 4, 0 if sh["rest"] is None else int(sh["rest"].shape[-1])
        g.layout = _lib.LAYOUTS["split"]
        if refresh:
            stamp = self._shadow_stamp()
            if sh["stamp"] != stamp:
                _lib.check(_lib.load().rf_convert_grid(C.byref(real), C.byref(g), torch.cuda.current_stream(d.device).cuda_stream), "rf_convert_grid")
                sh["stamp"] = stamp
        return sh, g

    def adopt_shadow(self, relayout: bool = True) -> None:
        """The split shadow is the newer copy (an optimizer updated it in place): re-layout it into the Parameters (raw-pointer
        write: their version counters do not move) and mark the two as in sync.  ``relayout=False``: the optimizer pass has written
        the Parameters itself (rf_brick_accumulate_adam_mirror) -- only the bookkeeping is left."""
        d, _ = self.kernel_tensors()
        sh, g = self._shadow(refresh=False)
        if relayout:
            real = self.to_rf_grid()
            _lib.check(_lib.load().rf_convert_grid(C.byref(g), C.byref(real), torch.cuda.current_stream(d.device).cuda_stream), "rf_convert_grid")
        self.invalidate_occupancy()  # the densities changed
        sh["stamp"] = self._shadow_stamp()

    def build_occupancy(self, threshold: float = 0.0) -> Tensor:
        """(Re)build the exact empty-cell bit mask used by RF_FLAG_OCCUPANCY_SKIP.  Must be called again
        whenever the densities change."""
        X, Y, Z = self.grid_dims
        ncell = (X + 1) * (Y + 1) * (Z + 1)
        dev = self.kernel_tensors()[0].device
        occ = torch.empty((ncell + 31) // 32, dtype=torch.int32, device=dev)
        grid = self.to_rf_grid()
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.load().rf_build_occupancy(grid, float(threshold), occ.data_ptr(), stream), "rf_build_occupancy")
        self._occupancy = occ
        d = self.kernel_tensors()[0]
        self._occupancy_stamp = (d.data_ptr(), d._version)
        return occ

    @property
    def occupancy(self) -> Optional[Tensor]:
        return self._occupancy

    def invalidate_occupancy(self) -> None:
        """The densities were changed behind autograd's back (a fused optimizer kernel writes through raw pointers): the
        mask has to be rebuilt before its next use."""
        self._occupancy_stamp = None
        self.__dict__["_shadow_epoch"] = self.__dict__.get("_shadow_epoch", 0) + 1  # (the split shadow of forward_rf_grid is stale too)

    def occupancy_current(self) -> bool:
        """True when a mask exists AND the density tensor is the one (same storage, same in-place version counter) it was
        built from.  ``torch.optim`` steps bump the version counter; the fused optimizer calls invalidate_occupancy()."""
        if self._occupancy is None:
            return False
        d = self.kernel_tensors()[0]
        return self.__dict__.get("_occupancy_stamp") == (d.data_ptr(), d._version)


class VoxelGrid(Module, KernelGridInterface):
    def __init__(
        self,
        densities: Tensor,
        features: Tensor,
        voxel_size: VoxelSize,
        grid_location: Optional[VoxelGridLocation] = VoxelGridLocation(),
        density_preactivation: Callable[[Tensor], Tensor] = torch.abs,
        density_postactivation: Callable[[Tensor], Tensor] = torch.nn.Identity(),
        feature_preactivation: Callable[[Tensor], Tensor] = torch.nn.Identity(),
        feature_postactivation: Callable[[Tensor], Tensor] = torch.nn.Identity(),
        radiance_transfer_function: Callable[[Tensor, Tensor], Tensor] = None,
        expected_density_scale: float = 1.0,
        tunable: bool = False,
        storage: str = "reference",
    ):
        """``storage`` (extension of this build) selects how the grid lives in HBM:
        "reference" keeps the reference's two tensors; "bricked" = "split" with the nodes in brick-major order
        (8^3-node bricks contiguous: cell corners ~1 KB apart, one contiguous write per brick in the binned backward);
        "split" keeps the MI355X-native pair
        base [X,Y,Z,4] = (density, degree-0 RGB) + rest [X,Y,Z,3(K-1)] (RF_LAYOUT_SPLIT).  Either way the
        constructor takes, and state_dict / .densities / .features present, reference-layout tensors."""
        if storage not in STORAGES:
            raise ValueError(f"storage must be one of {STORAGES}")
        if densities.dim() != 4 or densities.shape[-1] != 1:
            raise AssertionError(f"densities should be [W x D x H x 1], got {tuple(densities.shape)}")
        if features.dim() != 4 or features.shape[:3] != densities.shape[:3]:
            raise AssertionError(f"features should be [W x D x H x F], got {tuple(features.shape)}")
        if densities.device != features.device:
            raise AssertionError("densities and features are not on the same device")
        super().__init__()
        self._density_preactivation = density_preactivation
        self._density_postactivation = density_postactivation
        self._feature_preactivation = feature_preactivation
        self._feature_postactivation = feature_postactivation
        self._radiance_transfer_function = radiance_transfer_function
        self._grid_location = grid_location
        self._voxel_size = voxel_size
        self._expected_density_scale = expected_density_scale
        self._tunable = tunable
        # the activation pairs the fused kernels implement (the three the reference's trainer script builds + identity); any other
        # callables -- and non-identity feature activations / a radiance transfer function -- take the composed path
        # (composable.py: HIP interpolation of the pre-activated tensors, the callables applied by torch around it)
        try:
            self.density_mode = resolve_density_mode(density_preactivation, density_postactivation)
        except ValueError:
            self.density_mode = None

        densities = densities.detach().to(torch.float32).contiguous()
        features = features.detach().to(torch.float32).contiguous()
        self.storage = storage
        self._num_features = int(features.shape[-1])
        self.width_x, self.depth_y, self.height_z = (int(v) for v in features.shape[:3])
        if storage == "reference":
            self._register_grid_tensor("_densities", densities)
            self._register_grid_tensor("_features", features)
        else:
            base, rest = pack_storage(densities, features, storage)
            self._register_grid_tensor("_base", base)
            if rest is not None:
                self._register_grid_tensor("_rest", rest)
            else:
                self._rest = None
            # checkpoints keep the reference's keys and layout
            self._register_state_dict_hook(VoxelGrid._export_reference_state)
            self._register_load_state_dict_pre_hook(self._import_reference_state)
        self._aabb = self._setup_bounding_box_planes()
        self._occupancy: Optional[Tensor] = None

    def _register_grid_tensor(self, name: str, value: Tensor) -> None:
        if self._tunable:
            setattr(self, name, torch.nn.Parameter(value))
        else:
            # buffers, so that Module.to(device) moves a frozen grid as well
            self.register_buffer(name, value)

    @staticmethod
    def _export_reference_state(module, state, prefix, local_metadata):
        base = state.pop(prefix + "_base")
        rest = state.pop(prefix + "_rest", None)
        dens, feat = unpack_storage(base, rest, module.storage, module.grid_dims)
        state[prefix + u_DENSITIES], state[prefix + u_FEATURES] = dens, feat
        return state

    def _import_reference_state(self, state, prefix, local_metadata, strict, missing, unexpected, errors):
        if prefix + u_DENSITIES in state:
            base, rest = pack_storage(state.pop(prefix + u_DENSITIES), state.pop(prefix + u_FEATURES), self.storage)
            state[prefix + "_base"] = base
            if rest is not None:
                state[prefix + "_rest"] = rest

    def kernel_tensors(self):
        """The two tensors the kernels read, in storage order: (densities, features) or (base, rest).  (Waits for parameters a
        data-parallel step left in flight: the caller is about to read or describe them.)"""
        self.wait_for_parameters()
        return self._tensors()

    def _tensors(self):
        if self.storage == "reference":
            return self._densities, self._features
        return self._base, self._rest

    # readers that reach the Parameters without going through an accessor of this class
    def named_parameters(self, *args, **kwargs):
        self.wait_for_parameters()
        return super().named_parameters(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self.wait_for_parameters()
        return super()._apply(fn, *args, **kwargs)

    def state_dict(self, *args, **kwargs):
        self.wait_for_parameters()
        return super().state_dict(*args, **kwargs)

    def reference_gradients(self):
        """(dL/d densities, dL/d features) in the reference layout, whatever the storage."""
        a, b = self.kernel_tensors()
        if self.storage == "reference":
            return a.grad, b.grad
        if a.grad is None:
            return None, None
        return unpack_storage(a.grad, None if b is None else b.grad, self.storage, self.grid_dims)

    def unpack(self, first: Tensor, second: Optional[Tensor]):
        """Any pair of tensors shaped like ``kernel_tensors()`` (e.g. a gradient bucket) -> reference layout."""
        return unpack_storage(first, second, self.storage, self.grid_dims)

    def to_storage(self, storage: str) -> "VoxelGrid":
        """A new grid with the same content and configuration in the requested storage."""
        if storage == self.storage:
            return self
        return VoxelGrid(self.densities.detach(), self.features.detach(), self._voxel_size, **self.get_config_dict(), storage=storage)

    # ----- reference-compatible accessors -------------------------------------------------
    @property
    def densities(self) -> Tensor:
        """[X,Y,Z,1].  The Parameter itself with reference storage; a strided VIEW of the base tensor with
        split storage (in-place edits reach the grid, but it is not a leaf: use reference_gradients())."""
        self.wait_for_parameters()
        if self.storage == "reference":
            return self._densities
        if self.storage == "bricked":  # an assembled COPY (assign to the property to write)
            return unbrick_nodes(self._base[..., :1], self.grid_dims)
        return self._base[..., :1]

    @densities.setter
    def densities(self, value: Tensor) -> None:
        assert value.shape == (*self.grid_dims, 1), "new densities don't match the grid's dimensions"
        if self.storage == "reference":
            self._densities = torch.nn.Parameter(value) if self._tunable and not isinstance(value, torch.nn.Parameter) else value
        else:
            with torch.no_grad():
                if self.storage == "bricked":
                    self._base[..., :1].copy_(brick_nodes(value.to(self._base.device, torch.float32)))
                else:
                    self._base[..., :1].copy_(value)
        self._occupancy = None

    @property
    def features(self) -> Tensor:
        """[X,Y,Z,F], index = colour*K + k.  The Parameter itself with reference storage; an assembled COPY
        with split storage (assign to the property to write)."""
        self.wait_for_parameters()
        if self.storage == "reference":
            return self._features
        return unpack_storage(self._base, self._rest, self.storage, self.grid_dims)[1]

    @features.setter
    def features(self, value: Tensor) -> None:
        assert value.shape == (*self.grid_dims, self._num_features), "new features don't match the grid's dimensions"
        if self.storage == "reference":
            self._features = torch.nn.Parameter(value) if self._tunable and not isinstance(value, torch.nn.Parameter) else value
        else:
            with torch.no_grad():
                base, rest = pack_storage(self.densities, value.to(self._base.device), self.storage)
                self._base.copy_(base)
                if rest is not None:
                    self._rest.copy_(rest)

    @property
    def aabb(self) -> AxisAlignedBoundingBox:
        return self._aabb

    @property
    def grid_dims(self) -> Tuple[int, int, int]:
        return self.width_x, self.depth_y, self.height_z

    @property
    def voxel_size(self) -> VoxelSize:
        return self._voxel_size

    @voxel_size.setter
    def voxel_size(self, voxel_size: VoxelSize) -> None:
        self._voxel_size = voxel_size
        self._aabb = self._setup_bounding_box_planes()

    @property
    def expected_density_scale(self) -> float:
        return self._expected_density_scale

    @property
    def num_features(self) -> int:
        return self._num_features

    def fused_kernels_apply(self) -> bool:
        """True when the fused render kernels implement this grid's activations (else: the composed path)."""
        return self.density_mode is not None and _is_identity(self._feature_preactivation) and _is_identity(self._feature_postactivation)

    def forward(self, points: Tensor, viewdirs: Optional[Tensor] = None) -> Tensor:
        """[N, 3] -> [N, F+1] = cat(interpolated features, density)  (reference voxels.py:276-331).  With the activations the
        kernels implement: one rf_grid_query launch.  Otherwise the reference's composition with the HIP interpolation in the
        middle: post(interp(pre(D * rho))) | post_f(interp(pre_f(F))), then the radiance transfer function when ``viewdirs`` is given."""
        from .ops import grid_query, interpolate_tensors

        if self.fused_kernels_apply():
            out = grid_query(self, points)
            if self._radiance_transfer_function is None or viewdirs is None:
                return out
            feats, dens = out[..., :-1], out[..., -1:]
        else:
            pre_d = self._density_preactivation(self.densities * self._expected_density_scale)
            pre_f = self._feature_preactivation(self.features)
            both = interpolate_tensors(self, pre_d, pre_f, points)
            dens = self._density_postactivation(both[..., -1:])
            feats = self._feature_postactivation(both[..., :-1])
        if self._radiance_transfer_function is not None and viewdirs is not None:
            feats = self._radiance_transfer_function(feats, viewdirs)
        return torch.cat([feats, dens], dim=-1)

    def get_config_dict(self) -> Dict[str, Any]:
        return {
            "grid_location": self._grid_location,
            "density_preactivation": self._density_preactivation,
            "density_postactivation": self._density_postactivation,
            "feature_preactivation": self._feature_preactivation,
            "feature_postactivation": self._feature_postactivation,
            "radiance_transfer_function": self._radiance_transfer_function,
            "expected_density_scale": self._expected_density_scale,
            "tunable": self._tunable,
        }

    def get_save_config_dict(self) -> Dict[str, Any]:
        out = self.get_config_dict()
        out["voxel_size"] = self._voxel_size
        return out

    def _setup_bounding_box_planes(self) -> AxisAlignedBoundingBox:
        """centre +- dims * voxel_size / 2 in Python floats (reference voxels.py:187-212)."""
        ranges = []
        for n, v, c in zip(self.grid_dims, self._voxel_size, self._grid_location):
            half = (n * v) / 2
            ranges.append((c - half, c + half))
        return AxisAlignedBoundingBox(*ranges)

    def extra_repr(self) -> str:
        return (
            f"grid_dims: {self.grid_dims}, feature_dims: {self._num_features}, "
            f"voxel_size: {self._voxel_size}, grid_location: {self._grid_location}, "
            f"density: {self.density_mode}, tunable: {self._tunable}, storage: {self.storage}"
        )

    def test_inside_volume(self, points: Tensor) -> Tensor:
        """[N, 3] -> [N, 1] bool, strict inequalities (reference voxels.py:252-274)."""
        m = torch.ones_like(points[..., 0:1], dtype=torch.bool)
        for a, (lo, hi) in enumerate(self._aabb):
            m = m & (points[..., a : a + 1] > lo) & (points[..., a : a + 1] < hi)
        return m


class ForeignVoxelGridView(KernelGridInterface):
    """Kernel-side description of a grid module that is NOT this package's VoxelGrid but has the reference VoxelGrid's
    attribute names (duck typing; thre3d_atom/thre3d_reprs/voxels.py): ``densities`` [X,Y,Z,1] / ``features`` [X,Y,Z,F]
    (contiguous float32 on a HIP device, the reference's own layout = RF_LAYOUT_REFERENCE), ``aabb``,
    ``_expected_density_scale``, ``_density_preactivation`` / ``_density_postactivation`` (one of the supported pairs),
    identity feature activations and no radiance transfer function.  Everything is read from the module at call time, so
    re-assigned tensors or a changed voxel size are picked up; gradients flow to ``module.densities`` / ``module.features``
    through autograd like they do with the reference's own procedure."""

    storage = "reference"
    _grad_bucket = None

    def __init__(self, module):
        missing = [a for a in ("densities", "features", "aabb", "_expected_density_scale", "_density_preactivation", "_density_postactivation") if not hasattr(module, a)]
        if missing:
            raise TypeError(f"render_sh_voxel_grid (HIP) needs a VoxelGrid-like module; {type(module).__name__} lacks {missing}")
        self._module_ref = weakref.ref(module)  # (the view is cached weakly keyed by the module: no cycle that keeps it alive)
        self._occupancy = None

    @property
    def module(self):
        m = self._module_ref()
        if m is None:
            raise ReferenceError("the grid module behind this view no longer exists")
        return m

    def _check(self):
        m = self.module
        for name in ("_feature_preactivation", "_feature_postactivation"):
            if not _is_identity(getattr(m, name, None)):
                raise ValueError("the HIP render path supports identity feature activations only")
        if getattr(m, "_radiance_transfer_function", None) is not None:
            raise ValueError("radiance_transfer_function is not used on the SH render path")
        d, f = m.densities, m.features
        if d.dim() != 4 or d.shape[-1] != 1 or f.dim() != 4 or f.shape[:3] != d.shape[:3]:
            raise AssertionError(f"densities [X,Y,Z,1] / features [X,Y,Z,F] expected, got {tuple(d.shape)} / {tuple(f.shape)}")

    def kernel_tensors(self):
        self._check()
        return self.module.densities, self.module.features

    _tensors = kernel_tensors

    @property
    def grid_dims(self):
        return tuple(int(v) for v in self.module.densities.shape[:3])

    @property
    def _aabb(self):
        return AxisAlignedBoundingBox(*[(float(lo), float(hi)) for lo, hi in self.module.aabb])

    @property
    def aabb(self):
        return self._aabb

    @property
    def _expected_density_scale(self):
        return float(self.module._expected_density_scale)

    @property
    def expected_density_scale(self):
        return self._expected_density_scale

    @property
    def density_mode(self):
        return resolve_density_mode(self.module._density_preactivation, self.module._density_postactivation)

    @property
    def _num_features(self):
        return int(self.module.features.shape[-1])

    @property
    def num_features(self):
        return self._num_features


def as_kernel_grid(module) -> KernelGridInterface:
    """The object the operators talk to: a VoxelGrid of this package as it is, any other VoxelGrid-like module (the
    reference's) through a cached ForeignVoxelGridView."""
    if isinstance(module, KernelGridInterface):
        return module
    try:
        view = _RF_VIEW_CACHE.get(module)
    except TypeError:  # not weakly referenceable / not hashable: just do not cache
        return ForeignVoxelGridView(module)
    if view is None:
        view = ForeignVoxelGridView(module)
        _RF_VIEW_CACHE[module] = view
    return view


def scale_voxel_grid_with_required_output_size(
    voxel_grid: VoxelGrid, output_size: Tuple[int, int, int], mode: str = "trilinear"
) -> VoxelGrid:
    """Trilinear (align_corners=False) resampling of the whole [F+1]-channel volume, the voxel size
    shrinking so that the world extent is unchanged (reference voxels.py:334-373).  On the GPU: rf_upsample_grid
    (ATen's index / weight arithmetic, bit-identical to torch's CPU result) straight from the source storage into the
    new grid's storage -- no unified [X,Y,Z,F+1] tensor, no permutes.  CPU grids and other modes: F.interpolate."""
    old = voxel_grid.voxel_size
    new_voxel = VoxelSize(
        (old.x_size * voxel_grid.width_x) / output_size[0],
        (old.y_size * voxel_grid.depth_y) / output_size[1],
        (old.z_size * voxel_grid.height_z) / output_size[2],
    )
    first = voxel_grid.kernel_tensors()[0]
    if mode == "trilinear" and first.is_cuda:
        out = tuple(int(v) for v in output_size)
        new_grid = VoxelGrid(
            densities=torch.empty((*out, 1), dtype=torch.float32, device=first.device),
            features=torch.empty((*out, voxel_grid._num_features), dtype=torch.float32, device=first.device),
            voxel_size=new_voxel,
            **voxel_grid.get_config_dict(),
            storage=voxel_grid.storage,
        )
        stream = torch.cuda.current_stream(first.device).cuda_stream
        with torch.no_grad():
            _lib.check(_lib.load().rf_upsample_grid(voxel_grid.to_rf_grid(), new_grid.to_rf_grid(), stream), "rf_upsample_grid")
        return new_grid
    unified = torch.cat([voxel_grid.features, voxel_grid.densities], dim=-1).detach()
    new = torch.nn.functional.interpolate(
        unified.permute(3, 0, 1, 2)[None],
        size=tuple(output_size),
        mode=mode,
        align_corners=False,
        recompute_scale_factor=False,
    )[0].permute(1, 2, 3, 0)
    assert tuple(new.shape[:-1]) == tuple(output_size)
    return VoxelGrid(
        densities=new[..., -1:].contiguous(),
        features=new[..., :-1].contiguous(),
        voxel_size=new_voxel,
        **voxel_grid.get_config_dict(),
        storage=voxel_grid.storage,
    )


def create_voxel_grid_from_saved_info_dict(saved_info: Dict[str, Any], storage: str = "reference") -> VoxelGrid:
    """Rebuild a grid from the ``thre3d_repr`` entry of a checkpoint (reference voxels.py:376-383)."""
    state = saved_info[THRE3D_REPR][STATE_DICT]
    grid = VoxelGrid(
        densities=torch.empty_like(state[u_DENSITIES]),
        features=torch.empty_like(state[u_FEATURES]),
        **saved_info[THRE3D_REPR][CONFIG_DICT],
        storage=storage,
    )
    grid.load_state_dict(state)
    return grid
