"""The C-ABI shared library loads without a GPU and exports every symbol include/relu_field.h declares."""
import os
import re

import pytest

from thr3ed_atom_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def declared_functions():
    text = open(os.path.join(REPO, "include", "relu_field.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rf_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    assert len(names) >= 9
    for name in names:
        assert hasattr(lib, name), f"{name} declared in relu_field.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == names


def test_abi_version_and_error_strings(lib):
    assert lib.rf_abi_version() == 2
    assert lib.rf_error_string(0) == b"ok"
    for code in (-1, -2, -3, -4):
        assert lib.rf_error_string(code) not in (b"ok", b"unknown error code")
    assert lib.rf_error_string(-99) == b"unknown error code"


def test_argument_validation_needs_no_gpu(lib):
    """Error paths return codes before touching the device (never throw across the ABI)."""
    import ctypes as C

    g = _lib.RFGrid()
    r = _lib.RFRayBatch()
    o = _lib.RFRenderOut()
    assert lib.rf_render_forward(C.byref(g), C.byref(r), 0, C.byref(o), None) == -1  # null grid tensors
    g.densities_dev, g.features_dev = 16, 16
    g.dims[0], g.dims[1], g.dims[2] = 4, 4, 4
    g.num_features, g.density_stride, g.feature_stride = 5, 1, 5
    assert lib.rf_render_forward(C.byref(g), C.byref(r), 0, C.byref(o), None) == -3  # F=5 is no SH degree
    g.num_features, g.feature_stride = 27, 27
    g.dims[0] = 0
    assert lib.rf_render_forward(C.byref(g), C.byref(r), 0, C.byref(o), None) == -2  # bad shape
    g.dims[0] = 4
    r.num_rays, r.num_samples = 0, 0
    assert lib.rf_render_forward(C.byref(g), C.byref(r), 0, C.byref(o), None) == -2  # S < 1
    r.num_samples = 8
    assert lib.rf_render_forward(C.byref(g), C.byref(r), 0, C.byref(o), None) == 0  # zero rays: no-op
    assert lib.rf_cast_rays(4, 4, 1.0, None, None, None, None, None) == -1


def test_product_path_raises_without_gpu_tensors():
    import torch

    import thr3ed_atom_amd as rf

    grid = rf.VoxelGrid(torch.zeros(2, 2, 2, 1), torch.zeros(2, 2, 2, 3), rf.VoxelSize(1, 1, 1))
    cfg = rf.SHVoxGridRenderConfig(4, rf.CameraBounds(0.5, 4.0), perturb_sampled_points=False)
    rays = rf.Rays(torch.zeros(3, 3), torch.ones(3, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rf.render_sh_voxel_grid(grid, rays, cfg)
