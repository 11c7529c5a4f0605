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
    assert lib.rf_abi_version() == _lib.ABI_VERSION == 4
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
    # rf_upsample_grid: null / mismatching / aliased grids
    assert lib.rf_upsample_grid(None, C.byref(g), None) == -1
    h = _lib.RFGrid()
    h.densities_dev, h.features_dev = 4096, 8192
    h.dims[0], h.dims[1], h.dims[2] = 8, 8, 8
    h.num_features, h.density_stride, h.feature_stride = 3, 1, 3
    assert lib.rf_upsample_grid(C.byref(g), C.byref(h), None) == -2  # 27 vs 3 features
    h.num_features, h.feature_stride = 27, 27
    h.densities_dev = 16
    assert lib.rf_upsample_grid(C.byref(g), C.byref(h), None) == -2  # destination aliases the source


def test_node_count_guard_and_fused_optimizer_validation(lib):
    """check_grid rejects grids whose node count does not fit the kernels' 32-bit node indices; rf_brick_accumulate_adam
    validates layout / alignment / aliasing before any launch."""
    import ctypes as C

    g = _lib.RFGrid()
    g.densities_dev, g.features_dev = 16, 32
    g.num_features, g.density_stride, g.feature_stride = 3, 1, 3
    g.dims[0], g.dims[1], g.dims[2] = 2046, 2046, 2046  # 8.6e9 nodes
    r, o = _lib.RFRayBatch(), _lib.RFRenderOut()
    r.num_samples = 8
    assert lib.rf_render_forward(C.byref(g), C.byref(r), 0, C.byref(o), None) == -2
    g.dims[0], g.dims[1], g.dims[2] = 1500, 1500, 1500  # 3.4e9 nodes: fits
    assert lib.rf_render_forward(C.byref(g), C.byref(r), 0, C.byref(o), None) == 0
    # fused optimizer: reference layout is refused, so are misaligned moments and parameters that are not the grid's tensors
    g.dims[0], g.dims[1], g.dims[2] = 16, 16, 16
    g.num_features, g.density_stride, g.feature_stride = 27, 1, 27
    lists = (_lib.RFBrickList * 2)()
    for i in range(2):
        lists[i].records_sorted_dev, lists[i].offsets_dev, lists[i].render_diffuse = 64, 64, i
    st = _lib.RFAdamState()
    st.param_first_dev, st.param_second_dev = 16, 32
    st.exp_avg_first_dev = st.exp_avg_second_dev = st.exp_avg_sq_first_dev = st.exp_avg_sq_second_dev = 64
    st.lr, st.beta1, st.beta2, st.eps, st.step = 0.03, 0.9, 0.999, 1e-8, 1
    assert lib.rf_brick_accumulate_adam(C.byref(g), 8, lists, 2, C.byref(st), None) == -3  # reference layout
    g.layout, g.density_stride, g.feature_stride = 1, 4, 24
    st.exp_avg_first_dev = 68
    assert lib.rf_brick_accumulate_adam(C.byref(g), 8, lists, 2, C.byref(st), None) == -2  # misaligned
    st.exp_avg_first_dev, st.param_first_dev = 64, 48
    assert lib.rf_brick_accumulate_adam(C.byref(g), 8, lists, 2, C.byref(st), None) == -2  # not the grid's tensor
    st.param_first_dev, st.step = 16, 0
    assert lib.rf_brick_accumulate_adam(C.byref(g), 8, lists, 2, C.byref(st), None) == -2  # step counts from 1
    assert lib.rf_brick_accumulate_adam(C.byref(g), 8, lists, 2, None, None) == -1
    lists[0].render_diffuse, lists[1].render_diffuse = 1, 0
    assert lib.rf_brick_accumulate(C.byref(g), 8, lists, 2, 64, 64, 0, None) == -2  # the specular list comes first


def test_train_step_and_camera_validation_need_no_gpu(lib):
    """rf_train_step / RFRayBatch.camera: argument errors are reported as codes before any launch."""
    import ctypes as C

    g = _lib.RFGrid()
    g.densities_dev, g.features_dev = 16, 32
    g.dims[0], g.dims[1], g.dims[2] = 16, 16, 16
    g.num_features, g.density_stride, g.feature_stride, g.layout = 27, 4, 24, 1
    st = _lib.RFTrainStep()
    assert lib.rf_train_step(C.byref(g), None, None) == -1
    assert lib.rf_train_step(C.byref(g), C.byref(st), None) == 0  # zero rays: nothing to do
    st.num_rays, st.num_samples = 8, 16
    assert lib.rf_train_step(C.byref(g), C.byref(st), None) == -1  # ray / pixel / loss buffers missing
    st.origins_dev = st.directions_dev = st.pixels_dev = st.loss_sums_dev = 64
    assert lib.rf_train_step(C.byref(g), C.byref(st), None) == -1  # per-render scratch missing
    for i in range(2):
        ps = st.pass_[i]
        ps.grad_colour_dev = ps.cursor_dev = ps.offsets_dev = ps.records_sorted_dev = 64
        ps.out.key_hist_dev, ps.out.brick_size = 64, 8
    st.pass_[1].out.brick_size = 4
    assert lib.rf_train_step(C.byref(g), C.byref(st), None) == -2  # the two lists must use the same bricks
    st.pass_[0].out.brick_size = st.pass_[1].out.brick_size = 5
    assert lib.rf_train_step(C.byref(g), C.byref(st), None) == -3  # brick size 4 or 8
    # camera-generated rays: the pixel range must lie inside the frame
    cam = _lib.RFCamera()
    cam.height, cam.width, cam.focal = 10, 12, 20.0
    r, o = _lib.RFRayBatch(), _lib.RFRenderOut()
    r.num_rays, r.num_samples, r.t_vals_dev, r.camera, r.first_ray = 100, 8, 64, C.pointer(cam), 30
    assert lib.rf_render_forward(C.byref(g), C.byref(r), 0, C.byref(o), None) == -2  # 30 + 100 > 120 pixels
    r.first_ray = 20
    assert lib.rf_render_forward(C.byref(g), C.byref(r), 0, C.byref(o), None) == -1  # in range; output buffers missing
    r.camera = None
    assert lib.rf_render_forward(C.byref(g), C.byref(r), 0, C.byref(o), None) == -1  # neither rays nor a camera


def test_product_path_raises_without_gpu_tensors():
    import torch

    import thr3ed_atom_amd as rf

    grid = rf.VoxelGrid(torch.zeros(2, 2, 2, 1), torch.zeros(2, 2, 2, 3), rf.VoxelSize(1, 1, 1))
    cfg = rf.SHVoxGridRenderConfig(4, rf.CameraBounds(0.5, 4.0), perturb_sampled_points=False)
    rays = rf.Rays(torch.zeros(3, 3), torch.ones(3, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rf.render_sh_voxel_grid(grid, rays, cfg)


def test_frame_kernel_dispatch_rule_needs_no_gpu(lib, monkeypatch):
    """rf_frame_render_kernel (host-side only): frames go to the ray-packet kernel where an 8 x 8 pixel tile's footprint at the
    volume's centre is at most 2 voxels (3 with the occupancy mask) on split storage, else to the per-ray kernel;
    $RF_FRAME_TILES overrides where the packet kernel exists.  The bench configurations: configs[1] -> packets, configs[4] -> packets
    only with the mask."""
    import ctypes as C

    monkeypatch.delenv("RF_FRAME_TILES", raising=False)

    def grid_of(G, F=27, layout="split", occ=False):
        g = _lib.RFGrid()
        g.densities_dev, g.features_dev = 1 << 20, (1 << 20) + G * G * G * 16
        for a in range(3):
            g.dims[a] = G
            g.aabb_min[a], g.aabb_max[a] = -1.5, 1.5
            g.norm_scale[a], g.norm_bias[a] = 1.0 / 1.5, 0.0
        g.num_features = F
        split = layout != "reference"
        g.density_stride, g.feature_stride = (4, F - 3) if split else (1, F)
        g.layout = _lib.LAYOUTS[layout]
        g.density_scale, g.density_mode = 1.0, 0
        g.occupancy_dev = (1 << 19) if occ else None
        return g

    def cam_of(focal, dist=4.0311, hw=800):
        c = _lib.RFCamera()
        c.height, c.width, c.focal = hw, hw, focal
        for i, v in enumerate((1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, dist)):  # identity rotation, camera on the z axis
            c.pose[i] = float(v)
        return c

    def choice(g, c, flags=0):
        return lib.rf_frame_render_kernel(C.byref(g), C.byref(c), flags)

    cam = cam_of(1111.111)
    assert choice(grid_of(128), cam) == 1                                   # configs[1]: 1.24 voxels
    assert choice(grid_of(256), cam) == 0                                   # configs[4] without the mask: 2.48 voxels
    assert choice(grid_of(256, occ=True), cam, _lib.FLAG_OCCUPANCY_SKIP) == 1  # ... with it
    assert choice(grid_of(256, occ=True), cam, 0) == 0                      # (a mask that the render does not use)
    assert choice(grid_of(128), cam_of(88.9)) == 0                          # a 64-pixel camera: a tile spans many voxels
    assert choice(grid_of(128, layout="reference"), cam) == 0               # no packet kernel for the reference layout
    assert choice(grid_of(128, F=12), cam) == 1 and choice(grid_of(128, F=48), cam) == 1  # SH degree 1 / 3: the generic rest path (round 6)
    misaligned = grid_of(128)
    misaligned.densities_dev += 4
    assert choice(misaligned, cam) == 0                                     # the base records are fetched as aligned 16-byte quads
    assert choice(grid_of(64, F=3), cam_of(1111.111)) == 1                  # degree 0: the base-record instantiation
    monkeypatch.setenv("RF_FRAME_TILES", "0")
    assert choice(grid_of(128), cam) == 0
    monkeypatch.setenv("RF_FRAME_TILES", "1")
    assert choice(grid_of(256), cam) == 1 and choice(grid_of(128, layout="reference"), cam) == 0
    assert lib.rf_frame_render_kernel(C.byref(grid_of(128)), None, 0) == -1


def test_paired_entry_points_validate_before_any_launch(lib):
    """rf_render_forward_pair / rf_l1_loss_grad_pair / rf_bin_offsets_pair / rf_render_backward_emit_direct_pair (round 6: the two renders
    of an iteration as separate calls): argument errors come back as codes, and RF_ERR_UNSUPPORTED says "these two do not pair up"."""
    import ctypes as C

    g = _lib.RFGrid()
    g.densities_dev, g.features_dev = 16, 32
    g.dims[0], g.dims[1], g.dims[2] = 16, 16, 16
    g.num_features, g.density_stride, g.feature_stride, g.layout = 27, 4, 24, 1
    rays, outs, flags = (_lib.RFRayBatch * 2)(), (_lib.RFRenderOut * 2)(), (C.c_uint32 * 2)(0, _lib.FLAG_RENDER_DIFFUSE)
    assert lib.rf_render_forward_pair(C.byref(g), None, flags, outs, None) == -1
    for i in range(2):
        rays[i].origins_dev = rays[i].directions_dev = rays[i].t_vals_dev = 64
        rays[i].num_rays, rays[i].num_samples = 8, 16
    assert lib.rf_render_forward_pair(C.byref(g), rays, flags, outs, None) == -1  # output buffers missing
    for i in range(2):
        outs[i].colour_dev = outs[i].depth_dev = outs[i].acc_dev = outs[i].disparity_dev = 64
    assert lib.rf_render_forward_pair(C.byref(g), rays, flags, outs, None) == -3  # no sample caches: not the pair's business
    for i in range(2):
        outs[i].sample_cache_dev = outs[i].trans_cache_dev = outs[i].stop_cache_dev = outs[i].chunk_mask_dev = 64
    wrong = (C.c_uint32 * 2)(_lib.FLAG_RENDER_DIFFUSE, 0)
    assert lib.rf_render_forward_pair(C.byref(g), rays, wrong, outs, None) == -3  # [0] specular, [1] render_diffuse
    rays[1].num_rays = 9
    assert lib.rf_render_forward_pair(C.byref(g), rays, flags, outs, None) == -3  # different ray counts
    rays[1].num_rays = 8
    cam = _lib.RFCamera()
    rays[0].camera = C.pointer(cam)
    assert lib.rf_render_forward_pair(C.byref(g), rays, flags, outs, None) == -3  # frames are not paired
    rays[0].num_rays = rays[1].num_rays = 0
    rays[0].camera = None
    assert lib.rf_render_forward_pair(C.byref(g), rays, flags, outs, None) == 0  # zero rays: nothing to do
    # losses
    vp2 = C.c_void_p * 2
    assert lib.rf_l1_loss_grad_pair(None, 64, 8, 1.0, vp2(64, 64), 64, 64, None) == -1
    assert lib.rf_l1_loss_grad_pair(vp2(64, None), 64, 8, 1.0, vp2(64, 64), 64, 64, None) == -1
    assert lib.rf_l1_loss_grad_pair(vp2(64, 64), 64, 0, 1.0, vp2(64, 64), 64, 64, None) == -2
    # offsets
    assert lib.rf_bin_offsets_pair(None, 64, vp2(64, 64), vp2(64, 64), None) == -1
    assert lib.rf_bin_offsets_pair(vp2(64, 64), 0, vp2(64, 64), vp2(64, 64), None) == -2
    assert lib.rf_bin_offsets_pair(vp2(64, None), 64, vp2(64, 64), vp2(64, 64), None) == -1
    # adjoints
    passes = (_lib.RFPassScratch * 2)()
    assert lib.rf_render_backward_emit_direct_pair(C.byref(g), rays, flags, None, None) == -1
    passes[0].out.brick_size, passes[1].out.brick_size = 8, 4
    assert lib.rf_render_backward_emit_direct_pair(C.byref(g), rays, flags, passes, None) == -2  # one brick size for both lists
    passes[1].out.brick_size = 8
    assert lib.rf_render_backward_emit_direct_pair(C.byref(g), rays, flags, passes, None) == -1  # cursors / records / gradients missing
