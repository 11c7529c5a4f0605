"""Host-side storage conversions (no GPU): reference <-> split <-> bricked are exact inverses and the brick-major node
order matches the kernels' node_lin() formula (thr3ed_atom_amd/csrc/relu_field_kernels.hip)."""
import pytest
import torch

from thr3ed_atom_amd.voxels import BRICK, brick_nodes, pack_split, pack_storage, unbrick_nodes, unpack_split, unpack_storage


@pytest.mark.parametrize("dims", [(8, 8, 8), (5, 9, 17), (16, 24, 8), (1, 1, 1)])
@pytest.mark.parametrize("K", [1, 4, 9])
def test_pack_unpack_round_trip(dims, K):
    g = torch.Generator().manual_seed(3)
    d = torch.rand((*dims, 1), generator=g)
    f = torch.rand((*dims, 3 * K), generator=g)
    for storage in ("split", "bricked"):
        a, b = pack_storage(d, f, storage)
        assert (b is None) == (K == 1)
        d2, f2 = unpack_storage(a, b, storage, dims)
        assert torch.equal(d, d2) and torch.equal(f, f2)
    base, rest = pack_split(d, f)
    assert base.shape == (*dims, 4) and torch.equal(base[..., 0:1], d)
    assert torch.equal(base[..., 1:], f.unflatten(-1, (3, K))[..., 0])  # degree-0 coefficient of r, g, b
    d3, f3 = unpack_split(base, rest)
    assert torch.equal(d3, d) and torch.equal(f3, f)


def test_brick_major_order_matches_the_kernel_index_formula():
    dims = (5, 9, 17)
    t = torch.arange(5 * 9 * 17 * 4, dtype=torch.float32).reshape(*dims, 4)
    b = brick_nodes(t)
    nbx, nby, nbz = b.shape[:3]
    assert (nbx, nby, nbz) == (1, 2, 3) and b.shape[3:6] == (BRICK, BRICK, BRICK)
    flat = b.reshape(-1, 4)
    for x in range(dims[0]):
        for y in range(dims[1]):
            for z in range(dims[2]):
                lin = (((x >> 3) * nby + (y >> 3)) * nbz + (z >> 3)) * 512 + (((x & 7) << 6) | ((y & 7) << 3) | (z & 7))
                assert torch.equal(flat[lin], t[x, y, z])
    assert torch.equal(unbrick_nodes(b, dims), t)
    # padding nodes are zero and do not leak back
    assert float(b.sum()) == float(t.sum())


def test_composed_path_has_no_cpu_side():
    """thr3ed_atom_amd.composable (the path at the reference's plug-in points) runs on the device only: CPU tensors raise."""
    import pytest
    import thr3ed_atom_amd as rf
    from thr3ed_atom_amd import composable as cp

    rays = rf.Rays(torch.zeros(4, 3), torch.ones(4, 3))
    with pytest.raises(RuntimeError, match="HIP device"):
        cp.sample_uniform_points_on_rays(rays, rf.CameraBounds(1.0, 2.0), 8)
    with pytest.raises(RuntimeError, match="HIP device"):
        cp.accumulate_radiance_density_on_rays(cp.ProcessedPointsOnRays(torch.zeros(4, 8, 4), torch.zeros(4, 8)), rays)
    assert rf.render is cp.render and rf.SampledPointsOnRays is cp.SampledPointsOnRays
