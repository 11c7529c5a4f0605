"""World-size-2 tests of the data-parallel wiring on the gloo backend (CPU, runs everywhere).

The GPU path uses the same helpers with backend "nccl" (= RCCL): one all-reduce of the flat gradient bucket
per step, identical optimiser state on every rank, ray shards by contiguous range."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from thr3ed_atom_amd import distributed as rfdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, result_dir):
    os.environ.update(
        RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)
    )
    r, lr, w = rfdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and rfdist.rank() == rank and rfdist.world_size() == world

    # 1. flat-bucket gradient average == gradient of the concatenated batch
    torch.manual_seed(0)
    params = torch.randn(1000)
    data = torch.randn(world * 8, 1000)
    target = torch.randn(world * 8)
    p = params.clone().requires_grad_(True)
    lo, hi = rfdist.shard_range(world * 8)
    assert hi - lo == 8
    loss = ((data[lo:hi] @ p - target[lo:hi]) ** 2).mean()
    loss.backward()
    bucket = p.grad.clone()
    rfdist.all_reduce_mean_(bucket)
    pf = params.clone().requires_grad_(True)
    ((data @ pf - target) ** 2).mean().backward()
    assert torch.allclose(bucket, pf.grad, rtol=1e-5, atol=1e-6)

    # 2. identical Adam on every rank keeps replicas bit-identical after several steps
    m, v = torch.zeros_like(params), torch.zeros_like(params)
    q = params.clone()
    for step in range(1, 4):
        g = bucket * step
        m = m + (g - m) * 0.1
        v = v * 0.999 + g * g * 0.001
        q = q - 0.03 / (1 - 0.9**step) * (m / (v.sqrt() / (1 - 0.999**step) ** 0.5 + 1e-8))
    gathered = [torch.empty_like(q) for _ in range(world)]
    dist.all_gather(gathered, q)
    assert all(torch.equal(gathered[0], t) for t in gathered)

    # 2b. asynchronous mean all-reduce of the two parts of a bucket == one mean all-reduce of the whole bucket
    whole = torch.arange(100, dtype=torch.float32) * (rank + 1)
    parts = whole.clone()
    first = rfdist.all_reduce_mean_async(parts[60:])
    parts[:60].add_(0.0)  # work on the other part while the first collective is in flight
    second = rfdist.all_reduce_mean_async(parts[:60])
    first.wait()
    second.wait()
    rfdist.all_reduce_mean_(whole)
    assert torch.equal(parts, whole) and torch.allclose(whole, torch.arange(100.0) * (sum(range(1, world + 1)) / world))

    # 3. ragged row gather (rays of a frame split over ranks) and broadcast
    rows = torch.arange(10 * 3, dtype=torch.float32).reshape(10, 3)
    lo, hi = rfdist.shard_range(10)
    full = rfdist.all_gather_rows(rows[lo:hi] * 1.0)
    assert torch.equal(full, rows)
    # ... with the row counts known from the shard rule (what VolumetricModel.render passes): one all-gather, no count exchange;
    # ragged (11 rows) and even (12 rows) splits
    for total in (11, 12):
        rows = torch.arange(total * 6, dtype=torch.float32).reshape(total, 6)
        lo, hi = rfdist.shard_range(total)
        sizes = [b_ - a_ for a_, b_ in (rfdist.shard_range(total, r2, world) for r2 in range(world))]
        assert torch.equal(rfdist.all_gather_rows(rows[lo:hi], sizes), rows)
    b = torch.full((4,), float(rank))
    rfdist.broadcast_(b, src=0)
    assert torch.equal(b, torch.zeros(4))
    # 4. sharded optimizer exchange (ZeRO stage 1): reduce-scatter(mean) -> every rank updates its own chunk only ->
    #    all-gather of the chunks  ==  all-reduce(mean) followed by the full update on every rank
    grad = torch.arange(96.0) * (rank + 1)
    param = torch.zeros(96)
    assert rfdist.can_shard(96) and not rfdist.can_shard(97) or world == 1
    h = rfdist.reduce_scatter_mean_async(grad)
    h.wait()
    assert (h.hi - h.lo) * world == 96 and h.lo == rank * (96 // world)
    param[h.lo : h.hi] -= 0.1 * h.shard
    rfdist.all_gather_chunks_(param)
    assert torch.allclose(param, -0.1 * torch.arange(96.0) * (sum(range(1, world + 1)) / world))
    # 5. owner-computes exchange primitives: equal-sized all-gather of tables, personalised exchange of (overlapping) slices
    table = torch.arange(6, dtype=torch.int64).reshape(2, 3) + 100 * rank
    tables = torch.empty((world, 2, 3), dtype=torch.int64)
    assert rfdist.all_gather_rows_equal(tables, table, async_op=True) is None  # (gloo: completed on return)
    assert all(torch.equal(tables[r2], torch.arange(6).reshape(2, 3) + 100 * r2) for r2 in range(world))
    data = (torch.arange(40, dtype=torch.float32).reshape(10, 4) + 1000 * rank)
    # rank r sends rows [2 d, 2 d + 3 + r) to rank d: the slices of neighbouring destinations overlap
    send = [data[2 * d : 2 * d + 3 + rank] for d in range(world)]
    recv = [torch.full((3 + s_, 4), -1.0) for s_ in range(world)]
    rfdist.exchange_slices(send, recv)
    for s_ in range(world):
        if s_ == rank:
            assert torch.equal(recv[s_], torch.full((3 + s_, 4), -1.0))  # the own entry is left alone
        else:
            assert torch.equal(recv[s_], torch.arange(40, dtype=torch.float32).reshape(10, 4)[2 * rank : 2 * rank + 3 + s_] + 1000 * s_)
    open(os.path.join(result_dir, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_gradient_bucket_allreduce_and_sharding(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(world)]


def test_shard_range_covers_everything():
    for total in (0, 1, 7, 16384, 640000):
        for world in (1, 2, 3, 8):
            spans = [rfdist.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_helpers_are_noops():
    t = torch.ones(5)
    assert rfdist.world_size() == 1 and rfdist.rank() == 0
    assert torch.equal(rfdist.all_reduce_mean_(t.clone()), t)
    assert torch.equal(rfdist.all_gather_rows(t), t)
    h = rfdist.reduce_scatter_mean_async(t)
    h.wait()
    assert (h.lo, h.hi) == (0, 5) and rfdist.can_shard(5)
    assert torch.equal(rfdist.all_gather_chunks_(t.clone()), t)
