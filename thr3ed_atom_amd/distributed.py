"""Data-parallel rendering/training across the GPUs of one node: one process per GPU over RCCL/xGMI.

The reference has no distributed code at all (single process, single device).  Rays are independent, so the path shards
naturally: every rank renders its own ray batch against a replicated grid.  Frames: ``shard_range`` + ``all_gather_rows`` (the
[n, 6] per-ray results; VolumetricModel.render(data_parallel=True)).  Training adds exactly ONE exchange per iteration, in one of
two forms (trainers.TrainStepper(exchange=...)):

* ``"owner"`` -- the trainer's default where the binned adjoint applies: OWNER-COMPUTES.  The ranks exchange what their adjoints
  produce, gradient RECORDS already sorted by brick (``exchange_slices`` / ``fast_exchange``: an all-to-all of record slices, after
  an all-gather of the offset tables), every rank sums all ranks' records for its own x-slabs of bricks with Adam in the flush, and
  the updated parameters are all-gathered in place (``fast_all_gather_in_place`` / ``all_gather_chunks_``).
* ``"dense"`` -- the fall-back for grids the owner step does not cover: the flat gradient bucket (``FlatGrid.flat_grad``: 234.9 MB
  at 128^3 / SH degree 2) goes through a reduce-scatter (average), every rank applies Adam to its 1/N of the parameters (ZeRO
  stage 1) and the updated parameters are all-gathered; ``all_reduce_mean_`` + identical Adam everywhere when sharding is off.

The replicas stay bit-identical in every form.
Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests of the collective wiring.
"""
import os
from typing import Optional

import torch
import torch.distributed as dist
import torch.distributed.distributed_c10d as _c10d
from torch import Tensor


def env_world() -> tuple:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Join the process group described by torchrun's environment (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT).  Returns (rank, local_rank, world_size); a world of 1 initialises nothing.

    Test hooks (a multi-process run on a box with ONE GPU): ``RF_DIST_BACKEND=gloo`` selects the backend (RCCL refuses
    two ranks on one device; gloo carries device tensors through the host) and ``RF_SINGLE_DEVICE=1`` maps every rank to
    device 0 -- the returned local_rank is then 0."""
    rank, local_rank, world = env_world()
    if os.environ.get("RF_SINGLE_DEVICE"):
        local_rank = 0
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = os.environ.get("RF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
        # (a rank that raises while the others sit in a collective must end the job, not hang it for the default 10 minutes)
        import datetime

        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=int(os.environ.get("RF_DIST_TIMEOUT_S", "300"))))
    return rank, local_rank, world


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def all_reduce_mean_(bucket: Tensor) -> Tensor:
    """In-place average of one flat bucket over all ranks (no-op for a world of 1).  One collective per
    training step; RCCL picks ring / direct over the xGMI mesh."""
    n = world_size()
    if _collectives_on():
        if dist.get_backend() == "nccl":
            # RCCL averages inside the collective: no extra 235 MB read+write pass for the 1/n scaling
            dist.all_reduce(bucket, op=dist.ReduceOp.AVG)
        else:
            dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
            bucket.mul_(1.0 / n)
    return bucket


class _MeanHandle:
    """Completion handle of an asynchronous mean all-reduce (``wait()`` makes the current stream wait for it)."""

    def __init__(self, work, bucket: Tensor, scale: Optional[float]):
        self._work, self._bucket, self._scale = work, bucket, scale

    def wait(self) -> None:
        if self._work is not None:
            self._work.wait()
            if self._scale is not None:
                self._bucket.mul_(self._scale)
            self._work = None


def all_reduce_mean_async(bucket: Tensor) -> _MeanHandle:
    """Start averaging ``bucket`` over all ranks and return immediately.  The collective runs on the backend's own
    stream after everything already enqueued on the current stream, i.e. concurrently with kernels launched
    afterwards -- used to reduce the ``rest`` gradients while the diffuse pass (which only touches ``base``) runs."""
    n = world_size()
    if not _collectives_on():
        return _MeanHandle(None, bucket, None)
    if dist.get_backend() == "nccl":
        return _MeanHandle(dist.all_reduce(bucket, op=dist.ReduceOp.AVG, async_op=True), bucket, None)
    return _MeanHandle(dist.all_reduce(bucket, op=dist.ReduceOp.SUM, async_op=True), bucket, 1.0 / n)


class _ShardHandle:
    """Completion handle of an asynchronous mean reduce-scatter; after ``wait()``, ``shard`` holds the averaged
    elements [lo, hi) of the bucket -- the ones this rank is responsible for."""

    def __init__(self, work, shard: Tensor, lo: int, hi: int, scale: Optional[float]):
        self._work, self.shard, self.lo, self.hi, self._scale = work, shard, lo, hi, scale

    def wait(self) -> None:
        if self._work is not None:
            self._work.wait()
            if self._scale is not None:
                self.shard.mul_(self._scale)
            self._work = None


# tests on a single GPU set this to push world-size-1 calls through RCCL as well (API / shape / aliasing checks)
FORCE_COLLECTIVES = False


def _collectives_on() -> bool:
    return world_size() > 1 or (FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized())


def can_shard(numel: int) -> bool:
    """Equal shards only (reduce_scatter_tensor / all_gather_into_tensor operate on equal chunks)."""
    return numel % world_size() == 0


def reduce_scatter_mean_async(bucket: Tensor, out: Optional[Tensor] = None) -> _ShardHandle:
    """Start averaging ``bucket`` over all ranks such that rank r receives chunk r, into ``out`` (a separate buffer of
    numel/N elements; allocated when not given).  Half the traffic of an all-reduce; the other half is the all-gather
    of the updated parameters (``all_gather_chunks_``) after every rank has applied the optimizer to its own chunk
    only (ZeRO stage 1)."""
    n = world_size()
    if not _collectives_on():
        return _ShardHandle(None, bucket, 0, bucket.numel(), None)
    assert bucket.numel() % n == 0
    chunk = bucket.numel() // n
    lo = rank() * chunk
    if out is None:
        out = torch.empty(chunk, dtype=bucket.dtype, device=bucket.device)
    assert out.numel() == chunk
    if dist.get_backend() == "nccl":
        assert bucket.is_cuda and out.is_cuda and bucket.is_contiguous() and out.is_contiguous() and out.dtype == bucket.dtype == torch.float32
        work = dist.reduce_scatter_tensor(out, bucket, op=dist.ReduceOp.AVG, async_op=True)
        return _ShardHandle(work, out, lo, lo + chunk, None)
    # gloo has no reduce-scatter: reduce everything, keep the own chunk (CPU tests of the wiring)
    dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
    out.copy_(bucket[lo : lo + chunk])
    return _ShardHandle(None, out.mul_(1.0 / n), lo, lo + chunk, None)


def all_gather_chunks_(bucket: Tensor, async_op: bool = False):
    """Every rank contributes chunk r of ``bucket`` and receives all the others (in ``bucket``).  Returns ``bucket``, or with
    ``async_op`` the work handle (RCCL; None on backends that complete on return)."""
    n = world_size()
    work = None
    if _collectives_on():
        assert bucket.dim() == 1 and bucket.is_contiguous() and bucket.numel() % n == 0, (tuple(bucket.shape), bucket.is_contiguous(), n)
        chunk = bucket.numel() // n
        lo = rank() * chunk
        mine = bucket[lo : lo + chunk].clone()  # separate send buffer: no aliasing with the receive buffer
        if dist.get_backend() == "nccl":
            assert bucket.is_cuda and mine.numel() * n == bucket.numel() and mine.dtype == bucket.dtype
            work = dist.all_gather_into_tensor(bucket, mine, async_op=async_op)
        else:
            parts = [torch.empty(chunk, dtype=bucket.dtype, device=bucket.device) for _ in range(n)]
            dist.all_gather(parts, mine)
            for r, part in enumerate(parts):
                bucket[r * chunk : (r + 1) * chunk].copy_(part)
    return (work if async_op else bucket)


def all_gather_rows_equal(out: Tensor, mine: Tensor, async_op: bool = False):
    """``out`` [N, ...] <- every rank's ``mine`` [...] (same shape on all ranks), row r from rank r.  Returns the work handle when
    ``async_op`` (None when there is nothing to wait for)."""
    n = world_size()
    assert out.shape[0] == n and out[0].shape == mine.shape and out.is_contiguous() and mine.is_contiguous()
    if not _collectives_on():
        out[0].copy_(mine)
        return None
    if dist.get_backend() == "nccl":
        assert out.is_cuda and mine.is_cuda and out.dtype == mine.dtype and out.numel() == n * mine.numel()
        work = dist.all_gather_into_tensor(out, mine, async_op=async_op)
        return work if async_op else None
    parts = [out[r] for r in range(n)]  # gloo: list form (rows of a contiguous tensor are contiguous views)
    dist.all_gather(parts, mine)
    return None


def exchange_slices(send: list, recv: list, async_op: bool = False):
    """Personalised exchange: ``send[d]`` (a slice of this rank's data, any length, slices may overlap) goes to rank d and
    ``recv[s]`` (pre-sized by the caller, who knows every count) receives what rank s sends here.  Entries for the own rank are
    ignored (pass empty tensors).  RCCL: one grouped send/recv per pair (``all_to_all``) straight out of / into the given views
    over xGMI -- with ``async_op`` the work handle is returned (the exchange is ordered after the CURRENT stream's work and runs
    beside whatever is enqueued next); other backends (CPU tests of the wiring): staged through host memory, complete on return
    (None)."""
    n, me = world_size(), rank()
    assert len(send) == n and len(recv) == n
    if not _collectives_on():
        return None
    if dist.get_backend() == "nccl":
        empty = send[me][:0]
        ins = [empty if d == me else send[d] for d in range(n)]
        outs = [recv[me][:0] if s_ == me else recv[s_] for s_ in range(n)]
        # (every view contiguous, on the device, float32 rows of ONE width: RCCL moves bytes, a dtype or width mismatch between the
        # two ends of a pair would go unnoticed)
        width = tuple(send[me].shape[1:])
        for t_ in ins + outs:
            assert t_.is_cuda and t_.is_contiguous() and t_.dtype == torch.float32 and tuple(t_.shape[1:]) == width, (tuple(t_.shape), t_.dtype, width)
        work = dist.all_to_all(outs, ins, async_op=async_op)
        return work if async_op else None
    # gloo: all_to_all_single on host copies (gloo has no list all_to_all and no device all_to_all)
    width = tuple(send[0].shape[1:])
    ins = [send[d][:0] if d == me else send[d] for d in range(n)]
    packed = torch.cat([t.reshape((-1,) + width) for t in ins], dim=0).cpu()
    in_split = [0 if d == me else int(send[d].shape[0]) for d in range(n)]
    out_split = [0 if s_ == me else int(recv[s_].shape[0]) for s_ in range(n)]
    got = torch.empty((sum(out_split),) + width, dtype=packed.dtype)
    dist.all_to_all_single(got, packed, output_split_sizes=out_split, input_split_sizes=in_split)
    pos = 0
    for s_ in range(n):
        if out_split[s_]:
            recv[s_].copy_(got[pos : pos + out_split[s_]])
            pos += out_split[s_]
    return None


# ---- the per-step collectives of the owner-computes step, without torch.distributed's Python wrappers -------------------------------
# dist.all_gather_into_tensor / dist.all_to_all spend ~25-30 us per call in argument checks and group look-ups; the data-parallel step
# makes nine of them and is paced by its host side.  These go to the ProcessGroup object directly (the calls the wrappers end in),
# always asynchronously: the collective is ordered after the work already enqueued on the CURRENT stream and runs on the backend's own
# stream; ``work.wait()`` makes the then-current stream wait for it.  RCCL only; arguments are checked on the first calls.
_FAST = {"group": None, "ag": None, "a2a": None, "checks": 64}
# RF_DIST_FAST=0: the owner-computes step goes through torch.distributed's public functions only (all_gather_into_tensor with a staging
# copy, all_to_all) -- the conservative configuration bench.py falls back to before it gives up on the owner-computes exchange
FAST_COLLECTIVES = os.environ.get("RF_DIST_FAST", "1") != "0"


def _fast_group():
    if _FAST["group"] is None or _FAST["group"] is not dist.group.WORLD:
        _FAST["group"] = dist.group.WORLD
        _FAST["ag"], _FAST["a2a"] = _c10d.AllgatherOptions(), _c10d.AllToAllOptions()
        _FAST["ag"].asyncOp = True
        _FAST["a2a"].asyncOp = True
        _FAST["checks"] = 64
    return _FAST["group"]


def fast_path() -> bool:
    """True when the collectives above can be used: an initialised RCCL group."""
    return dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"


def fast_all_gather_in_place(bucket: Tensor):
    """``all_gather_chunks_`` without the staging copy: rank r's chunk of ``bucket`` is sent from where it lies (RCCL's in-place
    all-gather: the send buffer IS chunk r of the receive buffer).  Returns the work handle."""
    group = _fast_group()
    n, r = group.size(), group.rank()
    chunk = bucket.numel() // n
    if _FAST["checks"] > 0:
        _FAST["checks"] -= 1
        assert bucket.is_cuda and bucket.dim() == 1 and bucket.is_contiguous() and chunk * n == bucket.numel(), (tuple(bucket.shape), n)
    return group._allgather_base(bucket, bucket[r * chunk : (r + 1) * chunk], _FAST["ag"])


def fast_all_gather_rows(out: Tensor, mine: Tensor):
    group = _fast_group()
    if _FAST["checks"] > 0:
        _FAST["checks"] -= 1
        assert out.is_cuda and mine.is_cuda and out.is_contiguous() and mine.is_contiguous() and out.dtype == mine.dtype and out.numel() == group.size() * mine.numel()
    return group._allgather_base(out, mine, _FAST["ag"])


def fast_exchange(outs: list, ins: list):
    """``exchange_slices`` (list form of all_to_all: one grouped send/recv per pair straight out of / into the given views); the
    entries for the own rank must be empty views."""
    group = _fast_group()
    if _FAST["checks"] > 0:
        _FAST["checks"] -= 1
        n, me = group.size(), group.rank()
        assert len(outs) == n and len(ins) == n and outs[me].numel() == 0 and ins[me].numel() == 0
        width = tuple(ins[me].shape[1:])
        for t_ in ins + outs:
            assert t_.is_cuda and t_.is_contiguous() and t_.dtype == torch.float32 and tuple(t_.shape[1:]) == width, (tuple(t_.shape), t_.dtype, width)
    return group.alltoall(outs, ins, _FAST["a2a"])


def broadcast_(tensor: Tensor, src: int = 0) -> Tensor:
    if world_size() > 1:
        dist.broadcast(tensor, src=src)
    return tensor


def shard_range(total: int, rank_: Optional[int] = None, world: Optional[int] = None) -> tuple:
    """[start, stop) of this rank's contiguous shard of ``total`` items (remainder spread over the first
    ranks) -- used to split the rays of a full-image render."""
    r = rank() if rank_ is None else rank_
    w = world_size() if world is None else world
    base, rem = divmod(total, w)
    start = r * base + min(r, rem)
    return start, start + base + (1 if r < rem else 0)


def all_gather_rows(local: Tensor, sizes: Optional[list] = None) -> Tensor:
    """Concatenate per-rank row blocks (possibly of different lengths) along dim 0 on every rank.  ``sizes`` = every rank's row
    count when the caller knows them (the shards of a frame: ``shard_range``) -- the exchange is then ONE all-gather of equal padded
    blocks, with no host round trip; without it the counts are exchanged first."""
    w = world_size()
    if w == 1:
        return local
    if sizes is None:
        counts = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(w)]
        dist.all_gather(counts, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device))
        sizes = [int(s.item()) for s in counts]
    assert len(sizes) == w and sizes[rank()] == local.shape[0], (sizes, rank(), tuple(local.shape))
    width = max(sizes)
    if local.shape[0] == width:
        padded = local.contiguous()
    else:
        padded = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
    every = torch.empty((w, width) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    all_gather_rows_equal(every, padded)
    if all(s == width for s in sizes):
        return every.reshape((w * width,) + tuple(local.shape[1:]))
    return torch.cat([every[r, :s] for r, s in enumerate(sizes)], dim=0)


def replica_checksum(flat: Tensor) -> Tensor:
    """[3] float64 fingerprint of a parameter buffer: sum, sum of squares, and a position-weighted sum (a permutation of equal
    values moves it)."""
    x = flat.detach().reshape(-1).to(torch.float64)
    w = torch.arange(x.numel(), dtype=torch.float64, device=x.device).remainder_(8191.0).add_(1.0)
    return torch.stack([x.sum(), (x * x).sum(), (x * w).sum()])


def replicas_identical(flat: Tensor) -> bool:
    """True when every rank holds the same ``flat`` (compared through ``replica_checksum``; always True for a world of 1)."""
    if not _collectives_on():
        return True
    mine = replica_checksum(flat)
    every = [torch.empty_like(mine) for _ in range(world_size())]
    dist.all_gather(every, mine)
    return all(torch.equal(every[0], e) for e in every)


def assert_replicas_identical(flat: Tensor, what: str) -> None:
    if not replicas_identical(flat):
        raise RuntimeError(f"data-parallel replicas diverged ({what}): the ranks hold different parameters")
