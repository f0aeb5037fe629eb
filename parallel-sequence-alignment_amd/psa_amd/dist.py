"""One-process-per-GPU plumbing around libpsa_hip (torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" is used by the CPU tests of this module).

The reference splits the Seq2 list between exactly two MPI ranks by count and sends the
formatted result lines back to rank 0 (P/main.c:62-94, P/cFunctions.c:49-52).  Here any
number of ranks takes part:

* batch mode -- Seq2 are independent units: each rank scores the Seq2 that
  psa_partition assigns to it (cost-balanced), no collective touches the data path, and
  the 96-byte result records are gathered afterwards in input order;
* long-pair mode -- one Seq2 against a very long Seq1: each rank scores a contiguous
  range of offsets, and one all-gather of one record per rank followed by the max-loc
  combine (highest score, ties to the smallest (n, k, offset)) gives the answer.
"""
import ctypes as C

import numpy as np

from . import Result, combine_results, partition

RECORD_BYTES = C.sizeof(Result)


def _dist():
    import torch.distributed as dist
    return dist


def _device(group=None):
    import torch
    dist = _dist()
    backend = dist.get_backend(group)
    return torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")


def my_share(len2, len1, rank, world, stages=0):
    """Indices (ascending) of the Seq2 this rank scores (split balanced for the stages that run)."""
    owner = partition(len2, len1, world, stages)
    return [i for i in range(len(len2)) if owner[i] == rank], owner


def offset_shard(num_offsets, rank, world):
    """Contiguous range [lo, hi) of offsets of one long pair owned by this rank."""
    return num_offsets * rank // world, num_offsets * (rank + 1) // world


def records_to_array(results):
    if not results:
        return np.zeros((0, RECORD_BYTES), dtype=np.uint8)
    return np.stack([np.frombuffer(bytes(r), dtype=np.uint8) for r in results])


def array_to_records(arr):
    return [Result.from_buffer_copy(bytes(row)) for row in np.asarray(arr, dtype=np.uint8)]


def gather_results(local_results, my_indices, n_total, group=None):
    """Every rank passes the records of its own Seq2 (in the order of my_indices) and gets
    back all n_total records in input order."""
    import torch
    dist = _dist()
    world = dist.get_world_size(group)
    dev = _device(group)
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    counts[dist.get_rank(group)] = len(my_indices)
    dist.all_reduce(counts, group=group)
    cap = int(counts.max().item())
    buf = torch.zeros((cap, RECORD_BYTES + 8), dtype=torch.uint8, device=dev)
    if my_indices:
        rec = records_to_array(local_results)
        idx = np.asarray(my_indices, dtype=np.int64).view(np.uint8).reshape(-1, 8)
        buf[:len(my_indices)] = torch.from_numpy(np.concatenate([rec, idx], axis=1)).to(dev)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    out = [None] * n_total
    for r in range(world):
        rows = parts[r][:int(counts[r].item())].cpu().numpy()
        for row in rows:
            i = int(row[RECORD_BYTES:].copy().view(np.int64)[0])
            out[i] = Result.from_buffer_copy(bytes(row[:RECORD_BYTES]))
    assert all(o is not None for o in out), "a Seq2 was scored by no rank"
    return out


def allgather_bytes(group=None):
    """The all-gather callback of Context.score_range_allgather over torch.distributed: every rank
    passes the same number of bytes and gets the concatenation in rank order.  The exchanges of a long
    pair sit on its step's critical path (the shards wait for the maximum over all shards before they
    shortlist), so the buffers of each message size -- device tensors for RCCL, pinned staging on the
    host -- are made once and the gather is one collective into one tensor."""
    import torch
    dist = _dist()
    world = dist.get_world_size(group)
    dev = _device(group)
    cache = {}

    def buffers(n):
        b = cache.get(n)
        if b is None:
            host_in = torch.empty(n, dtype=torch.uint8)
            host_out = torch.empty(world * n, dtype=torch.uint8)
            if dev.type == "cuda":
                host_in, host_out = host_in.pin_memory(), host_out.pin_memory()
                b = (host_in, host_out, torch.empty(n, dtype=torch.uint8, device=dev),
                     torch.empty(world * n, dtype=torch.uint8, device=dev))
            else:
                b = (host_in, host_out, host_in, host_out)
            cache[n] = b
        return b

    def fn(send):
        n = len(send)
        host_in, host_out, mine, out = buffers(n)
        host_in.numpy()[:] = np.frombuffer(send, dtype=np.uint8)
        if mine is not host_in:
            mine.copy_(host_in, non_blocking=True)
        try:
            dist.all_gather_into_tensor(out, mine, group=group)
        except (RuntimeError, NotImplementedError):  # a backend without the single-tensor form
            parts = list(out.view(world, n).unbind(0))
            dist.all_gather(parts, mine, group=group)
        if out is not host_out:
            host_out.copy_(out)  # (waits for the collective on the current stream)
        return host_out.numpy().tobytes()
    return fn


def score_long_pair(ctx, seq2, stages=3, group=None):
    """One long pair on all ranks of the group: offsets sharded, one all-gather of one record per
    rank inside the library call, the same combined record on every rank."""
    dist = _dist()
    return ctx.score_range_allgather(seq2, dist.get_rank(group), dist.get_world_size(group),
                                     allgather_bytes(group), stages)


def combine_shards(local_result, group=None):
    """Max-loc all-reduce of the long-pair mode: one all-gather of one record per rank, then the
    same deterministic combine on every rank (psa_combine_results)."""
    import torch
    dist = _dist()
    world = dist.get_world_size(group)
    dev = _device(group)
    mine = torch.from_numpy(records_to_array([local_result])[0].copy()).to(dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    return combine_results(array_to_records(torch.stack(parts).cpu().numpy()))
