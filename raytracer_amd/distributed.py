"""Multi-GPU layer: photon-ID sharding and the single tally all-reduce.

The reference is single-process, single-GPU (no collective anywhere), so nothing here has a
counterpart in it. Photons are independent and every random number is keyed on the photon ID, so
rank r of R traces the contiguous ID range ``shard(total, r, R)`` against replicated grids and the
per-rank integer tallies are summed once: the result is bit-identical for every R.

One process per GPU, ``torch.distributed`` with backend ``"nccl"`` (RCCL over xGMI) on the GPUs;
``"gloo"`` on CPU tensors is what the CPU tests use to exercise the same code.
"""
from __future__ import annotations


def shard(total: int, rank: int, world: int):
    """Contiguous photon-ID range [begin, begin+count) of `rank`: floor split, remainder to the
    lowest ranks, every ID in [0,total) owned exactly once."""
    if world <= 0 or not (0 <= rank < world) or total < 0:
        raise ValueError(f"shard({total}, {rank}, {world})")
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, base + (1 if rank < rem else 0)


def allreduce_tallies(tally_tensor, group=None):
    """Sum the concatenated tally buffer (int64 view of the uint64 counts; two's-complement sums
    wrap identically) over all ranks, in place. No-op when torch.distributed is not initialised
    or the world size is 1."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return tally_tensor
    dist.all_reduce(tally_tensor, op=dist.ReduceOp.SUM, group=group)
    return tally_tensor


def trace_sharded(tracer, tally_tensor, seed: int, total_photons: int, first_photon_id: int = 0, group=None):
    """Trace this rank's shard of `total_photons` IDs starting at `first_photon_id` on `tracer`
    (whose tallies are bound to `tally_tensor`), then all-reduce. Returns (begin, count)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    begin, count = shard(total_photons, rank, world)
    tracer.trace(seed, first_photon_id + begin, count)
    allreduce_tallies(tally_tensor, group)
    return first_photon_id + begin, count
