"""Multi-GPU layer: photon-ID sharding and the single tally all-reduce.

The reference is single-process, single-GPU (no collective anywhere), so nothing here has a
counterpart in it. Photons are independent and every random number is keyed on the photon ID, so
rank r of R traces the contiguous ID range ``shard(total, r, R)`` against replicated grids and the
per-rank integer tallies are summed once: the result is bit-identical for every R.

One process per GPU, ``torch.distributed`` with backend ``"nccl"`` (RCCL over xGMI) on the GPUs;
``"gloo"`` is what the CPU tests and the one-GPU rehearsals use to exercise the same code.
``ShardedTracer`` is the step ``bench.py`` times and the tests check: trace this rank's shard, then ONE
all-reduce, double-buffered so that the all-reduce of job k runs beside the trace of job k+1, with the
counts on the wire as 32-bit words whenever they fit (half the bytes over xGMI's point-to-point links).
Several GPUs in ONE process (the C++ drivers' model) use ``rt_allreduce`` of the C ABI instead.
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


def _dist_state(group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group), dist.get_backend(group)
    return 0, 1, None


def allreduce_tallies(tally_tensor, group=None):
    """Sum the concatenated tally buffer (int64 view of the uint64 counts; two's-complement sums
    wrap identically) over all ranks, in place. No-op when torch.distributed is not initialised
    or the world size is 1."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return tally_tensor
    dist.all_reduce(tally_tensor, op=dist.ReduceOp.SUM, group=group)
    return tally_tensor


def _follow_torch_stream(tracer, tensor):
    """torch's collectives are ordered against torch's CURRENT stream; the tracer launches on its own. Make them
    the same stream, so that the all-reduce cannot read a buffer the tracer is still adding to."""
    if getattr(tensor, "is_cuda", False) and hasattr(tracer, "set_stream"):
        import torch
        tracer.set_stream(torch.cuda.current_stream(tensor.device).cuda_stream)


def trace_sharded(tracer, tally_tensor, seed: int, total_photons: int, first_photon_id: int = 0, group=None):
    """Trace this rank's shard of `total_photons` IDs starting at `first_photon_id` on `tracer`
    (whose tallies are bound to `tally_tensor`), then all-reduce. The tracer is put on torch's current
    stream first (see _follow_torch_stream). Returns (begin, count)."""
    rank, world, _ = _dist_state(group)
    begin, count = shard(total_photons, rank, world)
    _follow_torch_stream(tracer, tally_tensor)
    tracer.trace(seed, first_photon_id + begin, count)
    allreduce_tallies(tally_tensor, group)
    return first_photon_id + begin, count


# Resident waves per CU of the pool kernel that leave 160 - 22 * 6.1 = 26 KiB of every CU's LDS free (option
# reserve_lds, off by default). Round 2 first capped the tracer at this whenever a collective could be pending, after
# a probe with an LDS-only stand-in kernel (profiles/r1_overlap_probe.txt). The metadata of RCCL's own gfx950 code
# objects says that cannot work: rcclGenericKernel, the kernel behind every ring/tree collective, asks for 19 744 B of
# LDS AND 261-280 registers per lane (profiles/r2_rccl_kernel_resources.txt), so its 4-wave workgroup needs every SIMD
# of a CU down to 3 pool waves (12 per CU), not 22. The cap cost the trace 4 % and bought nothing, so the overlap
# rests on order instead: the all-reduce becomes ready when the narrowing kernel ends, the next trace only after the
# 2 GiB tally memset that follows it in the stream (about 0.5 ms). RCCL's stream stays at default priority: a
# high-priority one changed nothing in tools/overlap_probe.py and hung the world-size-1 RCCL test once.
POOL_WAVES_BESIDE_A_COLLECTIVE = 22


class ShardedTracer:
    """One rank's side of a multi-GPU job stream. `tracer` is a PhotonTracer (or anything with its
    nwords / bind_tally_tensor / reset / trace / narrow_tallies / widen_tallies); `device` is where the tally
    buffers live (the tracer's GPU, or "cpu" for a CPU stand-in).

    step(seed, total, first_id) zeroes a tally buffer, traces this rank's shard of the job into it and starts the
    job's ONE all-reduce; result(handle) returns the buffer holding the node total. With `overlap` (default for
    nccl at world size > 1) two buffers alternate and the all-reduce is asynchronous: it runs beside the next
    step's trace, and a step waits only for the all-reduce that used its buffer two steps earlier. With `wire32`
    the counts cross the wire as 32-bit words (narrow -> all-reduce; result() widens); if any rank holds a count too
    large for that (the narrowing kernel checks on the device, the flag is summed with the data) result() repeats the
    exchange on the 64-bit buffer, which the widening kernel then has left untouched. With `atmos32` (default) the
    tracer counts absorptions straight into the wire, so only the five 2-D arrays are narrowed at all. `reserve_lds` caps the pool
    kernel at POOL_WAVES_BESIDE_A_COLLECTIVE waves per CU while overlapping (an experiment, see above: off)."""

    def __init__(self, tracer, device, group=None, overlap=None, wire32=True, force_collective=False,
                 reserve_lds=False, atmos32=True):
        import torch
        self.torch = torch
        self.tracer, self.group = tracer, group
        self.device = torch.device(device)
        self.rank, self.world, self.backend = _dist_state(group)
        self.collective = self.world > 1 or (force_collective and self.backend is not None)
        # gloo cannot reduce device memory: stage through the host (rehearsals and tests on one GPU)
        self.stage_through_host = self.collective and self.backend == "gloo" and self.device.type == "cuda"
        if overlap is None:
            overlap = self.collective and self.backend == "nccl"
        self.overlap = bool(overlap) and self.collective and not self.stage_through_host
        self.wire32 = bool(wire32) and self.collective and not self.stage_through_host
        nbuf = 2 if self.overlap else 1
        n = tracer.nwords
        self.tallies = [torch.zeros(n, dtype=torch.int64, device=self.device) for _ in range(nbuf)]
        self.wire = [torch.zeros(n + 1, dtype=torch.int32, device=self.device) for _ in range(nbuf)] if self.wire32 else None
        # With 32-bit counts on the wire the tracer adds its absorptions straight into the wire (rt_bind_atmos_wire: 98 %
        # of the words need no narrowing pass, half the bytes to zero per job) for every job of fewer than 2^32 photons
        # over all ranks -- a photon is absorbed once, so no absorption word, and no sum of one over the ranks, can
        # wrap; the widening back to 64-bit counts is left to result(), the only reader (a bench that reads the last
        # job's result pays for it once, not per step).
        self.atmos32 = bool(atmos32) and self.wire32 and hasattr(tracer, "bind_atmos_wire")
        self.native = [False] * nbuf          # this buffer's absorption counts live in its wire
        self.widened = [True] * nbuf
        self.pending = [None] * nbuf
        self.exchanged = [False] * nbuf
        self.steps = 0
        self.step_stream = [None] * nbuf      # the stream each buffer's step was enqueued on (checked by result())
        self.timeout_s = float(__import__("os").environ.get("RT_COLLECTIVE_TIMEOUT_S", "600"))
        _follow_torch_stream(tracer, self.tallies[0])
        if self.overlap and reserve_lds and hasattr(tracer, "set_option"):
            tracer.set_option("pool_waves_per_cu", POOL_WAVES_BESIDE_A_COLLECTIVE)
        tracer.bind_tally_tensor(self.tallies[0])

    def span(self, total_photons, first_photon_id=0):
        begin, count = shard(total_photons, self.rank, self.world)
        return first_photon_id + begin, count

    def step(self, seed, total_photons, first_photon_id=0, before_trace=None, after_trace=None):
        """One job: returns the handle of its buffer (valid until the step after next reuses it)."""
        import torch.distributed as dist
        b = self.steps % len(self.tallies)
        self.steps += 1
        # The collective is ordered against torch's CURRENT stream at the moment of the call, so the tracer follows it
        # on every step (not once at construction): a caller inside `with torch.cuda.stream(s)` gets trace, narrowing,
        # all-reduce, widening and reset all on s. A buffer stays with the stream of its step until its result is in.
        _follow_torch_stream(self.tracer, self.tallies[b])
        self._finish(b)
        self.step_stream[b] = self._current_stream_id()
        tally = self.tallies[b]
        if len(self.tallies) > 1:
            self.tracer.bind_tally_tensor(tally)
        if self.atmos32:
            self.native[b] = total_photons < (1 << 32)
            self.tracer.bind_atmos_wire(self.wire[b] if self.native[b] else None)
        self.tracer.reset()
        begin, count = self.span(total_photons, first_photon_id)
        if before_trace is not None:
            before_trace()
        self.tracer.trace(seed, begin, count)
        if after_trace is not None:
            after_trace()
        self.exchanged[b] = False
        self.widened[b] = True
        if not self.collective:
            return b
        if self.stage_through_host:
            host = tally.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            tally.copy_(host)
            self.exchanged[b] = True
            return b
        buf = tally
        if self.wire32:
            buf = self.wire[b]
            self.tracer.narrow_tallies(buf, self.world, tally)
        work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=self.overlap)
        if self.overlap:
            self.pending[b] = work
        else:
            self._after_exchange(b)
        return b

    def _after_exchange(self, b):
        # (the sums are in the wire; result() widens them into the 64-bit buffer if anyone asks: a buffer that is only
        # zeroed again two steps later is never widened)
        self.widened[b] = not self.wire32
        self.exchanged[b] = True

    def _current_stream_id(self):
        if self.device.type != "cuda":
            return None
        return int(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _finish(self, b):
        if self.pending[b] is not None:
            # work.wait() makes torch's current stream wait for the collective (the host does not block), and the
            # widening kernel is enqueued behind it on the tracer's stream: both must be the stream the step ran on,
            # or the widening could run ahead of the all-reduce it depends on
            now = self._current_stream_id()
            if self.step_stream[b] is not None and now != self.step_stream[b]:
                raise RuntimeError(f"ShardedTracer: step of buffer {b} was enqueued on stream {self.step_stream[b]:#x}, its "
                                   f"result is asked for under stream {now:#x}; call step() and result() under one stream")
            _follow_torch_stream(self.tracer, self.tallies[b])
            self.pending[b].wait()
            self.pending[b] = None
            self._after_exchange(b)

    def _host_wait(self, what):
        """Blocks the host until everything enqueued so far on the current stream is done -- with a bound: a
        cross-stream wait that never completes becomes an error that says what was pending, not a silent hang
        (RT_COLLECTIVE_TIMEOUT_S, default 600)."""
        if self.device.type != "cuda":
            return
        import time
        ev = self.torch.cuda.Event()
        ev.record(self.torch.cuda.current_stream(self.device))
        t0, pause = time.monotonic(), 1e-5
        while not ev.query():
            if time.monotonic() - t0 > self.timeout_s:
                state = [f"buffer {b}: all-reduce {'pending' if w is not None and not w.is_completed() else 'done'}"
                         for b, w in enumerate(self.pending)]
                raise RuntimeError(f"ShardedTracer: {what} not complete after {self.timeout_s:.0f} s on stream "
                                   f"{self._current_stream_id():#x} ({'; '.join(state)})")
            time.sleep(pause)
            pause = min(pause * 2, 1e-3)

    def drain(self):
        """Every started all-reduce is complete (in stream order) when this returns."""
        for b in range(len(self.tallies)):
            self._finish(b)

    def result(self, handle):
        """The buffer of step `handle` with the node total in it (synchronises this rank's stream)."""
        import torch.distributed as dist
        self._finish(handle)
        tally = self.tallies[handle]
        if self.wire32 and self.exchanged[handle] and not self.widened[handle]:
            if self.atmos32:        # the widening kernel must know whether this wire holds native absorption counts
                self.tracer.bind_atmos_wire(self.wire[handle] if self.native[handle] else None)
            self.tracer.widen_tallies(self.wire[handle], tally)
            self.widened[handle] = True
        self._host_wait(f"result of buffer {handle} (trace, all-reduce, widening)")
        if self.wire32 and self.exchanged[handle] and int(self.wire[handle][-1].item()) != 0:
            # a count too large for 32 bits somewhere: every rank sees the same flag sum and repeats in 64 bits -- the
            # whole buffer, or the five 2-D arrays when the absorption counts were native to the wire (widened above)
            part = tally[:5 * self.tracer.ncol] if (self.atmos32 and self.native[handle]) else tally
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group)
            self.wire[handle][-1] = 0       # done: a second result() must not add again
        return tally
