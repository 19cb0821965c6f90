"""N > 1 path on CPU: photon-ID sharding + the single tally all-reduce, world_size 2 over gloo.
The per-rank 'device' work is done by the CPU oracle here (test stand-in for the GPU kernel; the
product's host logic under test is raytracer_amd.distributed)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from raytracer_amd import distributed as D
from raytracer_amd import scenes, split_tally_words


def test_shard_covers_every_id_once():
    for total in (0, 1, 7, 1 << 20, (1 << 30) + 3):
        for world in (1, 2, 3, 8):
            spans = [D.shard(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (b0, c0), (b1, _) in zip(spans, spans[1:]):
                assert b0 + c0 == b1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(ValueError):
        D.shard(10, 2, 2)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _OracleTracer:
    """Same interface as PhotonTracer.trace, accumulating into a bound int64 tensor."""

    def __init__(self, scene, tensor):
        self.scene, self.tensor = scene, tensor

    def trace(self, seed, begin, count):
        t, _ = O.photon_oracle(self.scene, seed, begin, count, nthreads=1)
        words = np.concatenate([t[n].ravel() for n in O.TALLY_NAMES]).view(np.int64)
        self.tensor += torch.from_numpy(words.copy())


def _worker(rank, world, port, total, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scenes.tiny()
    nwords = 5 * sc.ncol + 2 * sc.ncell
    tally = torch.zeros(nwords, dtype=torch.int64)
    begin, count = D.trace_sharded(_OracleTracer(sc, tally), tally, seed=21, total_photons=total, first_photon_id=5)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), tally.numpy())
    np.save(os.path.join(out_dir, f"span{rank}.npy"), np.array([begin, count]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_reproduce_single_process_tallies(tmp_path):
    total, world = 20001, 2
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    sc = scenes.tiny()
    whole, _ = O.photon_oracle(sc, 21, 5, total, nthreads=0)
    r0, r1 = (np.load(tmp_path / f"rank{r}.npy") for r in range(2))
    assert np.array_equal(r0, r1)                       # all-reduce leaves the node total on every rank
    got = split_tally_words(r0, sc.itot, sc.jtot, sc.ktot)
    for n in O.TALLY_NAMES:
        assert np.array_equal(got[n], whole[n]), n
    spans = [np.load(tmp_path / f"span{r}.npy") for r in range(2)]
    assert spans[0][0] == 5 and spans[0][0] + spans[0][1] == spans[1][0] and spans[0][1] + spans[1][1] == total


class _CpuTracer:
    """Stand-in with PhotonTracer's interface as ShardedTracer uses it, on CPU tensors: the oracle traces, and
    narrow / widen restate rt_narrow_kernel / rt_widen_kernel (the GPU tests check the kernels themselves)."""

    def __init__(self, scene, extra=0, atmos32=True):
        self.scene, self.nwords, self.tensor, self.extra = scene, 5 * scene.ncol + 2 * scene.ncell, None, extra
        self.ncol, self.atm = scene.ncol, None
        if atmos32:     # restates rt_bind_atmos_wire: absorptions counted straight into the wire's 32-bit words
            self.bind_atmos_wire = self._bind_atmos_wire

    def bind_tally_tensor(self, t):
        self.tensor = t

    def _bind_atmos_wire(self, wire):
        self.atm = wire

    def reset(self):
        if self.atm is None:
            self.tensor.zero_()
        else:
            self.tensor[:5 * self.ncol].zero_()
            self.tensor[5 * self.ncol:] = -12345        # stale in this mode: nothing may read it before the widening
            self.atm[5 * self.ncol:].zero_()

    def trace(self, seed, begin, count):
        t, _ = O.photon_oracle(self.scene, seed, begin, count, nthreads=1)
        words = torch.from_numpy(np.concatenate([t[n].ravel() for n in O.TALLY_NAMES]).view(np.int64).copy())
        n2 = 5 * self.ncol
        if self.atm is None:
            self.tensor += words
        else:
            self.tensor[:n2] += words[:n2]
            self.atm[n2:-1] += words[n2:].to(torch.int32)
        self.tensor[3] += self.extra               # a count that may not fit 32 bits on the wire

    def narrow_tallies(self, wire, world, tally=None):
        t = tally if tally is not None else self.tensor
        guard = 32 - int(np.ceil(np.log2(world))) if world > 1 else 32
        n = 5 * self.ncol if wire is self.atm else self.nwords
        wire[:n] = torch.from_numpy((t.numpy()[:n] & 0xFFFFFFFF).astype(np.uint32).view(np.int32))
        wire[-1] = int(bool(((t[:n] >> guard) != 0).any()))

    def widen_tallies(self, wire, tally=None):
        t = tally if tally is not None else self.tensor
        first = 0 if int(wire[-1]) == 0 else (5 * self.ncol if wire is self.atm else self.nwords)
        t[first:] = torch.from_numpy(wire[first:-1].numpy().view(np.uint32).astype(np.int64))


def _sharded_worker(rank, world, port, out_dir, overlap, wire32, extra, atmos32=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scenes.tiny()
    st = D.ShardedTracer(_CpuTracer(sc, extra if rank == 1 else 0, atmos32), "cpu", overlap=overlap, wire32=wire32)
    assert st.collective and st.overlap == overlap and st.wire32 == wire32 and len(st.tallies) == (2 if overlap else 1)
    assert st.atmos32 == (wire32 and atmos32)
    jobs = [(31, 3001, 7), (32, 2000, 100000), (33, 4099, 0)]          # (seed, photons, first id)
    handles = []
    for k, (seed, total, first) in enumerate(jobs):
        handles.append(st.step(seed, total, first))
        if k >= 1 or not overlap:            # a buffer is reused two steps later: read it before that
            kk = k - 1 if overlap else k
            np.save(os.path.join(out_dir, f"job{kk}_rank{rank}.npy"), st.result(handles[kk]).numpy().copy())
    if overlap:
        np.save(os.path.join(out_dir, f"job{len(jobs) - 1}_rank{rank}.npy"), st.result(handles[-1]).numpy().copy())
    st.drain()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap,wire32,extra,atmos32", [(False, False, 0, True), (True, True, 0, True), (False, True, 0, True),
                                                          (True, True, 1 << 31, True), (True, False, 1 << 33, True),
                                                          (True, True, 0, False), (False, True, 1 << 31, False)])
def test_sharded_tracer_steps_reproduce_single_process_tallies(tmp_path, overlap, wire32, extra, atmos32):
    """ShardedTracer (the step bench.py times): three jobs over two ranks, with and without the double-buffered
    asynchronous all-reduce, with 32-bit counts on the wire -- including a count of 2^31 on one rank, which the
    narrowing step must flag so that the exchange is repeated in 64 bits -- and with the absorption counts native to
    the wire (rt_bind_atmos_wire) or narrowed like the rest."""
    world = 2
    mp.spawn(_sharded_worker, args=(world, _free_port(), str(tmp_path), overlap, wire32, extra, atmos32), nprocs=world, join=True)
    sc = scenes.tiny()
    for k, (seed, total, first) in enumerate([(31, 3001, 7), (32, 2000, 100000), (33, 4099, 0)]):
        whole, _ = O.photon_oracle(sc, seed, first, total, nthreads=0)
        want = np.concatenate([whole[n].ravel() for n in O.TALLY_NAMES]).view(np.int64).copy()
        want[3] += extra
        r0, r1 = (np.load(tmp_path / f"job{k}_rank{r}.npy") for r in range(2))
        assert np.array_equal(r0, r1) and np.array_equal(r0, want), (k, np.flatnonzero(r0 != want)[:5])


def test_allreduce_is_noop_without_process_group():
    t = torch.arange(5, dtype=torch.int64)
    assert D.allreduce_tallies(t) is t and t.tolist() == [0, 1, 2, 3, 4]
