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


def test_allreduce_is_noop_without_process_group():
    t = torch.arange(5, dtype=torch.int64)
    assert D.allreduce_tallies(t) is t and t.tolist() == [0, 1, 2, 3, 4]
