"""The multi-GPU path with the HIP kernel as the per-rank worker (the CPU tests use the oracle in its place):
two ranks on the one GPU of the test box over gloo, the RCCL calls at world size 1, the 32-bit wire kernels, and
rt_allreduce of the C ABI. Summed tallies must be bit-identical to one call."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

JOBS = [(31, 200_001, 7), (32, 1 << 17, 5_000_000_000), (33, 77_777, 0)]      # (seed, photons, first id)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _words(t):
    return np.concatenate([t[n].ravel() for n in O.TALLY_NAMES]).view(np.int64)


def _single_call(rt, scene, **opts):
    out = []
    tr = rt.PhotonTracer.from_scene(scene, **opts)
    for seed, total, first in JOBS:
        tr.reset(); tr.trace(seed, first, total); tr.synchronize()
        out.append(_words(tr.tallies()).copy())
    tr.close()
    return out


def _rank_worker(rank, world, port, out_dir, launch_log2k):
    import torch
    import torch.distributed as dist
    import raytracer_amd as rt
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    scene = rt.scenes.tiny()
    tr = rt.PhotonTracer.from_scene(scene, device=0, launch_log2k=launch_log2k)
    st = rt.distributed.ShardedTracer(tr, "cuda:0")
    assert st.collective and st.stage_through_host and not st.overlap
    for k, (seed, total, first) in enumerate(JOBS):
        h = st.step(seed, total, first)
        np.save(os.path.join(out_dir, f"job{k}_rank{rank}.npy"), st.result(h).cpu().numpy())
    tr.synchronize()
    tr.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("launch_log2k", [-1, 2])
def test_two_ranks_on_one_gpu_reproduce_a_single_call(tmp_path, launch_log2k):
    """world size 2 over gloo, both ranks on cuda:0, each tracing its shard with the HIP kernel through
    ShardedTracer (the step bench.py times): every rank ends with the tallies of one rt_trace call over the whole
    range, bit for bit."""
    import torch.multiprocessing as mp
    import raytracer_amd as rt
    world = 2
    mp.spawn(_rank_worker, args=(world, _free_port(), str(tmp_path), launch_log2k), nprocs=world, join=True)
    want = _single_call(rt, rt.scenes.tiny(), launch_log2k=launch_log2k)
    for k in range(len(JOBS)):
        r0, r1 = (np.load(tmp_path / f"job{k}_rank{r}.npy") for r in range(2))
        assert np.array_equal(r0, r1) and np.array_equal(r0, want[k]), (k, np.flatnonzero(r0 != want[k])[:5])


def test_rccl_world_size_one_overlapped_32bit_exchange(tmp_path):
    """The exact path of bench.py --gpus N: backend nccl (RCCL), two tally buffers, asynchronous all-reduce of
    32-bit counts (narrow kernel -> ncclAllReduce -> widen kernel), here at world size 1 in this process."""
    import torch
    import torch.distributed as dist
    import raytracer_amd as rt
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        scene = rt.scenes.tiny()
        tr = rt.PhotonTracer.from_scene(scene, device=0)
        st = rt.distributed.ShardedTracer(tr, "cuda:0", force_collective=True, overlap=True, reserve_lds=True)
        assert st.collective and st.overlap and st.wire32 and len(st.tallies) == 2     # (reserve_lds: the optional cap)
        got, handles = [], []
        for k, (seed, total, first) in enumerate(JOBS):
            handles.append(st.step(seed, total, first))
            if k >= 1:
                got.append(st.result(handles[k - 1]).cpu().numpy().copy())
        got.append(st.result(handles[-1]).cpu().numpy().copy())
        tr.synchronize()
        tr.close()
    finally:
        dist.destroy_process_group()
    want = _single_call(rt, rt.scenes.tiny())
    for k in range(len(JOBS)):
        assert np.array_equal(got[k], want[k]), k


def test_rccl_exchange_under_a_side_stream_and_stream_mismatch_is_refused(tmp_path):
    """the collective is ordered against torch's CURRENT stream, so ShardedTracer follows it on every step
    (not once at construction). The same overlapped 32-bit exchange under `with torch.cuda.stream(side)`, created
    AFTER the ShardedTracer: bit-identical to single calls; a result asked for under another stream than its step's is
    refused (the widening kernel would not be ordered behind the all-reduce); and the bounded host wait reports
    instead of hanging when its limit is 0."""
    import torch
    import torch.distributed as dist
    import raytracer_amd as rt
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        scene = rt.scenes.tiny()
        tr = rt.PhotonTracer.from_scene(scene, device=0)
        st = rt.distributed.ShardedTracer(tr, "cuda:0", force_collective=True, overlap=True)
        side = torch.cuda.Stream()
        got, handles = [], []
        with torch.cuda.stream(side):
            for k, (seed, total, first) in enumerate(JOBS):
                handles.append(st.step(seed, total, first))
                assert st.step_stream[handles[-1]] == side.cuda_stream != 0
                if k >= 1:
                    got.append(st.result(handles[k - 1]).cpu().numpy().copy())
        with pytest.raises(RuntimeError, match="under one stream"):
            st.result(handles[-1])                      # default stream: not the step's
        with torch.cuda.stream(side):
            st.timeout_s = 0.0
            st.step(1, 1 << 24, 0)                      # something still running: the bounded wait must say so, not hang
            with pytest.raises(RuntimeError, match="not complete after 0 s"):
                st.result(handles[-1])
            st.timeout_s = 600.0
            got.append(st.result(handles[-1]).cpu().numpy().copy())
            st.drain()
        side.synchronize()
        tr.synchronize()
        tr.close()
    finally:
        dist.destroy_process_group()
    want = _single_call(rt, rt.scenes.tiny())
    for k in range(len(JOBS)):
        assert np.array_equal(got[k], want[k]), k


def test_narrow_and_widen_kernels(tmp_path):
    """rt_tallies_narrow / rt_tallies_widen either side of a 32-bit sum over two 'ranks' (two contexts on this GPU,
    the sum done with torch): equal to the 64-bit sum; a count at or above 2^31 raises the flag and the widening
    kernel then leaves the tallies alone."""
    import torch
    import raytracer_amd as rt
    scene = rt.scenes.tiny()
    trs = [rt.PhotonTracer.from_scene(scene, device=0) for _ in range(2)]
    n = trs[0].nwords
    tal = [torch.zeros(n, dtype=torch.int64, device="cuda:0") for _ in range(2)]
    wire = [torch.zeros(n + 1, dtype=torch.int32, device="cuda:0") for _ in range(2)]
    for r, tr in enumerate(trs):
        tr.set_stream(torch.cuda.current_stream().cuda_stream)
        tr.bind_tally_tensor(tal[r])
        tr.trace(5, r * 100_000, 100_000)
    want = (tal[0] + tal[1]).cpu().numpy()
    for r, tr in enumerate(trs):
        tr.narrow_tallies(wire[r], 2)
    total = wire[0] + wire[1]
    assert int(total[-1].item()) == 0
    for r, tr in enumerate(trs):
        tr.widen_tallies(total)
        assert np.array_equal(tal[r].cpu().numpy(), want)
    # one large count on rank 1: flagged, tallies untouched by the widening
    tal[1][7] = 1 << 31
    before = tal[1].cpu().numpy().copy()
    trs[1].narrow_tallies(wire[1], 2)
    assert int(wire[1][-1].item()) == 1 and int(wire[1][7].item()) == -(1 << 31)
    trs[1].widen_tallies(wire[1])
    assert np.array_equal(tal[1].cpu().numpy(), before)
    # at world size 1 the same count is fine (guard at 2^32)
    trs[1].narrow_tallies(wire[1], 1)
    assert int(wire[1][-1].item()) == 0
    with pytest.raises(rt.RtError):
        trs[0].narrow_tallies(torch.zeros(n, dtype=torch.int32, device="cuda:0"), 2)
    for tr in trs:
        tr.synchronize(); tr.close()


def test_absorptions_counted_straight_into_the_wire(tmp_path):
    """rt_bind_atmos_wire: two 'ranks' (two contexts on this GPU) count their absorptions as 32-bit
    words in their wires while the five 2-D arrays stay 64-bit; narrowing then touches the 2-D words only, the summed
    wires widened into either context's buffer equal the 64-bit sum of two ordinary runs bit for bit -- also when a 2-D
    count raises the flag (the absorption words are widened regardless, the 2-D words left for the 64-bit repeat).
    Refused: more photons than a 32-bit word can count, and the lane-resident kernel."""
    import torch
    import raytracer_amd as rt
    scene = rt.scenes.tiny()
    n2 = 5 * scene.ncol
    ref = []
    for r in range(2):
        tr = rt.PhotonTracer.from_scene(scene, device=0)
        tr.trace(5, r * 150_000, 150_000); tr.synchronize()
        ref.append(_words(tr.tallies()).copy())
        tr.close()
    want = ref[0] + ref[1]
    trs = [rt.PhotonTracer.from_scene(scene, device=0) for _ in range(2)]
    n = trs[0].nwords
    tal = [torch.full((n,), -7, dtype=torch.int64, device="cuda:0") for _ in range(2)]       # stale words must not survive
    wire = [torch.full((n + 1,), -9, dtype=torch.int32, device="cuda:0") for _ in range(2)]
    for r, tr in enumerate(trs):
        tr.set_stream(torch.cuda.current_stream().cuda_stream)
        tr.bind_tally_tensor(tal[r]); tr.bind_atmos_wire(wire[r])
        tr.reset()
        tr.trace(5, r * 150_000, 100_000); tr.trace_noalloc(5, r * 150_000 + 100_000, 50_000)     # accumulates across calls
        tr.narrow_tallies(wire[r], 2, tal[r])
    torch.cuda.synchronize()
    for r in range(2):      # the 64-bit buffer holds the 2-D arrays only; the absorption counts are in the wire
        assert np.array_equal(tal[r][:n2].cpu().numpy(), ref[r][:n2])
        assert np.array_equal(wire[r][n2:-1].cpu().numpy().astype(np.int64), ref[r][n2:])
    total = wire[0] + wire[1]
    assert int(total[-1].item()) == 0
    for r, tr in enumerate(trs):
        wire[r].copy_(total)
        tr.widen_tallies(wire[r], tal[r])
        assert np.array_equal(tal[r].cpu().numpy(), want)
    # a 2-D count too large for the 32-bit sum on rank 1: flagged; the widening then fills in the absorption words only
    tr = trs[1]
    tr.reset(); tr.trace(5, 150_000, 150_000)
    tal[1][7] += 1 << 31
    tr.narrow_tallies(wire[1], 2, tal[1])
    assert int(wire[1][-1].item()) == 1
    before2d = tal[1][:n2].cpu().numpy().copy()
    tr.widen_tallies(wire[1], tal[1])
    assert np.array_equal(tal[1][:n2].cpu().numpy(), before2d) and np.array_equal(tal[1][n2:].cpu().numpy(), ref[1][n2:])
    # refusals
    with pytest.raises(rt.RtError, match="overflow"):
        tr.trace(5, 0, (1 << 32) - 100_000)
    tr.set_option("pool", 0)
    with pytest.raises(rt.RtError, match="pool kernel"):
        tr.trace(5, 0, 10)
    tr.set_option("pool", 128)
    tr.bind_atmos_wire(None)        # unbound: an ordinary 64-bit run again
    tr.reset(); tr.trace(5, 150_000, 150_000); tr.synchronize()
    assert np.array_equal(tal[1].cpu().numpy(), ref[1])
    for t in trs:
        t.synchronize(); t.close()


@pytest.mark.parametrize("wire32", [1, 0])
def test_rt_allreduce_of_the_c_abi(wire32):
    """rt_allreduce (RCCL loaded at run time inside the library) on the one GPU there is: the sum over one context is
    that context's tallies, through the 32-bit path, the 64-bit path, and the 32-bit path falling back to 64 bits
    for a count of 2^32 + 5; two contexts on one GPU are refused."""
    import raytracer_amd as rt
    from raytracer_amd.tracer import allreduce
    scene = rt.scenes.tiny()
    tr = rt.PhotonTracer.from_scene(scene, wire32=wire32)
    tr.trace(9, 0, 1 << 16); tr.synchronize()
    want = tr.tallies()
    allreduce([tr])
    got = tr.tallies()
    for name in O.TALLY_NAMES:
        assert np.array_equal(got[name], want[name]), name
    big = {n: np.zeros_like(v) for n, v in want.items()}
    big["toa_up"][1, 2] = (1 << 32) + 5
    tr.add_tallies(big)
    allreduce([tr])
    got = tr.tallies()
    assert int(got["toa_up"][1, 2]) == int(want["toa_up"][1, 2]) + (1 << 32) + 5
    other = rt.PhotonTracer.from_scene(scene)
    with pytest.raises(rt.RtError, match="share GPU"):
        allreduce([tr, other])
    other.close(); tr.close()
