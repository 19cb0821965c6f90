#!/usr/bin/env python3
"""bench.py — photons/sec of the MI355X photon tracer on N GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W                 # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W     # one rank per GPU (RCCL)

A "step" is one complete job on the workload's grid: zero the tallies, trace this rank's batch of
photons (IDs keyed by step and rank, so every photon of the run is distinct), and — for N > 1 —
sum the tallies over ranks with ONE all-reduce (RCCL over xGMI; the path's only exchange step,
SURVEY.md §8e). Work per GPU is fixed as N grows (weak scaling). Inputs are resident in HBM before
the timed region (grids uploaded and packed once). For N > 1 the tallies are double-buffered and the
all-reduce of step k runs on RCCL's stream beside the trace of step k+1 (the way a zenith sweep
pipelines its jobs); all of it lies inside the timed region, which ends with every stream idle.

Prints ONE JSON line on rank 0 (metric of BASELINE.json: photons/sec, whole node) carrying
  roofline      algorithmic bytes per launch (from the kernel's own event counters, SURVEY.md
                §8d formula) / average kernel time measured with events on the launch stream
  cpu_baseline  oracle/cpu_ref.cpp (restatement of the reference's raytracer_3d.cpp, bit-pinned
                to it) timed on this host on a bounded sample of the same workload (N=1 only)
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (scene factory name, default log2 photons per GPU per step, description)
    "c1": ("block_cloud", 26, "C1 block cloud 128x128x128 (raytracer_3d.cu built-in scene)"),
    "c2": ("rayleigh_column", 24, "C2 clear-sky Rayleigh column 64x64x64"),
    "c3": ("cumulus_c3", 28, "C3 broken cumulus 256x256x128, HG cloud + Lambertian surface (2^28 photons on one GPU)"),
    "c4": ("cumulus_c4", 28, "C4 LES cumulus 1024x1024x128 (2^31 photons per step over 8 GPUs)"),
    # BASELINE config 5: the raytracer_3d_io flow as a zenith sweep with the full 3D absorption output per run
    # (handled by sweep_bench below: a step is one run of the sweep, output files included)
    "c5": ("io_default_grid", 26, "C5 zenith sweep of the raytracer_3d_io flow, 240x240x284 cells (raytracer_3d_io.cu:11-17)"),
}
C5_ZENITHS = (0.0, 15.0, 30.0, 45.0, 60.0, 75.0)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# The ceilings this kernel actually meets, measured on MI355X with tools/gather_rate.hip and
# tools/valu_rate.hip (outputs: profiles/r1_microbench_*.txt): sectors of a multi-GiB array gathered at
# random, random 64-bit atomics, and wave64 VALU instructions (2 cycles each at 256 CU x 4 SIMD x 2.4 GHz).
GATHER_PEAK_G = 54.7
ATOMIC_PEAK_G = 18.2
VALU_PEAK_GWAVEINST = 1228.8
# ... and per footprint (the same microbenchmarks on smaller arrays): a ceiling is priced for the bytes the accesses are
# spread over -- the packed cells for the gathers, the tally buffer for the atomics -- so that a small scene is not
# held against the multi-GiB figure (round 2's C1 line showed frac 1.23 that way)
GATHER_PEAK_BY_FOOTPRINT = ((16 << 20, 81.0), (128 << 20, 57.0), (2 << 30, 54.7))
ATOMIC_PEAK_BY_FOOTPRINT = ((16 << 20, 23.8), (2 << 30, 18.2))


def ceiling_for(footprint_bytes, table):
    """Measured rate (G/s) for accesses spread over footprint_bytes: log-linear between the measured footprints."""
    if footprint_bytes <= table[0][0]:
        return table[0][1]
    for (f0, r0), (f1, r1) in zip(table, table[1:]):
        if footprint_bytes <= f1:
            w = (math.log(footprint_bytes) - math.log(f0)) / (math.log(f1) - math.log(f0))
            return r0 + w * (r1 - r0)
    return table[-1][1]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 3; --workload c5: 12 = two sweeps of the six zenith angles)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--photons-log2", type=int, default=None, help="photons per GPU per step (log2)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-uniform-leg", action="store_true", help="skip the extra N=1 measurement with uniform launch positions")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target length of the CPU sample")
    ap.add_argument("--cpu-scenes", default="c1,c3",
                    help="scenes timed on the CPU besides the benchmarked one (SURVEY.md section 8d asks for C1 and C3)")
    ap.add_argument("--blocks-per-cu", type=int, default=None)
    ap.add_argument("--launch-log2k", type=int, default=None,
                    help="stratified launch: 2^k consecutive photon IDs per column (-1 = uniform launch positions, -2 = Sobol); "
                         "used only when every GPU's batch is a whole number of passes over the columns. Default: 8 on "
                         "grids beyond the caches (packed cells > 256 MiB), 5 on smaller ones, as the drivers' --launch=auto")
    ap.add_argument("--launch-tile-log2", type=int, default=None,
                    help="stratified launch: columns counted in tiles of 2^T columns square (0 = row by row). Default: 5 on "
                         "grids beyond the caches whose sides are multiples of 32, else 0")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="collective backend; gloo (tallies staged through host memory) only to rehearse "
                         "the multi-rank flow on a box with one GPU")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: wait for each step's all-reduce before the next step starts")
    ap.add_argument("--wire64", action="store_true",
                    help="N > 1: all-reduce the 64-bit tallies as they are (default: 32-bit counts on the wire)")
    ap.add_argument("--outdir", default=None, help="--workload c5: where the runs' output files go (default: the system's temporary directory)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: run the collective code path even at world size 1")
    return ap.parse_args()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_sample(scene, seconds, threads):
    """Time oracle/cpu_ref.cpp (the checker, here only as the reported CPU baseline) on this host: grow the
    sample until it runs for about `seconds`."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    s64 = O.scene_to_f64(scene)
    # (at least ~10 refills of the tracer's 65 536-photon batch: below that the tail of the last batch dominates)
    n, dt, t = max(250_000 * threads, 800_000), 0.0, None
    for _ in range(5):
        t0 = time.perf_counter()
        t, _ = O.cpu_ref(s64, n, nthreads=threads)
        dt = time.perf_counter() - t0
        if dt >= 0.7 * seconds:
            break
        n = int(min(n * max(1.5, 1.1 * seconds / max(dt, 1e-3)), 2e9))
    traced = int(t["toa_down"].sum())
    return {"value": traced / dt, "unit": "photons/s", "cores": threads, "photons": traced, "seconds": dt}


def reference_binary_sample(threads, arg):
    """The COMPILED REFERENCE itself (oracle/_ref/raytracer_3d_ref, built from /root/reference/src/raytracer_3d.cpp by
    oracle/Makefile; its only scene is the block cloud = workload c1) on this host: `arg` x 10^5 photons, its own
    'Duration' line, photons counted from the toa_down file it writes. None when the binary is not there."""
    import re
    import subprocess
    import tempfile
    import numpy as np
    exe = os.path.join(ROOT, "oracle", "_ref", "raytracer_3d_ref")
    if not os.access(exe, os.X_OK):
        return None
    try:
        with tempfile.TemporaryDirectory() as d:
            r = subprocess.run([exe, str(arg)], cwd=d, env={"PATH": "/usr/bin:/bin", "OMP_NUM_THREADS": str(threads)},
                               capture_output=True, text=True, timeout=180)
            m = re.search(r"Duration:\s*([0-9.eE+-]+)", r.stdout)
            if r.returncode != 0 or not m:
                return None
            traced = int(np.fromfile(os.path.join(d, "toa_down.bin"), dtype=np.uint64).sum())
        dur = float(m.group(1))
        return {"value": traced / dur, "unit": "photons/s", "cores": threads, "photons": traced, "seconds": dur,
                "kind": "reference", "command": f"oracle/_ref/raytracer_3d_ref {arg}"}
    except Exception:
        return None


def cpu_baseline(scene, seconds, workload, extra_scenes):
    """the bit-pinned restatement of raytracer_3d.cpp on all of this GPU's share of the host
    cores and on one thread, CPU model named; the headline entry is the all-cores figure on the benchmarked scene."""
    import raytracer_amd as rt
    cores = min(16, os.cpu_count() or 1)   # the GPU box gives one GPU's share of the host: 16 cores
    full = cpu_sample(scene, seconds, cores)
    one = cpu_sample(scene, seconds / 2, 1)
    out = {"value": full["value"], "unit": "photons/s", "cores": cores, "kind": "port",
           "sample": f"{full['photons']} photons of the same scene ({workload}) through oracle/cpu_ref.cpp "
                     f"(double, xoroshiro128+, global majorant; bit-pinned to raytracer_3d.cpp) in {full['seconds']:.2f} s",
           "cpu_model": cpu_model(),
           "one_thread": {k: one[k] for k in ("value", "unit", "cores", "photons", "seconds")}}
    def reference_binary():      # the compiled reference runs its own scene only: the block cloud (c1)
        ref = {"all_cores": reference_binary_sample(cores, 300), "one_thread": reference_binary_sample(1, 20)}
        return ref if ref["all_cores"] and ref["one_thread"] else None

    if workload == "c1":
        ref = reference_binary()
        if ref:
            out["reference_binary"] = ref
    others = {}
    for wl in extra_scenes:
        if wl == workload:
            continue
        sc = getattr(rt.scenes, WORKLOADS[wl][0])()
        others[wl] = {"all_cores": cpu_sample(sc, seconds / 3, cores), "one_thread": cpu_sample(sc, seconds / 3, 1)}
        if wl == "c1":
            ref = reference_binary()
            if ref:
                others[wl]["reference_binary"] = ref
    if others:
        out["other_scenes"] = others
    return out


def shared_scene(rt, factory, rank, world, share, dist):
    """The synthetic scene of this run. With several ranks on one node only rank 0 generates it (1024 x 1024 x 128:
    6.6 s and 5 GB of numpy temporaries) and leaves the four float32 fields in /dev/shm; the others map them read-only
    -- eight ranks generating side by side on shared host cores would each spend that ahead of the first GPU call."""
    if not share:
        return getattr(rt.scenes, factory)()
    import shutil
    d = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", f"rt_bench_scene_{os.environ.get('MASTER_PORT', '0')}_{factory}")
    names = ("k_gas", "k_cloud", "ssa", "asy")
    if rank == 0:
        sc = getattr(rt.scenes, factory)()
        os.makedirs(d, exist_ok=True)
        for name, a in zip(names, sc.fields()):
            np.save(os.path.join(d, name + ".npy"), np.ascontiguousarray(a, dtype=np.float32))
        with open(os.path.join(d, "meta.json"), "w") as fh:
            json.dump({"name": sc.name, "dims": [sc.itot, sc.jtot, sc.ktot], "spacing": [sc.dx, sc.dy, sc.dz],
                       "albedo": sc.albedo, "zenith_deg": sc.zenith_deg, "azimuth_deg": sc.azimuth_deg}, fh)
    dist.barrier()
    if rank != 0:
        with open(os.path.join(d, "meta.json")) as fh:
            m = json.load(fh)
        fields = [np.load(os.path.join(d, name + ".npy"), mmap_mode="r") for name in names]
        sc = rt.scenes.Scene(m["name"], *m["dims"], *m["spacing"], *fields)
        sc.albedo, sc.zenith_deg, sc.azimuth_deg = m["albedo"], m["zenith_deg"], m["azimuth_deg"]
    sc._shared_dir = d
    return sc


def release_shared_scene(scene, rank, dist):
    d = getattr(scene, "_shared_dir", None)
    if d is None:
        return
    import shutil
    dist.barrier()              # every rank has uploaded its copy to its GPU
    if rank == 0:
        shutil.rmtree(d, ignore_errors=True)


def sweep_bench(args, rt, torch):
    """--workload c5 (one GPU): runs of 2^26 photons under six solar zenith angles with the grids resident, each run
    followed by what raytracer_3d_io.cu:174-212 does after its kernel -- seven device-to-host copies and seven files.
    Pipelined as the driver raytracer_3d_io_gpu does it: the copy of run k leaves one tally buffer over the library's
    copy stream into pinned memory while run k+1 is traced into the other, and host threads write the files of up to
    four runs side by side (a run's files take four times as long as its trace and download). A
    step is one run; the timed region ends when the last file is closed. Measured beside it: the same runs strictly one
    after another (the reference's order), the trace alone, and the same sweep on the C3 grid."""
    import tempfile
    import threading
    n = 1 << (args.photons_log2 if args.photons_log2 is not None else WORKLOADS["c5"][1])

    def one_grid(scene, outdir, steps, warmup):
        tr = rt.PhotonTracer.from_scene(scene, device=0)
        k, tile = rt.choose_launch_order(scene.itot, scene.jtot, scene.ktot, n)
        tr.set_option("launch_log2k", k); tr.set_option("launch_tile_log2", tile)
        tr.prepare()
        # a ring of host sets: the files of a run (26 ms of write() for 264 MB on two threads) take four times as long as
        # its trace and download, so the writers of up to `depth` runs work side by side -- each on its own set of
        # pinned arrays and its own directory -- while the device alternates between its two tally buffers
        depth = max(2, int(os.environ.get("RT_C5_WRITERS", "4")))
        sets = [tr.pinned_tallies() for _ in range(depth)]
        dirs = [os.path.join(outdir, f"run_{i:02d}") for i in range(depth)]
        for d in dirs:
            os.makedirs(d, exist_ok=True)
        state = {"bytes": 0, "write_s": 0.0, "err": None}
        state_lock = threading.Lock()

        def write(ticket, arrays, d):
            try:
                if ticket is not None:
                    tr.tallies_wait(ticket)
                t0 = time.perf_counter()
                rt.save_tallies(arrays, d)
                with state_lock:
                    state["write_s"] += time.perf_counter() - t0
                    state["bytes"] += sum(a.nbytes for a in arrays.values())
            except Exception as exc:      # noqa: BLE001
                state["err"] = exc

        def sweep(jobs, first_job, pipelined):
            writers = [None] * depth
            for j in range(jobs):
                job = first_job + j
                zen = C5_ZENITHS[job % len(C5_ZENITHS)]
                tr.set_scene(zen, scene.azimuth_deg, scene.albedo)
                tr.reset()
                tr.trace(args.seed, job * n, n)
                arrays = sets[j % depth][0]
                if pipelined:
                    if writers[j % depth] is not None:      # this set's files of `depth` runs ago are written
                        writers[j % depth].join()
                    ticket = tr.tallies_async(arrays)
                    tr.swap_tally_buffer()
                    writers[j % depth] = threading.Thread(target=write, args=(ticket, arrays, dirs[j % depth]))
                    writers[j % depth].start()
                else:
                    tr.synchronize()
                    got = tr.tallies()          # seven blocking copies into pageable memory, as copy_from_gpu does
                    write(None, got, dirs[j % depth])
            for w in writers:
                if w is not None:
                    w.join()
            tr.synchronize()
            if state["err"] is not None:
                raise state["err"]

        sweep(warmup, 0, True)
        state.update(bytes=0, write_s=0.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sweep(steps, warmup, True)
        piped = time.perf_counter() - t0
        wrote, write_s = state["bytes"], state["write_s"]
        t0 = time.perf_counter()
        sweep(steps, warmup + steps, False)
        serial = time.perf_counter() - t0
        t0 = time.perf_counter()
        kernel_ms = []
        for j in range(steps):
            tr.set_scene(C5_ZENITHS[j % len(C5_ZENITHS)], scene.azimuth_deg, scene.albedo)
            tr.reset(); tr.trace(args.seed, (warmup + 2 * steps + j) * n, n); tr.synchronize()
            kernel_ms.append(tr.last_trace_ms())
        trace_only = time.perf_counter() - t0
        # the last run's files against the device's counts: what was written is what was traced
        t = {name: np.fromfile(os.path.join(dirs[(steps - 1) % depth], name + ".bin"), dtype=np.uint64) for name in rt.TALLY_NAMES}
        shaped = {name: v.reshape((scene.ktot, scene.jtot, scene.itot) if name.startswith("atmos") else (scene.jtot, scene.itot))
                  for name, v in t.items()}
        if rt.photon_balance(shaped) != 0 or int(t["toa_down"].sum()) != n:
            raise SystemExit("bench c5: the files of the last run do not balance")
        for s_ in sets:
            tr.free_pinned(s_[1])
        tr.set_option("stats", 1); tr.reset(); tr.trace(args.seed, 0, n); tr.synchronize(); ev = tr.events(); tr.set_option("stats", 0)
        tr.close()
        return {"grid": [scene.itot, scene.jtot, scene.ktot], "jobs": steps, "photons_per_job": n, "writers_in_flight": depth,
                "launch_log2k": k, "output_bytes_per_job": wrote // max(steps, 1),
                "end_to_end_photons_per_s": steps * n / piped, "jobs_per_s": steps / piped, "ms_per_job": piped / steps * 1e3,
                "not_pipelined_photons_per_s": steps * n / serial, "not_pipelined_ms_per_job": serial / steps * 1e3,
                "trace_only_photons_per_s": steps * n / trace_only, "trace_only_ms_per_job": trace_only / steps * 1e3,
                "kernel_ms": sum(kernel_ms) / len(kernel_ms), "file_write_s_per_job": write_s / steps,
                "file_write_GB_per_s": wrote / max(write_s, 1e-9) / 1e9, "events": ev}

    steps = args.steps if args.steps is not None else 2 * len(C5_ZENITHS)   # default: two sweeps of the six angles (the writers of four runs are in flight: one sweep is mostly the pipeline filling and draining)
    warmup = max(args.warmup, 1)
    with tempfile.TemporaryDirectory(dir=args.outdir) as d:
        io_scene = rt.scenes.cumulus(240, 240, 284, 100.0, 100.0, 30.0, n_clouds=60, seed=7, z_base=900.0, z_top=3000.0,
                                     name="io_default_grid")
        io = one_grid(io_scene, os.path.join(d, "io"), steps, warmup)
        c3 = one_grid(rt.scenes.cumulus_c3(), os.path.join(d, "c3"), steps, warmup)
        where = os.path.abspath(d)
    alg = rt.algorithmic_bytes(io["events"], io["photons_per_job"])
    achieved = alg / (io["kernel_ms"] * 1e-3) / 1e9
    out = {"metric": "photons/sec (whole node)", "value": io["end_to_end_photons_per_s"], "unit": "photons/s", "n_gpus": 1,
           "steps": steps, "warmup": warmup, "ms_per_step": io["ms_per_job"], "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": WORKLOADS["c5"][2] + f"; a step = one run of 2^{int(math.log2(n))} photons under the next of the zenith "
                      f"angles {list(C5_ZENITHS)} (azimuth 20 deg) INCLUDING the download of the seven tally arrays and the "
                      "seven output files (full 3D absorption fields) of raytracer_3d_io.cu:174-212, pipelined: copy and "
                      "file writes of run k beside the trace of run k+1; value = end to end, measured",
                      "photons_per_gpu_per_step": n, "parallelism": "one GPU (the sweep's runs are independent jobs)",
                      "files_written_under": where},
           "io_default_grid": {k: v for k, v in io.items() if k != "events"},
           "c3_grid": {k: v for k, v in c3.items() if k != "events"},
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": None, "kernel": "rt_trace_pool_kernel", "kernel_ms": io["kernel_ms"],
                        "algorithmic_bytes_per_photon": alg / io["photons_per_job"],
                        "note": "the trace kernel of one run; end to end this workload is bound by the output files "
                                f"({io['output_bytes_per_job'] / 1e6:.0f} MB per run at {io['file_write_GB_per_s']:.2f} GB/s)"}}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(io_scene, args.cpu_seconds, "c5", [])
    return out


def main():
    args = parse()
    # The contract is ONE JSON line on stdout. Libraries write there too (RCCL prints a five-line version banner at
    # its first collective), so file descriptor 1 is pointed at stderr for the whole run and the line goes to the
    # saved descriptor at the end.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly with --gpus N: become the launcher the contract describes (a child process; nothing here
        # has touched the GPU yet), one rank per GPU, and pass its exit status on
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(29600 + os.getpid() % 300),
               os.path.abspath(__file__), *sys.argv[1:]]
        raise SystemExit(subprocess.call(cmd))
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    import raytracer_amd as rt

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP library has no CPU fallback)")
    if args.workload != "c5" and args.steps is None:
        args.steps = 3
    if args.workload == "c5":
        if world != 1:
            raise SystemExit("--workload c5 is a one-GPU measurement (its runs are independent jobs)")
        torch.cuda.set_device(0)
        out = sweep_bench(args, rt, torch)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
        os.close(json_fd)
        return
    if args.single_device:
        if args.backend != "gloo":
            raise SystemExit("--single-device needs --backend gloo (RCCL refuses two ranks on one GPU)")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)      # RCCL over xGMI (default stream priority: a
            # high-priority RCCL stream bought nothing in tools/overlap_probe.py and hung a world-size-1 test once)
        else:
            dist.init_process_group("gloo")

    factory, default_log2, desc = WORKLOADS[args.workload]
    log2n = args.photons_log2 if args.photons_log2 is not None else default_log2
    n_per_gpu = 1 << log2n
    scene = shared_scene(rt, factory, rank, world, use_dist and world > 1, dist)

    tr = rt.PhotonTracer.from_scene(scene, device=local_rank)
    release_shared_scene(scene, rank, dist)
    if args.blocks_per_cu:
        tr.set_option("blocks_per_cu", args.blocks_per_cu)
    # Launch order: consecutive photon IDs sweep the columns of the domain top (the reference's launch is
    # a scrambled Sobol sequence over the same square), which keeps the cells a wave gathers L2-resident.
    # Every batch below is a whole number of such sweeps, so each step samples the domain top evenly.
    # Beyond the caches (C4) the columns are counted in tiles of 32 x 32 and a column takes 2^8 IDs at a time: the
    # photons of one tile keep its cells in the XCD's L2. Smaller grids keep rows and 2^5 (more photons per column at
    # a time contend for the same tally words there). Same rule as the drivers' --launch=auto.
    auto_k, auto_tile = rt.choose_launch_order(scene.itot, scene.jtot, scene.ktot, n_per_gpu)
    launch_tile_log2 = args.launch_tile_log2 if args.launch_tile_log2 is not None else auto_tile
    launch_log2k = args.launch_log2k
    if launch_log2k is None:
        launch_log2k = auto_k
        if launch_tile_log2 == 0 and launch_log2k > 5:      # tiles switched off by hand: rows take k <= 5
            launch_log2k = 5
    elif launch_log2k >= 0 and n_per_gpu % ((scene.itot * scene.jtot) << launch_log2k) != 0:
        launch_log2k = -1
    if launch_log2k < 0:
        launch_tile_log2 = 0
    tr.set_option("launch_log2k", launch_log2k)
    tr.set_option("launch_tile_log2", launch_tile_log2)
    tr.prepare()            # majorant tables now: every trace below is a pure asynchronous launch
    stream = torch.cuda.current_stream()
    # The step is raytracer_amd.distributed.ShardedTracer.step -- the code the multi-rank tests check: zero a
    # tally buffer, trace this rank's shard, start the ONE all-reduce (32-bit counts on the wire, two buffers so
    # that it runs beside the next step's trace). ShardedTracer puts the tracer on torch's current stream.
    sharded = rt.distributed.ShardedTracer(tr, dev, overlap=False if args.no_overlap else None,
                                           wire32=not args.wire64, force_collective=args.force_dist)
    overlap, use_dist = sharded.overlap, sharded.collective

    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    stops = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    def step(idx, timed_slot=None):
        # photon IDs of step idx: [idx * world * n_per_gpu, (idx + 1) * world * n_per_gpu), sharded over the ranks
        return sharded.step(args.seed, world * n_per_gpu, idx * world * n_per_gpu,
                            before_trace=(lambda: starts[timed_slot].record(stream)) if timed_slot is not None else None,
                            after_trace=(lambda: stops[timed_slot].record(stream)) if timed_slot is not None else None)

    def fence():
        sharded.drain()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        step(w)
    fence()
    t0 = time.perf_counter()
    last = None
    for k in range(args.steps):
        last = step(args.warmup + k, timed_slot=k)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    tr.synchronize()    # raises if any photon was abandoned by the step guard

    kernel_ms = [s.elapsed_time(e) for s, e in zip(starts, stops)]
    avg_kernel_ms = sum(kernel_ms) / len(kernel_ms)

    # sanity on the last step's result (after the all-reduce every rank holds the node total)
    words = sharded.result(last).cpu().numpy()
    t = rt.split_tally_words(words, scene.itot, scene.jtot, scene.ktot)
    launched = int(t["toa_down"].sum())
    balance = rt.photon_balance(t)
    if balance != 0 or launched != n_per_gpu * world:
        raise SystemExit(f"bench: invalid result (balance {balance}, launched {launched}, expected {n_per_gpu * world})")

    if sharded.atmos32:
        tr.bind_atmos_wire(None)        # what follows reads this context's own 64-bit tallies again
    out = None
    if rank == 0:
        # event counters of exactly one launch (same photon range as the last timed step of rank 0)
        tr.set_option("stats", 1)
        tr.reset()
        tr.trace(args.seed, ((args.warmup + args.steps - 1) * world) * n_per_gpu, n_per_gpu)
        tr.synchronize()
        ev = tr.events()
        tr.set_option("stats", 0)
        alg_bytes = rt.algorithmic_bytes(ev, n_per_gpu)
        achieved = alg_bytes / (avg_kernel_ms * 1e-3) / 1e9
        total_photons = n_per_gpu * world * args.steps
        value = total_photons / elapsed
        out = {
            "metric": "photons/sec (whole node)", "value": value, "unit": "photons/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{desc}; 2^{log2n} photons per GPU per step; "
                                   f"grid {scene.itot}x{scene.jtot}x{scene.ktot}; zenith {scene.zenith_deg} deg, "
                                   f"albedo {scene.albedo}; step = reset + trace" + (((" + RCCL all-reduce of tallies" + (" (32-bit counts on the wire)" if sharded.wire32 else " (64-bit)")) if args.backend == "nccl" else " + gloo all-reduce staged through the host (rehearsal)") if use_dist else "")
                                   + (" (all-reduce of step k beside the trace of step k+1, double-buffered)" if overlap else ""),
                       "photons_per_gpu_per_step": n_per_gpu, "parallelism": f"photon-id sharding x{world}",
                       "launch": (f"stratified, 2^{launch_log2k} consecutive photon ids per column, columns "
                                  + (f"in tiles of {1 << launch_tile_log2} x {1 << launch_tile_log2}, " if launch_tile_log2 > 0 else "row by row, ") +
                                  f"{n_per_gpu // ((scene.itot * scene.jtot) << launch_log2k)} sweeps of the domain top per GPU per step"
                                  if launch_log2k >= 0 else ("Sobol sequence indexed by the photon id (the reference's launch)" if launch_log2k == -2 else "uniform random positions") +
                                  "; photon ids taken in the order of their launch tile -- sorted on the device inside the timed region, kernel_ms includes the sort's passes")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "rt_trace_pool_kernel", "kernel_ms": avg_kernel_ms,
                         "algorithmic_bytes_per_photon": alg_bytes / n_per_gpu,
                         "events_per_photon": {k: v / n_per_gpu for k, v in ev.items()}},
            "kernel_photons_per_s_per_gpu": n_per_gpu / (avg_kernel_ms * 1e-3),
            "scheduler": tr.sched_stats(),
        }
        if world == 1 and launch_log2k >= 0 and not args.no_uniform_leg:
            # beside the headline: the same step with the two launch samplings that are functions of the photon ID alone
            # and work for any photon count on any grid -- the Sobol sequence (the reference's own launch,
            # raytracer_kernels.cu:239-253, 331; what the drivers' --launch=auto takes when no whole pass exists) and
            # uniform Philox positions (the library default) -- where the pool kernel sorts each batch's photon ids by
            # launch tile on the device, inside the timed region
            def launch_leg(mode, slot, what):
                tr.set_option("launch_log2k", mode)
                tr.prepare()
                base = (args.warmup + args.steps + 1 + slot * (args.steps + 1)) * n_per_gpu
                tr.reset(); tr.trace(args.seed, base, n_per_gpu); tr.synchronize()
                torch.cuda.synchronize()
                tu = time.perf_counter()
                for k in range(args.steps):
                    tr.reset(); tr.trace(args.seed, base + (k + 1) * n_per_gpu, n_per_gpu)
                tr.synchronize()
                torch.cuda.synchronize()
                tu = time.perf_counter() - tu
                tal = tr.tallies()
                td = tal["toa_down"].astype(np.float64)
                return {"value": n_per_gpu * args.steps / tu, "unit": "photons/s", "ms_per_step": tu / args.steps * 1e3,
                        "steps": args.steps, "balance": rt.photon_balance(tal),
                        "vs_headline": (n_per_gpu * args.steps / tu) / value,
                        "toa_down_variance_over_mean": float(td.var() / td.mean()),
                        "launch": what}
            out["sobol_launch"] = launch_leg(-2, 0, "Sobol sequence indexed by the photon id, digitally shifted per seed (the reference's "
                                             "launch, raytracer_kernels.cu:239-253); whole nets of ids taken box by box of the domain top "
                                             "through the net's inverse (no sort, no list), the rest sorted by launch tile")
            out["uniform_launch"] = launch_leg(-1, 1, "uniform random positions; photon ids sorted by launch tile on the device, inside the timed region")
            tr.set_option("launch_log2k", launch_log2k)
            tr.prepare()
            uprof = os.path.join(ROOT, "profiles", f"pmc_{args.workload}_uniform.json")
            if os.path.exists(uprof):
                try:
                    with open(uprof) as fh:
                        p = json.load(fh)
                    if p.get("photons_per_launch") == n_per_gpu:
                        out["uniform_launch"].update({
                            "tracer_kernel_ms": p.get("kernel_avg_ns", 0) / 1e6,
                            "traffic": p.get("hbm_bytes_per_launch"),
                            "fabric_reads_per_photon": p.get("fabric_reads_per_photon"),
                            "valu_wave_insts_per_photon": p.get("valu_wave_insts_per_photon"),
                            "counters_source": f"profiles/pmc_{args.workload}_uniform.json (rocprofv3 --pmc passes of "
                                               "bench.py --launch-log2k -1, tracer kernel only; replayed, not measured in this run)"})
                except Exception:
                    pass
        prof = os.path.join(ROOT, "profiles", f"pmc_{args.workload}.json")
        if os.path.exists(prof):
            try:
                with open(prof) as fh:
                    p = json.load(fh)
                if (p.get("photons_per_launch") == n_per_gpu and p.get("launch_log2k", -1) == launch_log2k
                        and p.get("launch_tile_log2", 0) == launch_tile_log2):
                    out["roofline"]["traffic"] = p.get("hbm_bytes_per_launch")
                    # counters cannot be read inside a plain run: these figures are REPLAYED from the committed
                    # rocprofv3 PMC passes of this same command, not measured by this process
                    out["roofline"]["traffic_source"] = out["counters_source"] = \
                        f"profiles/pmc_{args.workload}.json (rocprofv3 --pmc passes of this command, kernel {p.get('kernel_avg_ns', 0) / 1e6:.2f} ms there; replayed, not measured in this run)"
                    # the same profile's counters priced against the measured gather and VALU ceilings
                    kt = avg_kernel_ms * 1e-3
                    if p.get("fabric_reads_per_photon") is not None:
                        misses = p["fabric_reads_per_photon"] * n_per_gpu
                        atomics = p.get("atomics_per_photon", 0.0) * n_per_gpu
                        g_peak = ceiling_for(16 * scene.ncell, GATHER_PEAK_BY_FOOTPRINT)
                        a_peak = ceiling_for(8 * tr.nwords, ATOMIC_PEAK_BY_FOOTPRINT)
                        # Each of the two request streams alone needs its count / its measured rate: the LARGER of the two
                        # times is a true lower bound on the kernel's time (frac <= 1 by construction). Round 2 added the two
                        # times (no overlap at all), which the measurements refute: C3 runs 10 % FASTER than that sum allows
                        # (reads and atomics overlap in the memory system) -- kept as frac_additive for comparison only.
                        t_reads, t_atomics = misses / g_peak * 1e-9, atomics / a_peak * 1e-9
                        out["gather_roofline"] = {
                            "bound": "random 64-byte sector reads and 64-bit atomics (ceilings measured for this footprint, G/s); "
                                     "frac = the slower of the two streams alone / kernel time",
                            "sector_reads_per_photon": p["fabric_reads_per_photon"],
                            "atomics_per_photon": p.get("atomics_per_photon"),
                            "peak_sector_reads_G_per_s": g_peak, "peak_atomics_G_per_s": a_peak,
                            "footprint_bytes": {"cells": 16 * scene.ncell, "tallies": 8 * tr.nwords},
                            "binding_stream": "atomics" if t_atomics > t_reads else "sector reads",
                            "frac": max(t_reads, t_atomics) / kt,
                            "frac_additive": (t_reads + t_atomics) / kt}
                    if p.get("valu_wave_insts_per_photon") is not None:
                        valu_achieved = p["valu_wave_insts_per_photon"] * n_per_gpu / kt / 1e9
                        out["valu_roofline"] = {
                            "wave_insts_per_photon": p["valu_wave_insts_per_photon"],
                            "active_lanes": p.get("active_lanes_per_valu_inst"),
                            "peak_G_wave_insts_per_s": VALU_PEAK_GWAVEINST,
                            "sustained_by_microbenchmark_G_wave_insts_per_s": 1040.0,   # v_xor_b32, 8 waves/SIMD
                            "frac": valu_achieved / VALU_PEAK_GWAVEINST}
                        if p.get("valu_issue_cycles_per_photon") is not None:
                            # not every instruction is a 2-cycle one: Philox's 64-bit multiply-adds hold the SIMD 4 cycles,
                            # reciprocals and square roots 8 (tools/valu_rate.hip); with the instruction mix the counters
                            # give, the share of the SIMDs' cycles that issue this kernel's VALU work
                            cyc = p["valu_issue_cycles_per_photon"] * n_per_gpu / kt
                            out["valu_roofline"].update({
                                "issue_cycles_per_photon": p["valu_issue_cycles_per_photon"],
                                "int64_insts_per_photon": p.get("valu_int64_insts_per_photon"),
                                "trans_insts_per_photon": p.get("valu_trans_insts_per_photon"),
                                "frac_issue_cycles": cyc / (1024 * 2.4e9)})
                        # roofline.achieved = max(bytes / BW_peak, valu / VALU_peak), `bound` names
                        # the binding ceiling. The streaming-HBM figure (measured live) stays in roofline.hbm; the VALU
                        # instruction count per photon is the replayed counter, the kernel time is this run's.
                        hbm = {k: out["roofline"][k] for k in ("achieved", "peak", "unit", "frac")}
                        out["roofline"]["hbm"] = hbm
                        if out["valu_roofline"].get("frac_issue_cycles") is not None:
                            # with the instruction mix known, the VALU ceiling is priced in issue cycles (a 64-bit multiply-add
                            # is not a 2-cycle instruction): SIMD cycles that issue this kernel's VALU work / all SIMD cycles
                            cyc_g = out["valu_roofline"]["issue_cycles_per_photon"] * n_per_gpu / kt / 1e9
                            if out["valu_roofline"]["frac_issue_cycles"] > hbm["frac"]:
                                out["roofline"].update({"bound": "valu", "achieved": cyc_g, "peak": 1024 * 2.4,
                                                        "unit": "G SIMD issue cycles/s (wave64 VALU instruction 2 cycles, 64-bit integer 4, "
                                                                "transcendental 8: tools/valu_rate.hip; 1024 SIMDs x 2.4 GHz)",
                                                        "frac": out["valu_roofline"]["frac_issue_cycles"]})
                        elif valu_achieved / VALU_PEAK_GWAVEINST > hbm["frac"]:
                            out["roofline"].update({"bound": "valu", "achieved": valu_achieved, "peak": VALU_PEAK_GWAVEINST,
                                                    "unit": "G wave-instructions/s (wave64 VALU, 2-cycle issue x 1024 SIMDs x 2.4 GHz)",
                                                    "frac": valu_achieved / VALU_PEAK_GWAVEINST})
                        # ... and (i) of that section for grids inside the caches: the random-access ceiling measured by
                        # the microbenchmarks over the same footprint (64-byte sector reads + 64-bit atomics, additive)
                        g = out.get("gather_roofline")
                        if g and g["frac"] > out["roofline"]["frac"]:
                            per_photon = (g["atomics_per_photon"] or 0.0) if g["binding_stream"] == "atomics" else g["sector_reads_per_photon"]
                            acc = per_photon * n_per_gpu / kt / 1e9
                            out["roofline"].update({"bound": "gather", "achieved": acc, "peak": acc / g["frac"],
                                                    "unit": "G random accesses/s of the binding stream (" + g["binding_stream"] + "; rate "
                                                            "measured by tools/gather_rate.hip for this footprint)",
                                                    "frac": g["frac"]})
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, args.cpu_seconds, args.workload,
                                               [w for w in args.cpu_scenes.split(",") if w in WORKLOADS])
    tr.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
