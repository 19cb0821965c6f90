#!/usr/bin/env python3
"""Does a kernel that needs LDS (and, with regs=1, 284 registers per lane like RCCL's ring kernel) get onto the GPU
while the persistent tracer runs? (stand-in for RCCL's collective kernels beside the next step's trace).
Usage: python tools/overlap_probe.py [probe_first=1] [regs=1] [prio=1] [scene=c4] [nprobes=1] [probe_ms=6] [lds=19]
       [blocks=32] [tracer option=value ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liboverlap_probe.so"))
lib.probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
lib.probe_launch_regs.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
opts = dict(kv.split("=") for kv in sys.argv[1:])
probe_first = int(opts.pop("probe_first", 0))     # 1: the LDS kernel is already resident when the trace starts
probe_ms = int(opts.pop("probe_ms", 1))
lds_list = [int(v) for v in opts.pop("lds", "0,8,21,24,32").split(",")]
launch = lib.probe_launch_regs if int(opts.pop("regs", 0)) else lib.probe_launch
opts_reps = opts.pop("reps", 1)
prio, nprobes, blocks = int(opts.pop("prio", 0)), int(opts.pop("nprobes", 8)), int(opts.pop("blocks", 32))
reserve = 0
if opts.pop("scene", "c3") == "c4":
    scene, n = rt.scenes.cumulus_c4(), 1 << 28
    tr = rt.PhotonTracer.from_scene(scene, launch_log2k=8, launch_tile_log2=5, **{k: int(v) for k, v in opts.items()})
else:
    scene, n = rt.scenes.cumulus_c3(), 1 << 27
    tr = rt.PhotonTracer.from_scene(scene, launch_log2k=5, **{k: int(v) for k, v in opts.items()})
main = torch.cuda.current_stream()
if reserve > 0:
    main = torch.cuda.ExternalStream(tr.create_stream(reserve))
else:
    tr.set_stream(main.cuda_stream)
print("reserved CUs:", reserve)
side = torch.cuda.Stream(priority=-1 if prio else 0)
tr.trace(1, 0, n); tr.synchronize()                 # a full-size warm-up: the first trace of a process runs slower
reps = int(opts_reps)
for lds_kb in [v for v in lds_list for _ in range(reps)]:
    rc = 0
    e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
    ps = [torch.cuda.Event(enable_timing=True) for _ in range(nprobes)]
    torch.cuda.synchronize()
    e0.record(main)
    if probe_first:
        side.wait_event(e0)
        for p in ps:
            rc = launch(side.cuda_stream, lds_kb * 1024, blocks, probe_ms * 1000)
            p.record(side)
        tr.reset()                                  # what a bench step does first: a memset of the tallies
    tr.trace(1, 0, n)
    e1.record(main)
    if not probe_first:
        side.wait_event(e0)
        for p in ps:
            rc = launch(side.cuda_stream, lds_kb * 1024, blocks, probe_ms * 1000)
            p.record(side)
    torch.cuda.synchronize()
    print(f"probes with {lds_kb:2d} KiB LDS: rc={rc}  trace {e0.elapsed_time(e1):7.2f} ms; probes finished at",
          " ".join(f"{e0.elapsed_time(p):.1f}" for p in ps), "ms after the trace began")
