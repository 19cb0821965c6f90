#!/usr/bin/env python3
"""Does a kernel that needs LDS get onto the GPU while the persistent tracer runs? (stand-in for RCCL's
collective kernels beside the next step's trace). Usage: python tools/overlap_probe.py [option=value ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liboverlap_probe.so"))
lib.probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
opts = dict(kv.split("=") for kv in sys.argv[1:])
probe_first = int(opts.pop("probe_first", 0))     # 1: the LDS kernel is already resident when the trace starts
probe_ms = int(opts.pop("probe_ms", 1))
lds_list = [int(v) for v in opts.pop("lds", "0,8,21,24,32").split(",")]
scene = rt.scenes.cumulus_c3()
reserve = 0
tr = rt.PhotonTracer.from_scene(scene, launch_log2k=5, **{k: int(v) for k, v in opts.items()})
main = torch.cuda.current_stream()
if reserve > 0:
    main = torch.cuda.ExternalStream(tr.create_stream(reserve))
else:
    tr.set_stream(main.cuda_stream)
print("reserved CUs:", reserve)
side = torch.cuda.Stream()
n = 1 << 27
tr.trace(1, 0, 1 << 20); tr.synchronize()
for lds_kb in lds_list:
    e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
    ps = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
    torch.cuda.synchronize()
    e0.record(main)
    if probe_first:
        side.wait_event(e0)
        for p in ps:
            rc = lib.probe_launch(side.cuda_stream, lds_kb * 1024, 32, probe_ms * 1000)
            p.record(side)
        tr.reset()                                  # what a bench step does first: a memset of the tallies
    tr.trace(1, 0, n)
    e1.record(main)
    if not probe_first:
        side.wait_event(e0)
        for p in ps:
            rc = lib.probe_launch(side.cuda_stream, lds_kb * 1024, 32, probe_ms * 1000)       # 32 blocks x 1 ms each
            p.record(side)
    torch.cuda.synchronize()
    print(f"probes with {lds_kb:2d} KiB LDS: rc={rc}  trace {e0.elapsed_time(e1):7.2f} ms; probes finished at",
          " ".join(f"{e0.elapsed_time(p):.1f}" for p in ps), "ms after the trace began")
