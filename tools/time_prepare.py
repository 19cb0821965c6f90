#!/usr/bin/env python3
"""Wall time of rt_prepare (majorant tables: rt_majorant_kernel + rt_boxes_kernel + one host wait) on a workload's
scene, for experimental builds of the library. Usage: RT_TUNE_LIB=tools/variants/x.so python tools/time_prepare.py c4"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
if os.environ.get("RT_TUNE_LIB"):
    rt.tracer.LIB_PATH = os.path.abspath(os.environ["RT_TUNE_LIB"])
wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
scene = {"c1": rt.scenes.block_cloud, "c3": rt.scenes.cumulus_c3, "c4": rt.scenes.cumulus_c4}[wl]()
tr = rt.PhotonTracer.from_scene(scene)
tr.set_option("launch_log2k", 5)
ts = []
for r in range(3):
    tr.set_option("merge_tau_milli", 750 + r)        # makes the tables stale without changing much
    tr.synchronize(); t0 = time.perf_counter(); tr.prepare(); tr.synchronize(); ts.append(time.perf_counter() - t0)
print(wl, os.environ.get("RT_TUNE_LIB", "default"), "prepare ms:", " ".join("%.1f" % (t * 1e3) for t in ts))
