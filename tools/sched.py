#!/usr/bin/env python3
"""Print scheduler statistics of one stats launch. Usage: python tools/sched.py c3 22 [th_event th_scatter low]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
wl, log2n = sys.argv[1], int(sys.argv[2])
scene = {"c1": rt.scenes.block_cloud, "c2": rt.scenes.rayleigh_column, "c3": rt.scenes.cumulus_c3, "c4": rt.scenes.cumulus_c4}[wl]()
tr = rt.PhotonTracer.from_scene(scene, stats=1, pool=int(os.environ.get("POOL", "0")))
if len(sys.argv) > 5:
    tr.set_option("th_event", int(sys.argv[3])); tr.set_option("th_scatter", int(sys.argv[4])); tr.set_option("low_traverse", int(sys.argv[5]))
n = 1 << log2n
tr.trace(1, 0, n); tr.synchronize()
st, ev = tr.sched_stats(), tr.events()
print(wl, "ms", tr.last_trace_ms(), {k: round(v / n, 2) for k, v in ev.items()})
for k in "TES":
    r, l = st["rounds"][k], st["lanes"][k]
    print(f"stage {k}: wave-rounds/photon {r / n:.4f}  lanes/round {l / max(r, 1):.1f}  lane-events/photon {l / n:.2f}")
