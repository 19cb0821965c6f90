#!/usr/bin/env python3
"""Sweep scheduling thresholds / supergrid sizes / blocks per CU on one GPU (interleaved rounds in
one process, median of kernel times). Usage: python tools/tune.py c4 24 [what]"""
import itertools, json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
if os.environ.get("RT_TUNE_LIB"):      # an experimental build of the library (tools only)
    rt.tracer.LIB_PATH = os.path.abspath(os.environ["RT_TUNE_LIB"])

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
what = sys.argv[3] if len(sys.argv) > 3 else "thresholds"
scene = {"c1": rt.scenes.block_cloud, "c2": rt.scenes.rayleigh_column, "c3": rt.scenes.cumulus_c3, "c4": rt.scenes.cumulus_c4}[wl]()
n = 1 << log2n
tr = rt.PhotonTracer.from_scene(scene)

def run(opts, rounds=3):
    for k, v in opts.items():
        tr.set_option(k, v)
    ts = []
    for r in range(rounds):
        tr.reset(); tr.trace(1, r * n, n); tr.synchronize(); ts.append(tr.last_trace_ms())
    return statistics.median(ts)

base = dict(pool=128, chunk=256, levels=-1, merge_tau_milli=750, launch_log2k=8, launch_tile_log2=5, th_launch=40, pool_waves_per_cu=32, th_event=20, th_scatter=16, low_traverse=20, blocks_per_cu=6, supergrid_x=-1, supergrid_y=-1, supergrid_z=-1)
for kv in os.environ.get("RT_TUNE_OPTS", "").split(","):      # e.g. RT_TUNE_OPTS=launch_log2k=3,chunk=512
    if "=" in kv:
        base[kv.split("=")[0]] = int(kv.split("=")[1])
run(base, 1)
res = []
if what == "thresholds":
    for te, ts_, lo in itertools.product((8, 16, 24, 32, 48), (8, 16, 24, 32), (8, 16, 32)):
        o = dict(base, th_event=te, th_scatter=ts_, low_traverse=lo); res.append((run(o), te, ts_, lo))
elif what == "grid":
    for sg in ((0, 0, 0), (3, 3, 2), (4, 4, 2), (4, 4, 4), (6, 6, 4), (8, 8, 4), (8, 8, 8), (12, 12, 8), (16, 16, 8)):
        o = dict(base, supergrid_x=sg[0], supergrid_y=sg[1], supergrid_z=sg[2]); res.append((run(o),) + sg)
elif what == "pool":
    for pl, b in ((0, 32), (64, 32), (64, 24), (96, 32), (96, 20), (96, 16), (128, 32), (128, 16), (128, 12), (160, 32), (160, 12), (192, 32), (256, 32)):
        o = dict(base, pool=pl, pool_waves_per_cu=b); res.append((run(o), pl, b))
elif what == "pool2":
    for pl, b in ((64, 32), (64, 24), (96, 23), (96, 20), (96, 16), (128, 32), (128, 18), (128, 16), (128, 14), (128, 12), (160, 14), (160, 12), (192, 12), (192, 10), (256, 8)):
        o = dict(base, pool=pl, pool_waves_per_cu=b); res.append((run(o), pl, b))
elif what == "launch2":
    for lk in (-1, 3, 5, 6, 7, 8):
        o = dict(base, pool=128, launch_log2k=lk); res.append((run(o, 3), lk))
elif what == "merge":
    for sg in ((4, 4, 4), (4, 4, 2), (8, 8, 4), (2, 2, 2)):
        for tau in (100, 500, 2000):
            o = dict(base, pool=128, levels=4, merge_tau_milli=tau, supergrid_x=sg[0], supergrid_y=sg[1], supergrid_z=sg[2])
            res.append((run(o, 3),) + sg + (tau,))
elif what == "merge2":
    for sg, lv in (((4, 4, 2), 4), ((2, 2, 2), 5), ((4, 4, 4), 4), ((2, 2, 1), 5), ((4, 4, 2), 5)):
        for tau in (2000, 4000, 8000, 16000):
            o = dict(base, pool=128, levels=lv, merge_tau_milli=tau, supergrid_x=sg[0], supergrid_y=sg[1], supergrid_z=sg[2])
            res.append((run(o, 3),) + sg + (lv, tau))
elif what == "merge3":
    for sg, lv in (((4, 4, 2), 4), ((3, 3, 2), 5), ((4, 4, 3), 4), ((2, 2, 2), 5), ((3, 3, 3), 5), ((4, 2, 2), 5)):
        for tau in (1000, 2000, 3000):
            o = dict(base, pool=128, levels=lv, merge_tau_milli=tau, supergrid_x=sg[0], supergrid_y=sg[1], supergrid_z=sg[2])
            res.append((run(o, 3),) + sg + (lv, tau))
elif what == "merge4":
    for sg, lv in (((3, 3, 2), 5), ((2, 2, 2), 5), ((3, 3, 1), 5), ((2, 2, 1), 6)):
        for tau in (250, 500, 750, 1000, 1250, 1500):
            o = dict(base, levels=lv, merge_tau_milli=tau, supergrid_x=sg[0], supergrid_y=sg[1], supergrid_z=sg[2])
            res.append((run(o, 3),) + sg + (lv, tau))
elif what == "regions":
    # anisotropic majorant regions (round 3): supervoxel size x largest level x merge criterion; with the events per photon
    def events(o):
        for k, v in o.items():
            tr.set_option(k, v)
        tr.set_option("stats", 1); tr.reset(); tr.trace(1, 0, n); tr.synchronize(); ev = tr.events(); tr.set_option("stats", 0)
        return "steps %.2f coll %.2f trans %.2f" % (ev["steps"] / n, ev["collision_tests"] / n, ev["transition"] / n)
    combos = os.environ.get("RT_TUNE_REGIONS", "3x3x2:5:1000,3x3x2:7:1000,3x3x2:7:500,3x3x2:7:2000,2x2x2:7:1000,2x2x1:7:1000,3x3x1:7:1000,4x4x2:7:1000")
    for item in combos.split(","):
        sgs, lv, tau = item.split(":")
        sg = tuple(int(v) for v in sgs.split("x"))
        o = dict(base, levels=int(lv), merge_tau_milli=int(tau), supergrid_x=sg[0], supergrid_y=sg[1], supergrid_z=sg[2])
        res.append((run(o, 3), item, events(o)))
elif what == "thl":
    for thl in (16, 24, 32, 40, 48, 64):
        o = dict(base, pool=128, th_launch=thl); res.append((run(o, 3), thl))
elif what == "levels":
    for lv in (0, 1, 2, 3, 4):
        o = dict(base, pool=128, levels=lv); res.append((run(o, 3), lv))
elif what == "one":
    o = dict(base); res.append((run(o, 5), "default"))
elif what == "chunk":
    for ch in (64, 128, 256, 512, 1024, 2048, 4096, 16384):
        o = dict(base, pool=128, chunk=ch); res.append((run(o, 5), ch))
elif what == "launch":
    for lk, thl in ((-1, 32), (-1, 16), (-1, 48), (-1, 64), (-1, 96), (5, 32), (3, 32), (7, 32), (0, 32), (10, 32), (5, 64)):
        o = dict(base, pool=128, launch_log2k=lk, th_launch=thl); res.append((run(o, 5), lk, thl))
elif what == "blocks":
    for b in (2, 3, 4, 5, 6, 7, 8, 10, 12):
        o = dict(base, blocks_per_cu=b); res.append((run(o), b))
res.sort()
for r in res[:24]:
    print(wl, what, "ms=%.3f" % r[0], "Gph/s=%.3f" % (n / r[0] / 1e6), r[1:])
print("worst", res[-1])
