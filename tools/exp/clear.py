import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import raytracer_amd as rt
rt.tracer.LIB_PATH = os.path.abspath("tools/exp/librt_CLEAR.so")
for name in ("cumulus_c4", "cumulus_c3", "block_cloud"):
    sc = getattr(rt.scenes, name)()
    tr = rt.PhotonTracer.from_scene(sc, stats=1)
    n = 1 << 24
    tr.trace(1, 0, n); tr.synchronize()
    ev = tr.events()
    print(name, {k: round(v / n, 3) for k, v in ev.items()})
    tr.close()
