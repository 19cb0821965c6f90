#!/usr/bin/env python3
"""Randomised parity sweep on one GPU: random small scenes (ragged grids, clouds that touch the surface or the
domain top, cloud-free and gas-free columns, a single slab) x random options (pool size, supervoxel size, region
levels, launch order, planar scattering, fixed dot, chunk, launch threshold) x random photon ID ranges, the HIP
kernels against oracle/photon_oracle.c, every tally and event counter bit for bit. Later case ranges add options
without changing what earlier case numbers mean: launch sort and tiled launch order (all), kernels without event
counters and several sort batches (from 300 000), grids of whole 32-column tiles (400 000), split calls through
rt_trace_noalloc (500 000), grids of one to four cells per axis (700 000), any sun angle and a context used twice
with a new sun or new optics (1 100 000), the Sobol launch (2 000 000).

Test infrastructure like tests/ (it calls the oracle as the checker). Usage, on the GPU box:
    python tools/fuzz_parity.py [cases=200] [first_case=0]
Exit status 1 and the failing case's parameters on the first difference."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O          # noqa: E402
import raytracer_amd as rt      # noqa: E402


def random_scene(rng, large=False, degenerate=False):
    S = rt.scenes
    f = np.float32
    itot, jtot, ktot = int(rng.integers(3, 41)), int(rng.integers(3, 41)), int(rng.integers(2, 25))
    if degenerate:  # one or two cells along some axes: a single column, a single level, the planar grid (jtot = 1)
        itot, jtot, ktot = int(rng.integers(1, 5)), int(rng.integers(1, 4)), int(rng.integers(1, 4))
    if large:       # sides that are whole numbers of 32-column tiles: the tiled launch orders and many sort tiles
        itot, jtot, ktot = 32 * int(rng.integers(1, 6)), 32 * int(rng.integers(1, 6)), int(rng.integers(8, 41))
    dx, dy, dz = float(rng.uniform(20, 200)), float(rng.uniform(20, 200)), float(rng.uniform(10, 120))
    ztop = ktot * dz
    kind = rng.choice(["clouds", "deck", "clear", "to_ground", "to_top", "everywhere"])
    k_gas = float(rng.choice([0.0, 1e-5, 1e-4, 1e-3]))
    ssa_gas = float(rng.choice([1.0, 0.5, 0.9]))
    asy_cloud = float(rng.choice([0.85, 0.0, -0.4, 0.5, 0.95, 5e-4]))
    ssa_cloud = float(rng.choice([1.0, 0.9, 0.6]))
    kc_lo = float(rng.choice([1e-3, 5e-3, 2e-2]))
    if kind == "clear":
        sc = S.cumulus(itot, jtot, ktot, dx, dy, dz, n_clouds=0, seed=int(rng.integers(1 << 30)), k_gas=max(k_gas, 1e-5),
                       ssa_gas=ssa_gas, name="fuzz_clear")
    else:
        zb, zt = {"clouds": (0.25 * ztop, 0.7 * ztop), "deck": (0.3 * ztop, 0.5 * ztop), "to_ground": (0.0, 0.5 * ztop),
                  "to_top": (0.5 * ztop, ztop), "everywhere": (0.0, ztop)}[kind]
        sc = S.cumulus(itot, jtot, ktot, dx, dy, dz, n_clouds=int(rng.integers(1, 8)), seed=int(rng.integers(1 << 30)),
                       z_base=zb, z_top=zt, k_gas=k_gas, ssa_gas=ssa_gas, ssa_cloud=ssa_cloud, asy_cloud=asy_cloud,
                       k_cloud_range=(kc_lo, kc_lo * 8), name="fuzz_" + kind)
        if kind in ("deck", "everywhere"):
            # a closed layer: every column of some levels is cloudy
            k0, k1 = int(zb / dz), max(int(zb / dz) + 1, int(zt / dz))
            kc = sc.k_cloud.copy()
            kc[k0:k1] = np.maximum(kc[k0:k1], f(kc_lo))
            ssa, asy = S._mix(sc.k_gas, ssa_gas, 0.0, kc, ssa_cloud, asy_cloud)
            sc = S.Scene(sc.name, itot, jtot, ktot, dx, dy, dz, sc.k_gas, kc, ssa, asy)
    sc.albedo = float(rng.choice([0.0, 0.2, 0.7, 1.0]))
    sc.zenith_deg = float(rng.choice([0.0, 20.0, 50.0, 75.0, 88.0]))
    sc.azimuth_deg = float(rng.choice([0.0, 20.0, 90.0, 213.0]))
    return sc, kind


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    for case in range(first, first + cases):
        rng = np.random.default_rng(1000 + case)
        rng2 = np.random.default_rng(77000 + case)          # options added later: a stream of its own, so that the
        large = case >= 400000 and rng2.random() < 0.15     # cases of earlier sweeps keep their definitions
        degenerate = case >= 700000 and not large and rng2.random() < 0.15
        scene, kind = random_scene(rng, large, degenerate)
        pool = int(rng.choice([0, 64, 96, 128, 128, 160, 192, 256]))
        sg = (0, 0, 0) if rng.random() < 0.15 else tuple(int(v) for v in rng.integers(1, 9, size=3))
        levels = int(rng.integers(0, 6))
        tau_milli = int(rng.choice([0, 37, 500, 2000, 100000]))
        log2k = int(rng.choice([-1, -1, 0, 2, 5]))
        planar = bool(pool != 0 and rng.random() < 0.2)
        fixed_dot = bool(rng.random() < 0.3)
        chunk = int(rng.choice([64, 256, 1024]))
        thl = int(rng.choice([1, 16, 40, 100]))
        n = int(rng.choice([1, 63, 1000, 20000, 60000]))
        pid0 = int(rng.integers(0, 1 << 39))
        seed = int(rng.integers(0, 1 << 62))
        sort = int(rng2.choice([-1, 0, 1, 1]))              # launch sort (round 2)
        sort_tile, sort_batch = int(rng2.integers(0, 7)), int(rng2.choice([16, 28]))
        tiles = [t for t in ((0, 1, 2, 3, 4, 5) if large else (0, 1, 2, 3)) if scene.itot % (1 << t) == 0 and scene.jtot % (1 << t) == 0]
        ltile = int(rng2.choice(tiles)) if log2k >= 0 else 0          # tiled stratified launch order
        stats = 1
        if case >= 300000:       # later additions (earlier case numbers keep their definitions): the kernels without
            stats = int(rng2.random() < 0.5)        # event counters (what production runs), and calls of several sort batches
            if rng2.random() < 0.08:
                n = int(rng2.choice([70001, 150001, 262144]))
        second = None
        if case >= 1100000:      # any sun and surface, tiles too large for the sort's toa_down pass, and a context that is
            rng3 = np.random.default_rng(55000 + case)      # used twice: a new sun or new optics, counts accumulating
            if rng3.random() < 0.5:
                scene.zenith_deg = float(rng3.uniform(0.0, 89.5))
                scene.azimuth_deg = float(rng3.uniform(0.0, 360.0))
                scene.albedo = float(rng3.uniform(0.0, 1.0))
            if rng3.random() < 0.3:
                sort_tile = int(rng3.integers(5, 10))
            if rng3.random() < 0.35:
                second = dict(kind=str(rng3.choice(["sun", "optics", "both"])), zenith=float(rng3.uniform(0.0, 89.5)),
                              azimuth=float(rng3.uniform(0.0, 360.0)), albedo=float(rng3.choice([0.0, 0.35, 1.0])),
                              scale=float(rng3.choice([0.25, 3.0])), n=int(rng3.choice([1, 999, 30000])),
                              seed=int(rng3.integers(0, 1 << 62)), upload=bool(rng3.random() < 0.5))
        if case >= 2000000:      # round 3: the Sobol launch (uniform-launch cases become Sobol with probability 1/2) and
            rng4 = np.random.default_rng(33000 + case)      # ID ranges around multiples of 2^32, where its shift changes
            if log2k < 0 and rng4.random() < 0.5:
                log2k = -2
                if rng4.random() < 0.4:
                    pid0 = (int(rng4.integers(1, 100)) << 32) - int(rng4.integers(0, max(n, 2)))
        desc = dict(case=case, kind=kind, grid=(scene.itot, scene.jtot, scene.ktot), albedo=scene.albedo, zenith=scene.zenith_deg,
                    pool=pool, supergrid=sg, levels=levels, tau_milli=tau_milli, log2k=log2k, planar=planar,
                    fixed_dot=fixed_dot, chunk=chunk, th_launch=thl, n=n, pid0=pid0, stats=stats,
                    launch_sort=sort, sort_tile_log2=sort_tile, sort_batch_log2=sort_batch, launch_tile_log2=ltile,
                    second=second)
        tr = rt.PhotonTracer.from_scene(scene, stats=stats, pool=pool, levels=levels, merge_tau_milli=tau_milli,
                                        launch_log2k=log2k, planar=int(planar), fixed_dot=int(fixed_dot), chunk=chunk,
                                        th_launch=thl, launch_sort=sort, sort_tile_log2=sort_tile,
                                        sort_batch_log2=sort_batch, launch_tile_log2=ltile)
        tr.set_supergrid(*sg)
        cuts = [0, n]
        if case >= 500000 and n > 2 and rng2.random() < 0.3:        # the range in two or three calls that accumulate,
            cuts = sorted({0, n, *(int(c) for c in rng2.integers(1, n, size=int(rng2.integers(1, 3))))})   # the later
        try:                                                        # ones through the entry point that allocates nothing
            for piece, (b, e) in enumerate(zip(cuts, cuts[1:])):
                if piece > 0 and pool > 0:
                    tr.trace_noalloc(seed, pid0 + b, e - b)
                else:
                    tr.trace(seed, pid0 + b, e - b)
            tr.synchronize()
            scene2 = None
            if second is not None:
                S = rt.scenes
                scene2 = scene
                if second["kind"] in ("optics", "both"):        # thinner or thicker clouds, darker droplets
                    kc = (scene.k_cloud * np.float32(second["scale"])).astype(np.float32)
                    ssa, asy = S._mix(scene.k_gas, 0.8, 0.0, kc, 0.95, 0.7)
                    scene2 = S.Scene(scene.name + "_2", scene.itot, scene.jtot, scene.ktot, scene.dx, scene.dy, scene.dz,
                                     scene.k_gas, kc, ssa, asy)
                    scene2.albedo, scene2.zenith_deg, scene2.azimuth_deg = scene.albedo, scene.zenith_deg, scene.azimuth_deg
                    tr.upload_optics(*scene2.fields())
                if second["kind"] in ("sun", "both"):
                    if scene2 is scene:
                        scene2 = S.Scene(scene.name + "_2", scene.itot, scene.jtot, scene.ktot, scene.dx, scene.dy, scene.dz,
                                         *scene.fields())
                    scene2.albedo, scene2.zenith_deg, scene2.azimuth_deg = second["albedo"], second["zenith"], second["azimuth"]
                    tr.set_scene(scene2.zenith_deg, scene2.azimuth_deg, scene2.albedo)
                if second["upload"]:                              # the counts so far leave the device and come back
                    held = tr.tallies()
                    tr.reset()
                    tr.add_tallies(held)
                tr.trace(second["seed"], pid0 + n, second["n"])
                tr.synchronize()
        except Exception as exc:
            print("ERROR", desc, "seed:", seed, exc, flush=True)
            return 1
        got, gev = tr.tallies(), tr.events()
        tr.close()
        want, wev = O.photon_oracle(scene, seed, pid0, n, fixed_dot=fixed_dot, supergrid=sg, launch_log2k=log2k,
                                    planar=planar, levels=levels, merge_tau=np.float32(tau_milli / 1000.0),
                                    launch_tile_log2=ltile)
        if second is not None:                  # the oracle adds into the same arrays; a reset in between zeroes the
            want, e2 = O.photon_oracle(scene2, second["seed"], pid0 + n, second["n"], tallies=want,    # event counters
                                       fixed_dot=fixed_dot, supergrid=sg, launch_log2k=log2k, planar=planar, levels=levels,
                                       merge_tau=np.float32(tau_milli / 1000.0), launch_tile_log2=ltile)
            wev = e2 if second["upload"] else {k: wev[k] + e2[k] for k in wev}
        bad = [name for name in O.TALLY_NAMES if not np.array_equal(got[name], want[name])]
        if bad or (stats and gev != wev):
            print("MISMATCH", desc, "tallies:", bad, "events gpu/oracle:", gev, wev, flush=True)
            return 1
        if case % 20 == 0:
            print("ok", desc, flush=True)
    print(f"{cases} cases bit-identical", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
