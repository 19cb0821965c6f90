#!/usr/bin/env python3
"""North-star-sized statistical fixtures (run in the build container, where /root/reference exists;
about 15 minutes on 8 cores).

reference_c1_2e8.npz  the COMPILED REFERENCE itself, `oracle/_ref/raytracer_3d_ref 2000` (2*10^8 photons of
                      its built-in block-cloud scene, 8 OpenMP threads; CLI raytracer_3d.cpp:541-554, outputs
                      :531-537): global sums, per-level absorption (direct, diffuse), 4x4-column-block sums of
                      the five 2-D tallies, and the per-cell absorption of the 20x20x20 cloud block.
cpu_ref_c3_1e8.npz    oracle/cpu_ref.cpp (bit-pinned to raytracer_3d.cpp by cpu_ref_100k.json) on
                      scenes.cumulus_c3(), >= 1.2*10^8 photons, 8 threads: sums, per-level sums, 4x4 blocks.
cpu_ref_c4_6e7.npz    the same on scenes.cumulus_c4() (1024x1024x128), >= 6*10^7 photons: sums, per-level
                      sums, 16x16-column-block sums.
cpu_ref_sun0_5e7.npz, cpu_ref_sun75_5e7.npz   the same on a 96x80x48 cumulus scene under two other suns: overhead
                      (zenith 0, azimuth 0: two direction components exactly zero) over a bright surface (albedo 0.7),
                      and low (zenith 75, azimuth 213) over a black one; 5*10^7 photons each, 2x2-column blocks.
reference_2d_2e8.npz  the COMPILED 2D REFERENCE, `oracle/_ref/raytracer_2d_ref 2000` (2*10^8 photons of the built-in scene
                      of raytracer.cpp, 8 threads): sums of its six tallies, the five 1-D tallies per column, and
                      its atmos array (128 levels x 256 columns, direct + diffuse) cell by cell.
Each file also holds the md5 of the scene's four float32 fields, so a drifting scene generator is noticed.
Fixtures are counts only; no reference source is stored."""
import hashlib
import os
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np  # noqa: E402
import oracle_lib as O  # noqa: E402
from raytracer_amd import scenes  # noqa: E402

TWO_D = O.TALLY_NAMES[:5]


def scene_md5(scene):
    h = hashlib.md5()
    for a in scene.fields():
        h.update(np.ascontiguousarray(a, dtype=np.float32).tobytes())
    return h.hexdigest()


def blocks(a, b):
    j, i = a.shape
    return a.reshape(j // b, b, i // b, b).sum(axis=(1, 3)).astype(np.uint64)


def summary(t, block):
    out = {"sums": np.array([int(t[n].sum()) for n in O.TALLY_NAMES], dtype=np.uint64),
           "level_direct": t["atmos_direct"].sum(axis=(1, 2)).astype(np.uint64),
           "level_diffuse": t["atmos_diffuse"].sum(axis=(1, 2)).astype(np.uint64),
           "block": np.array(block)}
    for n in TWO_D:
        out["blocks_" + n] = blocks(t[n], block)
    return out


def other_sun(zenith, azimuth, albedo):
    sc = scenes.cumulus(96, 80, 48, 50.0, 50.0, 25.0, n_clouds=12, seed=3, z_base=300.0, z_top=900.0)
    sc.zenith_deg, sc.azimuth_deg, sc.albedo = zenith, azimuth, albedo
    return sc


which = sys.argv[1:] or ["c1", "c3", "c4", "sun0", "sun75", "2d"]
if "2d" in which:
    t0 = time.time()
    with tempfile.TemporaryDirectory() as d:
        r = O.run_reference_binary_2d(2000, d, 8)
    s2 = {"sums": np.array([int(r[n].sum()) for n in O.TALLY_NAMES_2D], dtype=np.uint64),
          "atmos": r["atmos"].astype(np.uint32), "threads": np.array(8)}
    for n in O.TALLY_NAMES_2D[:5]:
        s2["columns_" + n] = r[n].astype(np.uint64)
    np.savez_compressed(os.path.join(HERE, "reference_2d_2e8.npz"), **s2)
    print("2d", int(r["toa_down"].sum()), "photons", time.time() - t0, "s", flush=True)
if "c1" in which:
    t0 = time.time()
    with tempfile.TemporaryDirectory() as d:
        r = O.run_reference_binary(2000, d, 8)
    s = summary(r, 4)
    # the cloud block of raytracer_3d.cpp:186-193: x in (4000, 5000), y in (2700, 3700), z in (1000, 1500)
    s["cloud_cells_direct"] = r["atmos_direct"][40:60, 54:74, 80:100].astype(np.uint32)
    s["cloud_cells_diffuse"] = r["atmos_diffuse"][40:60, 54:74, 80:100].astype(np.uint32)
    s["scene_md5"] = np.array(scene_md5(scenes.block_cloud()))
    s["threads"] = np.array(8)
    np.savez_compressed(os.path.join(HERE, "reference_c1_2e8.npz"), **s)
    print("c1", int(r["toa_down"].sum()), "photons", time.time() - t0, "s", flush=True)
for key, factory, n, block, fname in (("c3", scenes.cumulus_c3, 120_000_000, 4, "cpu_ref_c3_1e8.npz"),
                                      ("c4", scenes.cumulus_c4, 60_000_000, 16, "cpu_ref_c4_6e7.npz"),
                                      ("sun0", lambda: other_sun(0.0, 0.0, 0.7), 50_000_000, 2, "cpu_ref_sun0_5e7.npz"),
                                      ("sun75", lambda: other_sun(75.0, 213.0, 0.0), 50_000_000, 2, "cpu_ref_sun75_5e7.npz")):
    if key not in which:
        continue
    t0 = time.time()
    sc = factory()
    t, _ = O.cpu_ref(O.scene_to_f64(sc), n, nthreads=8)
    assert O.balance(t) == 0
    s = summary(t, block)
    s["scene_md5"] = np.array(scene_md5(sc))
    s["threads"] = np.array(8)
    np.savez_compressed(os.path.join(HERE, fname), **s)
    print(key, int(t["toa_down"].sum()), "photons", time.time() - t0, "s", flush=True)
