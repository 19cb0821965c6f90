#!/usr/bin/env python3
"""Why is mean(z^2) of the C1 run against the compiled reference's fixture 0.925? One GPU, a few seconds.

Traces C1 (2^28 photons, uniform launch) under several seeds and compares (a) every run with the fixture
reference_c1_2e8.npz and (b) the runs with each other, with the z-score statistic of tests/test_gpu_parity.py. If
the GPU path were under-dispersed, (b) would be below 1 as well; if the fixture's one fixed sample happens to lie
close to the truth, (a) is low for EVERY seed (the fixture's share of each z is the same numbers every time) while
(b) sits at 1. The five 2-D tallies of a block are fed by the same photons, so their z-scores are correlated and
the pooled statistic has fewer independent terms than bins: its spread is taken from (b), not from 2 / bins.
Test infrastructure (reads tests/golden). Usage: python tools/dispersion_check.py [seeds=6] > profiles/r3_c1_dispersion.json"""
import itertools
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracer_amd as rt      # noqa: E402
import oracle_lib as O          # noqa: E402

TWO_D = O.TALLY_NAMES[:5]


def blocks(a, b=4):
    j, i = a.shape
    return a.reshape(j // b, b, i // b, b).sum(axis=(1, 3)).astype(np.float64)


def z(xa, na, xb, nb, min_count=50):
    xa, xb = np.asarray(xa, dtype=np.float64).ravel(), np.asarray(xb, dtype=np.float64).ravel()
    m = xb >= min_count
    pooled = (xa[m] + xb[m]) / (na + nb)
    return (xa[m] / na - xb[m] / nb) / np.sqrt(pooled * (1 / na + 1 / nb))


def summary(t):
    s = {"blocks_" + n: blocks(t[n]) for n in TWO_D}
    s["level_direct"] = t["atmos_direct"].sum(axis=(1, 2)).astype(np.float64)
    s["level_diffuse"] = t["atmos_diffuse"].sum(axis=(1, 2)).astype(np.float64)
    return s


def z2(a, na, b, nb):
    zs = [z(a[k], na, b[k], nb) for k in list(a)]
    allz = np.concatenate(zs)
    return float(np.mean(allz ** 2)), {k: float(np.mean(v ** 2)) for k, v in zip(a, zs)}, int(allz.size)


def main():
    nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    scene = rt.scenes.block_cloud()
    n = 1 << 28
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_c1_2e8.npz"))
    nb = int(g["sums"][0])
    fixture = {k: g[k].astype(np.float64) for k in ["blocks_" + t for t in TWO_D] + ["level_direct", "level_diffuse"]}
    runs = []
    tr = rt.PhotonTracer.from_scene(scene)
    for s in range(nseeds):
        tr.reset(); tr.trace(2024 + 1000 * s, 0, n); tr.synchronize()
        runs.append(summary(tr.tallies()))
    tr.close()
    vs_fixture = [z2(r, n, fixture, nb) for r in runs]
    pairs = [z2(runs[i], n, runs[j], n) for i, j in itertools.combinations(range(nseeds), 2)]
    pa = np.array([p[0] for p in pairs])
    out = {"gpu_photons": n, "reference_photons": nb, "bins": vs_fixture[0][2],
           "gpu_vs_fixture_z2_all": [v[0] for v in vs_fixture],
           "gpu_vs_fixture_z2_by_tally_seed0": vs_fixture[0][1],
           "gpu_vs_gpu_z2_all": [p[0] for p in pairs],
           "gpu_vs_gpu_mean": float(pa.mean()), "gpu_vs_gpu_std_of_one_comparison": float(pa.std(ddof=1)),
           "naive_std_sqrt_2_over_bins": float(np.sqrt(2.0 / vs_fixture[0][2]))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
