"""SURVEY.md §8 row f4 — the reference's 2D tracer (src/raytracer.cpp).

Parity chain of the 2D mode, mirroring the 3D one:
  compiled reference (oracle/_ref/raytracer_2d_ref)  ==bit-exact==  oracle/cpu_ref_2d.cpp
  oracle/cpu_ref_2d.cpp  ~~statistical (4 sigma)~~  oracle/photon_oracle.c with planar = 1
  photon_oracle.c planar  ==bit-exact==  HIP pool kernel, option "planar"   (tests/test_gpu_parity.py)"""
import hashlib
import json
import os
import tempfile

import numpy as np
import pytest

import oracle_lib as O
from raytracer_amd import scenes

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "cpu_ref_2d_100k.json")))


def scene_2d():
    """The 2D built-in scene as a jtot = 1 grid of the 3D tracer (float32, sun in the x-z plane)."""
    s = O.block_cloud_2d_f64()
    f = [s[k].astype(np.float32).reshape(s["ktot"], 1, s["itot"]) for k in ("k_gas", "k_cloud", "ssa", "asy")]
    return scenes.Scene("block_cloud_2d", s["itot"], 1, s["ktot"], s["dx"], s["dx"], s["dx"], *f,
                        albedo=0.2, zenith_deg=50.0, azimuth_deg=0.0)


@pytest.fixture(scope="module")
def run_100k():
    return O.cpu_ref_2d(O.block_cloud_2d_f64(), 100000, nthreads=1)[0]


def test_2d_restatement_bit_exact_against_golden_digests(run_100k):
    for n in O.TALLY_NAMES_2D:
        assert int(run_100k[n].sum()) == GOLD["pins"][n]["sum"], n
        assert hashlib.md5(run_100k[n].tobytes()).hexdigest() == GOLD["pins"][n]["md5"], n
    assert O.balance_2d(run_100k) == 0


@pytest.mark.skipif(not os.path.exists(O.REF2D_BIN), reason="compiled 2D reference (oracle/_ref) not built here")
def test_2d_restatement_bit_exact_against_compiled_reference_live(run_100k):
    with tempfile.TemporaryDirectory() as d:
        ref = O.run_reference_binary_2d(1, d, nthreads=1)
    for n in O.TALLY_NAMES_2D:
        assert np.array_equal(ref[n], run_100k[n]), n


def test_planar_float_oracle_agrees_with_2d_reference_statistics():
    """float32 + Philox + supergrid traversal + in-plane rotation by angle-sum formulas against the compiled
    reference's sums at 4*10^6 photons: global fractions within 4 sigma, per-level absorption z-scores ~ N(0,1)."""
    n = 2_000_000
    t, ev = O.photon_oracle(scene_2d(), 3, 0, n, planar=True)
    assert O.balance(t) == 0 and int(t["toa_down"].sum()) == n
    assert not t["atmos_direct"][:, 0, :].sum() == 0
    g, ng = GOLD["sums_4e6"], GOLD["sums_4e6"]["toa_down"]
    mine = {k: int(t[k].sum()) for k in ("toa_up", "surface_down_direct", "surface_down_diffuse", "surface_up")}
    mine["atmos"] = int(t["atmos_direct"].sum() + t["atmos_diffuse"].sum())
    for name, v in mine.items():
        p, pg = v / n, g[name] / ng
        sig = np.sqrt(pg * (1 - pg) * (1 / n + 1 / ng))
        assert abs(p - pg) < 4 * sig, (name, p, pg, sig)
    lev = (t["atmos_direct"] + t["atmos_diffuse"]).sum(axis=(1, 2)).astype(np.float64)
    levg = np.array(GOLD["atmos_level_sums_4e6"], dtype=np.float64)
    pooled = (lev + levg) / (n + ng)
    z = (lev / n - levg / ng) / np.sqrt(pooled * (1 / n + 1 / ng))
    assert abs(z.mean()) < 0.5 and 0.6 < z.std() < 1.5 and np.abs(z).max() < 5, (z.mean(), z.std(), np.abs(z).max())
    # directions never leave the plane: no photon is ever tallied with a y index other than 0 (jtot = 1),
    # and the planar rule differs from the 3D rule on the same scene
    t3, _ = O.photon_oracle(scene_2d(), 3, 0, 200_000, planar=False)
    tp, _ = O.photon_oracle(scene_2d(), 3, 0, 200_000, planar=True)
    assert any(not np.array_equal(t3[k], tp[k]) for k in O.TALLY_NAMES)


def test_large_2d_fixture_is_consistent(run_100k):
    """tests/golden/reference_2d_2e8.npz (make_golden_large.py: 2*10^8 photons of the compiled 2D reference): the
    sums are those of the stored arrays, the photons balance, and a fresh 10^5-photon run of the restatement falls
    within 4 sigma of its fractions."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_2d_2e8.npz"))
    s = dict(zip(O.TALLY_NAMES_2D, (int(v) for v in g["sums"])))
    assert s["toa_down"] >= 200_000_000
    assert s["toa_down"] == s["surface_down_direct"] + s["surface_down_diffuse"] - s["surface_up"] + s["toa_up"] + s["atmos"]
    assert int(g["atmos"].sum(dtype=np.int64)) == s["atmos"] and g["atmos"].shape == (128, 256)
    for name in O.TALLY_NAMES_2D[:5]:
        assert int(g["columns_" + name].sum(dtype=np.int64)) == s[name] and g["columns_" + name].size == 256
    n, nb = int(run_100k["toa_down"].sum()), s["toa_down"]
    for name in O.TALLY_NAMES_2D[1:]:
        p, q = int(run_100k[name].sum()) / n, s[name] / nb
        assert abs(p - q) < 4 * np.sqrt(q * (1 - q) * (1 / n + 1 / nb)), name
