"""Pins oracle/photon_oracle.c (float32 / Philox restatement of the CUDA kernel's life cycle):
Philox known answers, accuracy of its libm-free math, a frozen golden vector, physics closures,
and statistical agreement with oracle/cpu_ref.cpp (itself bit-pinned to the reference)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
from raytracer_amd import scenes

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    kat = [([0] * 4, [0] * 2, [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, want in kat:
        o = (C.c_uint32 * 4)()
        O.lib().oracle_philox4x32_10((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), o)
        assert list(o) == want


def test_math_accuracy():
    L = O.lib()
    u = np.arange(0, 1 << 24, 4099, dtype=np.uint32).astype(np.float32) * np.float32(2.0 ** -24)
    x = (np.float32(1) - u) + np.float32(1.1920928955078125e-07)
    got = np.array([L.oracle_logf(float(v)) for v in x])
    assert np.max(np.abs(got - np.log(x.astype(np.float64)))) < 2e-6
    assert np.max(np.abs(got - np.log(x.astype(np.float64))) / np.maximum(np.abs(np.log(x.astype(np.float64))), 1e-3)) < 1e-5
    s, c = C.c_float(), C.c_float()
    for v in u[::7]:
        L.oracle_sincos2pi(float(v), C.byref(s), C.byref(c))
        assert abs(s.value - np.sin(2 * np.pi * float(v))) < 5e-7 and abs(c.value - np.cos(2 * np.pi * float(v))) < 5e-7
    for v in np.linspace(0.2, 4.3, 2001, dtype=np.float32):
        assert abs(L.oracle_inv_cbrtf(float(v)) - float(v) ** (-1 / 3)) < 3e-7 * float(v) ** (-1 / 3)


def test_sampler_moments():
    """<mu>_HG = g, <mu>_Rayleigh = 0, <mu^2>_Rayleigh = 2/5."""
    L = O.lib()
    u = (np.arange(200000, dtype=np.float64) + 0.5) / 200000
    ray = np.array([L.oracle_rayleigh_mu(float(v)) for v in u.astype(np.float32)])
    assert abs(ray.mean()) < 1e-4 and abs((ray ** 2).mean() - 0.4) < 1e-4 and np.abs(ray).max() <= 1.0 + 1e-6
    for g in (0.85, 0.5, -0.3, 0.0):
        hg = np.array([L.oracle_henyey_mu(g, float(v)) for v in u.astype(np.float32)])
        assert abs(hg.mean() - g) < 2e-4 and np.abs(hg).max() <= 1.0 + 1e-5


def test_frozen_golden_vector():
    gold = np.load(os.path.join(GOLD, "photon_oracle_tiny.npz"))
    t, ev = O.photon_oracle(scenes.tiny(), 11, 0, 1 << 14, nthreads=0)
    for n in O.TALLY_NAMES:
        assert np.array_equal(t[n], gold[n]), n
    assert [ev[k] for k in O.EVENT_NAMES] == [int(v) for v in gold["events"]]


def test_partition_and_thread_independence():
    sc = scenes.tiny()
    whole, _ = O.photon_oracle(sc, 3, 100, 30000, nthreads=1)
    parts = O.new_tallies(sc)
    for b, c in ((100, 7), (107, 9993), (10100, 20000)):
        O.photon_oracle(sc, 3, b, c, nthreads=0, tallies=parts)
    for n in O.TALLY_NAMES:
        assert np.array_equal(whole[n], parts[n])
    assert O.balance(whole) == 0 and int(whole["toa_down"].sum()) == 30000


def test_clear_sky_closure_and_beer_lambert():
    """C2. Conservative gas (ssa = 1): nothing absorbed in the air. Absorbing gas: the direct
    beam at the surface follows exp(-tau / mu0)."""
    n = 1 << 20
    t, ev = O.photon_oracle(scenes.rayleigh_column(ssa=1.0), 1, 0, n)
    assert int(t["atmos_direct"].sum()) + int(t["atmos_diffuse"].sum()) == 0
    assert O.balance(t) == 0 and ev["abort"] == 0
    sc = scenes.rayleigh_column(ssa=0.5)
    t, _ = O.photon_oracle(sc, 2, 0, n)
    tau = 1.0e-4 * sc.ktot * sc.dz
    expect = np.exp(-tau / np.cos(np.deg2rad(50.0)))
    p = int(t["surface_down_direct"].sum()) / n
    assert abs(p - expect) < 4 * np.sqrt(expect * (1 - expect) / n) + 1e-4


def test_surface_albedo_and_lambert():
    sc = scenes.rayleigh_column(ssa=1.0)
    t, _ = O.photon_oracle(sc, 4, 0, 1 << 20)
    down = int(t["surface_down_direct"].sum()) + int(t["surface_down_diffuse"].sum())
    up = int(t["surface_up"].sum())
    assert abs(up / down - 0.2) < 4 * np.sqrt(0.16 / down)


@pytest.mark.parametrize("name", ["block_cloud", "tiny"])
def test_statistical_parity_with_cpu_ref(name):
    """global fractions within 4 sigma of the reference restatement, per-column
    and per-level z-scores consistent with N(0,1)."""
    sc = scenes.block_cloud() if name == "block_cloud" else scenes.tiny()
    n = 1 << 22
    a, ev = O.photon_oracle(sc, 99, 0, n)
    b, _ = O.cpu_ref(O.scene_to_f64(sc), n, nthreads=0)
    assert ev["abort"] == 0 and O.balance(a) == 0 and O.balance(b) == 0
    na, nb = int(a["toa_down"].sum()), int(b["toa_down"].sum())
    assert na == n
    for t in O.TALLY_NAMES[1:]:
        pa, pb = int(a[t].sum()) / na, int(b[t].sum()) / nb
        sig = np.sqrt(max(pb * (1 - pb), 1e-12) * (1 / na + 1 / nb))
        assert abs(pa - pb) < 4 * sig, (t, pa, pb, (pa - pb) / sig)

    def zscores(xa, xb):
        xa, xb = xa.astype(np.float64).ravel(), xb.astype(np.float64).ravel()
        m = (xa + xb) >= 20
        pooled = (xa + xb) / (na + nb)
        return ((xa / na - xb / nb) / np.sqrt(pooled * (1 / na + 1 / nb)))[m]
    def gate(z):
        # mean(z^2) of N standard normal z-scores scatters by sqrt(2 / N) around 1: 0.15, or 3.5 of those if larger
        # (the tiny scene has 160 level-by-row bins only)
        return abs(np.mean(z ** 2) - 1.0) < max(0.15, 3.5 * np.sqrt(2.0 / z.size))
    for t in ("surface_down_direct", "surface_down_diffuse", "toa_up"):
        z = zscores(a[t], b[t])
        assert z.size > 50 and np.max(np.abs(z)) < 6.0 and gate(z), (t, np.mean(z ** 2))
    atm_a = (a["atmos_direct"] + a["atmos_diffuse"]).sum(axis=2)
    atm_b = (b["atmos_direct"] + b["atmos_diffuse"]).sum(axis=2)
    z = zscores(atm_a, atm_b)
    assert np.max(np.abs(z)) < 6.0 and gate(z), (np.mean(z ** 2), z.size)


def test_fixed_dot_changes_and_conserves():
    sc = scenes.tiny()
    a, _ = O.photon_oracle(sc, 5, 0, 1 << 16, fixed_dot=False)
    b, _ = O.photon_oracle(sc, 5, 0, 1 << 16, fixed_dot=True)
    assert O.balance(a) == 0 and O.balance(b) == 0
    assert any(not np.array_equal(a[n], b[n]) for n in O.TALLY_NAMES)


def test_slab_derivation():
    sl = O.oracle_slabs(scenes.block_cloud())
    assert sl.nslab == 3 and list(sl.zface) == [0.0, 1000.0, 1500.0, 3200.0] and (sl.k_lo, sl.k_hi) == (40, 59)
    assert sl.kmaj[0] == np.float32(1e-4) and sl.kmaj[1] == np.float32(1e-4) + np.float32(5e-3)
    sl = O.oracle_slabs(scenes.rayleigh_column())
    assert sl.nslab == 1 and (sl.k_lo, sl.k_hi) == (-1, -1) and sl.zface[1] == 3200.0


# ---- Sobol launch (reference: raytracer_kernels.cu:239-253, 331) ------------------
def _elementary_counts_ok(words, m):
    """every elementary interval 2^-a x 2^-(m-a) of the unit square holds exactly one of the 2^m points"""
    x, y = words[:, 0].astype(np.uint64), words[:, 1].astype(np.uint64)
    for a in range(m + 1):
        box = ((x >> np.uint64(32 - a)) << np.uint64(m - a)) | (y >> np.uint64(32 - (m - a))) if 0 < a < m else \
            (y >> np.uint64(32 - m) if a == 0 else x >> np.uint64(32 - m))
        if not np.array_equal(np.sort(box), np.arange(1 << m, dtype=np.uint64)):
            return False
    return True


def test_sobol_words_are_the_joe_kuo_sequence_of_scipy():
    """The unshifted launch words against an independent published implementation: scipy.stats.qmc.Sobol holds the
    Joe-Kuo direction numbers (dimensions 0 and 1 are the tables cuRAND hands the reference, raytracer_3d.cu:135-140)
    and walks them in Gray-code order, so its point n is the oracle's point n ^ (n >> 1)."""
    qmc = pytest.importorskip("scipy.stats.qmc")
    n = 1 << 14
    want = qmc.Sobol(d=2, scramble=False, bits=32).random(n)
    ids = np.arange(n)
    got = O.sobol_words(0, ids ^ (ids >> 1), shifted=False)
    assert np.array_equal(got.astype(np.float64) / 2.0 ** 32, want)


@pytest.mark.parametrize("m,first", [(4, 0), (7, 5 << 7), (10, 3 << 10), (12, (1 << 32) - (1 << 12)), (12, (7 << 32) + (9 << 12))])
def test_sobol_launch_fills_every_elementary_interval(m, first):
    """(0,2)-sequence property, digital shift included: an aligned block of 2^m consecutive photon IDs puts exactly
    one point into each elementary interval 2^-a x 2^-(m-a), a = 0..m -- also in the last block below a multiple of
    2^32 and in blocks beyond it (their shift is another draw)."""
    w = O.sobol_words(99, np.arange(first, first + (1 << m)), shifted=True)
    assert _elementary_counts_ok(w, m)
    assert not _elementary_counts_ok(O.sobol_words(99, np.arange(first + 1, first + 1 + (1 << m))), m) or m < 3


def test_sobol_launch_is_even_on_a_grid_that_is_no_power_of_two():
    """toa_down of the oracle on 16 x 12 columns (192: no whole-pass k exists for a power-of-two count): with the
    Sobol launch the per-column counts of 2^16 photons vary far less than Poisson (variance = mean = 341), for a
    range that starts anywhere; with uniform positions they are Poisson."""
    sc = scenes.tiny()
    n = 1 << 16
    for begin in (0, 12345):
        t, _ = O.photon_oracle(sc, 5, begin, n, launch_log2k=-2)
        td = t["toa_down"].astype(np.float64)
        assert int(td.sum()) == n and O.balance(t) == 0
        assert td.var() < 0.05 * td.mean(), (begin, td.var(), td.mean())
    t, _ = O.photon_oracle(sc, 5, 0, n)
    td = t["toa_down"].astype(np.float64)
    assert 0.7 * td.mean() < td.var() < 1.4 * td.mean()


def test_sobol_launch_changes_only_where_photons_enter():
    """Same seed, same IDs: the Sobol launch moves the entry points, the global fractions stay within noise."""
    sc = scenes.tiny()
    n = 200_000
    a, _ = O.photon_oracle(sc, 8, 0, n)
    b, _ = O.photon_oracle(sc, 8, 0, n, launch_log2k=-2)
    assert O.balance(b) == 0 and int(b["toa_down"].sum()) == n
    for name in O.TALLY_NAMES[1:]:
        pa, pb = int(a[name].sum()) / n, int(b[name].sum()) / n
        assert abs(pa - pb) < 5 * np.sqrt(max(pa * (1 - pa), 1e-9) * 2 / n), name
