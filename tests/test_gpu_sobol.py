"""the Sobol launch keyed on the photon ID (option launch_log2k = -2 / launch_sobol = 1),
the reference's scrambled-Sobol launch (raytracer_kernels.cu:239-253, 331) as a pure function of the ID. HIP against
the oracle bit for bit in both kernels; the (0,2)-sequence property of the device's own arithmetic; toa_down on the
reference's 240 x 240-column io grid; BASELINE config 3 with it against the reference-side fixture."""
import numpy as np
import pytest

import oracle_lib as O
from test_gpu_parity import SCENES, TWO_D, _assert_same, _check_against_fixture
from test_oracle_photon import _elementary_counts_ok

pytestmark = pytest.mark.gpu

SOBOL = -2


@pytest.fixture(scope="module")
def rt():
    import raytracer_amd as rt
    rt.load_library()
    return rt


@pytest.mark.parametrize("pool", [0, 128])
@pytest.mark.parametrize("scene_name,begin,n", [
    ("tiny", 0, 1 << 16), ("tiny", 123_456_789_012, 50_001), ("tiny", (1 << 32) - 30_000, 70_000),
    ("tiny", (5 << 32) - 1, 2), ("block_cloud", 1 << 20, 1 << 18), ("cumulus_small", 777, 1 << 17),
    ("rayleigh", (3 << 32) - 100_000, 200_000), ("cloud_to_ground", 99, 40_000)])
def test_sobol_launch_bit_identical_to_oracle(rt, pool, scene_name, begin, n):
    """Both kernels against the oracle (which walks the textbook recurrence; the device uses a bit reversal and five
    shift-XOR steps): ranges that start anywhere, cross a multiple of 2^32 (the launch is cut there: another digital
    shift) and lie beyond it."""
    scene = SCENES[scene_name](rt.scenes)
    tr = rt.PhotonTracer.from_scene(scene, stats=1, pool=pool, launch_log2k=SOBOL)
    tr.trace(4242, begin, n); tr.synchronize()
    want, ev = O.photon_oracle(scene, 4242, begin, n, launch_log2k=SOBOL)
    _assert_same(tr.tallies(), want)
    assert tr.events() == ev and int(want["toa_down"].sum()) == n
    tr.close()


@pytest.mark.parametrize("n,batch_log2,tile_log2", [(300_001, 16, 2), (1 << 19, 30, 5), ((1 << 19) + 77, 18, 0)])
def test_sobol_launch_sorted_by_tile_is_still_the_oracle(rt, n, batch_log2, tile_log2):
    """The launch sort orders any ID -> position function by launch tile: with Sobol positions (no Philox block in
    the sort's passes) every ID is still traced exactly once, across sort batches and across a multiple of 2^32; and
    the option by its name."""
    scene = rt.scenes.cumulus(300, 70, 8, 100.0, 80.0, 60.0, n_clouds=6, seed=3, z_base=120.0, z_top=360.0)
    begin = (9 << 32) - n // 3
    tr = rt.PhotonTracer.from_scene(scene, stats=1, launch_sobol=1, launch_sort=1, sort_batch_log2=batch_log2,
                                    sort_tile_log2=tile_log2)
    tr.trace(7, begin, n); tr.synchronize()
    want, ev = O.photon_oracle(scene, 7, begin, n, launch_log2k=SOBOL)
    _assert_same(tr.tallies(), want)
    assert tr.events() == ev
    tr.set_option("launch_sobol", 0)        # back to uniform positions
    tr.reset(); tr.trace(7, begin, 50_000); tr.synchronize()
    want, _ = O.photon_oracle(scene, 7, begin, 50_000)
    _assert_same(tr.tallies(), want)
    tr.close()


@pytest.mark.parametrize("grid,begin,n", [((300, 70), (9 << 32) - 200_000, 500_000), ((64, 64), 4096 * 5, 4096 * 37),
                                          ((33, 5), 7, 100_000), ((1, 1), 0, 5000), ((256, 2), 1 << 20, (1 << 18) + 3)])
def test_sobol_launch_in_box_order_takes_every_id_once(rt, grid, begin, n):
    """Whole nets of a Sobol launch (2^m consecutive IDs from a multiple of 2^m, 2^m boxes ~ columns) are taken box by
    box through the inverse of the net (option sobol_boxes, default on) instead of through a sorted list: scheduling
    only, so the tallies and event counts are the oracle's -- ragged grids whose boxes are not their columns, ranges with
    a head and a tail outside the whole nets and across a multiple of 2^32 -- and equal those of the sorted path."""
    scene = rt.scenes.cumulus(grid[0], grid[1], 8, 100.0, 80.0, 60.0, n_clouds=4, seed=3, z_base=120.0, z_top=360.0)
    want, ev = O.photon_oracle(scene, 21, begin, n, launch_log2k=SOBOL)
    for boxes in (1, 0):
        tr = rt.PhotonTracer.from_scene(scene, stats=1, launch_log2k=SOBOL, sobol_boxes=boxes)
        tr.trace(21, begin, n); tr.synchronize()
        _assert_same(tr.tallies(), want)
        assert tr.events() == ev
        if boxes:       # the box order needs no list, so the entry point that allocates nothing takes it too
            tr.reset(); tr.prepare()
            tr.trace_noalloc(21, begin, n); tr.synchronize()
            _assert_same(tr.tallies(), want)
        tr.close()


def test_device_sobol_words_equal_the_oracle_and_fill_every_elementary_interval(rt):
    """rt_selftest_launch: the device's words for IDs either side of 2^32 equal the oracle's; and every aligned block
    of 2^m consecutive IDs puts exactly one point into each elementary interval 2^-a x 2^-(m-a), a = 0..m, for
    m = 4 .. 20 -- on the words and, for the 2^-24 lattice the positions live on, on the positions themselves."""
    scene = rt.scenes.tiny()
    tr = rt.PhotonTracer.from_scene(scene, launch_log2k=SOBOL)
    xs, ys = scene.itot * scene.dx, scene.jtot * scene.dy
    for first in (0, 1_000_000, (1 << 32) - 4096, (77 << 32) + 12_345):
        xy, w = tr.launch_positions(31, first, 4096)
        assert np.array_equal(w, O.sobol_words(31, np.arange(first, first + 4096)))
        assert np.array_equal(xy[:, 0], (np.float32(xs) * ((w[:, 0] >> 8).astype(np.float32) * np.float32(2.0 ** -24))))
        assert np.array_equal(xy[:, 1], (np.float32(ys) * ((w[:, 1] >> 8).astype(np.float32) * np.float32(2.0 ** -24))))
    rng = np.random.default_rng(5)
    for m in (4, 9, 13, 16, 20):
        for block in [0, 1] + [int(b) for b in rng.integers(0, 1 << (40 - m), size=3)]:
            _, w = tr.launch_positions(31, block << m, 1 << m)
            assert _elementary_counts_ok(w, m), (m, block)
            _, w = tr.launch_positions(31, (block << m) + (1 << (m - 1)), 1 << m)       # not aligned: no net
            assert not _elementary_counts_ok(w, m), (m, block)
    # uniform positions are no net at all
    tr.set_option("launch_log2k", -1)
    _, w = tr.launch_positions(31, 0, 1 << 10)
    assert not _elementary_counts_ok(w, 10)
    with pytest.raises(rt.RtError, match="2\\^32"):
        tr.launch_positions(31, (1 << 32) - 5, 10)
    tr.close()


@pytest.mark.parametrize("n", [1 << 24, 10_000_000])
def test_sobol_toa_down_on_the_io_grid_is_far_more_even_than_poisson(rt, n):
    """240 x 240 columns (the reference's io grid, raytracer_3d_io.cu:15-17; 57600 is no power of
    two, so no stratified pass exists for 2^n photons) with 2^24 photons, and with a count that is no power of two:
    variance of toa_down per column <= 0.1 x Poisson (= the mean); uniform positions give Poisson."""
    scene = rt.scenes.rayleigh_column(240, 240, 4, 100.0, 100.0, 30.0)
    tr = rt.PhotonTracer.from_scene(scene, launch_log2k=SOBOL)
    tr.trace(3, 0, n); tr.synchronize()
    t = tr.tallies()
    td = t["toa_down"].astype(np.float64)
    assert int(td.sum()) == n and rt.photon_balance(t) == 0
    assert td.var() <= 0.1 * td.mean(), (td.var(), td.mean())
    tr.set_option("launch_log2k", -1)
    tr.reset(); tr.trace(3, 0, n); tr.synchronize()
    tu = tr.tallies()["toa_down"].astype(np.float64)
    assert 0.9 * tu.mean() < tu.var() < 1.1 * tu.mean()
    tr.close()


def test_c3_cumulus_2p28_photons_sobol_launch_against_reference_restatement(rt):
    """BASELINE config 3 at its full 2^28 photons with the Sobol launch against 1.2*10^8 photons of oracle/cpu_ref.cpp
    (fixture cpu_ref_c3_1e8.npz): exact balance, global fractions within 4 sigma, z-scores of the column blocks of the
    four outgoing 2-D tallies (toa_down is even here and Poisson in the fixture: its evenness is asserted instead; the
    direct beam at the surface inherits that evenness and is gated from above only) and of the per-level absorption."""
    scene = rt.scenes.cumulus_c3()
    n = 1 << 28
    tr = rt.PhotonTracer.from_scene(scene, launch_log2k=SOBOL)
    tr.trace(17, 0, n); tr.synchronize()
    a = tr.tallies()
    tr.close()
    assert rt.photon_balance(a) == 0 and int(a["toa_down"].sum()) == n
    per_col = n // scene.ncol
    assert int(np.abs(a["toa_down"].astype(np.int64) - per_col).max()) <= 4       # 2^16 columns: a (0,2)-net per 2^16 IDs
    _, rep = _check_against_fixture(a, n, "cpu_ref_c3_1e8.npz", scene, {"blocks": TWO_D[1:], "sub_poisson": ("surface_down_direct",)})
    assert rep["z2_surface_down_direct"] < 0.95      # the direct beam at the surface is visibly less noisy than Poisson


def test_c4_les_grid_2p30_photons_sobol_launch_against_reference_restatement(rt):
    """north_star's target with the reference's own launch: 2^30 photons through the 1024x1024x128 LES grid as four
    accumulating calls of 2^28 (the shards of four ranks), Sobol launch in box order. Exact balance; toa_down EXACTLY 1024 per
    column (2^20 columns: every aligned 2^20 IDs are a net with one point per column; positions live on the 2^-24 lattice,
    on which a column is a whole number of steps); against 6*10^7 photons of oracle/cpu_ref.cpp (fixture
    cpu_ref_c4_6e7.npz): global fractions within 4 sigma, z-scores of the 16x16-column-block sums and per-level absorption."""
    scene = rt.scenes.cumulus_c4()
    n = 1 << 30
    tr = rt.PhotonTracer.from_scene(scene, launch_log2k=SOBOL)
    for r in range(4):
        tr.trace(23, r * (n // 4), n // 4)
    tr.synchronize()
    a = tr.tallies()
    tr.close()
    assert rt.photon_balance(a) == 0 and int(a["toa_down"].sum()) == n
    assert int(np.abs(a["toa_down"].astype(np.int64) - n // scene.ncol).max()) <= 8
    _check_against_fixture(a, n, "cpu_ref_c4_6e7.npz", scene, {"blocks": TWO_D[1:], "sub_poisson": ("surface_down_direct",)})
