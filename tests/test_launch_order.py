"""Host logic: the automatic launch order (raytracer_amd.choose_launch_order), the rule bench.py uses and
drivers/driver_common.h restates for --launch=auto."""
import raytracer_amd as rt


def test_grids_beyond_the_caches_take_tiles_and_up_to_256_ids_per_column():
    # BASELINE config 4: one pass of 2^20 columns x 2^8 IDs per 2^28-photon step and GPU, at any GPU count
    assert rt.choose_launch_order(1024, 1024, 128, 1 << 28) == (8, 5)
    assert rt.choose_launch_order(1024, 1024, 128, 1 << 31, ngpus=8) == (8, 5)
    assert rt.choose_launch_order(1024, 1024, 128, 1 << 26) == (6, 5)          # fewer photons: smaller k, still tiles
    # sides that are not multiples of 32 keep rows and k <= 5 (the io driver's default grid)
    assert rt.choose_launch_order(240, 240, 284, 57600 << 5) == (5, 0)
    # exactly 256 MiB of packed cells is not "beyond"
    assert rt.choose_launch_order(1024, 1024, 16, 1 << 28) == (5, 0)
    assert rt.choose_launch_order(1024, 1024, 20, 1 << 28) == (8, 5)


def test_small_grids_keep_rows():
    assert rt.choose_launch_order(256, 256, 128, 1 << 28) == (5, 0)
    assert rt.choose_launch_order(128, 128, 128, 1 << 18) == (4, 0)            # 2^18 = 2^14 columns x 2^4
    assert rt.choose_launch_order(128, 128, 128, 1 << 20, ngpus=2) == (5, 0)


def test_no_whole_pass_means_the_sobol_launch():
    """when no whole-pass k exists the automatic order is the Sobol sequence (even for any count
    on any grid), not Poisson positions -- in particular for every power-of-two photon count on the reference's own
    240 x 240-column io grid (raytracer_3d_io.cu:15-17), where 57600 * 2^k never divides 2^n."""
    S = rt.tracer.LAUNCH_SOBOL
    assert S == -2
    assert rt.choose_launch_order(16, 12, 10, 1 << 16) == (S, 0)              # 192 columns never divide 2^16
    assert rt.choose_launch_order(1024, 1024, 128, (1 << 28) + 1) == (S, 0)
    assert rt.choose_launch_order(1024, 1024, 128, 1 << 19) == (S, 0)         # fewer photons than columns
    for e in range(15, 41):
        assert rt.choose_launch_order(240, 240, 284, 1 << e) == (S, 0), e
    for k, tile in (rt.choose_launch_order(16, 12, 10, 192 * 7 << 3),):
        assert (k, tile) == (3, 0)
