"""rt_trace_noalloc — the one entry point that allocates nothing — against rt_trace, through the C ABI.

It takes the device words a pool launch needs beyond its inputs and outputs (photon-ID cursors, cold block) from
a static device array of the library instead of the context's allocation. The array is shared by every call on a
GPU and keeps the last call's contents, so the test makes two calls in a row, the second smaller, and asks for
tallies bit-identical to rt_trace's both times. Named so that pytest collects it last."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TALLIES = 7


def _ctx(lib, scene, launch_log2k):
    h = C.c_void_p()
    assert lib.rt_create(0, scene.itot, scene.jtot, scene.ktot, scene.dx, scene.dy, scene.dz, C.byref(h)) == 0, lib.rt_last_error()
    f32p = C.POINTER(C.c_float)
    fields = [np.ascontiguousarray(a, dtype=np.float32) for a in scene.fields()]
    assert lib.rt_upload_optics(h, *[a.ctypes.data_as(f32p) for a in fields]) == 0, lib.rt_last_error()
    rad = np.float32(np.pi / 180.0)
    assert lib.rt_set_scene(h, np.float32(scene.zenith_deg) * rad, np.float32(scene.azimuth_deg) * rad, scene.albedo) == 0
    assert lib.rt_set_option(h, b"launch_log2k", launch_log2k) == 0
    return h


def _download(lib, h, scene):
    ncol, ncell = scene.ncol, scene.ncell
    out = [np.zeros(ncol if t < 5 else ncell, dtype=np.uint64) for t in range(TALLIES)]
    u64p = C.POINTER(C.c_uint64)
    ptrs = (u64p * TALLIES)(*[a.ctypes.data_as(u64p) for a in out])
    assert lib.rt_download_tallies(h, ptrs) == 0, lib.rt_last_error()
    return out


@pytest.mark.parametrize("launch_log2k", [-1, 3])
def test_noalloc_entry_point_matches_rt_trace_twice_in_a_row(launch_log2k):
    import raytracer_amd as rt
    lib = rt.load_library()
    scene = rt.scenes.cumulus(96, 80, 48, 50.0, 50.0, 25.0, n_clouds=12, seed=3, z_base=300.0, z_top=900.0)
    a, b = _ctx(lib, scene, launch_log2k), _ctx(lib, scene, launch_log2k)
    try:
        # refused, not allocated for: the majorant tables do not exist yet
        assert lib.rt_trace_noalloc(b, 1, 0, 10) == 3 and b"rt_prepare" in lib.rt_last_error()
        assert lib.rt_prepare(b) == 0, lib.rt_last_error()
        # first a large call, then a smaller one on the same (now used) scratch; then one after an rt_trace on the
        # same context has left its launch-sort list behind (uniform launch: the no-alloc call sorts into it)
        for seed, begin, count in ((21, 5_000_000_000, 1 << 19), (22, 123, 40_001), (23, 77, (1 << 19) + 3)):
            if seed == 23:
                assert lib.rt_trace(b, 5, 0, 1 << 20) == 0 and lib.rt_synchronize(b) == 0, lib.rt_last_error()
            assert lib.rt_reset_tallies(a) == 0 and lib.rt_reset_tallies(b) == 0
            assert lib.rt_trace(a, seed, begin, count) == 0, lib.rt_last_error()
            assert lib.rt_trace_noalloc(b, seed, begin, count) == 0, lib.rt_last_error()
            assert lib.rt_synchronize(a) == 0 and lib.rt_synchronize(b) == 0, lib.rt_last_error()
            want, got = _download(lib, a, scene), _download(lib, b, scene)
            assert int(want[0].sum()) == count
            for t in range(TALLIES):
                assert np.array_equal(got[t], want[t]), f"tally {t} differs after {count} photons"
        # limits are errors with a message, not allocations or truncations
        assert lib.rt_trace_noalloc(b, 1, (1 << 40) - 5, 10) == 1 and b"40 bits" in lib.rt_last_error()
        assert lib.rt_set_option(b, b"pool", 0) == 0
        assert lib.rt_trace_noalloc(b, 1, 0, 10) == 1 and b"pool" in lib.rt_last_error()
        assert lib.rt_set_option(b, b"pool", 128) == 0
        assert lib.rt_set_option(b, b"levels", 2) == 0             # an option change invalidates the tables
        assert lib.rt_trace_noalloc(b, 1, 0, 10) == 3
        assert lib.rt_trace_noalloc(None, 1, 0, 10) == 1
    finally:
        lib.rt_destroy(a)
        lib.rt_destroy(b)
