"""Host side of the MI355X photon tracer: a thin ctypes layer over librt_photon.so.

Mirrors the host flow of the reference drivers (``run_ray_tracer`` in
``/root/reference/src/raytracer_3d.cu:3-204`` and ``raytracer_3d_io.cu:3-213``): stage the grids,
derive cloud bounds and majorants, launch the tracer, fetch the seven tally arrays, write them as
``<name>.bin``. All compute happens in the HIP library; if the library is missing this module
raises — there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_PHOTON_LIB: an experimental build of the library (tools/cmp_libs.sh, kernel experiments); default: the in-tree one
LIB_PATH = os.environ.get("RT_PHOTON_LIB") or os.path.join(_HERE, "librt_photon.so")

TALLY_NAMES = ("toa_down", "toa_up", "surface_down_direct", "surface_down_diffuse", "surface_up",
               "atmos_direct", "atmos_diffuse")        # file names of raytracer_3d.cu:197-203
EVENT_NAMES = ("steps", "collision_tests", "real", "scatter", "cloud_scatter", "surface", "toa",
               "absorb", "transition", "abort")


LAUNCH_UNIFORM, LAUNCH_SOBOL = -1, -2      # values of option "launch_log2k" below zero (include/rt_photon.h)


class RtError(RuntimeError):
    pass


class SlabInfo(C.Structure):
    _fields_ = [("nslab", C.c_int), ("zface", C.c_float * 4), ("kmaj", C.c_float * 3),
                ("k_lo", C.c_int), ("k_hi", C.c_int), ("cloud_min", C.c_float), ("cloud_max", C.c_float),
                ("sun", C.c_float * 3), ("gslab", C.c_int), ("nci", C.c_int), ("ncj", C.c_int), ("nck", C.c_int),
                ("hx", C.c_float), ("hy", C.c_float), ("hz", C.c_float)]


_lib = None


def load_library():
    """Load librt_photon.so (built in-tree by __graft_entry__.build()). Fails loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # PyTorch-ROCm brings its own HIP runtime and this library uses the system's. Both work in one process when
    # torch's is initialised FIRST; the other way round torch finds "No HIP GPUs". So if torch is already imported,
    # let it initialise before this library makes its first HIP call (a process that imports torch only later must
    # do `import torch; torch.cuda.init()` before it creates a PhotonTracer: INTEGRATION.md).
    torch = sys.modules.get("torch")
    if torch is not None:
        try:
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass
    lib = C.CDLL(LIB_PATH)
    vp, u64, f32p, u64p = C.c_void_p, C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_uint64)
    sig = {
        "rt_last_error": (C.c_char_p, []),
        "rt_device_count": (C.c_int, [C.POINTER(C.c_int)]),
        "rt_create": (C.c_int, [C.c_int] * 4 + [C.c_float] * 3 + [C.POINTER(vp)]),
        "rt_destroy": (C.c_int, [vp]),
        "rt_upload_optics": (C.c_int, [vp] + [f32p] * 4),
        "rt_upload_optics_device": (C.c_int, [vp] * 5),
        "rt_upload_optics_multi": (C.c_int, [C.POINTER(vp), C.c_int] + [f32p] * 4),
        "rt_set_scene": (C.c_int, [vp, C.c_float, C.c_float, C.c_float]),
        "rt_set_option": (C.c_int, [vp, C.c_char_p, C.c_longlong]),
        "rt_set_stream": (C.c_int, [vp, vp]),
        "rt_trace": (C.c_int, [vp, u64, u64, u64]),
        "rt_trace_noalloc": (C.c_int, [vp, u64, u64, u64]),
        "rt_prepare": (C.c_int, [vp]),
        "rt_synchronize": (C.c_int, [vp]),
        "rt_download_tallies": (C.c_int, [vp, C.POINTER(u64p)]),
        "rt_upload_tallies": (C.c_int, [vp, C.POINTER(u64p)]),
        "rt_reset_tallies": (C.c_int, [vp]),
        "rt_download_tallies_async": (C.c_int, [vp, C.POINTER(u64p), C.POINTER(C.c_int)]),
        "rt_download_wait": (C.c_int, [vp, C.c_int]),
        "rt_swap_tally_buffer": (C.c_int, [vp]),
        "rt_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(vp)]),
        "rt_host_free": (C.c_int, [vp]),
        "rt_tally_device_buffer": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]),
        "rt_bind_tally_buffer": (C.c_int, [vp, vp, C.c_size_t]),
        "rt_allreduce": (C.c_int, [C.POINTER(vp), C.c_int]),
        "rt_bind_atmos_wire": (C.c_int, [vp, vp, C.c_size_t]),
        "rt_tallies_narrow": (C.c_int, [vp, vp, vp, C.c_int]),
        "rt_tallies_widen": (C.c_int, [vp, vp, vp]),
        "rt_get_slabs": (C.c_int, [vp, C.POINTER(SlabInfo)]),
        "rt_get_majorants": (C.c_int, [vp, f32p, C.c_size_t]),
        "rt_get_region_boxes": (C.c_int, [vp, C.POINTER(C.c_uint32), C.c_size_t]),
        "rt_get_events": (C.c_int, [vp, u64p]),
        "rt_get_sched_stats": (C.c_int, [vp, u64p]),
        "rt_last_trace_ms": (C.c_int, [vp, C.POINTER(C.c_float)]),
        "rt_selftest_philox": (C.c_int, [C.c_int, C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
        "rt_selftest_math": (C.c_int, [C.c_int, C.c_int, f32p, C.c_size_t, f32p]),
        "rt_selftest_exact": (C.c_int, [C.c_int, C.c_int, C.c_uint32, C.c_uint64, u64p]),
        "rt_selftest_launch": (C.c_int, [vp, u64, u64, C.c_size_t, f32p, C.POINTER(C.c_uint32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise RtError(f"[rt_status {rc}] " + load_library().rt_last_error().decode(errors="replace"))


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class PhotonTracer:
    """One tracer context on one GPU (rt_context). Not thread-safe; one per GPU."""

    def __init__(self, itot, jtot, ktot, dx, dy, dz, device=0):
        self.lib = load_library()
        self.itot, self.jtot, self.ktot = int(itot), int(jtot), int(ktot)
        self.dx, self.dy, self.dz = float(dx), float(dy), float(dz)
        self.ncol = self.itot * self.jtot
        self.ncell = self.ncol * self.ktot
        self.nwords = 5 * self.ncol + 2 * self.ncell
        self.device = device
        h = C.c_void_p()
        _check(self.lib.rt_create(device, self.itot, self.jtot, self.ktot, self.dx, self.dy, self.dz, C.byref(h)))
        self._h = h
        self._bound = None

    @classmethod
    def from_scene(cls, scene, device=0, **options):
        t = cls(scene.itot, scene.jtot, scene.ktot, scene.dx, scene.dy, scene.dz, device=device)
        t.upload_optics(*scene.fields())
        t.set_scene(scene.zenith_deg, scene.azimuth_deg, scene.albedo)
        for k, v in options.items():
            t.set_option(k, v)
        return t

    # -- staging ---------------------------------------------------------------------------
    def upload_optics(self, k_gas, k_cloud, ssa, asy):
        arrs = []
        for name, a in (("k_ext_gas", k_gas), ("k_ext_cloud", k_cloud), ("ssa", ssa), ("asy", asy)):
            a = np.ascontiguousarray(a, dtype=np.float32)
            if a.size != self.ncell:
                raise RtError(f"{name}: {a.size} values, the grid has {self.ncell}")
            arrs.append(a)
        _check(self.lib.rt_upload_optics(self._h, *[_f32p(a) for a in arrs]))

    def upload_optics_device(self, d_gas, d_cloud, d_ssa, d_asy):
        """Device pointers (ints) of four float32 arrays already on this GPU."""
        _check(self.lib.rt_upload_optics_device(self._h, *[C.c_void_p(int(p)) for p in (d_gas, d_cloud, d_ssa, d_asy)]))

    def set_scene(self, zenith_deg=50.0, azimuth_deg=20.0, albedo=0.2):
        f = np.float32
        # angles become floats before the cosines are taken, raytracer_3d.cu:25-26
        _check(self.lib.rt_set_scene(self._h, f(zenith_deg * (math.pi / 180.0)), f(azimuth_deg * (math.pi / 180.0)), f(albedo)))

    def set_option(self, name, value):
        _check(self.lib.rt_set_option(self._h, name.encode(), int(value)))

    def set_stream(self, hip_stream_ptr):
        _check(self.lib.rt_set_stream(self._h, C.c_void_p(int(hip_stream_ptr) if hip_stream_ptr else None)))

    def bind_tally_tensor(self, tensor):
        """Accumulate into a caller-owned int64 device tensor of nwords elements (for collectives)."""
        if tensor.numel() != self.nwords or tensor.element_size() != 8 or not tensor.is_contiguous():
            raise RtError(f"tally tensor must be contiguous, 8-byte, {self.nwords} elements")
        _check(self.lib.rt_bind_tally_buffer(self._h, C.c_void_p(tensor.data_ptr()), self.nwords))
        self._bound = tensor

    # -- the hot path -----------------------------------------------------------------------
    def trace(self, seed, photon_id_begin, photon_count):
        _check(self.lib.rt_trace(self._h, int(seed), int(photon_id_begin), int(photon_count)))

    def trace_noalloc(self, seed, photon_id_begin, photon_count):
        """rt_trace for callers that cannot allocate (needs prepare() first; scratch in a static device array)."""
        _check(self.lib.rt_trace_noalloc(self._h, int(seed), int(photon_id_begin), int(photon_count)))

    def narrow_tallies(self, wire_tensor, world_size, tally_tensor=None):
        """Low words of the tallies (of `tally_tensor`, default: the bound buffer) into an int32 device tensor of
        nwords + 1 elements; the last one becomes 1 if a count is too large for a 32-bit sum over world_size ranks."""
        if wire_tensor.numel() != self.nwords + 1 or wire_tensor.element_size() != 4 or not wire_tensor.is_contiguous():
            raise RtError(f"wire tensor must be contiguous, 4-byte, {self.nwords + 1} elements")
        src = C.c_void_p(tally_tensor.data_ptr()) if tally_tensor is not None else None
        _check(self.lib.rt_tallies_narrow(self._h, src, C.c_void_p(wire_tensor.data_ptr()), int(world_size)))

    def widen_tallies(self, wire_tensor, tally_tensor=None):
        """The summed 32-bit words back into the 64-bit tallies, unless the flag word says a rank refused."""
        dst = C.c_void_p(tally_tensor.data_ptr()) if tally_tensor is not None else None
        _check(self.lib.rt_tallies_widen(self._h, dst, C.c_void_p(wire_tensor.data_ptr())))

    def bind_atmos_wire(self, wire_tensor):
        """While an int32 device tensor of nwords + 1 elements is bound, absorptions are counted straight into it as
        32-bit words (rt_bind_atmos_wire); None unbinds."""
        if wire_tensor is None:
            _check(self.lib.rt_bind_atmos_wire(self._h, None, 0))
            self._atm_wire = None
            return
        if wire_tensor.numel() != self.nwords + 1 or wire_tensor.element_size() != 4 or not wire_tensor.is_contiguous():
            raise RtError(f"wire tensor must be contiguous, 4-byte, {self.nwords + 1} elements")
        _check(self.lib.rt_bind_atmos_wire(self._h, C.c_void_p(wire_tensor.data_ptr()), self.nwords + 1))
        self._atm_wire = wire_tensor

    def prepare(self):
        """Build the majorant tables now, so that later trace() calls are pure asynchronous launches."""
        _check(self.lib.rt_prepare(self._h))

    def synchronize(self):
        _check(self.lib.rt_synchronize(self._h))

    def reset(self):
        _check(self.lib.rt_reset_tallies(self._h))

    def last_trace_ms(self):
        ms = C.c_float()
        _check(self.lib.rt_last_trace_ms(self._h, C.byref(ms)))
        return ms.value

    # -- results ----------------------------------------------------------------------------
    def _host_tallies(self):
        out = {}
        for n in TALLY_NAMES:
            shape = (self.ktot, self.jtot, self.itot) if n.startswith("atmos") else (self.jtot, self.itot)
            out[n] = np.zeros(shape, dtype=np.uint64)
        return out

    def tallies(self):
        out = self._host_tallies()
        ptrs = (C.POINTER(C.c_uint64) * 7)(*[out[n].ctypes.data_as(C.POINTER(C.c_uint64)) for n in TALLY_NAMES])
        _check(self.lib.rt_download_tallies(self._h, ptrs))
        return out

    def pinned_tallies(self):
        """The seven arrays in ONE block of pinned host memory (rt_host_alloc): what tallies_async() copies into while the
        GPU goes on. Returns (dict of numpy views, handle to pass to free_pinned)."""
        p = C.c_void_p()
        _check(self.lib.rt_host_alloc(self.nwords * 8, C.byref(p)))
        block = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(self.nwords,))
        return split_tally_words(block, self.itot, self.jtot, self.ktot), p

    def free_pinned(self, handle):
        _check(self.lib.rt_host_free(handle))

    def tallies_async(self, out):
        """Start the copy of the CURRENT tally buffer into `out` (dict of the seven arrays, ideally pinned_tallies())
        behind everything enqueued so far, on the library's copy stream; returns the ticket for tallies_wait()."""
        ptrs = (C.POINTER(C.c_uint64) * 7)(*[out[n].ctypes.data_as(C.POINTER(C.c_uint64)) for n in TALLY_NAMES])
        ticket = C.c_int()
        _check(self.lib.rt_download_tallies_async(self._h, ptrs, C.byref(ticket)))
        return ticket.value

    def tallies_wait(self, ticket):
        _check(self.lib.rt_download_wait(self._h, int(ticket)))

    def swap_tally_buffer(self):
        """Later launches use the context's other tally buffer (in stream order; see rt_swap_tally_buffer)."""
        _check(self.lib.rt_swap_tally_buffer(self._h))

    def add_tallies(self, tallies):
        arrs = [np.ascontiguousarray(tallies[n], dtype=np.uint64) for n in TALLY_NAMES]
        ptrs = (C.POINTER(C.c_uint64) * 7)(*[a.ctypes.data_as(C.POINTER(C.c_uint64)) for a in arrs])
        _check(self.lib.rt_upload_tallies(self._h, ptrs))

    def events(self):
        ev = np.zeros(len(EVENT_NAMES), dtype=np.uint64)
        _check(self.lib.rt_get_events(self._h, ev.ctypes.data_as(C.POINTER(C.c_uint64))))
        return dict(zip(EVENT_NAMES, (int(v) for v in ev)))

    def sched_stats(self):
        """Rounds and served lanes per stage of the wave scheduler (stats launches only)."""
        v = np.zeros(8, dtype=np.uint64)
        _check(self.lib.rt_get_sched_stats(self._h, v.ctypes.data_as(C.POINTER(C.c_uint64))))
        r = [int(x) for x in v]
        return {"rounds": dict(T=r[0], E=r[1], S=r[2], slab=r[6]), "lanes": dict(T=r[3], E=r[4], S=r[5], slab=r[7])}

    def slabs(self):
        s = SlabInfo()
        _check(self.lib.rt_get_slabs(self._h, C.byref(s)))
        return s

    def majorants(self):
        """The majorant grid of the cloud slab as an array [nck, ncj, nci, 2] of {M, 1/M} (or None)."""
        s = self.slabs()
        n = s.nci * s.ncj * s.nck
        if n == 0:
            return None
        out = np.zeros((s.nck, s.ncj, s.nci, 2), dtype=np.float32)
        _check(self.lib.rt_get_majorants(self._h, _f32p(out), n))
        return out

    def region_boxes(self):
        """The majorant box of every supervoxel, uint32 [nck, ncj, nci] (or None): six extents of 5 bits,
        xm | xp << 5 | ym << 10 | yp << 15 | zm << 20 | zp << 25 -- a photon in supervoxel (ci, cj, ck) crosses the
        supervoxels [ci - xm, ci + xp] x [cj - ym, cj + yp] x [ck - zm, ck + zp] as one region; majorants() holds
        the box's pair."""
        s = self.slabs()
        n = s.nci * s.ncj * s.nck
        if n == 0:
            return None
        out = np.zeros((s.nck, s.ncj, s.nci), dtype=np.uint32)
        _check(self.lib.rt_get_region_boxes(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32)), n))
        return out

    def launch_positions(self, seed, photon_id_begin, n):
        """Where photons [begin, begin + n) enter the domain top under the current launch option (uniform or Sobol):
        (xy float32 [n, 2] in metres, words uint32 [n, 2] they were scaled from). Test hook."""
        xy = np.zeros((int(n), 2), dtype=np.float32)
        words = np.zeros((int(n), 2), dtype=np.uint32)
        _check(self.lib.rt_selftest_launch(self._h, int(seed), int(photon_id_begin), int(n), _f32p(xy),
                                           words.ctypes.data_as(C.POINTER(C.c_uint32))))
        return xy, words

    def set_supergrid(self, sx, sy, sz):
        """Supervoxel size in cells of the majorant grid over the cloud slab; (0,0,0) disables it."""
        for axis, v in zip("xyz", (sx, sy, sz)):
            self.set_option("supergrid_" + axis, v)

    def tally_device_buffer(self):
        p, n = C.c_void_p(), C.c_size_t()
        _check(self.lib.rt_tally_device_buffer(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def close(self):
        if getattr(self, "_h", None):
            self.lib.rt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def allreduce(tracers):
    """rt_allreduce over several PhotonTracer objects of ONE process (one per GPU): every tracer ends up holding
    the summed tallies (RCCL inside the library)."""
    lib = load_library()
    arr = (C.c_void_p * len(tracers))(*[t._h for t in tracers])
    _check(lib.rt_allreduce(arr, len(tracers)))


def choose_launch_order(itot, jtot, ktot, photons_per_call, ngpus=1):
    """The launch order the drivers' --launch=auto and bench.py use: (launch_log2k, launch_tile_log2).

    Stratified by column with the largest k that gives every GPU whole passes over the columns (ncol * 2^k photon
    IDs each). On a grid beyond the caches (16 bytes of packed cell per cell > 256 MiB) whose sides are multiples
    of 32 the columns are counted in tiles of 32 x 32 and k may go up to 8: a tile's 2^(10+k) photons keep its
    cells in one XCD's L2 (1024 x 1024 x 128: a third of the memory traffic, +6 % photons/s). Smaller grids keep
    rows and k <= 5 (more photons per column at a time contend for the same tally words there). No k with whole
    passes (any photon count on a grid whose column count is not a power of two, e.g. the 240 x 240 columns of
    raytracer_3d_io.cu:15-17): (LAUNCH_SOBOL, 0) -- the Sobol sequence indexed by the photon ID, even for every
    count, which the library sorts by launch tile on the device."""
    ncol = itot * jtot
    beyond_caches = 16 * ncol * ktot > (256 << 20)
    tile = 5 if beyond_caches and itot % 32 == 0 and jtot % 32 == 0 else 0
    for k in range(8 if tile else 5, -1, -1):
        if photons_per_call % (ngpus * (ncol << k)) == 0:
            return k, tile
    return LAUNCH_SOBOL, 0


def split_tally_words(words, itot, jtot, ktot):
    """Concatenated tally buffer (host array of nwords) -> dict of the seven arrays."""
    ncol, ncell = itot * jtot, itot * jtot * ktot
    out, off = {}, 0
    for n in TALLY_NAMES:
        if n.startswith("atmos"):
            out[n] = np.asarray(words[off:off + ncell]).view(np.uint64).reshape(ktot, jtot, itot)
            off += ncell
        else:
            out[n] = np.asarray(words[off:off + ncol]).view(np.uint64).reshape(jtot, itot)
            off += ncol
    return out


def save_tallies(tallies, directory="."):
    """Write the seven ``<name>.bin`` files of raytracer_3d.cu:184-203 (uint64, truncating). The two 3D absorption
    fields are all but 2 % of the bytes: one of them goes out on a thread of its own."""
    import threading
    os.makedirs(directory, exist_ok=True)
    errors = []

    def write(n):
        path = os.path.join(directory, n + ".bin")
        try:
            np.ascontiguousarray(tallies[n], dtype=np.uint64).tofile(path)
        except OSError as exc:
            errors.append((n, exc))

    other = threading.Thread(target=write, args=(TALLY_NAMES[-1],))
    other.start()
    for n in TALLY_NAMES[:-1]:
        write(n)
    other.join()
    if errors:
        raise RuntimeError(f'Cannot write file "{errors[0][0]}.bin"') from errors[0][1]


def photon_balance(t) -> int:
    """balance_in - balance_out of plot_results_3d.py:41-44; exactly 0 for a complete run."""
    inn = int(t["toa_down"].sum())
    out = int(t["surface_down_direct"].sum()) + int(t["surface_down_diffuse"].sum()) - int(t["surface_up"].sum()) \
        + int(t["toa_up"].sum()) + int(t["atmos_direct"].sum()) + int(t["atmos_diffuse"].sum())
    return inn - out


def algorithmic_bytes(events, n_photons) -> int:
    """4 B per collision test (k_tot) + 4 B per real collision (ssa) + 8 B per
    scattering ({asy, k_cloud}) + 16 B (8-byte read-modify-write) per tally update: one launch
    tally per photon plus one per surface arrival, surface reflection (= surface arrivals that
    did not end the photon), TOA escape and absorption."""
    e = events
    reflections = e["surface"] - (n_photons - e["toa"] - e["absorb"])
    tally_updates = n_photons + e["surface"] + reflections + e["toa"] + e["absorb"]
    return 4 * e["collision_tests"] + 4 * e["real"] + 8 * e["scatter"] + 16 * tally_updates
