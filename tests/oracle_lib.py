"""ctypes bindings of oracle/liboracle.so for the tests (and bench.py's cpu_baseline leg).

The oracle is the CHECKER. Nothing under raytracer_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
# RT_ORACLE_LIB: another build of the checker (tools/ubsan.sh: the UBSan build of oracle/Makefile's `ubsan` target)
LIB_PATH = os.environ.get("RT_ORACLE_LIB") or os.path.join(ORACLE_DIR, "liboracle.so")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "raytracer_3d_ref")

TALLY_NAMES = ("toa_down", "toa_up", "surface_down_direct", "surface_down_diffuse", "surface_up",
               "atmos_direct", "atmos_diffuse")
EVENT_NAMES = ("steps", "collision_tests", "real", "scatter", "cloud_scatter", "surface", "toa",
               "absorb", "transition", "abort")


class OracleScene(C.Structure):
    _fields_ = [("itot", C.c_int), ("jtot", C.c_int), ("ktot", C.c_int),
                ("dx", C.c_float), ("dy", C.c_float), ("dz", C.c_float),
                ("zenith_rad", C.c_float), ("azimuth_rad", C.c_float), ("albedo", C.c_float),
                ("fixed_dot", C.c_int), ("sx", C.c_int), ("sy", C.c_int), ("sz", C.c_int),
                ("launch_log2k", C.c_int), ("planar", C.c_int), ("levels", C.c_int), ("merge_tau", C.c_float),
                ("launch_tile_log2", C.c_int)]


class OracleSlabs(C.Structure):
    _fields_ = [("nslab", C.c_int), ("zface", C.c_float * 4), ("kmaj", C.c_float * 3),
                ("inv_kmaj", C.c_float * 3), ("k_lo", C.c_int), ("k_hi", C.c_int),
                ("sun", C.c_float * 3), ("gslab", C.c_int), ("nci", C.c_int), ("ncj", C.c_int), ("nck", C.c_int),
                ("hx", C.c_float), ("hy", C.c_float), ("hz", C.c_float), ("maj", C.POINTER(C.c_float)),
                ("box", C.POINTER(C.c_uint32))]


_lib = None


def build_oracle() -> None:
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build_oracle()
        _lib = C.CDLL(LIB_PATH)
        u64p = C.POINTER(C.c_uint64)
        f32p = C.POINTER(C.c_float)
        f64p = C.POINTER(C.c_double)
        _lib.cpu_ref_trace.restype = C.c_int
        _lib.cpu_ref_trace.argtypes = [C.c_int] * 3 + [C.c_double] * 3 + [f64p] * 4 + [C.c_double] * 3 + \
            [C.c_uint64, C.c_int, C.c_int] + [u64p] * 8
        _lib.photon_oracle_trace.restype = C.c_int
        _lib.photon_oracle_trace.argtypes = [C.POINTER(OracleScene)] + [f32p] * 4 + \
            [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int] + [u64p] * 8
        _lib.oracle_build_slabs.restype = None
        _lib.oracle_build_slabs.argtypes = [C.POINTER(OracleScene)] + [f32p] * 3 + [C.POINTER(OracleSlabs)]
        _lib.oracle_free_slabs.restype = None
        _lib.oracle_free_slabs.argtypes = [C.POINTER(OracleSlabs)]
        _lib.oracle_philox4x32_10.restype = None
        _lib.oracle_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
        _lib.oracle_sobol_words.restype = None
        _lib.oracle_sobol_words.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_uint32)]
        _lib.oracle_logf.restype = C.c_float
        _lib.oracle_logf.argtypes = [C.c_float]
        _lib.oracle_inv_cbrtf.restype = C.c_float
        _lib.oracle_inv_cbrtf.argtypes = [C.c_float]
        _lib.oracle_rayleigh_mu.restype = C.c_float
        _lib.oracle_rayleigh_mu.argtypes = [C.c_float]
        _lib.oracle_henyey_mu.restype = C.c_float
        _lib.oracle_henyey_mu.argtypes = [C.c_float, C.c_float]
        _lib.oracle_sincos2pi.restype = None
        _lib.oracle_sincos2pi.argtypes = [C.c_float, f32p, f32p]
    return _lib


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def new_tallies(scene):
    out = {}
    for n in TALLY_NAMES:
        shape = (scene.ktot, scene.jtot, scene.itot) if n.startswith("atmos") else (scene.jtot, scene.itot)
        out[n] = np.zeros(shape, dtype=np.uint64)
    return out


DEFAULT_SUPERGRID = "auto"        # must equal the product's default (rt_set_option "supergrid_*")


DEFAULT_LEVELS = "auto"      # the library's defaults of options "levels" and "merge_tau_milli"
DEFAULT_MERGE_TAU = 0.75


def resolve_auto(scene, supergrid, levels, launch_log2k, pool=128, launch_sort=-1):
    """The library's automatic choices (wants_fine / build_majorants in rt_photon.hip), restated: fine supervoxels
    with merged regions when the launch is stratified, the packed cells fit the Infinity Cache or the pool kernel
    may sort its launches (its default), else coarse and unmerged."""
    fine = launch_log2k >= 0 or 16 * scene.ncell <= (256 << 20) or (pool > 0 and launch_sort != 0)

    def nearest(target, d):       # whole number of cells nearest to the target length, 1..8
        return min(max(int(target / float(d) + 0.5), 1), 8)
    if isinstance(supergrid, str):
        sx, sy, sz = (nearest(50.0, scene.dx), nearest(50.0, scene.dy), nearest(50.0, scene.dz)) if fine else (8, 8, 4)
        while -(-scene.itot // sx) > 32767:      # (the library coarsens an automatic size until its indices fit)
            sx *= 2
        while -(-scene.jtot // sy) > 32767:
            sy *= 2
        supergrid = (sx, sy, sz)
    if isinstance(levels, str):
        levels = 5 if fine else 0
    return supergrid, levels


def oracle_scene(scene, fixed_dot=False, supergrid=DEFAULT_SUPERGRID, launch_log2k=-1, planar=False,
                 levels=DEFAULT_LEVELS, merge_tau=DEFAULT_MERGE_TAU, launch_tile_log2=0) -> OracleScene:
    f = np.float32
    supergrid, levels = resolve_auto(scene, supergrid, levels, launch_log2k)
    sx, sy, sz = supergrid if supergrid else (0, 0, 0)
    return OracleScene(scene.itot, scene.jtot, scene.ktot, scene.dx, scene.dy, scene.dz,
                       f(scene.zenith_deg * (math.pi / 180.0)), f(scene.azimuth_deg * (math.pi / 180.0)),
                       scene.albedo, int(fixed_dot), sx, sy, sz, int(launch_log2k), int(planar), int(levels), float(merge_tau),
                       int(launch_tile_log2))


def photon_oracle(scene, seed, pid_begin, count, nthreads=0, tallies=None, fixed_dot=False,
                  supergrid=DEFAULT_SUPERGRID, launch_log2k=-1, planar=False, levels=DEFAULT_LEVELS,
                  merge_tau=DEFAULT_MERGE_TAU, launch_tile_log2=0):
    """Run oracle/photon_oracle.c; returns (tallies dict, events dict). Adds into `tallies`."""
    t = tallies if tallies is not None else new_tallies(scene)
    ev = np.zeros(len(EVENT_NAMES), dtype=np.uint64)
    sc = oracle_scene(scene, fixed_dot, supergrid, launch_log2k, planar, levels, merge_tau, launch_tile_log2)
    fields = [np.ascontiguousarray(a, dtype=np.float32) for a in scene.fields()]
    rc = lib().photon_oracle_trace(C.byref(sc), *[_p(a, C.c_float) for a in fields],
                                   seed, pid_begin, count, nthreads,
                                   *[_p(t[n], C.c_uint64) for n in TALLY_NAMES], _p(ev, C.c_uint64))
    if rc != 0:
        raise RuntimeError(f"photon_oracle_trace failed: {rc}")
    return t, dict(zip(EVENT_NAMES, (int(v) for v in ev)))


def sobol_words(seed, pids, shifted=True):
    """The oracle's Sobol launch words of the given photon IDs: uint32 [n, 2]."""
    out = np.zeros((len(pids), 2), dtype=np.uint32)
    w = (C.c_uint32 * 2)()
    fn = lib().oracle_sobol_words
    for r, pid in enumerate(pids):
        fn(int(seed), int(pid), int(shifted), w)
        out[r] = (w[0], w[1])
    return out


def oracle_slabs(scene, supergrid=DEFAULT_SUPERGRID, levels=DEFAULT_LEVELS, merge_tau=DEFAULT_MERGE_TAU) -> OracleSlabs:
    """Slabs + majorant grid; `.maj_array` holds a copy of the {M, 1/M} pairs [nck, ncj, nci, 2] and
    `.box_array` the supervoxels' boxes (six extents of 5 bits) [nck, ncj, nci]."""
    sc = oracle_scene(scene, supergrid=supergrid, levels=levels, merge_tau=merge_tau)
    sl = OracleSlabs()
    kg, kc, _, asy = [np.ascontiguousarray(a, dtype=np.float32) for a in scene.fields()]
    lib().oracle_build_slabs(C.byref(sc), _p(kg, C.c_float), _p(kc, C.c_float), _p(asy, C.c_float), C.byref(sl))
    n = sl.nci * sl.ncj * sl.nck
    sl.maj_array = np.ctypeslib.as_array(sl.maj, shape=(2 * n,)).copy().reshape(sl.nck, sl.ncj, sl.nci, 2) if n else None
    sl.box_array = np.ctypeslib.as_array(sl.box, shape=(n,)).copy().reshape(sl.nck, sl.ncj, sl.nci) if n else None
    lib().oracle_free_slabs(C.byref(sl))
    return sl


def cpu_ref(scene_f64, n_photons, nthreads=1, serial_seed=False):
    """Run oracle/cpu_ref.cpp on double-precision fields.

    scene_f64: dict with itot,jtot,ktot,dx,dy,dz,k_gas,k_cloud,ssa,asy (float64 arrays),
    albedo, zenith_rad, azimuth_rad. Returns (tallies, events[8])."""
    s = scene_f64
    class _S:  # shape helper
        itot, jtot, ktot = s["itot"], s["jtot"], s["ktot"]
    t = new_tallies(_S)
    ev = np.zeros(8, dtype=np.uint64)
    arrs = [np.ascontiguousarray(s[k], dtype=np.float64) for k in ("k_gas", "k_cloud", "ssa", "asy")]
    rc = lib().cpu_ref_trace(s["itot"], s["jtot"], s["ktot"], s["dx"], s["dy"], s["dz"],
                             *[_p(a, C.c_double) for a in arrs],
                             s["albedo"], s["zenith_rad"], s["azimuth_rad"],
                             n_photons, nthreads, int(serial_seed),
                             *[_p(t[n], C.c_uint64) for n in TALLY_NAMES], _p(ev, C.c_uint64))
    if rc != 0:
        raise RuntimeError(f"cpu_ref_trace failed: {rc}")
    return t, ev


def block_cloud_f64():
    """The built-in scene of raytracer_3d.cpp:168-220 in double precision, value for value."""
    itot = jtot = ktot = 128
    dx = dy = 50.0
    dz = 25.0
    k_ext_gas, ssa_gas, asy_gas = 1.0e-4, 0.5, 0.0
    k_ext_cloud, ssa_cloud, asy_cloud = 5.0e-3, 0.9, 0.85
    k_gas = np.full((ktot, jtot, itot), k_ext_gas)
    k_cloud = np.zeros((ktot, jtot, itot))
    ssa = np.full((ktot, jtot, itot), ssa_gas)
    asy = np.zeros((ktot, jtot, itot))
    x = (np.arange(itot) + 0.5) * dx
    y = (np.arange(jtot) + 0.5) * dy
    z = (np.arange(ktot) + 0.5) * dz
    m = ((z > 1000.) & (z < 1500.))[:, None, None] & ((y > 2700.) & (y < 3700.))[None, :, None] \
        & ((x > 4000.) & (x < 5000.))[None, None, :]
    k_cloud[m] = k_ext_cloud
    ssa[m] = (ssa_gas * k_ext_gas + ssa_cloud * k_ext_cloud) / (k_ext_gas + k_ext_cloud)
    asy[m] = (asy_gas * ssa_gas * k_ext_gas + asy_cloud * ssa_cloud * k_ext_cloud) \
        / (ssa_gas * k_ext_gas + ssa_cloud * k_ext_cloud)
    return dict(itot=itot, jtot=jtot, ktot=ktot, dx=dx, dy=dy, dz=dz, k_gas=k_gas, k_cloud=k_cloud,
                ssa=ssa, asy=asy, albedo=0.2, zenith_rad=50. * (math.pi / 180.), azimuth_rad=20. * (math.pi / 180.))


def scene_to_f64(scene):
    """float32 Scene -> the dict cpu_ref() takes (values widened, not recomputed)."""
    return dict(itot=scene.itot, jtot=scene.jtot, ktot=scene.ktot, dx=scene.dx, dy=scene.dy, dz=scene.dz,
                k_gas=scene.k_gas.astype(np.float64), k_cloud=scene.k_cloud.astype(np.float64),
                ssa=scene.ssa.astype(np.float64), asy=scene.asy.astype(np.float64), albedo=scene.albedo,
                zenith_rad=scene.zenith_deg * (math.pi / 180.), azimuth_rad=scene.azimuth_deg * (math.pi / 180.))


def run_reference_binary(arg: int, workdir: str, nthreads: int = 1):
    """Run the compiled reference (oracle/_ref/raytracer_3d_ref <arg>) in workdir; returns tallies."""
    env = {"PATH": "/usr/bin:/bin", "OMP_NUM_THREADS": str(nthreads)}    # (compiled from public source: a minimal environment)
    subprocess.run([REF_BIN, str(arg)], cwd=workdir, check=True, env=env, stdout=subprocess.DEVNULL)
    out = {}
    for n in TALLY_NAMES:
        shape = (128, 128, 128) if n.startswith("atmos") else (128, 128)
        out[n] = np.fromfile(os.path.join(workdir, n + ".bin"), dtype=np.uint64).reshape(shape)
    return out


def balance(t) -> int:
    """Photon balance of plot_results_3d.py:41-44 (must be exactly 0)."""
    inn = int(t["toa_down"].sum())
    out = int(t["surface_down_direct"].sum()) + int(t["surface_down_diffuse"].sum()) - int(t["surface_up"].sum()) \
        + int(t["toa_up"].sum()) + int(t["atmos_direct"].sum()) + int(t["atmos_diffuse"].sum())
    return inn - out


# ---- 2D tracer (reference src/raytracer.cpp) -------------------------------
REF2D_BIN = os.path.join(ROOT, "oracle", "_ref", "raytracer_2d_ref")
TALLY_NAMES_2D = ("toa_down", "toa_up", "surface_down_direct", "surface_down_diffuse", "surface_up", "atmos")


def block_cloud_2d_f64():
    """The built-in scene of raytracer.cpp:126-173 in double precision, value for value (its cloud mask
    tests the lower cell edges i*dx, k*dx, not the centres)."""
    itot, ktot, dx = 256, 128, 25.0
    k_ext_gas, ssa_gas, asy_gas = 1.0e-4, 0.5, 0.0
    k_ext_cloud, ssa_cloud, asy_cloud = 5.0e-3, 0.9, 0.85
    k_gas = np.full((ktot, itot), k_ext_gas)
    k_cloud = np.zeros((ktot, itot))
    ssa = np.full((ktot, itot), ssa_gas)
    asy = np.zeros((ktot, itot))
    xi = np.arange(itot) * dx
    zk = np.arange(ktot) * dx
    mask = ((zk > 1000) & (zk < 1500.))[:, None] & ((xi > 4000) & (xi < 5000))[None, :]
    k_cloud[mask] = k_ext_cloud
    ssa[mask] = (ssa_gas * k_ext_gas + ssa_cloud * k_ext_cloud) / (k_ext_gas + k_ext_cloud)
    asy[mask] = (asy_gas * ssa_gas * k_ext_gas + asy_cloud * ssa_cloud * k_ext_cloud) / (ssa_gas * k_ext_gas + ssa_cloud * k_ext_cloud)
    return dict(itot=itot, ktot=ktot, dx=dx, k_gas=k_gas, k_cloud=k_cloud, ssa=ssa, asy=asy,
                albedo=0.2, zenith_rad=50. * (np.pi / 180.))


def cpu_ref_2d(s, n_photons, nthreads=1, serial_seed=False):
    """Run oracle/cpu_ref_2d.cpp; returns (tallies dict with the six 2D names, events[8])."""
    itot, ktot = s["itot"], s["ktot"]
    t = {n: np.zeros((ktot, itot) if n == "atmos" else (itot,), dtype=np.uint64) for n in TALLY_NAMES_2D}
    ev = np.zeros(8, dtype=np.uint64)
    arrs = [np.ascontiguousarray(s[k], dtype=np.float64) for k in ("k_gas", "k_cloud", "ssa", "asy")]
    fn = lib().cpu_ref2d_trace
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_int, C.c_double] + [C.c_void_p] * 4 + [C.c_double, C.c_double, C.c_uint64, C.c_int, C.c_int] \
        + [C.c_void_p] * 7
    rc = fn(itot, ktot, s["dx"], *[a.ctypes.data for a in arrs], s["albedo"], s["zenith_rad"],
            n_photons, nthreads, int(serial_seed), *[t[n].ctypes.data for n in TALLY_NAMES_2D], ev.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"cpu_ref2d_trace failed: {rc}")
    return t, ev


def run_reference_binary_2d(arg: int, workdir: str, nthreads: int = 1):
    """Run the compiled reference (oracle/_ref/raytracer_2d_ref <arg>) in workdir; returns its six tallies."""
    env = {"PATH": "/usr/bin:/bin", "OMP_NUM_THREADS": str(nthreads)}
    subprocess.run([REF2D_BIN, str(arg)], cwd=workdir, check=True, env=env, stdout=subprocess.DEVNULL)
    return {n: np.fromfile(os.path.join(workdir, n + ".bin"), dtype=np.uint64).reshape((128, 256) if n == "atmos" else (256,))
            for n in TALLY_NAMES_2D}


def balance_2d(t) -> int:
    """Photon balance of plot_results.py:27-31 (must be exactly 0)."""
    return int(t["toa_down"].sum()) - (int(t["surface_down_direct"].sum()) + int(t["surface_down_diffuse"].sum())
                                       - int(t["surface_up"].sum()) + int(t["toa_up"].sum()) + int(t["atmos"].sum()))
