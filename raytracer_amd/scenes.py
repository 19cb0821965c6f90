"""Synthetic optical-property scenes in the reference's input format.

Every generator returns a :class:`Scene` whose four float32 fields have the memory layout of the
``make_binary.py`` files (``/root/reference/python/make_binary.py:7-17``): headerless C-order
``[k][j][i]`` with ``i`` fastest. ``Scene.write(dir)`` writes ``k_ext_gas.bin``,
``k_ext_cloud.bin``, ``ssa.bin`` and ``asy.bin`` exactly as ``raytracer_3d_io.cu:54-57`` reads them.

Scenes:
  C1 ``block_cloud``      the built-in scene of raytracer_3d.cu:11-65 / raytracer_3d.cpp:168-220
  C2 ``rayleigh_column``  clear-sky homogeneous gas column (energy closure / Beer-Lambert)
  C3 ``cumulus``          broken cumulus field: union of random ellipsoids, 256x256x128
  C4 ``cumulus`` at 1024x1024x128 (LES-like)
"""
from __future__ import annotations

import dataclasses
import math
import os

import numpy as np


@dataclasses.dataclass
class Scene:
    name: str
    itot: int
    jtot: int
    ktot: int
    dx: float
    dy: float
    dz: float
    k_gas: np.ndarray      # float32 [ktot, jtot, itot]
    k_cloud: np.ndarray
    ssa: np.ndarray
    asy: np.ndarray
    albedo: float = 0.2                     # raytracer_3d.cu:24
    zenith_deg: float = 50.0                # raytracer_3d.cu:25
    azimuth_deg: float = 20.0               # raytracer_3d.cu:26

    @property
    def ncell(self) -> int:
        return self.itot * self.jtot * self.ktot

    @property
    def ncol(self) -> int:
        return self.itot * self.jtot

    def fields(self):
        return self.k_gas, self.k_cloud, self.ssa, self.asy

    def write(self, directory: str, sidecar: bool = True, dtype=np.float32) -> None:
        """Write the four input files of the raytracer_3d_io flow (raytracer_3d_io.cu:54-57) and, unless
        sidecar=False, the header file grid.hdr next to them (the data files themselves stay headerless,
        so the reference's own executable still reads them)."""
        os.makedirs(directory, exist_ok=True)
        for name, arr in (("k_ext_gas", self.k_gas), ("k_ext_cloud", self.k_cloud),
                          ("ssa", self.ssa), ("asy", self.asy)):
            np.ascontiguousarray(arr, dtype=dtype).tofile(os.path.join(directory, name + ".bin"))
        if sidecar:
            write_sidecar(directory, self.itot, self.jtot, self.ktot, self.dx, self.dy, self.dz,
                          dtype=np.dtype(dtype).name, albedo=self.albedo, zenith_deg=self.zenith_deg,
                          azimuth_deg=self.azimuth_deg)

    @staticmethod
    def read(directory: str, itot: int = None, jtot: int = None, ktot: int = None, dx: float = None,
             dy: float = None, dz: float = None, name: str = "from_files", **kw) -> "Scene":
        """Load the four headerless files; the size must match exactly (the reference over-reads by 2x
        and relies on EOF, raytracer_3d_io.cu:41 — not reproduced). Dimensions and spacings come from
        the arguments or, for those left out, from the sidecar grid.hdr (which may also say that the
        files hold float64; they are narrowed to float32 as the tracer computes in float32)."""
        hdr = read_sidecar(directory)
        given = dict(itot=itot, jtot=jtot, ktot=ktot, dx=dx, dy=dy, dz=dz)
        for key, val in given.items():
            if val is None:
                if key not in hdr:
                    raise RuntimeError(f"Scene.read: {key} not given and no grid.hdr in {directory} provides it")
                given[key] = hdr[key]
        itot, jtot, ktot = int(given["itot"]), int(given["jtot"]), int(given["ktot"])
        dtype = np.dtype(hdr.get("dtype", "float32"))
        for key in ("albedo", "zenith_deg", "azimuth_deg"):
            if key in hdr and key not in kw:
                kw[key] = float(hdr[key])
        n = itot * jtot * ktot
        arrs = []
        for fname in ("k_ext_gas", "k_ext_cloud", "ssa", "asy"):
            path = os.path.join(directory, fname + ".bin")
            if not os.path.exists(path):
                raise RuntimeError(f'Cannot read file "{fname}.bin"')   # raytracer_3d_io.cu:44
            if os.path.getsize(path) != dtype.itemsize * n:
                raise RuntimeError(f'File "{fname}.bin" has {os.path.getsize(path)} bytes, expected {dtype.itemsize*n}')
            arrs.append(np.fromfile(path, dtype=dtype).reshape(ktot, jtot, itot).astype(np.float32))
        return Scene(name, itot, jtot, ktot, float(given["dx"]), float(given["dy"]), float(given["dz"]), *arrs, **kw)


SIDECAR = "grid.hdr"
_SIDECAR_INT = ("itot", "jtot", "ktot")
_SIDECAR_FLOAT = ("dx", "dy", "dz", "albedo", "zenith_deg", "azimuth_deg")


def write_sidecar(directory: str, itot: int, jtot: int, ktot: int, dx: float, dy: float, dz: float,
                  dtype: str = "float32", **extra) -> str:
    """grid.hdr: `key = value` lines describing the four headerless data files beside it — what
    raytracer_3d_io.cu:11-17 fixes at compile time. Read by Scene.read and by drivers/raytracer_3d_io_gpu."""
    if dtype not in ("float32", "float64"):
        raise ValueError(f"sidecar dtype {dtype!r}: float32 or float64")
    path = os.path.join(directory, SIDECAR)
    with open(path, "w") as fh:
        fh.write("# sidecar of k_ext_gas.bin k_ext_cloud.bin ssa.bin asy.bin: headerless, C order [k][j][i]\n")
        for key, val in (("itot", itot), ("jtot", jtot), ("ktot", ktot)):
            fh.write(f"{key} = {int(val)}\n")
        for key, val in (("dx", dx), ("dy", dy), ("dz", dz)):
            fh.write(f"{key} = {float(val)!r}\n")
        fh.write(f"dtype = {dtype}\norder = kji\n")
        for key, val in extra.items():
            if key not in _SIDECAR_FLOAT:
                raise ValueError(f"sidecar key {key!r} unknown")
            fh.write(f"{key} = {float(val)!r}\n")
    return path


def read_sidecar(directory: str) -> dict:
    """Parse grid.hdr ({} if absent). Unknown keys and malformed lines are errors, not ignored."""
    path = os.path.join(directory, SIDECAR)
    if not os.path.exists(path):
        return {}
    out = {}
    with open(path) as fh:
        for lineno, line in enumerate(fh, 1):
            line = line.split("#", 1)[0].strip()
            if not line:
                continue
            if "=" not in line:
                raise RuntimeError(f"{path}:{lineno}: expected `key = value`")
            key, val = (t.strip() for t in line.split("=", 1))
            if key in _SIDECAR_INT:
                out[key] = int(val)
            elif key in _SIDECAR_FLOAT:
                out[key] = float(val)
            elif key == "dtype":
                if val not in ("float32", "float64"):
                    raise RuntimeError(f"{path}:{lineno}: dtype {val!r}: float32 or float64")
                out[key] = val
            elif key == "order":
                if val != "kji":
                    raise RuntimeError(f"{path}:{lineno}: order {val!r}: only kji (C order [k][j][i]) is supported")
            else:
                raise RuntimeError(f"{path}:{lineno}: unknown key {key!r}")
    return out


def _mix(k_gas, ssa_gas, asy_gas, k_cloud, ssa_cloud, asy_cloud):
    """Mixed single-scattering albedo and asymmetry, raytracer_3d.cu:60-63 (float32 arithmetic)."""
    f = np.float32
    k_gas = k_gas.astype(f); k_cloud = k_cloud.astype(f)
    sg = f(ssa_gas) * k_gas
    sc = f(ssa_cloud) * k_cloud
    ssa = ((sg + sc) / (k_gas + k_cloud)).astype(f)
    with np.errstate(invalid="ignore", divide="ignore"):
        # association of raytracer_3d.cu:62: (asy*ssa)*k, so the float32 values match the C++ driver's
        asy = (((f(asy_gas) * f(ssa_gas)) * k_gas + (f(asy_cloud) * f(ssa_cloud)) * k_cloud) / (sg + sc)).astype(f)
    asy = np.where(k_cloud > 0, asy, f(0)).astype(f)
    ssa = np.where(k_cloud > 0, ssa, f(ssa_gas)).astype(f)
    return ssa, asy


def block_cloud(itot=128, jtot=128, ktot=128, dx=50.0, dy=50.0, dz=25.0) -> Scene:
    """C1: one 1000 x 1000 x 500 m block cloud (raytracer_3d.cu:11-65)."""
    f = np.float32
    k_gas = np.full((ktot, jtot, itot), 1.0e-4, dtype=f)
    k_cloud = np.zeros((ktot, jtot, itot), dtype=f)
    x = (np.arange(itot) + 0.5) * dx
    y = (np.arange(jtot) + 0.5) * dy
    z = (np.arange(ktot) + 0.5) * dz
    mx = (x > 4000.0) & (x < 5000.0)
    my = (y > 2700.0) & (y < 3700.0)
    mz = (z > 1000.0) & (z < 1500.0)
    mask = mz[:, None, None] & my[None, :, None] & mx[None, None, :]
    k_cloud[mask] = f(5.0e-3)
    ssa, asy = _mix(k_gas, 0.5, 0.0, k_cloud, 0.9, 0.85)
    return Scene("block_cloud", itot, jtot, ktot, dx, dy, dz, k_gas, k_cloud, ssa, asy)


def rayleigh_column(itot=64, jtot=64, ktot=64, dx=100.0, dy=100.0, dz=50.0,
                    k_gas=1.0e-4, ssa=1.0, scale_height=None) -> Scene:
    """C2: homogeneous (or exponentially stratified) gas column, no cloud."""
    f = np.float32
    kg = np.full((ktot, jtot, itot), k_gas, dtype=f)
    if scale_height:
        z = (np.arange(ktot) + 0.5) * dz
        kg *= np.exp(-z / scale_height).astype(f)[:, None, None]
    zeros = np.zeros_like(kg)
    return Scene("rayleigh_column", itot, jtot, ktot, dx, dy, dz, kg, zeros,
                 np.full_like(kg, ssa), zeros.copy())


def cumulus(itot=256, jtot=256, ktot=128, dx=50.0, dy=50.0, dz=25.0, n_clouds=40, seed=0,
            z_base=1000.0, z_top=2000.0, k_gas=1.0e-4, ssa_gas=0.5, ssa_cloud=0.9, asy_cloud=0.85,
            k_cloud_range=(5.0e-3, 5.0e-2), target_cover=None, name=None) -> Scene:
    """C3/C4: broken cumulus = union of random ellipsoids between z_base and z_top, periodic in
    x and y. Built per-ellipsoid on its bounding box so 1024x1024x128 stays cheap."""
    f = np.float32
    rng = np.random.Generator(np.random.Philox(seed))
    kg = np.full((ktot, jtot, itot), k_gas, dtype=f)
    kc = np.zeros((ktot, jtot, itot), dtype=f)
    xs, ys = itot * dx, jtot * dy
    for _ in range(n_clouds):
        cx, cy = rng.uniform(0, xs), rng.uniform(0, ys)
        rz = rng.uniform(0.25, 0.5) * (z_top - z_base)
        cz = rng.uniform(z_base + rz, z_top - rz) if z_top - z_base > 2 * rz else 0.5 * (z_base + z_top)
        rx = rng.uniform(0.6, 1.6) * 2.0 * rz
        ry = rng.uniform(0.6, 1.6) * 2.0 * rz
        kval = f(math.exp(rng.uniform(math.log(k_cloud_range[0]), math.log(k_cloud_range[1]))))
        i0, i1 = int(math.floor((cx - rx) / dx)), int(math.ceil((cx + rx) / dx))
        j0, j1 = int(math.floor((cy - ry) / dy)), int(math.ceil((cy + ry) / dy))
        k0, k1 = max(0, int(math.floor((cz - rz) / dz))), min(ktot, int(math.ceil((cz + rz) / dz)))
        ii = np.arange(i0, i1); jj = np.arange(j0, j1); kk = np.arange(k0, k1)
        ex = (((ii + 0.5) * dx - cx) / rx) ** 2
        ey = (((jj + 0.5) * dy - cy) / ry) ** 2
        ez = (((kk + 0.5) * dz - cz) / rz) ** 2
        inside = (ez[:, None, None] + ey[None, :, None] + ex[None, None, :]) < 1.0
        sub = kc[np.ix_(kk, jj % jtot, ii % itot)]
        kc[np.ix_(kk, jj % jtot, ii % itot)] = np.where(inside, np.maximum(sub, kval), sub)
    ssa, asy = _mix(kg, ssa_gas, 0.0, kc, ssa_cloud, asy_cloud)
    nm = name or f"cumulus_{itot}x{jtot}x{ktot}"
    return Scene(nm, itot, jtot, ktot, dx, dy, dz, kg, kc, ssa, asy)


def cumulus_c3(seed=0) -> Scene:
    return cumulus(256, 256, 128, 50.0, 50.0, 25.0, n_clouds=40, seed=seed, name="C3_cumulus_256x256x128")


def cumulus_c4(seed=0) -> Scene:
    return cumulus(1024, 1024, 128, 25.0, 25.0, 25.0, n_clouds=600, seed=seed, name="C4_les_1024x1024x128")


def tiny(itot=16, jtot=12, ktot=10, seed=1) -> Scene:
    """Small ragged (non power-of-two, non-square) scene with a cloud deck for fast tests."""
    return cumulus(itot, jtot, ktot, 100.0, 80.0, 60.0, n_clouds=3, seed=seed, z_base=180.0,
                   z_top=480.0, k_gas=2.0e-4, k_cloud_range=(4.0e-3, 2.0e-2), name="tiny")


def sun_direction(zenith_deg: float, azimuth_deg: float):
    """Direction cosines of the direct beam in float32, as raytracer_3d.cu:25-30 computes them."""
    f = np.float32
    zen = f(zenith_deg * (math.pi / 180.0))
    azi = f(azimuth_deg * (math.pi / 180.0))
    return (f(-np.sin(zen) * np.cos(azi)), f(-np.sin(zen) * np.sin(azi)), f(-np.cos(zen)))
