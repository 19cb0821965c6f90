"""Converter to the tracer's input files — the netCDF-free equivalent of python/make_binary.py:6-17.

The reference script slices one g-point out of four netCDF variables [gpt, z, y, x] and dumps them with
`.tofile` (headerless, C order [k][j][i]), after patching levels 265-268 (asy := 0, ssa := ssa of level
264; make_binary.py:11-12). This module does the same from arrays the caller already holds, or from a
.npz / four .npy files, and also writes the sidecar grid.hdr so that no dimension has to be compiled in
(raytracer_3d_io.cu:11-17). netCDF input is accepted only if the netCDF4 module is importable; it is
not a dependency.

    python -m raytracer_amd.convert fields.npz --outdir run1 --dx 100 --dy 100 --dz 30 --gpt 19 --patch-levels 265:269
"""
from __future__ import annotations

import argparse
import os

import numpy as np

from .scenes import Scene

# variable names of the reference's netCDF file (make_binary.py:7-10) -> tracer field names
NC_NAMES = {"k_ext_cloud": "kext_cloud", "k_ext_gas": "kext_gas", "ssa": "ssa", "asy": "g"}
FIELDS = ("k_ext_gas", "k_ext_cloud", "ssa", "asy")


def patch_levels(ssa: np.ndarray, asy: np.ndarray, lo: int, hi: int):
    """make_binary.py:11-12: asy[lo:hi] = 0 and ssa[lo:hi] = ssa[lo-1] (arrays [z, y, x]; copies)."""
    ktot = ssa.shape[0]
    if not (0 < lo < hi <= ktot):
        raise ValueError(f"patch levels {lo}:{hi} outside 1..{ktot}")
    ssa, asy = ssa.copy(), asy.copy()
    asy[lo:hi] = 0
    ssa[lo:hi] = ssa[lo - 1]
    return ssa, asy


def to_scene(fields: dict, dx: float, dy: float, dz: float, gpt: int | None = None,
             patch: tuple | None = None, name: str = "converted", **kw) -> Scene:
    """Build a Scene from a mapping with keys k_ext_gas, k_ext_cloud, ssa, asy (tracer names) or
    kext_gas, kext_cloud, ssa, g (the reference's netCDF names); arrays [z, y, x] or [gpt, z, y, x]."""
    arrs = {}
    for f in FIELDS:
        key = f if f in fields else NC_NAMES[f]
        if key not in fields:
            raise KeyError(f"field {f!r} (or {NC_NAMES[f]!r}) missing")
        a = np.asarray(fields[key])
        if a.ndim == 4:
            if gpt is None:
                raise ValueError(f"{key} is 4-D [gpt, z, y, x]: pass gpt")
            a = a[gpt]
        elif a.ndim != 3:
            raise ValueError(f"{key} has {a.ndim} dimensions, expected [z, y, x] or [gpt, z, y, x]")
        arrs[f] = a
    shapes = {a.shape for a in arrs.values()}
    if len(shapes) != 1:
        raise ValueError(f"fields differ in shape: {shapes}")
    ssa, asy = arrs["ssa"], arrs["asy"]
    if patch is not None:
        ssa, asy = patch_levels(ssa, asy, *patch)
    ktot, jtot, itot = ssa.shape
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return Scene(name, itot, jtot, ktot, float(dx), float(dy), float(dz),
                 f32(arrs["k_ext_gas"]), f32(arrs["k_ext_cloud"]), f32(ssa), f32(asy), **kw)


def load_fields(path: str) -> dict:
    """.npz with the four arrays, a directory of four .npy files, or (netCDF4 permitting) a .nc file."""
    if os.path.isdir(path):
        out = {}
        for f in FIELDS:
            for cand in (f, NC_NAMES[f]):
                p = os.path.join(path, cand + ".npy")
                if os.path.exists(p):
                    out[f] = np.load(p, mmap_mode="r")
        return out
    if path.endswith(".npz"):
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
    if path.endswith(".nc"):
        try:
            import netCDF4
        except ImportError as e:
            raise RuntimeError("reading .nc needs the netCDF4 module, which is not installed here; "
                               "export the four variables to .npz/.npy instead") from e
        ds = netCDF4.Dataset(path, "r")
        return {f: np.asarray(ds.variables[NC_NAMES[f]][:]) for f in FIELDS}
    raise RuntimeError(f"{path}: expected a .npz file, a directory of .npy files or a .nc file")


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("input")
    ap.add_argument("--outdir", default=".")
    ap.add_argument("--dx", type=float, required=True)
    ap.add_argument("--dy", type=float, required=True)
    ap.add_argument("--dz", type=float, required=True)
    ap.add_argument("--gpt", type=int, default=None, help="g-point index for 4-D inputs (make_binary.py:4 uses 19)")
    ap.add_argument("--patch-levels", default=None, metavar="LO:HI",
                    help="asy := 0 and ssa := ssa[LO-1] in levels LO..HI-1 (make_binary.py:11-12 uses 265:269)")
    ap.add_argument("--no-sidecar", action="store_true", help="write only the four headerless files")
    args = ap.parse_args(argv)
    patch = tuple(int(t) for t in args.patch_levels.split(":")) if args.patch_levels else None
    sc = to_scene(load_fields(args.input), args.dx, args.dy, args.dz, gpt=args.gpt, patch=patch)
    sc.write(args.outdir, sidecar=not args.no_sidecar)
    print(f"wrote {sc.itot} x {sc.jtot} x {sc.ktot} float32 fields to {args.outdir}")


if __name__ == "__main__":
    main()
