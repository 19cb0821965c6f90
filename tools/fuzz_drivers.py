#!/usr/bin/env python3
"""Randomised check of the drop-in driver raytracer_3d_io_gpu on one GPU: random small scenes written in the
reference's file format (with the sidecar header), random --seed / --launch / --launch-tile, optionally the run split
in two with --resume; the seven output files against oracle/photon_oracle.c with the launch order the driver must
have chosen (raytracer_amd.choose_launch_order restates its --launch=auto), bit for bit.

Test infrastructure like tests/ (it calls the oracle as the checker). Usage, on the GPU box:
    python tools/fuzz_drivers.py [cases=100] [first_case=0]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_parity  # noqa: E402
import oracle_lib as O  # noqa: E402
import raytracer_amd as rt  # noqa: E402

IO = os.path.join(ROOT, "drivers", "raytracer_3d_io_gpu")


def read(d, sc):
    return {n: np.fromfile(os.path.join(d, n + ".bin"), dtype=np.uint64).reshape(
        (sc.ktot, sc.jtot, sc.itot) if n.startswith("atmos") else (sc.jtot, sc.itot)) for n in O.TALLY_NAMES}


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    for case in range(first, first + cases):
        rng = np.random.default_rng(31000 + case)
        with np.errstate(all="ignore"):
            scene, kind = fuzz_parity.random_scene(rng, large=rng.random() < 0.1)
        e = int(rng.choice([16, 17]))
        n = 1 << e
        seed = int(rng.integers(0, 1 << 40))
        launch = str(rng.choice(["auto", "auto", "uniform", "0", "2", "4", "sobol"]))
        split = bool(rng.random() < 0.3)
        runs = [e - 1, e - 1] if split else [e]
        ncol = scene.itot * scene.jtot
        # the order the driver must choose: auto looks at the photons of ONE run
        if launch == "auto":
            k, tile = rt.choose_launch_order(scene.itot, scene.jtot, scene.ktot, 1 << runs[0])
        elif launch == "uniform":
            k, tile = -1, 0
        elif launch == "sobol":
            k, tile = -2, 0
        else:
            k, tile = int(launch), 0
        args = [f"--seed={seed}", f"--launch={launch}"]
        tiles = [t for t in (1, 2, 3) if scene.itot % (1 << t) == 0 and scene.jtot % (1 << t) == 0]
        if k >= 0 and tiles and rng.random() < 0.3:
            tile = int(rng.choice(tiles))
            args.append(f"--launch-tile={tile}")
        desc = dict(case=case, kind=str(kind), grid=(scene.itot, scene.jtot, scene.ktot), e=e, seed=seed, launch=launch,
                    k=k, tile=tile, split=split, zenith=scene.zenith_deg, azimuth=scene.azimuth_deg, albedo=scene.albedo)
        with tempfile.TemporaryDirectory() as d:
            scene.write(d)
            for i, ex in enumerate(runs):
                r = subprocess.run([IO, str(ex), *args, *(["--resume"] if i else [])], cwd=d, capture_output=True,
                                   text=True, timeout=300)
                if r.returncode != 0:
                    print("ERROR", desc, r.stderr[-500:], flush=True)
                    return 1
            got = read(d, scene)
        want, _ = O.photon_oracle(scene, seed, 0, n, launch_log2k=k, launch_tile_log2=tile)
        bad = [name for name in O.TALLY_NAMES if not np.array_equal(got[name], want[name])]
        if bad:
            print("MISMATCH", desc, bad, flush=True)
            return 1
        if case % 10 == 0:
            print("ok", desc, flush=True)
    print(f"{cases} driver cases bit-identical", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
