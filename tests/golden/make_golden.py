#!/usr/bin/env python3
"""Regenerates the golden fixtures (run in the build container, where /root/reference exists).

cpu_ref_100k.json   digests + sums of the seven tallies of the COMPILED REFERENCE
                    (oracle/_ref/raytracer_3d_ref 1, OMP_NUM_THREADS=1): the bit-exact pin of
                    oracle/cpu_ref.cpp. Same values as SURVEY.md §8c.
reference_1e7.json  global sums and per-level absorption sums of the compiled reference at
                    `raytracer_3d_ref 100` (10^7 photons, 8 threads): statistical pin.
photon_oracle_tiny.npz  tallies of oracle/photon_oracle.c on scenes.tiny(), seed 11, IDs [0, 2^14):
                    frozen vector for the float/Philox path (CPU oracle and GPU must both match).
cpu_ref_2d_100k.json  digests + sums of the six tallies of the compiled 2D reference
                    (oracle/_ref/raytracer_2d_ref 1, OMP_NUM_THREADS=1): the bit-exact pin of
                    oracle/cpu_ref_2d.cpp; plus its sums at `raytracer_2d_ref 40` (4*10^6 photons, 8 threads).
Fixtures are data only (counts and digests); no reference source is stored."""
import hashlib, json, os, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np
import oracle_lib as O
from raytracer_amd import scenes

with tempfile.TemporaryDirectory() as d:
    r = O.run_reference_binary(1, d, 1)
json.dump({n: {"sum": int(r[n].sum()), "md5": hashlib.md5(r[n].tobytes()).hexdigest()} for n in O.TALLY_NAMES},
          open(os.path.join(HERE, "cpu_ref_100k.json"), "w"), indent=1)
with tempfile.TemporaryDirectory() as d:
    r = O.run_reference_binary(100, d, 8)
atm = r["atmos_direct"] + r["atmos_diffuse"]
json.dump({"threads": 8, "sums": {n: int(r[n].sum()) for n in O.TALLY_NAMES},
           "atmos_level_sums": [int(v) for v in atm.sum(axis=(1, 2))],
           "surface_down_direct_row_sums": [int(v) for v in r["surface_down_direct"].sum(axis=1)]},
          open(os.path.join(HERE, "reference_1e7.json"), "w"), indent=1)
t, ev = O.photon_oracle(scenes.tiny(), 11, 0, 1 << 14, nthreads=1)
np.savez_compressed(os.path.join(HERE, "photon_oracle_tiny.npz"), **t, events=np.array([ev[k] for k in O.EVENT_NAMES], dtype=np.uint64))
with tempfile.TemporaryDirectory() as d:
    r = O.run_reference_binary_2d(1, d, 1)
with tempfile.TemporaryDirectory() as d:
    r40 = O.run_reference_binary_2d(40, d, 8)
json.dump({"pins": {n: {"sum": int(r[n].sum()), "md5": hashlib.md5(r[n].tobytes()).hexdigest()} for n in O.TALLY_NAMES_2D},
           "sums_4e6": {n: int(r40[n].sum()) for n in O.TALLY_NAMES_2D},
           "atmos_level_sums_4e6": [int(v) for v in r40["atmos"].sum(axis=1)]},
          open(os.path.join(HERE, "cpu_ref_2d_100k.json"), "w"), indent=1)
print("golden fixtures written")
