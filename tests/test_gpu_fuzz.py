"""Randomised parity on the GPU: a slice of tools/fuzz_parity.py (random ragged scenes x random options x random
photon ID ranges; HIP kernels against oracle/photon_oracle.c, tallies and event counters bit for bit). The tool
itself was run over 3 400 cases on the round-1 kernel and some 975 000 on round 2's ("Randomised
parity"); the suite keeps 200 of them, half from the ranges whose cases also draw the options added in round 2
(launch sort, tiled launch order, counter-free kernels, several sort batches, 32-column tiles, split calls through
rt_trace_noalloc; from case 1 100 000 any sun and surface, and a context used twice: a new sun or new optics, counts
accumulating on the device or through rt_download_tallies / rt_upload_tallies)."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.fixture(scope="module")
def rt():
    import raytracer_amd
    raytracer_amd.load_library()      # fails loudly without the HIP library
    return raytracer_amd


@pytest.mark.parametrize("first", [0, 1000, 2000, 3000, 300000, 400000, 500000, 543210, 700000, 1100000, 2000000, 2000020])
def test_random_scenes_and_options_bit_identical_to_oracle(rt, first, capsys):
    import fuzz_parity
    argv = sys.argv
    sys.argv = ["fuzz_parity.py", "20", str(first)]
    try:
        with np.errstate(all="ignore"):
            rc = fuzz_parity.main()
    finally:
        sys.argv = argv
    out = capsys.readouterr().out
    assert rc == 0, out
    assert "20 cases bit-identical" in out


def test_random_driver_runs_bit_identical_to_oracle(rt, capsys):
    """A slice of tools/fuzz_drivers.py: the C++ driver raytracer_3d_io_gpu on random scenes with random --seed,
    --launch, --launch-tile and --resume splits against the oracle (the tool was run over 6 250 cases)."""
    import fuzz_drivers
    argv = sys.argv
    sys.argv = ["fuzz_drivers.py", "12", "1000"]
    try:
        rc = fuzz_drivers.main()
    finally:
        sys.argv = argv
    out = capsys.readouterr().out
    assert rc == 0, out
    assert "12 driver cases bit-identical" in out
