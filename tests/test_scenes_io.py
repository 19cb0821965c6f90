"""File formats on both sides of the path: the four float32 inputs of the raytracer_3d_io flow
(make_binary.py layout) and the seven uint64 outputs read the way plot_results_3d.py reads them."""
import os

import numpy as np
import pytest

import oracle_lib as O
from raytracer_amd import scenes, save_tallies, photon_balance, algorithmic_bytes, TALLY_NAMES


def test_input_files_round_trip(tmp_path):
    sc = scenes.tiny()
    sc.write(str(tmp_path))
    for f in ("k_ext_gas", "k_ext_cloud", "ssa", "asy"):
        assert os.path.getsize(tmp_path / f"{f}.bin") == 4 * sc.ncell     # headerless float32
    back = scenes.Scene.read(str(tmp_path), sc.itot, sc.jtot, sc.ktot, sc.dx, sc.dy, sc.dz)
    for a, b in zip(sc.fields(), back.fields()):
        assert np.array_equal(a, b)
    raw = np.fromfile(tmp_path / "k_ext_cloud.bin", dtype=np.float32)
    k, j, i = 3, 2, 5
    assert raw[i + j * sc.itot + k * sc.itot * sc.jtot] == sc.k_cloud[k, j, i]   # i fastest, raytracer_3d_io.cu:66


def test_missing_and_wrong_size_inputs(tmp_path):
    sc = scenes.tiny()
    with pytest.raises(RuntimeError, match='Cannot read file "k_ext_gas.bin"'):
        scenes.Scene.read(str(tmp_path), sc.itot, sc.jtot, sc.ktot, sc.dx, sc.dy, sc.dz)
    sc.write(str(tmp_path))
    with pytest.raises(RuntimeError, match="bytes"):
        scenes.Scene.read(str(tmp_path), sc.itot + 1, sc.jtot, sc.ktot, sc.dx, sc.dy, sc.dz)


def test_block_cloud_matches_reference_scene():
    sc = scenes.block_cloud()
    ref = O.block_cloud_f64()
    assert np.array_equal(sc.k_cloud > 0, ref["k_cloud"] > 0) and int((sc.k_cloud > 0).sum()) == 20 * 20 * 20
    assert np.allclose(sc.ssa, ref["ssa"], rtol=1e-6) and np.allclose(sc.asy, ref["asy"], rtol=1e-6)
    assert sc.k_gas.dtype == np.float32 and sc.k_gas.shape == (128, 128, 128)
    # the float32 mixing of raytracer_3d.cu:60-63 as g++ evaluates it (drivers/raytracer_3d_gpu.cpp)
    assert float(sc.ssa.max()).hex() == "0x1.c8c8c80000000p-1" and float(sc.asy.max()).hex() == "0x1.ae6ae60000000p-1"


def test_outputs_read_like_plot_results_3d(tmp_path):
    """plot_results_3d.py:28-48: np.fromfile(..., uint64).reshape and the balance formula."""
    sc = scenes.tiny()
    t, ev = O.photon_oracle(sc, 8, 0, 5000)      # any valid tallies will do for the format check
    save_tallies(t, str(tmp_path))
    rd = {}
    for n in TALLY_NAMES:
        shape = (sc.ktot, sc.jtot, sc.itot) if n.startswith("atmos") else (sc.jtot, sc.itot)
        rd[n] = np.fromfile(tmp_path / f"{n}.bin", dtype=np.uint64).reshape(shape)
        assert np.array_equal(rd[n], t[n])
    balance_in = np.int64(rd["toa_down"].sum())
    balance_out = np.int64((rd["surface_down_direct"] + rd["surface_down_diffuse"]).sum() - rd["surface_up"].sum()
                           + rd["toa_up"].sum() + (rd["atmos_direct"] + rd["atmos_diffuse"]).sum())
    assert balance_in - balance_out == 0 == photon_balance(rd)
    # algorithmic bytes formula on the oracle's counters
    b = algorithmic_bytes(ev, 5000)
    refl = int(t["surface_up"].sum())
    assert b == 4 * ev["collision_tests"] + 4 * ev["real"] + 8 * ev["scatter"] + 16 * (5000 + ev["surface"] + refl + ev["toa"] + ev["absorb"])


def test_cumulus_generator_is_deterministic_and_periodic():
    a, b = scenes.cumulus(32, 32, 16, n_clouds=4, seed=2, z_base=100, z_top=300), scenes.cumulus(32, 32, 16, n_clouds=4, seed=2, z_base=100, z_top=300)
    assert np.array_equal(a.k_cloud, b.k_cloud) and (a.k_cloud > 0).any()
    assert np.all(a.asy[a.k_cloud == 0] == 0) and np.all(a.asy[a.k_cloud > 0] > 0)
    assert np.all((a.ssa > 0) & (a.ssa <= 1))
