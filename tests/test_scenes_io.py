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


def test_sidecar_header_round_trip_and_float64_files(tmp_path):
    """f3: grid.hdr carries what raytracer_3d_io.cu:11-17 compiles in; the data files stay headerless."""
    sc = scenes.tiny()
    sc.albedo, sc.zenith_deg = 0.35, 30.0
    sc.write(str(tmp_path))
    hdr = scenes.read_sidecar(str(tmp_path))
    assert (hdr["itot"], hdr["jtot"], hdr["ktot"]) == (sc.itot, sc.jtot, sc.ktot)
    assert (hdr["dx"], hdr["dy"], hdr["dz"], hdr["dtype"]) == (sc.dx, sc.dy, sc.dz, "float32")
    back = scenes.Scene.read(str(tmp_path))                       # nothing but the directory
    assert (back.itot, back.jtot, back.ktot, back.dx, back.dy, back.dz) == (sc.itot, sc.jtot, sc.ktot, sc.dx, sc.dy, sc.dz)
    assert (back.albedo, back.zenith_deg, back.azimuth_deg) == (0.35, 30.0, sc.azimuth_deg)
    for a, b in zip(sc.fields(), back.fields()):
        assert np.array_equal(a, b)
    # arguments override the sidecar (and a wrong one is caught by the size check)
    with pytest.raises(RuntimeError, match="bytes"):
        scenes.Scene.read(str(tmp_path), ktot=sc.ktot + 1)
    # float64 files are narrowed to float32
    d64 = tmp_path / "f64"
    sc.write(str(d64), dtype=np.float64)
    assert os.path.getsize(d64 / "ssa.bin") == 8 * sc.ncell
    back64 = scenes.Scene.read(str(d64))
    for a, b in zip(sc.fields(), back64.fields()):
        assert b.dtype == np.float32 and np.array_equal(a, b)
    # headerless flow still works, and without a sidecar the dimensions must be given
    bare = tmp_path / "bare"
    sc.write(str(bare), sidecar=False)
    assert not os.path.exists(bare / "grid.hdr")
    with pytest.raises(RuntimeError, match="grid.hdr"):
        scenes.Scene.read(str(bare))
    with open(tmp_path / "grid.hdr", "a") as fh:
        fh.write("colour = blue\n")
    with pytest.raises(RuntimeError, match="unknown key"):
        scenes.read_sidecar(str(tmp_path))


def test_converter_follows_make_binary(tmp_path):
    """python -m raytracer_amd.convert: slice one g-point of [gpt, z, y, x] variables, patch a band of
    levels the way make_binary.py:11-12 does (asy := 0, ssa := ssa of the level below the band), dump
    headerless float32 [k][j][i] + sidecar."""
    from raytracer_amd import convert
    rng = np.random.default_rng(5)
    shape = (3, 12, 5, 7)                                          # gpt, z, y, x
    nc = {"kext_cloud": rng.random(shape), "kext_gas": rng.random(shape), "ssa": rng.random(shape), "g": rng.random(shape)}
    np.savez(tmp_path / "fields.npz", **nc)
    convert.main([str(tmp_path / "fields.npz"), "--outdir", str(tmp_path / "out"), "--dx", "100", "--dy", "100",
                  "--dz", "30", "--gpt", "2", "--patch-levels", "6:9"])
    sc = scenes.Scene.read(str(tmp_path / "out"))
    assert (sc.itot, sc.jtot, sc.ktot, sc.dx, sc.dz) == (7, 5, 12, 100.0, 30.0)
    # what make_binary.py would have written, step by step
    asy = nc["g"][2].copy(); ssa = nc["ssa"][2].copy()
    asy[6:9, :, :] = 0
    ssa[6:9, :, :] = ssa[5, :, :]
    for got, want in ((sc.k_cloud, nc["kext_cloud"][2]), (sc.k_gas, nc["kext_gas"][2]), (sc.ssa, ssa), (sc.asy, asy)):
        assert np.array_equal(got, want.astype(np.float32))
    raw = np.fromfile(tmp_path / "out" / "asy.bin", dtype=np.float32)
    assert raw.size == 12 * 5 * 7 and raw[3 + 2 * 7 + 4 * 35] == np.float32(asy[4, 2, 3])
    with pytest.raises(ValueError, match="gpt"):
        convert.to_scene(nc, 1, 1, 1)
    with pytest.raises(RuntimeError, match="netCDF4|expected"):
        convert.load_fields(str(tmp_path / "missing.nc"))


def _plot_script_reads(directory, itot=128, jtot=128, ktot=128):
    """What the reference's plot script does with a run's output before it plots (plot_results_3d.py:8-10, 28-48),
    restated: seven headerless uint64 files by name, the five 2-D ones reshaped (jtot, itot), the two 3-D ones
    (ktot, jtot, itot); photon balance in - out with out = surface_down - surface_up + toa_up + atmos."""
    def rd(name, shape):
        return np.fromfile(os.path.join(directory, name + ".bin"), dtype=np.uint64).reshape(shape)
    surface_down = rd("surface_down_direct", (jtot, itot)) + rd("surface_down_diffuse", (jtot, itot))
    surface_up, toa_up, toa_down = rd("surface_up", (jtot, itot)), rd("toa_up", (jtot, itot)), rd("toa_down", (jtot, itot))
    atmos = rd("atmos_direct", (ktot, jtot, itot)) + rd("atmos_diffuse", (ktot, jtot, itot))
    balance_in = np.int64(toa_down.sum())
    balance_out = np.int64(surface_down.sum() - surface_up.sum() + toa_up.sum() + atmos.sum())
    return int(balance_in), int(balance_out), int(balance_in - balance_out), float(toa_down.mean())


def test_t7_outputs_read_the_way_the_plot_script_reads_them(tmp_path):
    """SURVEY.md section 4, T7, wherever the tests run: files written by save_tallies for the reference's 128^3 scene,
    read by name, dtype and shape as plot_results_3d.py:28-35 reads them, give its photon balance of 0 (:41-48) and
    the normalisation it divides by (:52)."""
    sc = scenes.block_cloud()
    t, _ = O.photon_oracle(sc, 2, 0, 400_000)
    save_tallies(t, str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == sorted(n + ".bin" for n in O.TALLY_NAMES)
    b_in, b_out, net, norm = _plot_script_reads(str(tmp_path))
    assert (b_in, b_out, net) == (400_000, 400_000, 0) and norm == 400_000 / (128 * 128)


REF_PLOT = "/root/reference/python/plot_results_3d.py"


@pytest.mark.skipif(not os.path.exists(REF_PLOT), reason="the reference tree is only present in the build container")
def test_t7_reference_plot_script_runs_unmodified_on_our_files(tmp_path):
    """The same with the reference's own plot_results_3d.py run as it is (headless) in that directory -- where the
    reference tree exists; elsewhere parity with the script itself rests on the restatement above. The script is public,
    untrusted code: it gets an environment of four variables, not the test process's."""
    import subprocess
    import sys
    sc = scenes.block_cloud()
    t, _ = O.photon_oracle(sc, 2, 0, 400_000)
    save_tallies(t, str(tmp_path))
    env = {"PATH": "/usr/bin:/bin", "HOME": str(tmp_path), "MPLBACKEND": "Agg", "MPLCONFIGDIR": str(tmp_path)}
    r = subprocess.run([sys.executable, REF_PLOT], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = dict(l.split(":", 1) for l in r.stdout.splitlines() if ":" in l)
    assert int(lines["in"]) == 400_000 and int(lines["out"]) == 400_000 and int(lines["balance"]) == 0
    assert (int(lines["in"]), int(lines["out"]), int(lines["balance"])) == _plot_script_reads(str(tmp_path))[:3]
