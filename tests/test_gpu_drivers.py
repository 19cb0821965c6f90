"""The two drop-in executables (drivers/) on the GPU: CLI contract, stdout lines, file formats,
and tallies bit-identical to the CPU oracle on the same scene and photon IDs."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from raytracer_amd import scenes, TALLY_NAMES

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GPU = os.path.join(ROOT, "drivers", "raytracer_3d_gpu")
IO = os.path.join(ROOT, "drivers", "raytracer_3d_io_gpu")
GPU2D = os.path.join(ROOT, "drivers", "raytracer_gpu")
# --launch=auto when the photon count is no whole number of passes over the columns: the Sobol launch (f1)
SOBOL = -2


def _read(d, sc):
    out = {}
    for n in TALLY_NAMES:
        shape = (sc.ktot, sc.jtot, sc.itot) if n.startswith("atmos") else (sc.jtot, sc.itot)
        out[n] = np.fromfile(os.path.join(d, n + ".bin"), dtype=np.uint64).reshape(shape)
    return out


def test_builtin_scene_driver(tmp_path):
    r = subprocess.run([GPU, "18"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert lines[0].startswith("Shooting 262144 photons (8 per thread)")
    assert lines[1].startswith("Duration cloud-mask computation:") and lines[2] == "1000 - 1500"
    assert any(l.startswith("Duration: ") and l.endswith("(s)") for l in lines)
    sc = scenes.block_cloud()
    got = _read(tmp_path, sc)
    # the default launch is --launch=auto (the reference launches quasi-randomly, raytracer_kernels.cu:331):
    # 2^18 photons are 16 = 2^4 whole passes over the 128 x 128 columns
    want, _ = O.photon_oracle(sc, 0, 0, 1 << 18, launch_log2k=4)
    for n in TALLY_NAMES:
        assert np.array_equal(got[n], want[n]), n
    assert O.balance(got) == 0
    assert int(np.abs(got["toa_down"].astype(np.int64) - 16).max()) <= 2          # even over the domain top
    ck = dict(l.split(" = ") for l in open(tmp_path / "checkpoint").read().splitlines() if " = " in l)
    assert ck["next_photon_id"] == str(1 << 18) and ck["launch_log2k"] == "4" and ck["seed"] == "0"


def test_cli_contract(tmp_path):
    r = subprocess.run([GPU], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "please add the exponent n" in r.stdout
    r = subprocess.run([GPU, "10"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "must be larger than 32768" in r.stderr
    r = subprocess.run([IO, "16", "--itot=8", "--jtot=8", "--ktot=8"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and 'Cannot read file "k_ext_gas.bin"' in r.stderr


def test_file_input_driver_and_zenith_sweep(tmp_path):
    sc = scenes.tiny()
    sc.write(str(tmp_path))
    dims = [f"--itot={sc.itot}", f"--jtot={sc.jtot}", f"--ktot={sc.ktot}", f"--dx={sc.dx}", f"--dy={sc.dy}", f"--dz={sc.dz}"]
    r = subprocess.run([IO, "16", "--seed=3", *dims], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    want, _ = O.photon_oracle(sc, 3, 0, 1 << 16, launch_log2k=SOBOL)     # 192 columns: no whole pass, so auto = Sobol
    got = _read(tmp_path, sc)
    ck = dict(l.split(" = ") for l in open(tmp_path / "checkpoint").read().splitlines() if " = " in l)
    assert ck["launch_log2k"] == "-2"
    for n in TALLY_NAMES:
        assert np.array_equal(got[n], want[n]), n
    r = subprocess.run([IO, "15", "--zenith=0,30,60", *dims], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    for zen in (0, 30, 60):
        sc.zenith_deg = float(zen)
        want, _ = O.photon_oracle(sc, 0, 0, 1 << 15, launch_log2k=SOBOL)
        got = _read(tmp_path / f"zenith_{zen:02d}", sc)
        for n in TALLY_NAMES:
            assert np.array_equal(got[n], want[n]), (zen, n)


def test_resume_continues_the_photon_sequence(tmp_path):
    """f2 (the restart raytracer_3d.cu:89, 98-104 prepares for): two runs of 2^17 photons, the second with --resume,
    leave exactly the files of one run of 2^18 -- tallies and checkpoint -- for the built-in scene driver (stratified
    launch, K fixed by the first run) and for the file driver (Sobol launch, chosen by auto and kept by --resume)."""
    one, two = tmp_path / "one", tmp_path / "two"
    one.mkdir(); two.mkdir()
    sc = scenes.block_cloud()
    r = subprocess.run([GPU, "18", "--launch=3", "--seed=11"], cwd=one, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([GPU, "17", "--launch=3", "--seed=11"], cwd=two, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    half = _read(two, sc)
    assert int(half["toa_down"].sum()) == 1 << 17
    r = subprocess.run([GPU, "17", "--seed=11", "--resume"], cwd=two, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert f"Resuming . at photon {1 << 17}" in r.stdout
    a, b = _read(one, sc), _read(two, sc)
    for n in TALLY_NAMES:
        assert np.array_equal(a[n], b[n]), n
    assert int(b["toa_down"].sum()) == 1 << 18 and O.balance(b) == 0
    assert open(one / "checkpoint").read() == open(two / "checkpoint").read()
    want, _ = O.photon_oracle(sc, 11, 0, 1 << 18, launch_log2k=3)
    for n in TALLY_NAMES:
        assert np.array_equal(b[n], want[n]), n
    # what does not continue the same sequence is refused
    for bad, msg in ((["--seed=12"], "another seed"), (["--seed=11", "--zenith=40"], "another seed, sun position"),
                     (["--seed=11", "--launch=2"], "--launch=3"), (["--seed=11", "--itot=64"], "another grid")):
        r = subprocess.run([GPU, "17", "--resume", *bad], cwd=two, capture_output=True, text=True, timeout=120)
        assert r.returncode == 1 and msg in r.stderr, (bad, r.stderr)
    r = subprocess.run([GPU, "17", "--resume"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "checkpoint" in r.stderr
    # a third leg on the file driver, Sobol launch, three legs of 2^15 == one call of 3 * 2^15 photon ids
    tiny = scenes.tiny()
    d = tmp_path / "io"
    tiny.write(str(d))
    for leg in range(3):
        r = subprocess.run([IO, "15", "--seed=4"] + (["--resume"] if leg else []), cwd=d, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
    want, _ = O.photon_oracle(tiny, 4, 0, 3 << 15, launch_log2k=SOBOL)
    got = _read(d, tiny)
    for n in TALLY_NAMES:
        assert np.array_equal(got[n], want[n]), n


def test_zenith_and_azimuth_sweep(tmp_path):
    """f2: --zenith=a,b --azimuth=c,d runs every pair with the grids resident, one directory each."""
    sc = scenes.tiny()
    sc.write(str(tmp_path))
    r = subprocess.run([IO, "15", "--zenith=20,55", "--azimuth=0,135"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    for zen in (20, 55):
        for azi in (0, 135):
            sc.zenith_deg, sc.azimuth_deg = float(zen), float(azi)
            want, _ = O.photon_oracle(sc, 0, 0, 1 << 15, launch_log2k=SOBOL)
            got = _read(tmp_path / f"zenith_{zen:02d}_azimuth_{azi:03d}", sc)
            for n in TALLY_NAMES:
                assert np.array_equal(got[n], want[n]), (zen, azi, n)
    r = subprocess.run([IO, "15", "--azimuth=90,270"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "azimuth_090" / "atmos_direct.bin").exists() and (tmp_path / "azimuth_270" / "checkpoint").exists()


def test_io_driver_at_its_default_grid_with_a_zenith_sweep(tmp_path):
    """BASELINE config 5 at the grid raytracer_3d_io.cu:11-17 compiles in (240 x 240 x 284 cells of 100/100/30 m): no
    dimension options, no sidecar, four headerless float32 files of 65 MB each, two solar zenith angles with the
    full 3D absorption output per run; bit-identical to the oracle. 57600 columns admit no whole pass for 2^20
    photons: --launch=auto (the default) takes the Sobol launch here, and toa_down is far more even than Poisson."""
    sc = scenes.cumulus(240, 240, 284, 100.0, 100.0, 30.0, n_clouds=60, seed=7, z_base=900.0, z_top=3000.0,
                        name="io_default_grid")
    sc.write(str(tmp_path), sidecar=False)
    r = subprocess.run([IO, "20", "--seed=6", "--zenith=20,65"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for zen in (20, 65):
        sc.zenith_deg = float(zen)
        want, _ = O.photon_oracle(sc, 6, 0, 1 << 20, launch_log2k=SOBOL)
        got = _read(tmp_path / f"zenith_{zen:02d}", sc)
        for n in TALLY_NAMES:
            assert np.array_equal(got[n], want[n]), (zen, n)
        td = got["toa_down"].astype(np.float64)
        assert td.var() < 0.2 * td.mean(), (td.var(), td.mean())      # 18.2 photons per column on average
        ck = dict(l.split(" = ") for l in open(tmp_path / f"zenith_{zen:02d}" / "checkpoint").read().splitlines() if " = " in l)
        assert ck["launch_log2k"] == "-2"
        assert O.balance(got) == 0 and got["atmos_direct"].shape == (284, 240, 240)


def test_stratified_launch_option_of_the_drivers(tmp_path):
    """--launch=auto picks the largest K <= 5 with whole passes of ncol * 2^K photons, else the Sobol launch;
    --launch=K, --launch=sobol and --launch=uniform are explicit."""
    sc = scenes.tiny()
    sc.write(str(tmp_path))
    ncol = sc.itot * sc.jtot
    dims = [f"--itot={sc.itot}", f"--jtot={sc.jtot}", f"--ktot={sc.ktot}", f"--dx={sc.dx}", f"--dy={sc.dy}", f"--dz={sc.dz}"]
    n = 1 << 16
    k_auto = max((k for k in range(6) if n % (ncol << k) == 0), default=SOBOL)
    assert k_auto == SOBOL
    for arg, k in (("--launch=auto", k_auto), ("--launch=2", 2), ("--launch=uniform", -1), ("--launch=sobol", SOBOL)):
        r = subprocess.run([IO, "16", "--seed=5", arg, *dims], cwd=tmp_path, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        want, _ = O.photon_oracle(sc, 5, 0, n, launch_log2k=k)
        got = _read(tmp_path, sc)
        for name in TALLY_NAMES:
            assert np.array_equal(got[name], want[name]), (arg, name)


def test_tiled_launch_order_of_the_drivers(tmp_path):
    """--launch-tile=T counts the columns of the stratified launch in tiles of 2^T x 2^T (the library's
    launch_tile_log2): bit-identical to the oracle with the same order, recorded in the checkpoint, refused when the
    grid is not a whole number of tiles. On a grid beyond the caches whose sides are multiples of 32, --launch=auto
    chooses tiles of 32 x 32 and up to 2^8 IDs per column by itself (1024 x 1024 x 20 cells: 2^28 photons are one
    pass; toa_down is then 256 per column up to edge rounding, the balance is exact)."""
    sc = scenes.tiny()
    sc.write(str(tmp_path))
    dims = [f"--itot={sc.itot}", f"--jtot={sc.jtot}", f"--ktot={sc.ktot}", f"--dx={sc.dx}", f"--dy={sc.dy}", f"--dz={sc.dz}"]
    r = subprocess.run([IO, "16", "--seed=5", "--launch=3", "--launch-tile=2", *dims], cwd=tmp_path, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    want, _ = O.photon_oracle(sc, 5, 0, 1 << 16, launch_log2k=3, launch_tile_log2=2)
    got = _read(tmp_path, sc)
    for name in TALLY_NAMES:
        assert np.array_equal(got[name], want[name]), name
    ck = dict(l.split(" = ") for l in open(tmp_path / "checkpoint").read().splitlines() if " = " in l)
    assert ck["launch_log2k"] == "3" and ck["launch_tile_log2"] == "2"
    r = subprocess.run([IO, "16", "--launch=3", "--launch-tile=3", *dims], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "multiples of 8" in r.stderr
    r = subprocess.run([IO, "16", "--seed=5", "--launch=3", "--launch-tile=1", "--resume", *dims], cwd=tmp_path,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "--launch-tile=2" in r.stderr

    big = tmp_path / "big"
    big.mkdir()
    sc = scenes.cumulus(1024, 1024, 20, 25.0, 25.0, 100.0, n_clouds=200, seed=11, z_base=400.0, z_top=1200.0, name="big_flat")
    sc.write(str(big))
    r = subprocess.run([IO, "28", "--seed=2"], cwd=big, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    ck = dict(l.split(" = ") for l in open(big / "checkpoint").read().splitlines() if " = " in l)
    assert ck["launch_log2k"] == "8" and ck["launch_tile_log2"] == "5"
    got = _read(big, sc)
    assert O.balance(got) == 0 and int(got["toa_down"].sum()) == 1 << 28
    assert int(np.abs(got["toa_down"].astype(np.int64) - 256).max()) <= 8


def test_io_driver_reads_the_sidecar_header(tmp_path):
    """f3: with grid.hdr beside the data files the driver needs no --itot/--dx options; float64 files work."""
    sc = scenes.tiny()
    sc.zenith_deg = 35.0
    sc.write(str(tmp_path), dtype=np.float64)
    r = subprocess.run([IO, "16", "--seed=8"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    want, _ = O.photon_oracle(sc, 8, 0, 1 << 16, launch_log2k=SOBOL)
    got = _read(tmp_path, sc)
    for n in TALLY_NAMES:
        assert np.array_equal(got[n], want[n]), n
    with open(tmp_path / "grid.hdr", "a") as fh:
        fh.write("colour = blue\n")
    r = subprocess.run([IO, "16"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "unknown key colour" in r.stderr


def test_2d_driver_writes_what_plot_results_reads(tmp_path):
    """f4: raytracer_gpu <multiple of 100 000> — CLI of raytracer.cpp:424-433, the six files of :416-421 read the
    way plot_results.py:17-23 reads them, counts bit-identical to the planar oracle."""
    from test_oracle_2d import scene_2d
    r = subprocess.run([GPU2D], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "Add the multiple of 100,000 photons as an argument!" in r.stdout
    r = subprocess.run([GPU2D, "3", "--seed=4"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert any(l.startswith("Duration: ") and l.endswith("(s)") for l in r.stdout.splitlines())
    itot, ktot = 256, 128
    atmos = np.fromfile(tmp_path / "atmos.bin", dtype=np.uint64).reshape((ktot, itot))       # plot_results.py:23
    rows = {n: np.fromfile(tmp_path / f"{n}.bin", dtype=np.uint64) for n in TALLY_NAMES[:5]}
    assert all(v.shape == (itot,) for v in rows.values())
    want, _ = O.photon_oracle(scene_2d(), 4, 0, 300000, planar=True)
    for n in TALLY_NAMES[:5]:
        assert np.array_equal(rows[n], want[n][0]), n
    assert np.array_equal(atmos, (want["atmos_direct"] + want["atmos_diffuse"])[:, 0, :])
    balance = int(rows["toa_down"].sum()) - (int(rows["surface_down_direct"].sum()) + int(rows["surface_down_diffuse"].sum())
                                             - int(rows["surface_up"].sum()) + int(rows["toa_up"].sum()) + int(atmos.sum()))
    assert balance == 0                                                                       # plot_results.py:27-31


def test_pipelined_sweep_writes_the_same_bytes_and_a_json_summary(tmp_path):
    """BASELINE config 5: a zenith sweep traces job k+1 while job k's seven arrays come down over a
    copy stream into pinned memory and a host thread writes its files (rt_download_tallies_async +
    rt_swap_tally_buffer; raytracer_3d_io.cu:174-212 does the three strictly one after another). Every file of every
    run -- tallies and checkpoints -- is byte-identical with --pipeline=0, and equal to the oracle; the last stdout
    line is one JSON object (metrics/logging)."""
    import json
    sc = scenes.cumulus(48, 40, 30, 100.0, 100.0, 30.0, n_clouds=8, seed=9, z_base=240.0, z_top=600.0, name="sweep")
    zens = (0, 15, 30, 45, 60, 75)
    out = {}
    for mode in ("1", "0", "2", "3"):       # 1: four jobs' files in flight (automatic), 2 / 3: that many, 0: the reference's order
        d = tmp_path / f"pipe{mode}"
        d.mkdir()
        sc.write(str(d))
        r = subprocess.run([IO, "17", "--seed=12", f"--pipeline={mode}", "--zenith=" + ",".join(str(z) for z in zens)], cwd=d,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        summary = json.loads(r.stdout.splitlines()[-1])
        assert summary["jobs"] == 6 and summary["photons_per_job"] == 1 << 17 and summary["pipeline"] == (mode != "0")
        assert summary["jobs_in_flight"] == {"1": 4, "0": 1, "2": 2, "3": 3}[mode]
        assert summary["grid"] == [48, 40, 30] and summary["end_to_end_photons_per_s"] > 0
        assert sum(l.startswith("Duration: ") for l in r.stdout.splitlines()) == (1 if mode != "0" else 6)
        out[mode] = d
    for zen in zens:
        sub = f"zenith_{zen:02d}"
        for name in list(TALLY_NAMES) + ["checkpoint"]:
            fn = name + ".bin" if name != "checkpoint" else name
            for mode in ("1", "2", "3"):
                assert open(out[mode] / sub / fn, "rb").read() == open(out["0"] / sub / fn, "rb").read(), (mode, zen, name)
        sc.zenith_deg = float(zen)
        k, _ = __import__("raytracer_amd").choose_launch_order(sc.itot, sc.jtot, sc.ktot, 1 << 17)
        want, _ = O.photon_oracle(sc, 12, 0, 1 << 17, launch_log2k=k)
        got = _read(out["1"] / sub, sc)
        for n in TALLY_NAMES:
            assert np.array_equal(got[n], want[n]), (zen, n)
    # resumed pipelined sweep == one run of twice the photons
    d2 = tmp_path / "twice"
    d2.mkdir()
    sc.write(str(d2))
    r = subprocess.run([IO, "17", "--seed=12", "--resume", "--zenith=" + ",".join(str(z) for z in zens)], cwd=out["1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([IO, "18", "--seed=12", "--launch=sobol", "--zenith=" + ",".join(str(z) for z in zens)], cwd=d2,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    ck = dict(l.split(" = ") for l in open(out["1"] / "zenith_00" / "checkpoint").read().splitlines() if " = " in l)
    if ck["launch_log2k"] == "-2":
        for zen in zens:
            for name in TALLY_NAMES:
                a = open(out["1"] / f"zenith_{zen:02d}" / (name + ".bin"), "rb").read()
                assert a == open(d2 / f"zenith_{zen:02d}" / (name + ".bin"), "rb").read(), (zen, name)
