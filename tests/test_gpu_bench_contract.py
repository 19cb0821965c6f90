"""bench.py's contract with the driver: ONE JSON line on stdout, with the metric, the roofline object and (when asked
for) the CPU baseline -- a small workload, as a subprocess, the way the driver runs it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c1", "--photons-log2", "22",
                        "--steps", "2", "--warmup", "1", *extra], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[:500]
    return json.loads(lines[0])


def test_one_json_line_with_roofline_and_cpu_baseline():
    d = _run("--cpu-seconds", "2", "--cpu-scenes", "c1")
    assert d["metric"] == "photons/sec (whole node)" and d["unit"] == "photons/s" and d["n_gpus"] == 1
    assert d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["value"] > 1e8 and abs(d["value"] - (1 << 22) / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    roof = d["roofline"]
    # bound names the binding ceiling: "valu" only when the committed counters of exactly this
    # workload and photon count are replayed; the streaming-HBM figure is then kept in roofline.hbm
    assert roof["bound"] in ("hbm", "valu", "gather") and roof["kernel_ms"] > 0 and 0 < roof["frac"] <= 1.0
    hbm = roof if roof["bound"] == "hbm" else roof["hbm"]
    assert hbm["unit"] == "GB/s" and hbm["peak"] == 8000.0 and abs(hbm["frac"] - hbm["achieved"] / hbm["peak"]) < 1e-12
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 1e4 and cpu["one_thread"]["cores"] == 1
    assert "sample" in cpu and "cpu_model" in cpu
    if "reference_binary" in cpu:          # the compiled reference travels with the tree when it was built
        assert cpu["reference_binary"]["all_cores"]["kind"] == "reference"
    assert d["uniform_launch"]["balance"] == 0 and d["uniform_launch"]["value"] > 1e8
    assert d["sobol_launch"]["balance"] == 0 and d["sobol_launch"]["toa_down_variance_over_mean"] < 0.1
    assert 0.7 < d["uniform_launch"]["toa_down_variance_over_mean"] < 1.4


def test_collective_path_keeps_stdout_clean():
    """--force-dist runs the step with the RCCL all-reduce at world size 1; RCCL's banner must not reach stdout."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c1", "--photons-log2", "22",
                        "--steps", "2", "--warmup", "1", "--force-dist", "--no-cpu-baseline", "--no-uniform-leg"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[:500]
    d = json.loads(lines[0])
    assert "RCCL all-reduce" in d["config"]["workload"] and d["value"] > 1e8


def test_two_ranks_started_plainly_with_gpus_2():
    """`python bench.py --gpus 2` without a launcher starts its own ranks (torch.distributed.run, as the driver would);
    here both on the one GPU of the test box with the gloo stand-in for RCCL. Still ONE JSON line, from rank 0, with the
    whole job's photons."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-device", "--backend", "gloo",
                        "--workload", "c1", "--photons-log2", "21", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[:500]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["photons_per_gpu_per_step"] == 1 << 21
    assert abs(d["value"] - 2 * (1 << 21) / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_c5_workload_measures_the_sweep_end_to_end(tmp_path):
    """--workload c5 (BASELINE config 5): runs under six zenith angles on the io flow's own grid, each with the download of
    the seven arrays and the seven output files inside the timed region, pipelined; beside it the same runs one thing
    after another, the trace alone, and the C3 grid."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c5", "--photons-log2", "20",
                        "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--outdir", str(tmp_path)],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[:500]
    d = json.loads(lines[0])
    assert d["metric"] == "photons/sec (whole node)" and d["steps"] == 3 and d["config"]["photons_per_gpu_per_step"] == 1 << 20
    io, c3 = d["io_default_grid"], d["c3_grid"]
    assert io["grid"] == [240, 240, 284] and io["output_bytes_per_job"] == 8 * (5 * 240 * 240 + 2 * 240 * 240 * 284)
    assert c3["grid"] == [256, 256, 128] and d["value"] == io["end_to_end_photons_per_s"]
    for g in (io, c3):
        assert g["trace_only_photons_per_s"] > g["end_to_end_photons_per_s"] > 0 and g["not_pipelined_photons_per_s"] > 0
    assert not os.listdir(tmp_path)         # the runs' files are removed again
