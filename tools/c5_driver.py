#!/usr/bin/env python3
"""Config C5 through the C++ drop-in driver (raytracer_3d_io_gpu) on the io flow's own 240 x 240 x 284 grid: a sweep of
zenith angles with the full output per run, for --pipeline = 0 (the reference's order), 2, 4 (jobs' files in flight).
Prints the drivers' JSON summaries. Usage, on the GPU box: python tools/c5_driver.py [log2 photons = 26] [angles = 12]"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
nang = int(sys.argv[2]) if len(sys.argv) > 2 else 12
sc = rt.scenes.cumulus(240, 240, 284, 100.0, 100.0, 30.0, n_clouds=60, seed=7, z_base=900.0, z_top=3000.0, name="io_default_grid")
zens = ",".join(f"{75.0 * i / max(nang - 1, 1):.3f}" for i in range(nang))
for mode in ("0", "2", "4", "6"):
    with tempfile.TemporaryDirectory() as d:
        sc.write(d)
        r = subprocess.run([os.path.join(ROOT, "drivers", "raytracer_3d_io_gpu"), str(log2n), "--pipeline=" + mode, "--zenith=" + zens],
                           cwd=d, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            print(r.stderr[-2000:]); sys.exit(1)
        s = json.loads(r.stdout.splitlines()[-1])
        print(json.dumps({"pipeline": mode, "jobs_in_flight": s["jobs_in_flight"], "ms_per_job": s["sweep_seconds"] / s["jobs"] * 1e3,
                          "end_to_end_G_photons_per_s": s["end_to_end_photons_per_s"] / 1e9}), flush=True)
