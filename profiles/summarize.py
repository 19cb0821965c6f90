#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (written by profiles/run_rocprof.sh) into
profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats) and profiles/<tag>_pmc.json
(per-launch counter values of rt_trace_kernel), and refresh profiles/pmc_<workload>.json that
bench.py reads for roofline.traffic.

HBM traffic per launch = (FETCH_SIZE + WRITE_SIZE) * 1024 bytes (rocprofv3 reports KiB).
MI355X_MICROARCH.md §HBM: FETCH_SIZE under-reports WIDE COALESCED streaming reads by 2x on gfx950
and is uncalibrated for other widths. This kernel's reads are 4/8-byte random gathers; calibration
on its own pattern: TCC_MISS_sum * 64 B agrees with FETCH_SIZE * 1024 (one 64-B request per gather
miss), so no doubling is applied here."""
import collections, csv, glob, json, os, shutil, sys

tag, workload, photons = sys.argv[1], sys.argv[2], int(sys.argv[3])
launch_log2k = int(sys.argv[4]) if len(sys.argv) > 4 else -1     # bench.py --launch-log2k of the profiled run
src = f"gpurun_out/prof_{tag}"
here = os.path.dirname(os.path.abspath(__file__))
newest = lambda pattern: max(glob.glob(pattern), key=os.path.getmtime)     # gpurun_out/ keeps earlier runs' files beside the last one's
shutil.copy(newest(f"{src}/trace/*/*_kernel_stats.csv"), f"{here}/{tag}_kernel_stats.csv")
pmc = {}
for d in sorted(glob.glob(f"{src}/pmc_*/")):
    for f in [newest(d + "*/*_counter_collection.csv")]:
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if ("rt_trace_kernel<false, false>" in kn) or ("rt_trace_pool_kernel<0, false" in kn) or ("rt_trace_pool_kernel<false, false" in kn):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                pmc["grid"] = int(r["Grid_Size"])      # (register counts: see the compiler's remark in *_resource_usage.txt)
        for k, v in agg.items():
            pmc[k] = sum(v) / len(v)
stats = [r for r in csv.DictReader(open(f"{here}/{tag}_kernel_stats.csv")) if ("rt_trace_kernel<false, false>" in r["Name"]) or ("rt_trace_pool_kernel<0, false" in r["Name"]) or ("rt_trace_pool_kernel<false, false" in r["Name"])]
pmc["kernel_avg_ns"] = float(stats[0]["AverageNs"]); pmc["kernel_calls"] = int(stats[0]["Calls"])
pmc["photons_per_launch"] = photons
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    pmc["hbm_bytes_per_launch"] = (pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024
if "SQ_THREAD_CYCLES_VALU" in pmc and "SQ_INSTS_VALU" in pmc:
    pmc["active_lanes_per_valu_inst"] = pmc["SQ_THREAD_CYCLES_VALU"] / pmc["SQ_INSTS_VALU"]
    pmc["valu_wave_insts_per_photon"] = pmc["SQ_INSTS_VALU"] / photons
if "SQ_INSTS_VALU_TRANS_F32" in pmc and "SQ_INSTS_VALU_INT64" in pmc and "SQ_INSTS_VALU" in pmc:
    # issue cycles of the VALU work as tools/valu_rate.hip measures them (profiles/r1_microbench_valu_rate.txt): a wave64
    # instruction holds its SIMD 2 cycles, a 64-bit integer one (Philox's v_mad_u64_u32) 4, a transcendental (v_rcp_f32,
    # v_sqrt_f32) 8
    trans, i64 = pmc["SQ_INSTS_VALU_TRANS_F32"], pmc["SQ_INSTS_VALU_INT64"]
    pmc["valu_trans_insts_per_photon"] = trans / photons
    pmc["valu_int64_insts_per_photon"] = i64 / photons
    pmc["valu_issue_cycles_per_photon"] = (2.0 * (pmc["SQ_INSTS_VALU"] - trans - i64) + 4.0 * i64 + 8.0 * trans) / photons
pmc["launch_log2k"] = launch_log2k
pmc["launch_tile_log2"] = int(sys.argv[5]) if len(sys.argv) > 5 else 0     # bench.py --launch-tile-log2 of the profiled run
if "FETCH_SIZE" in pmc:
    pmc["fabric_reads_per_photon"] = pmc["FETCH_SIZE"] * 1024 / 64 / photons      # 64-byte read requests
if "TCC_EA0_ATOMIC_sum" in pmc:
    pmc["atomics_per_photon"] = pmc["TCC_EA0_ATOMIC_sum"] / photons
if "SQ_INSTS_SALU" in pmc:
    pmc["salu_insts_per_photon"] = pmc["SQ_INSTS_SALU"] / photons
if "SQ_INSTS_LDS" in pmc and "SQ_LDS_BANK_CONFLICT" in pmc:
    pmc["lds_insts_per_photon"] = pmc["SQ_INSTS_LDS"] / photons
    pmc["lds_conflict_cycles_per_lds_inst"] = pmc["SQ_LDS_BANK_CONFLICT"] / pmc["SQ_INSTS_LDS"]
json.dump(pmc, open(f"{here}/{tag}_pmc.json", "w"), indent=1, sort_keys=True)
keys = ("photons_per_launch", "launch_log2k", "launch_tile_log2", "hbm_bytes_per_launch", "fabric_reads_per_photon", "atomics_per_photon",
        "valu_wave_insts_per_photon", "valu_issue_cycles_per_photon", "valu_trans_insts_per_photon", "valu_int64_insts_per_photon", "active_lanes_per_valu_inst", "salu_insts_per_photon", "lds_insts_per_photon",
        "lds_conflict_cycles_per_lds_inst", "kernel_avg_ns")
json.dump(dict({k: pmc.get(k) for k in keys}, source=f"profiles/{tag}_pmc.json"),
          open(f"{here}/pmc_{workload}.json", "w"), indent=1)
print(json.dumps(pmc, indent=1, sort_keys=True))
