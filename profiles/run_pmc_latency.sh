#!/bin/bash
# Memory-latency counter passes for the bench command (pool kernel). Usage: bash profiles/run_pmc_latency.sh <tag> [bench args...]
set -u
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
BENCH="python3 bench.py --no-cpu-baseline $*"
timeout -s KILL 200 rocprofv3 --pmc VmemLatency --output-format csv -d $out/pmc_vlat -- $BENCH > $out/pmc_vlat.log 2>&1 || { echo "pmc vlat failed"; tail -5 $out/pmc_vlat.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc LdsLatency --output-format csv -d $out/pmc_llat -- $BENCH > $out/pmc_llat.log 2>&1 || { echo "pmc llat failed"; tail -5 $out/pmc_llat.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum --output-format csv -d $out/pmc_tcp -- $BENCH > $out/pmc_tcp.log 2>&1 || { echo "pmc tcp failed"; tail -5 $out/pmc_tcp.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $out/pmc_sq3 -- $BENCH > $out/pmc_sq3.log 2>&1 || { echo "pmc sq3 failed"; tail -5 $out/pmc_sq3.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $out/pmc_tcc2 -- $BENCH > $out/pmc_tcc2.log 2>&1 || { echo "pmc tcc2 failed"; tail -5 $out/pmc_tcc2.log; exit 1; }
python3 - <<PY
import csv, glob, collections
for d in ('pmc_vlat','pmc_llat','pmc_tcp','pmc_sq3','pmc_tcc2'):
    for f in glob.glob('$out/'+d+'/*/*_counter_collection.csv'):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'rt_trace_pool_kernel' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in sorted(agg.items()): print(d, k, len(v), '%.6g' % (sum(v)/len(v)))
PY
