#!/bin/bash
# SQ counter passes for the bench command. Usage: bash profiles/run_pmc_sq.sh <tag> [bench args...]
set -u
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
BENCH="python3 bench.py --no-cpu-baseline $*"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $out/pmc_sq -- $BENCH > $out/pmc_sq.log 2>&1 || { echo "pmc sq failed"; tail -5 $out/pmc_sq.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES --output-format csv -d $out/pmc_sq2 -- $BENCH > $out/pmc_sq2.log 2>&1 || { echo "pmc sq2 failed"; tail -5 $out/pmc_sq2.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum --output-format csv -d $out/pmc_tcc -- $BENCH > $out/pmc_tcc.log 2>&1 || { echo "pmc tcc failed"; tail -5 $out/pmc_tcc.log; exit 1; }
python3 - <<PY
import csv, glob, collections
for d in ('pmc_sq','pmc_sq2','pmc_tcc'):
    for f in glob.glob('$out/'+d+'/*/*_counter_collection.csv'):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'rt_trace_pool_kernel<0, false' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in sorted(agg.items()): print(d, k, len(v), '%.4g' % (sum(v)/len(v)))
PY
