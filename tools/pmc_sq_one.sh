#!/bin/bash
# SQ instruction counters of "tune.py <wl> <log2n> one" for one library build. Usage: pmc_sq_one.sh tag wl log2n [lib]
# (lib under tools/exp/; options through RT_TUNE_OPTS, e.g. launch_log2k=5). Prints per-launch averages.
tag=$1; wl=$2; n=$3; lib=${4:-}
export TMPDIR=/tmp
[ -n "$lib" ] && export RT_TUNE_LIB=tools/exp/$lib
out=gpurun_out/pmc_sq_$tag
mkdir -p $out
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM --output-format csv -d $out -- python3 tools/tune.py $wl $n one > $out/log.txt 2>&1 || { echo fail; tail -3 $out/log.txt; exit 1; }
python3 - <<PY
import csv, glob, collections
for f in glob.glob('$out/*/*_counter_collection.csv'):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'rt_trace_pool_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    n = float(1 << $n)
    for k,v in sorted(agg.items()): print('$tag', k, len(v), '%.6g' % (sum(v)/len(v)), 'per photon %.4g' % (sum(v)/len(v)/n))
PY
