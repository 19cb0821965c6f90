#!/bin/bash
# L2 hit/miss counters of "tune.py <wl> <log2n> one" under given options. Usage: pmc_one.sh tag wl log2n
tag=$1; wl=$2; n=$3
export TMPDIR=/tmp
out=gpurun_out/pmc_one_$tag
mkdir -p $out
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $out -- python3 tools/tune.py $wl $n one > $out/log.txt 2>&1 || { echo fail; tail -3 $out/log.txt; exit 1; }
python3 - <<PY
import csv, glob, collections
for f in glob.glob('$out/*/*_counter_collection.csv'):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'rt_trace_pool_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print('$tag', k, len(v), '%.6g' % (sum(v)/len(v)))
PY
