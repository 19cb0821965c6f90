#!/bin/bash
# L2 (TCC) hit / miss / request counts of the tracer for one launch stratification k (bench.py --launch-log2k k):
# shows how little the kernel time follows the miss count (k = 3 -> 5 -> 8: misses 1.91 -> 1.52 -> 1.34 G per 2^28
# photons on C4, 5.28 -> 5.36 -> 5.33 G photons/s). Run on the GPU box: bash tools/pmc_launch_k.sh 5
# usage: pmc_k.sh <k>
export TMPDIR=/tmp
k=$1
out=gpurun_out/pmc_k$k
mkdir -p $out
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --launch-log2k $k > $out.log 2>&1 || { tail -3 $out.log; exit 1; }
python3 - <<PY
import csv, glob, collections
for f in glob.glob('$out/*/*_counter_collection.csv'):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'rt_trace_pool_kernel<false, false' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print('k=$k', {k: '%.4g' % (sum(v)/len(v)) for k, v in sorted(agg.items())})
PY
tail -1 $out.log | cut -c1-90
