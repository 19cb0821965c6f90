#!/bin/bash
# Per-kernel times (rocprofv3 --kernel-trace --stats) of the bench command with the given extra arguments.
# usage (on the GPU box): tools/trace_kernels.sh <tag> [bench args...]; prints the kernel statistics table
tag=$1; shift
out=gpurun_out/kt_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --no-cpu-baseline "$@" > $out/log.txt 2>&1 || { echo "trace failed"; tail -5 $out/log.txt; exit 1; }
f=$(find $out -name '*kernel_stats.csv' | head -1)
cut -d, -f1-6 $f | cut -c1-150
