#!/bin/bash
# Round-3 measurement set (run on the GPU box: gpurun -- 'bash profiles/run_r3.sh'): bench lines of the four single-GPU
# workloads, the C5 sweep, the rehearsed multi-GPU step, and the rocprofv3 passes (kernel trace + PMC) of each workload's
# bench command. Results land under gpurun_out/; profiles/collect_r3.sh copies what is committed.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=${1:-r3}
out=gpurun_out/$tag
mkdir -p $out
for wl in c4 c3 c1 c2; do
  extra="--no-cpu-baseline"; [ $wl = c4 ] && extra=""
  timeout -k 10 500 python3 bench.py --workload $wl --steps 5 --warmup 2 $extra > $out/bench_$wl.json 2> $out/bench_$wl.err || { echo "bench $wl failed"; tail -3 $out/bench_$wl.err; exit 1; }
  echo "bench $wl done"
done
timeout -k 10 500 python3 bench.py --workload c5 > $out/bench_c5.json 2> $out/bench_c5.err || { echo "bench c5 failed"; tail -3 $out/bench_c5.err; exit 1; }
echo "bench c5 done"
timeout -k 10 300 python3 bench.py --force-dist --no-cpu-baseline --no-uniform-leg --steps 5 --warmup 2 > $out/bench_c4_force_dist.json 2> $out/bench_c4_force_dist.err || { echo "force-dist failed"; exit 1; }
echo "force-dist done"
for wl in c4 c3 c1 c2; do
  bash profiles/run_rocprof.sh ${tag}_$wl --workload $wl --no-uniform-leg > $out/prof_$wl.log 2>&1 || { echo "rocprof $wl failed"; tail -5 $out/prof_$wl.log; exit 1; }
  echo "rocprof $wl done"
done
