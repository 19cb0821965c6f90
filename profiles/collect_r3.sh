#!/bin/bash
# Copies the round-3 measurement set from gpurun_out/<tag>/ (profiles/run_r3.sh) into profiles/ under committed names.
set -eu
cd "$(dirname "$0")/.."
tag=${1:-r3}
for wl in c1 c2 c3 c4 c5; do cp gpurun_out/$tag/bench_$wl.json profiles/${tag}_bench_$wl.json; done
cp gpurun_out/$tag/bench_c4_force_dist.json profiles/${tag}_bench_c4_force_dist.json
for wl in c4 c3 c1 c2; do
  k=5; t=0; n=268435456
  [ $wl = c4 ] && k=8 && t=5
  [ $wl = c1 ] && n=67108864
  [ $wl = c2 ] && n=16777216
  python3 profiles/summarize.py ${tag}_$wl $wl $n $k $t > /dev/null
done
ls profiles | grep "^${tag}_"
