#!/bin/bash
# Compare experimental builds of the library on one GPU (kernel time of tools/tune.py "one").
# usage: tools/cmp_libs.sh <workload> <log2 photons> lib1.so lib2.so ...   (libraries under tools/variants/, which is
# git-ignored but travels with gpurun); options for every run through RT_TUNE_OPTS, e.g. launch_log2k=5
wl=$1; n=$2; shift 2
for l in "$@"; do
  echo -n "$l: "
  RT_TUNE_LIB=${RT_LIB_DIR:-tools/variants}/$l timeout -k 10 120 python tools/tune.py $wl $n one 2>&1 | grep " one " || exit 1
done
