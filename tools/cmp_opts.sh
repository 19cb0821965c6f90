#!/bin/bash
# Kernel time of tools/tune.py "one" for a list of "lib:opt=val,opt=val" pairs. usage: tools/cmp_opts.sh <workload> <log2 photons> lib.so:opts ...
wl=$1; n=$2; shift 2
for spec in "$@"; do
  l=${spec%%:*}; o=${spec#*:}
  echo -n "$spec: "
  RT_TUNE_LIB=tools/exp/$l RT_TUNE_OPTS=$o timeout -k 10 120 python tools/tune.py $wl $n one 2>&1 | grep " one " || exit 1
done
