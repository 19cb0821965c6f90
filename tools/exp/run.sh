#!/bin/bash
# run tune "one" for each experimental library
for v in "$@"; do
  echo "== $v"
  RT_TUNE_LIB=tools/exp/librt_$v.so timeout -k 10 120 python tools/tune.py c4 27 one || exit 1
  RT_TUNE_LIB=tools/exp/librt_$v.so timeout -k 10 120 python tools/tune.py c3 26 one || exit 1
done
