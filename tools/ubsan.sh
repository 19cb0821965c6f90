#!/bin/bash
# Undefined-behaviour passes where a sanitizer can run (GPU-side sanitizers are not available on this pool).
#   tools/ubsan.sh cpu        build oracle/liboracle_ubsan.so (make -C oracle ubsan) and run the CPU test suites on it:
#                             the three checker sources under -fsanitize=undefined,float-cast-overflow,bounds, any
#                             report aborts the run
#   tools/ubsan.sh build      build raytracer_amd/librt_photon_ubsan.so: the library's HOST code under
#                             -fsanitize=undefined,bounds,float-cast-overflow with traps (device code unchanged)
#   tools/ubsan.sh gpu        run the GPU suite through that library (on the GPU box:
#                             gpurun -- 'bash tools/ubsan.sh gpu'); a trap ends the run with SIGILL
set -eu
cd "$(dirname "$0")/.."
case "${1:-cpu}" in
  cpu)
    make -C oracle -s ubsan
    RT_ORACLE_LIB=$PWD/oracle/liboracle_ubsan.so UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
      python -m pytest tests -q -m "not gpu" -x -k "oracle or golden or scenes or distributed or launch_order"
    ;;
  build)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -shared -fPIC \
      -Xarch_host -fsanitize=undefined,bounds,float-cast-overflow -Xarch_host -fsanitize-trap=all \
      raytracer_amd/csrc/rt_photon.hip -o raytracer_amd/librt_photon_ubsan.so
    ;;
  gpu)
    RT_PHOTON_LIB=$PWD/raytracer_amd/librt_photon_ubsan.so timeout -k 10 1100 \
      python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread \
      --deselect tests/test_gpu_drivers.py --deselect tests/test_gpu_bench_contract.py
    ;;
  *) echo "usage: tools/ubsan.sh cpu|build|gpu"; exit 2;;
esac
