#!/bin/bash
# Profiling recipe for the bench command (run on the GPU box through gpurun).
# Usage: bash profiles/run_rocprof.sh <tag> [bench args...]; writes under gpurun_out/prof_<tag>/
set -u
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
BENCH="python3 bench.py --no-cpu-baseline $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $BENCH > $out/trace.log 2>&1 || { echo "trace failed"; tail -5 $out/trace.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $BENCH > $out/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; tail -5 $out/pmc_fetch.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $BENCH > $out/pmc_write.log 2>&1 || { echo "pmc write failed"; tail -5 $out/pmc_write.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $out/pmc_sq -- $BENCH > $out/pmc_sq.log 2>&1 || { echo "pmc sq failed"; tail -5 $out/pmc_sq.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY --output-format csv -d $out/pmc_sq2 -- $BENCH > $out/pmc_sq2.log 2>&1 || { echo "pmc sq2 failed"; tail -5 $out/pmc_sq2.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_sum --output-format csv -d $out/pmc_tcc -- $BENCH > $out/pmc_tcc.log 2>&1 || { echo "pmc tcc failed"; tail -5 $out/pmc_tcc.log; exit 1; }
# instruction mix of the VALU work (which instructions hold the SIMD longer than the two cycles of a plain one)
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT --output-format csv -d $out/pmc_mix -- $BENCH > $out/pmc_mix.log 2>&1 || { echo "pmc mix failed"; tail -5 $out/pmc_mix.log; exit 1; }
find $out -name '*.csv' | head -40
