#!/bin/bash
# Profiling recipe used for the summaries under profiles/ (run on the GPU box through gpurun):
#   profiles/run_profile.sh <tag>
# Pass 1: kernel trace + stats.  Passes 2-4: PMC counters, each in a run of its own (no tracing).
set -u
TAG=${1:-r1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq1 -- $BENCH > $OUT/pmc_sq1.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/pmc_tcc -- $BENCH > $OUT/pmc_tcc.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1 || exit 1
find $OUT -name "*.csv" | head -50
