#!/bin/bash
# Profiling recipe used for the summaries under profiles/ (run on the GPU box through gpurun):
#   profiles/run_profile.sh <tag> [passes...]      passes: trace sq1 sq2 tcc fetch write (default: all)
# Pass "trace": kernel trace + stats.  The others: PMC counters, each in a run of its own (no tracing).
# PROFILE_CMD names another command than the default headline bench (it must start with the program itself: python3 ...),
# e.g. PROFILE_CMD="python3 tools/render_config4.py transparent_cornell" profiles/run_profile.sh r3_c4
set -u
TAG=${1:-r1}
shift
PASSES=${@:-trace sq1 sq2 tcc fetch write}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
BENCH=${PROFILE_CMD:-"python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra"}
echo "command: $BENCH" > $OUT/command.txt
for p in $PASSES; do
  case $p in
    trace) ARGS="--kernel-trace --stats" ;;
    sq1) ARGS="--pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" ;;
    sq2) ARGS="--pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM" ;;
    tcc) ARGS="--pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" ;;
    fetch) ARGS="--pmc FETCH_SIZE" ;;
    write) ARGS="--pmc WRITE_SIZE" ;;
  esac
  timeout -s KILL 300 rocprofv3 $ARGS --output-format csv -d $OUT/$p -- $BENCH > $OUT/$p.log 2>&1 || exit 1
done
{ cat $OUT/command.txt; python3 profiles/summarize.py $OUT; } > $OUT/summary.txt
cat $OUT/summary.txt
