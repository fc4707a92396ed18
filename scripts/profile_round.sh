#!/bin/bash
# usage: scripts/profile_round.sh <tag>   (run on the GPU box through gpurun)
# kernel trace + HBM traffic counters, each in its own rocprofv3 pass (MI355X guide: PMC passes
# must not be combined with trace domains; FETCH_SIZE and WRITE_SIZE do not fit one pass).
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-real-problem"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $CMD > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $CMD > $OUT/bench_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -o sq -- $CMD > $OUT/bench_sq.log 2>&1
find $OUT -name "*.csv" | head -20
tail -1 $OUT/bench_trace.log | cut -c1-200
