#!/bin/bash
# usage: scripts/profile_round.sh <tag> [extra bench args]   (run on the GPU box through gpurun)
# kernel trace + HBM traffic counters + SQ counters, each in its own rocprofv3 pass (MI355X guide: PMC passes
# must not be combined with trace domains; FETCH_SIZE and WRITE_SIZE do not fit one pass; 8 SQ slots per pass).
TAG=${1:-r02}
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-real-problem --no-latency --no-centroidal $*"
rocprofv3 -L > $OUT/counters_available.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $CMD > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $CMD > $OUT/bench_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_sq -o sq -- $CMD > $OUT/bench_sq.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o sq2 -- $CMD > $OUT/bench_sq2.log 2>&1
python scripts/pmc_summary.py $OUT > $OUT/pmc_summary.csv 2> $OUT/pmc_summary.err
find $OUT -name "*stats*.csv" | head -5
# the bench line of the PROFILED run (the log also holds rocprofv3's own messages: take the JSON line, not the last line)
grep -E '^\{"metric"' $OUT/bench_trace.log | tail -1 > $OUT/bench_under_rocprof.json
cut -c1-300 $OUT/bench_under_rocprof.json
