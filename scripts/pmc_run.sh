#!/bin/bash
# usage (GPU box): bash scripts/pmc_run.sh <name> <command ...>   - rocprofv3 kernel statistics + two SQ counter passes + FETCH / WRITE of ANY
# command (each in its own pass: PMC passes are never combined with trace domains), summary -> gpurun_out/pmc_<name>/
NAME=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$NAME
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- "$@" > $OUT/run_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_sq -o sq -- "$@" > $OUT/run_sq.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o sq2 -- "$@" > $OUT/run_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- "$@" > $OUT/run_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- "$@" > $OUT/run_write.log 2>&1
python scripts/pmc_summary.py $OUT > $OUT/pmc_summary.csv 2> $OUT/pmc_summary.err
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/trace $OUT/pmc_sq $OUT/pmc_sq2 $OUT/pmc_fetch $OUT/pmc_write
head -8 $OUT/kernel_stats.csv | cut -c1-160
