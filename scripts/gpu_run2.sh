cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r02b}
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/tests_$TAG.log 2>&1; echo "tests rc $?"
tail -15 gpurun_out/tests_$TAG.log
timeout 400 python bench.py --no-traffic > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc $?"
python - <<PY
import json
o=json.load(open('gpurun_out/bench_$TAG.json'))
print('value', o['value'], 'ms', o['ms_per_step'], 'frac', o['roofline']['frac'], 'launch_ms', o['roofline']['avg_launch_ms'])
print(o['kernel_time_ms_per_step']); print(o['solver_iters']); print(o.get('latency_b1')); print({k:v['mpc_steps_per_s'] for k,v in o.get('mpc_loop_b1',{}).items()}); print(o.get('real_problem',{}).get('value'))
PY
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-real-problem --no-latency --no-traffic --no-centroidal"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_sq -o sq -- $CMD > $OUT/bench_sq.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o sq2 -- $CMD > $OUT/bench_sq2.log 2>&1
python scripts/pmc_summary.py $OUT > $OUT/pmc_summary.csv 2> $OUT/pmc_summary.err
grep ip_queue $OUT/pmc_summary.csv
cut -c1-160 $OUT/trace/trace_kernel_stats.csv | head -6
