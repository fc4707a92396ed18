cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-real-problem --no-latency --no-traffic --no-centroidal"
for W in 4 2 1; do
  OUT=gpurun_out/prof_waves$W
  mkdir -p $OUT
  CIMPC_WAVES=$W rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc -o sq -- $CMD > $OUT/bench.log 2>&1
  python scripts/pmc_summary.py $OUT | grep ip_queue | cut -c60- | tr '\n' ' '; echo " WAVES=$W"; tail -1 $OUT/bench.log | python -c "import json,sys; o=json.loads(sys.stdin.read()); print('ms', o['ms_per_step'], 'launch', o['roofline']['avg_launch_ms'])"
done
OUT=gpurun_out/prof_extra
mkdir -p $OUT
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc -o sq -- $CMD > $OUT/bench.log 2>&1
python scripts/pmc_summary.py $OUT | grep ip_queue | cut -c60-
