# round 4: centroidal sweep (BASELINE configs[4]) - sensitivity interleave width, per-launch trace, PMC counters; re-linearisation tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_round4.py -m gpu -q -k relinearisation > gpurun_out/tests_r04g.log 2>&1; echo "tests rc $?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/tests_r04g.log | tail -5
grep -E "^E  " gpurun_out/tests_r04g.log | head -12
for r in 1 2; do for L in libcimpc_hip.so libcimpc_ilp3.so libcimpc_ilp4.so; do TAG=$L CIMPC_ASYNC=0 CIMPC_LIB=$PWD/contactimplicitmpc/jl_amd/$L python scripts/cent_knob.py 64 2>/dev/null | tail -1; done; done > gpurun_out/cent_ilp.log 2>&1
cat gpurun_out/cent_ilp.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_r04_cent; mkdir -p $OUT
CIMPC_ASYNC=0 CIMPC_DEBUG_ROUNDS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python scripts/cent_knob.py 64 > $OUT/trace.log 2> $OUT/rounds.log
CIMPC_ASYNC=0 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc1 -o p -- python scripts/cent_knob.py 64 > /dev/null 2>&1
CIMPC_ASYNC=0 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc2 -o p -- python scripts/cent_knob.py 64 > /dev/null 2>&1
python scripts/pmc_summary.py $OUT > $OUT/pmc_summary.csv 2> $OUT/pmc_summary.err
grep "ip_queue" $OUT/pmc_summary.csv | cut -c50-
grep -c "sweep launch" $OUT/rounds.log; grep "ip_queue" $OUT/trace/*kernel_stats.csv | cut -c1-200
