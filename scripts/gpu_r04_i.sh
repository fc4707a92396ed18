cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for r in 1 2; do for E in "X=1" "CIMPC_KKT_PIPE=1" "CIMPC_ASYNC=0" "CIMPC_ASYNC=0 CIMPC_KKT_PIPE=1"; do TAG="$E" env $E python scripts/cent_knob.py 64 2>/dev/null | tail -1; done; done > gpurun_out/cent_kkt_pipe.log 2>&1
for E in "X=1" "CIMPC_KKT_PIPE=1"; do TAG="$E" env $E python scripts/cent_knob.py 128 2>/dev/null | tail -1; TAG="$E" env $E python scripts/cent_knob.py 16 2>/dev/null | tail -1; done >> gpurun_out/cent_kkt_pipe.log 2>&1
cat gpurun_out/cent_kkt_pipe.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_r04_cent2; mkdir -p $OUT
CIMPC_ASYNC=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python scripts/cent_knob.py 64 > $OUT/trace.log 2>&1
CIMPC_ASYNC=0 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc1 -o p -- python scripts/cent_knob.py 64 > /dev/null 2>&1
CIMPC_ASYNC=0 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc2 -o p -- python scripts/cent_knob.py 64 > /dev/null 2>&1
python scripts/pmc_summary.py $OUT > $OUT/pmc_summary.csv 2> $OUT/pmc_summary.err
grep "ip_queue" $OUT/pmc_summary.csv | cut -c50-
grep -E "ip_queue|kkt" $OUT/trace/*kernel_stats.csv | cut -c1-220
