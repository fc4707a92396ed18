cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2; do bash scripts/knob.sh "" "CIMPC_ASYNC_TAIL=48" "CIMPC_ASYNC_TAIL=64" "CIMPC_ASYNC_TAIL=128" "CIMPC_ITER_CAP=20" "CIMPC_ITER_CAP=36" "CIMPC_DRAIN_PCT=90" "CIMPC_SPEC_FIRST=1"; done > gpurun_out/knobs_r04.log 2>&1
cat gpurun_out/knobs_r04.log
bash scripts/batch_sizes.sh > gpurun_out/batch_sizes_r04.log 2>&1; cat gpurun_out/batch_sizes_r04.log
