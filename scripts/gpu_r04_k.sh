cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{ for E in "X=1" "CIMPC_SWEEP_WGS=256" "CIMPC_SWEEP_WGS=192" "CIMPC_SWEEP_WGS=320"; do echo "== $E"; env $E python scripts/exp_subbatch.py 1 2 2>&1 | grep "sub-batches"; done; echo "== CIMPC_SWEEP_WGS=128 k=4"; CIMPC_SWEEP_WGS=128 python scripts/exp_subbatch.py 4 2>&1 | grep sub-batches; } > gpurun_out/exp_subbatch_r04.log 2>&1
cat gpurun_out/exp_subbatch_r04.log
