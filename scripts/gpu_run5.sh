cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/contactimplicitmpc/jl_amd
bash scripts/sweep_knobs.sh CIMPC_LIB=$L/libcimpc_v2.so CIMPC_LIB=$L/libcimpc_v3.so CIMPC_LIB=$L/libcimpc_v4.so 2>&1 | tee gpurun_out/sweep_r02g.log
python bench.py --steps 5 --warmup 2 --rollouts 2048 --no-cpu-baseline --no-real-problem --no-latency --no-traffic | python -c "import json,sys; o=json.loads(sys.stdin.read()); print('B=2048', round(o['value']), round(o['ms_per_step'],2))"
