cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for B in 64 128 256; do
  TAG=prev CIMPC_ASYNC=0 CIMPC_LIB=$PWD/contactimplicitmpc/jl_amd/libcimpc_prev.so python scripts/cent_knob.py $B 2>/dev/null | tail -1
  TAG=window_w4 CIMPC_ASYNC=0 python scripts/cent_knob.py $B 2>/dev/null | tail -1
  TAG=window_w8 CIMPC_ASYNC=0 CIMPC_WAVES32=8 python scripts/cent_knob.py $B 2>/dev/null | tail -1
done > gpurun_out/cent_w8b.log 2>&1
TAG=prev_async32 CIMPC_LIB=$PWD/contactimplicitmpc/jl_amd/libcimpc_prev.so python scripts/cent_knob.py 32 2>/dev/null | tail -1 >> gpurun_out/cent_w8b.log
TAG=window_async32 python scripts/cent_knob.py 32 2>/dev/null | tail -1 >> gpurun_out/cent_w8b.log
cat gpurun_out/cent_w8b.log
CIMPC_WAVES32=8 timeout 600 python -m pytest tests -m gpu -q -k "centroidal or config4" > gpurun_out/tests_r04q.log 2>&1; echo "tests rc $?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/tests_r04q.log | tail -4
