# round 4: 32-lane sweep with LDS-staged broadcasts - parity tests of the centroidal model, timing against the previous build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -k "centroidal or config4 or mixed or plant" > gpurun_out/tests_r04h.log 2>&1; echo "tests rc $?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/tests_r04h.log | tail -8
grep -E "^E  " gpurun_out/tests_r04h.log | head -20
for r in 1 2; do for L in libcimpc_ilp2.so libcimpc_hip.so; do TAG=$L CIMPC_ASYNC=0 CIMPC_LIB=$PWD/contactimplicitmpc/jl_amd/$L python scripts/cent_knob.py 64 2>/dev/null | tail -1; done; done > gpurun_out/cent_stage.log 2>&1
for B in 8 32 128; do for L in libcimpc_ilp2.so libcimpc_hip.so; do TAG=$L CIMPC_LIB=$PWD/contactimplicitmpc/jl_amd/$L python scripts/cent_knob.py $B 2>/dev/null | tail -1; done; done >> gpurun_out/cent_stage.log 2>&1
cat gpurun_out/cent_stage.log
