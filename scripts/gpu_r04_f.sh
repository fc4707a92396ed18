cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python scripts/dbg/relin_debug.py 2>&1 | grep "^step" > gpurun_out/relin_debug.log
python scripts/dbg/relin_debug.py norelin 2>&1 | grep "^step" | sed 's/^/NORELIN /' >> gpurun_out/relin_debug.log
CIMPC_LIB=$PWD/contactimplicitmpc/jl_amd/libcimpc_r03.so python scripts/dbg/relin_debug.py norelin 2>&1 | grep "^step" | sed 's/^/R03-NORELIN /' >> gpurun_out/relin_debug.log
cat gpurun_out/relin_debug.log
