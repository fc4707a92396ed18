cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for B in 64 128 256; do
  TAG=prev CIMPC_ASYNC=0 CIMPC_LIB=$PWD/contactimplicitmpc/jl_amd/libcimpc_prev.so python scripts/cent_knob.py $B 2>/dev/null | tail -1
  TAG=head CIMPC_ASYNC=0 python scripts/cent_knob.py $B 2>/dev/null | tail -1
done > gpurun_out/cent_two_builds.log 2>&1
TAG=head_default_sched python scripts/cent_knob.py 64 2>/dev/null | tail -1 >> gpurun_out/cent_two_builds.log
TAG=head_default_sched python scripts/cent_knob.py 128 2>/dev/null | tail -1 >> gpurun_out/cent_two_builds.log
cat gpurun_out/cent_two_builds.log
timeout 900 python -m pytest tests -m gpu -q -k "centroidal or config4 or mixed or plant or callback or round3" > gpurun_out/tests_r04r.log 2>&1; echo "tests rc $?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/tests_r04r.log | tail -4
grep -E "^E  " gpurun_out/tests_r04r.log | head -10
