cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { env $1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-latency --no-traffic --no-centroidal 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['real_problem']
print('[$1] synthetic ms/step %.3f rounds %.1f | real gait2 ms/step %.3f value %.0f sweeps %.3f'%(d['ms_per_step'], d['solver_iters']['lockstep_rounds_per_step'], r['ms_per_step'], r['value'], r['sweeps_per_step']))"; }
for i in 1 2; do
CIMPC_LIB=$PWD/contactimplicitmpc/jl_amd/libcimpc_prefix.so run "X=prefix"
run "X=head"
run "CIMPC_TAIL_DIV=16"
run "CIMPC_TAIL_DIV=32"
run "CIMPC_TAIL_DIV=0"
done > gpurun_out/knob_tail_div.log 2>&1
cat gpurun_out/knob_tail_div.log
