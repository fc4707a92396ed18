# round 4, first GPU call: full GPU suite, A/B of the round-3 library against HEAD, default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/tests_r04a.log 2>&1; echo "tests rc $?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/tests_r04a.log | tail -12
grep -E "^E  " gpurun_out/tests_r04a.log | head -30
bash scripts/ab.sh contactimplicitmpc/jl_amd/libcimpc_r03.so contactimplicitmpc/jl_amd/libcimpc_hip.so 3 > gpurun_out/ab_hoist.log 2>&1
cat gpurun_out/ab_hoist.log
timeout 600 python bench.py > gpurun_out/bench_r04a.json 2> gpurun_out/bench_r04a.err; echo "bench rc $?"
python - <<PY
import json
o=json.load(open('gpurun_out/bench_r04a.json'))
print('value', o['value'], 'ms', o['ms_per_step'], 'roofline', {k:o['roofline'][k] for k in ('bound','achieved','frac','step_frac','traffic','avg_launch_ms','units_per_launch')})
print(o['kernel_time_ms_per_step']); print(o.get('latency_b1')); print({k:v['mpc_steps_per_s'] for k,v in o.get('mpc_loop_b1',{}).items()}); print(o.get('headline'))
print(json.dumps(o.get('centroidal_payload_h60'))[:1200]); print(o.get('cpu_baseline',{}).get('value'), o.get('speedup_vs_cpu_1thread'))
PY
