# usage (GPU box): bash scripts/gpu_round.sh <tag>  - the round's evidence in one call: GPU tests, the default bench line,
# the rocprofv3 passes of scripts/profile_round.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r02}
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/tests_$TAG.log 2>&1; echo "tests rc $?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/tests_$TAG.log | tail -12
grep -E "^E  " gpurun_out/tests_$TAG.log | head -20
timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc $?"
python - <<PY
import json
o=json.load(open('gpurun_out/bench_$TAG.json'))
print('value', o['value'], 'ms', o['ms_per_step'], 'roofline', {k:o['roofline'][k] for k in ('bound','achieved','frac','traffic','avg_launch_ms')}, 'hbm', o['roofline']['hbm'])
print(o['kernel_time_ms_per_step']); print(o.get('latency_b1')); print({k:v['mpc_steps_per_s'] for k,v in o.get('mpc_loop_b1',{}).items()}); print(o.get('real_problem',{}).get('value'))
print(json.dumps(o.get('centroidal_payload_h60'))[:1500]); print(o.get('cpu_baseline',{}).get('value'), o.get('speedup_vs_cpu_1thread'))
PY
bash scripts/profile_round.sh $TAG
