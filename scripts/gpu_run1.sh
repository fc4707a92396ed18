cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q > gpurun_out/tests_r02a.log 2>&1; echo "tests rc $?"
tail -30 gpurun_out/tests_r02a.log
timeout 400 python bench.py > gpurun_out/bench_r02a.json 2> gpurun_out/bench_r02a.err; echo "bench rc $?"; cut -c1-1500 gpurun_out/bench_r02a.json
CIMPC_BENCH_BACKEND=gloo timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --rollouts 128 > gpurun_out/bench_2rank_weak.json 2> gpurun_out/bench_2rank_weak.err; echo "2rank weak rc $?"; cut -c1-600 gpurun_out/bench_2rank_weak.json
CIMPC_BENCH_BACKEND=gloo timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --rollouts 128 --scaling strong > gpurun_out/bench_2rank_strong.json 2> gpurun_out/bench_2rank_strong.err; echo "2rank strong rc $?"; cut -c1-600 gpurun_out/bench_2rank_strong.json
bash scripts/profile_round.sh r02_before --no-traffic
