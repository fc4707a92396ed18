# kernel timeline of one headline solve (rocprofv3 --kernel-trace): bash scripts/dbg/timeline.sh <tag> [bench args / env via TL_ENV]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=$1; shift
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl_$TAG && env $TL_ENV CIMPC_BENCH_NOPROF=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$TAG -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-latency --no-real-problem --no-traffic --no-centroidal "$@" > /tmp/tl_$TAG.json 2>/tmp/tl_$TAG.err )
tail -c 600 /tmp/tl_$TAG.json; echo
F=$(find /tmp/tl_$TAG -name "*kernel_trace.csv" | head -1)
python scripts/dbg/headline_timeline.py $F 2 > gpurun_out/timeline_$TAG.txt; head -3 gpurun_out/timeline_$TAG.txt
python scripts/trace_gaps.py $F | head -12
