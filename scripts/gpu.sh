# One parametrised script for the GPU box (replaces the one-shot gpu_r04_*.sh files of round 4):
#   gpurun -- 'bash scripts/gpu.sh <tag> <step> [<step> ...]'
# steps:  t5        round-5 device tests only            tests     the whole -m gpu suite
#         bench     the default bench line               ab:<KNOB>:<v0>:<v1>[:flags]   scripts/ab_env.py on one box
#         prof      scripts/profile_round.sh <tag>       k:<pytest -k expression>      a subset of the suite
#         b1stats:<KNOB>:<v>   rocprofv3 --kernel-trace --stats of 50 cold B = 1 solves under KNOB=v
# Everything lands under gpurun_out/ (<step>_<tag>.log); every step runs under its own timeout.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=$1; shift
for STEP in "$@"; do
  case "$STEP" in
    t5) timeout 900 python -m pytest tests/test_gpu_round5.py -m gpu -q -x > gpurun_out/t5_$TAG.log 2>&1; echo "t5 rc $?"
        grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/t5_$TAG.log | tail -12; grep -E "^E  " gpurun_out/t5_$TAG.log | head -30 ;;
    tests) timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/tests_$TAG.log 2>&1; echo "tests rc $?"
        grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/tests_$TAG.log | tail -20; grep -E "^E  " gpurun_out/tests_$TAG.log | head -30 ;;
    k:*) timeout 900 python -m pytest tests -m gpu -q -k "${STEP#k:}" > gpurun_out/k_$TAG.log 2>&1; echo "k rc $?"
        grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/k_$TAG.log | tail -20; grep -E "^E  " gpurun_out/k_$TAG.log | head -30 ;;
    bench) timeout 900 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc $?"; head -c 1500 gpurun_out/bench_$TAG.json; echo ;;
    ab:*) IFS=: read -r _ KNOB V0 V1 FL <<< "$STEP"
        timeout 900 python scripts/ab_env.py $KNOB $V0 $V1 $(echo $FL | tr , ' ') > gpurun_out/ab_${KNOB}_$TAG.log 2> gpurun_out/ab_${KNOB}_$TAG.err; echo "ab rc $?"
        cat gpurun_out/ab_${KNOB}_$TAG.log; tail -5 gpurun_out/ab_${KNOB}_$TAG.err ;;
    prof) bash scripts/profile_round.sh $TAG ;;
    looptrace:*) IFS=: read -r _ KNOB V WHICH <<< "$STEP"     # kernel trace (durations + gaps) of the warm B = 1 loop under KNOB=V
        ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/lt_${TAG}_$V && env $KNOB=$V rocprofv3 --kernel-trace --output-format csv -d /tmp/lt_${TAG}_$V -o p -- python $GRAFT_REPO_ROOT/scripts/loop_b1.py $WHICH 40 > /tmp/lt_${TAG}_$V.log 2>&1 )
        F=$(find /tmp/lt_${TAG}_$V -name "*kernel_trace.csv" | head -1); grep ms_per /tmp/lt_${TAG}_$V.log | tail -1
        python scripts/trace_gaps.py $F | tee gpurun_out/loop_trace_${WHICH}_${KNOB}_${V}_$TAG.txt ;;
    b1stats:*) IFS=: read -r _ KNOB V <<< "$STEP"     # rocprofv3 kernel statistics of the B = 1 cold solve under KNOB=V
        ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/b1_$TAG_$V && env $KNOB=$V rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/b1_${TAG}_$V -o p -- python $GRAFT_REPO_ROOT/scripts/b1_cold.py 50 > /tmp/b1_${TAG}_$V.log 2>&1 )
        F=$(find /tmp/b1_${TAG}_$V -name "*kernel_stats.csv" | head -1); cp $F gpurun_out/kernel_stats_b1_${KNOB}_${V}_$TAG.csv; tail -2 /tmp/b1_${TAG}_$V.log
        python - <<PY
import csv
for r in list(csv.DictReader(open("$F")))[:10]:
    print("%-70s calls %6s  avg %9.1f us  total %8.2f ms  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
PY
        ;;
    *) echo "unknown step $STEP" ;;
  esac
done
