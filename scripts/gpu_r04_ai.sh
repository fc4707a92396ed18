# rocprofv3 view of the velocity-objective leg (configs[4] with the example's own objective): kernel trace + SQ counters of kkt_banded_kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/prof_vel
CMD="python scripts/dbg/vel_leg.py 64"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_vel/trace -o trace -- $CMD > gpurun_out/prof_vel/trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d gpurun_out/prof_vel/pmc_sq -o sq -- $CMD > gpurun_out/prof_vel/sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/prof_vel/pmc_sq2 -o sq2 -- $CMD > gpurun_out/prof_vel/sq2.log 2>&1
f=$(find gpurun_out/prof_vel/trace -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-220
python - <<'PY'
import csv, glob, collections
for tag in ("pmc_sq", "pmc_sq2"):
    fs = glob.glob("gpurun_out/prof_vel/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs: print(tag, "no file"); continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(fs[0])):
        if "kkt_banded" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in sorted(acc.items()): print("%-32s per launch %.4g  (launch records %d)" % (k, v / max(n, 1), n))
PY
for B in 1 64 256; do timeout 300 python scripts/dbg/vel_leg.py $B 2>/dev/null | tail -1; done
