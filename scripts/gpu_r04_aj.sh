# HBM traffic of kkt_banded_kernel (velocity-objective leg, 64 rollouts): FETCH_SIZE and WRITE_SIZE in separate passes (MI355X guide)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/prof_vel
CMD="python scripts/dbg/vel_leg.py 64"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_vel/pmc_fetch -o fetch -- $CMD > gpurun_out/prof_vel/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_vel/pmc_write -o write -- $CMD > gpurun_out/prof_vel/write.log 2>&1
python - <<'PY'
import csv, glob, collections
for tag in ("pmc_fetch", "pmc_write"):
    fs = glob.glob("gpurun_out/prof_vel/%s/**/*counter_collection.csv" % tag, recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(fs[0])):
        k = "kkt_banded" if "kkt_banded" in r["Kernel_Name"] else ("ip_queue" if "ip_queue" in r["Kernel_Name"] else None)
        if k:
            a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k in acc:
        for c, (v, n) in acc[k].items(): print("%-12s %-12s per launch %.5g  (records %d)" % (k, c, v / max(n, 1), n))
PY
