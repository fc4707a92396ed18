"""Timeline of ONE solve from a rocprofv3 --kernel-trace csv: every kernel between two reset launches in start order with its
start offset, duration and queue, then per lock-step round (= one decision launch) the span of its sweep / KKT / decision kernels.
usage: python scripts/dbg/headline_timeline.py <kernel_trace.csv> [which-solve-from-the-end, default 2]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 2


def short(n):
    n = n.split("(")[0]
    for k in ("ip_queue_kernel", "kkt_kernel_packed", "kkt_kernel_twisted", "kkt_kernel_pipe", "kkt_kernel_duo", "resid_slot_kernel", "resid_decide_kernel",
              "newton_async_kernel", "reset_kernel", "solve_finish_kernel", "async_handoff", "dz_commit"):
        if k in n:
            return k
    return n[-40:]


starts = [i for i, r in enumerate(rows) if "reset_kernel" in r["Kernel_Name"]]
if len(starts) < which + 1:
    raise SystemExit("not enough solves in the trace")
a, b = starts[-which - 1], starts[-which]
sel = rows[a:b]
t0 = int(sel[0]["Start_Timestamp"])
qs = sorted({r.get("Queue_Id", "?") for r in sel})
print("solve of %d kernels, %.1f us from the reset launch to the next one" % (len(sel), (int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
rnd = 0
first = {}
for r in sel:
    n = short(r["Kernel_Name"])
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%8.1f  %8.1f us  q%-2d r%-3d %-24s grid %s" % (s, e - s, qs.index(r.get("Queue_Id", "?")), rnd, n, r.get("Grid_Size", "?")))
    first.setdefault(rnd, s)
    if n == "resid_decide_kernel":
        rnd += 1
