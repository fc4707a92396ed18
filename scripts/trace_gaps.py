"""Per-kernel durations AND the idle gap behind each kernel from a rocprofv3 --kernel-trace csv: python scripts/trace_gaps.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = defaultdict(list); gap = defaultdict(list)
for a, b in zip(rows, rows[1:] + [None]):
    n = a["Kernel_Name"].split("(")[0][-48:]
    dur[n].append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
    if b is not None:
        gap[n].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print("%d kernels over %.1f us" % (len(rows), span))
for n in sorted(dur, key=lambda k: -sum(dur[k])):
    g = gap[n] or [0.0]
    g2 = sorted(g)
    print("%-50s calls %5d  avg %8.1f us  total %9.1f us | gap behind it: median %6.1f  mean %6.1f us" % (n, len(dur[n]), sum(dur[n]) / len(dur[n]), sum(dur[n]), g2[len(g2) // 2], sum(g) / len(g)))
