#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the `--stats` style table:
per kernel name: calls, total / average / min / max duration (us), percentage.
usage: rocpd_stats.py results.db [out.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("""
    select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start),
           min(d.end - d.start), max(d.end - d.start), max(s.arch_vgpr_count), max(s.sgpr_count),
           max(d.group_segment_size), max(d.workgroup_size_x)
    from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
    group by s.kernel_name order by 3 desc""").fetchall()
tot = sum(r[2] for r in rows) or 1
lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,VGPRs,SGPRs,LDS_bytes,WG_size"]
for r in rows:
    name = r[0].replace(",", ";")
    lines.append("\"%s\",%d,%d,%.1f,%d,%d,%.2f,%d,%d,%d,%d" % (name, r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot,
                                                            r[6], r[7], r[8], r[9]))
out = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out)
sys.stdout.write(out)
