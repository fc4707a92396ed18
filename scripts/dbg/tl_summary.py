"""per-round summary of timelines written by scripts/dbg/timeline.sh: python scripts/dbg/tl_summary.py <tag> ..."""
import sys
for t in sys.argv[1:]:
    lines = open(f"gpurun_out/timeline_{t}.txt").read().splitlines()
    rows = [l.split() for l in lines[1:]]
    print(t, lines[0].strip())
    cur = {}
    for r in rows:
        cur.setdefault(r[4], {})[r[5]] = (float(r[0]), float(r[1]))
    tot = {"sweep": 0.0, "kkt": 0.0, "slot": 0.0, "decide": 0.0}
    for rnd, d in cur.items():
        k = [v for n, v in d.items() if n.startswith("kkt_kernel")]
        sw = d.get("ip_queue_kernel"); dec = d.get("resid_decide_kernel"); sl = d.get("resid_slot_kernel")
        if sw:
            tot["sweep"] += sw[1]; tot["kkt"] += k[0][1] if k else 0; tot["slot"] += sl[1] if sl else 0; tot["decide"] += dec[1] if dec else 0
            print(" ", rnd, "sweep %4.0f@%5.0f" % (sw[1], sw[0]), "kkt %s" % (("%4.0f@%5.0f" % (k[0][1], k[0][0])) if k else "     -    "),
                  "slot %3.0f" % (sl[1] if sl else 0), ("decide %3.0f@%5.0f" % (dec[1], dec[0])) if dec else "")
        for n, v in d.items():
            if n.startswith("newton_async"):
                print("  tail %.0f@%.0f" % (v[1], v[0]))
    print("  totals", {k: round(v) for k, v in tot.items()})
