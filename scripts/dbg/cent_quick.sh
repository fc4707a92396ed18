# configs[4] legs (scripts/cent_legs.py) under the library CIMPC_LIB names, reduced to the numbers that matter
timeout 300 python scripts/cent_legs.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: continue
    print({k:(round(v,3) if isinstance(v,float) else v) for k,v in j.items() if k in ('ms_per_step','ip_sweep_ms_per_step','kkt_ms_per_step','newton_iters_per_step','value')})
"
