"""Per-kernel PMC averages from a rocprofv3 rocpd db: python scratch/pmc_summary.py db [kernel-substring]"""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1]); flt = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [n for (n,) in db.execute("select name from sqlite_master where type in ('table','view')")]
def T(prefix):
    c = [t for t in tabs if t == prefix] or [t for t in tabs if t.startswith(prefix)]
    return c[0]
pmc_info, pmc_ev, disp, sym = T("rocpd_info_pmc"), T("rocpd_pmc_event"), T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")
cols = [r[1] for r in db.execute(f"pragma table_info({disp})")]
q = f"""select s.kernel_name, d.id, d.end - d.start, d.grid_size_x, i.name, sum(e.value) from {pmc_ev} e join {pmc_info} i on e.pmc_id = i.id
        join {disp} d on e.event_id = d.event_id join {sym} s on d.kernel_id = s.id group by d.id, i.name"""
acc = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
for name, did, dt, grid, cname, val in db.execute(q):
    if flt not in name: continue
    key = (name[:60], grid)
    acc[key][cname].append(val); 
for key, cs in sorted(acc.items(), key=lambda kv: -len(next(iter(kv[1].values())))):
    n = len(next(iter(cs.values())))
    if n < 5: continue
    print(key, "dispatches", n)
    for c, v in sorted(cs.items()):
        v = sorted(v)[len(v) // 2:]  # upper half: skips the short final launch
        print(f"   {c:40s} {sum(v) / len(v):16.1f}")
