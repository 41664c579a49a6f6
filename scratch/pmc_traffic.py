"""profiles/r<N>_pmc_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) per microbench shape.
usage: python scratch/pmc_traffic.py out.json  B:fetch_db:write_db [B:fetch_db:write_db ...]      (K = 50)"""
import json, re, sqlite3, sys
from collections import defaultdict

K = 50
def per_kernel(dbpath, counter):
    db = sqlite3.connect(dbpath)
    tabs = [n for (n,) in db.execute("select name from sqlite_master where type in ('table','view')")]
    T = lambda p: ([t for t in tabs if t == p] or [t for t in tabs if t.startswith(p)])[0]
    q = f"""select s.kernel_name, d.id, d.grid_size_x * d.grid_size_y, sum(e.value) from {T('rocpd_pmc_event')} e join {T('rocpd_info_pmc')} i on e.pmc_id = i.id
            join {T('rocpd_kernel_dispatch')} d on e.event_id = d.event_id join {T('rocpd_info_kernel_symbol')} s on d.kernel_id = s.id
            where i.name = '{counter}' group by d.id"""
    acc = defaultdict(list)
    for name, _, grid, val in db.execute(q):
        acc[name].append((grid, val))
    out = {}
    for name, v in acc.items():
        g = max(x[0] for x in v)
        vals = [x[1] for x in v if x[0] == g]
        out[name] = (sum(vals) / len(vals), len(vals))
    return out

def short(name):
    m = re.search(r"\d+([a-z_0-9]+_kernel)(?:ILi(\d+))?", name)
    return (m.group(1) + (f"<{m.group(2)}>" if m.group(2) else "")) if m else name

res = {"command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python scratch/microbench.py B 50   (second pass: --pmc WRITE_SIZE)",
       "note": "FETCH_SIZE / WRITE_SIZE in KB per dispatch as rocprofv3 reports them (L2 memory-side requests; Infinity-Cache hits are counted). "
               "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads 1/2 of the bytes of wide (16 B/lane) coalesced reads -> corrected = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024.",
       "shapes": {}}
for spec in sys.argv[2:]:
    B, fdb, wdb = spec.split(":")
    B = int(B); R = B * K
    f, w = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    shape = {}
    for name in sorted(f):
        shape[short(name)] = {"FETCH_SIZE_KB_per_dispatch": round(f[name][0], 1), "WRITE_SIZE_KB_per_dispatch": round(w.get(name, (0, 0))[0], 1), "dispatches": f[name][1]}
    res["shapes"][f"B={B}"] = shape
    for kname, v in shape.items():
        if kname.startswith("pc_step"):
            raw = (v["FETCH_SIZE_KB_per_dispatch"] + v["WRITE_SIZE_KB_per_dispatch"]) * 1024
            corr = (2 * v["FETCH_SIZE_KB_per_dispatch"] + v["WRITE_SIZE_KB_per_dispatch"]) * 1024
            alg = R * 216 + 1040 * 1024 + 25 * 1024 + B * 768 * 4 + B * 12  # rows (x, score, 2 x noise in; x, score out) + weights + tvec/biases + cvec + centre
            res[f"{kname}@{R}"] = {"raw_bytes_per_launch": int(raw), "corrected_bytes_per_launch": int(corr), "algorithmic_bytes_per_launch": int(alg)}
json.dump(res, open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if "@" in k}, indent=1))
