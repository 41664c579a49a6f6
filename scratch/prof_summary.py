"""Summarise a rocprofv3 rocpd sqlite db: per-kernel calls / total / avg / min / max (us) and % of kernel time."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = cur.execute("""select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start),
                      max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(d.group_segment_size), max(d.grid_size_x), max(d.workgroup_size_x)
                      from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 3 desc""").fetchall()
tot = sum(r[2] for r in rows)
print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>8s} {'%':>6s} vgpr agpr lds grid wg")
for r in rows[:40]:
    name = r[0][:70]
    print(f"{name:70s} {r[1]:6d} {r[2]/1e6:9.3f} {r[3]/1e3:9.2f} {r[4]/1e3:8.2f} {r[5]/1e3:8.2f} {100*r[2]/tot:6.2f} {r[6]} {r[7]} {r[8]} {r[9]} {r[10]}")
print("total kernel ms:", tot/1e6)
