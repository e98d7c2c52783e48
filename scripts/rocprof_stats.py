"""Summarise a rocprofv3 rocpd database (kernel trace) into a per-kernel table (markdown).
usage: python scripts/rocprof_stats.py <results.db> [out.md]"""
import sqlite3, sys
db = sys.argv[1]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start), "
                 "max(vgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name "
                 "order by sum(end-start) desc").fetchall()
tot = sum(r[5] for r in rows) or 1
lines = ["| kernel | calls | avg us | min us | max us | total ms | % | vgpr | lds B | grid_x | wg_x |", "|---|---|---|---|---|---|---|---|---|---|---|"]
for r in rows[:40]:
    lines.append(f"| {r[0][:80]} | {r[1]} | {r[2]/1e3:.2f} | {r[3]/1e3:.2f} | {r[4]/1e3:.2f} | {r[5]/1e6:.3f} | {r[5]/tot*100:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} |")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
