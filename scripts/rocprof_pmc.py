"""Per-kernel average of ONE PMC counter from a rocprofv3 rocpd database.
usage: python scripts/rocprof_pmc.py <results.db> [name-filter]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)").fetchall()]
rows = c.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection "
                 "group by kernel_name, counter_name order by sum(value) desc").fetchall()
print("| kernel | counter | dispatches | avg / dispatch | total |")
print("|---|---|---|---|---|")
for r in rows:
    if flt in r[0]:
        print(f"| {r[0][:70]} | {r[1]} | {r[2]} | {r[3]:.1f} | {r[4]:.1f} |")
