#!/usr/bin/env python3
"""Per-kernel summary (calls, total, avg, min, max, %) from a rocprofv3 rocpd .db
(the ROCm 7.2 default output of `rocprofv3 --kernel-trace --stats`).  Usage: rocpd_stats.py x.db [top_n]"""
import sqlite3, sys
db = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
con = sqlite3.connect(db)
rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"# {db}: {sum(r[1] for r in rows)} dispatches, {tot/1e6:.3f} ms total GPU kernel time")
print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}  kernel")
for name, n, s, a, lo, hi in rows[:top]:
    print(f"{n:7d} {s/1e6:10.3f} {a/1e3:10.2f} {lo/1e3:9.2f} {hi/1e3:9.2f} {100*s/tot:6.2f}  {name[:150]}")
