"""Summarise a rocprofv3 rocpd .db (kernel-trace) into per-kernel stats (text)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), avg(end-start)/1000.0, min(end-start)/1000.0, max(end-start)/1000.0, sum(end-start)/1000.0 "
                  "from kernels group by name order by 6 desc").fetchall()
tot = sum(r[5] for r in rows)
print("%-70s %6s %10s %10s %10s %11s %6s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "total_us", "pct"))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print("%-70s %6d %10.1f %10.1f %10.1f %11.1f %5.1f%%" % (r[0][:70], r[1], r[2], r[3], r[4], r[5], 100 * r[5] / tot))
