"""Per-kernel summary of a rocprofv3 rocpd database (rocprofv3 --kernel-trace ... writes *_results.db)."""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                 "group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"{'%':>6} {'calls':>6} {'avg_us':>9} {'min_us':>9} {'max_us':>9}  kernel")
for n, cnt, s, a, lo, hi in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    print(f"{s / tot * 100:6.1f} {cnt:6d} {a / 1e3:9.1f} {lo / 1e3:9.1f} {hi / 1e3:9.1f}  {n[:110]}")
print(f"total kernel time {tot / 1e6:.3f} ms")
