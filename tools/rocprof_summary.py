#!/usr/bin/env python3
"""Turn a rocprofv3 (rocpd sqlite) result into the per-kernel stats table we commit under profiles/.
usage: rocprof_summary.py <results.db> [out.md]"""
import sqlite3, sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
for n, c, t, a, mn, mx in rows:
    n = n if len(n) < 110 else n[:107] + "..."
    lines.append(f"| `{n}` | {c} | {t/1e6:.3f} | {a/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*t/tot:.1f} |")
txt = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
print(txt)
