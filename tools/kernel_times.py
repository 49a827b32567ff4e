"""Per-dispatch kernel durations from a rocprofv3 rocpd database: python tools/kernel_times.py <db> [name-substring]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cur = con.cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall() if False else None
try:
    rows = cur.execute("select name, (end - start) from kernels order by start").fetchall()
except sqlite3.Error:
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    print("tables:", tabs)
    sys.exit(0)
from collections import defaultdict
d = defaultdict(list)
for n, dur in rows:
    if pat in n:
        d[n.split("(")[0]].append(dur / 1000.0)
for n, v in d.items():
    print(f"{n[:50]:50s} n={len(v)} us: " + " ".join(f"{x:.0f}" for x in v[:14]))
