"""Timeline of the last parses in a rocprofv3 rocpd database: kernel name, start relative to the first kernel of the parse,
duration, gap to the previous kernel: python tools/timeline.py <db> [parses]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
# a parse starts with k_s1_prepare
starts = [i for i, r in enumerate(rows) if "k_s1_prepare" in r[0]]
want = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for s in starts[-want:]:
    e = next((x for x in starts if x > s), len(rows))
    t0 = rows[s][1]
    prev_end = t0
    print("-- parse")
    for name, a, b in rows[s:e]:
        print(f"  {name.split('(')[0][:44]:44s} start {(a - t0) / 1000:8.1f} us  dur {(b - a) / 1000:7.1f} us  gap {(a - prev_end) / 1000:6.1f} us")
        prev_end = b
    print(f"  total {(rows[e - 1][2] - t0) / 1000:.1f} us")
