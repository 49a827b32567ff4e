"""Kernel timeline of the LAST `reps` repetitions of a launch chain from a rocprofv3 rocpd database: per kernel the start relative to
the chain's first kernel, the duration and the gap to the kernel in front (microseconds): python tools/timeline.py <db> [chains]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
want = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = con.execute("select name, start, end from kernels order by start").fetchall()
# a chain begins with stage 1 (until the middle of round 6: with its preparation kernel)
starts = [i for i, r in enumerate(rows) if "k_s1_prepare" in r[0]] or [i for i, r in enumerate(rows) if "stage1_kernel" in r[0]]
for ci in starts[-want:]:
    nxt = [s for s in starts if s > ci]
    chain = rows[ci:(nxt[0] if nxt else len(rows))]
    t0 = chain[0][1]
    prev_end = t0
    print(f"-- chain of {len(chain)} launches, {(chain[-1][2] - t0) / 1e3:.1f} us from first start to last end")
    for name, s, e in chain:
        print(f"  {name.split('(')[0][:44]:44s} start {(s - t0) / 1e3:7.1f}  dur {(e - s) / 1e3:6.1f}  gap {(s - prev_end) / 1e3:5.1f}")
        prev_end = e
