"""Summarise a SJHIP_S1_TRACE dump: 8 x u64 per tile written by the scouts (wall_clock64 ticks of 10 ns):
0 ticket drawn, 1 aggregate published, 2 look-back resolved, 3 look-back attempts, 4 first look-back attempt."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.int64)
n = len(a)
t0 = a[:, 0].min()
tk, ag, rs, first = [(a[:, i] - t0) / 100.0 for i in (0, 1, 2, 4)]
print(f"tiles {n}  span {rs.max():.1f} us")
def row(nm, x):
    print(f"  {nm:40s} mean {x.mean():7.2f}  p10 {np.percentile(x,10):7.2f}  p50 {np.percentile(x,50):7.2f}  p90 {np.percentile(x,90):7.2f}  max {x.max():7.2f}")
row("ticket -> aggregate published (us)", ag - tk)
row("aggregate -> first look-back attempt", first[1:] - ag[1:])
row("aggregate -> resolved", rs - ag)
row("look-back attempts", a[:, 3].astype(float))
run = np.maximum.accumulate(ag)
row("slowest predecessor agg - own agg", np.maximum(np.concatenate([[0], run[:-1] - ag[1:]]), 0))
row("resolved - slowest predecessor agg", rs[1:] - run[:-1])
# tickets per us over time
print("  ticket rate per us (10 bins):", np.round(np.histogram(tk, bins=10)[0] / (tk.max() / 10 + 1e-9), 1))
# tiles with ticket but unresolved, sampled
ev = np.concatenate([np.stack([tk, np.ones(n)], 1), np.stack([rs, -np.ones(n)], 1)])
ev = ev[np.argsort(ev[:, 0])]
conc = np.cumsum(ev[:, 1])
print(f"  tiles ticketed-but-unresolved: mean {np.average(conc[:-1], weights=np.diff(ev[:,0]) + 1e-9):.0f} max {conc.max():.0f}")
