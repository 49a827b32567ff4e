"""Summarise a SJHIP_S1_TRACE dump: [tile][16 waves][16 stamps] u64, wall_clock64 ticks of 10 ns.
stamps: 0 phase A start, 1 phase A done, 2 aggregate published, 3 look-back done, 4 flatten done."""
import sys
import numpy as np
waves = int(sys.argv[2]) if len(sys.argv) > 2 else 8
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16, 16).astype(np.int64)[:, :waves, :]
n = len(a)
t0 = a[:, :, 0][a[:, :, 0] > 0].min()
ts = (a - t0) / 100.0  # us
print(f"tiles {n}  span {ts[:,:,4].max():.1f} us")
def row(nm, x):
    x = x.ravel()
    print(f"  {nm:34s} mean {x.mean():7.2f}  p10 {np.percentile(x,10):7.2f}  p50 {np.percentile(x,50):7.2f}  p90 {np.percentile(x,90):7.2f}  max {x.max():7.2f} us")
row("phase A (per wave)", ts[:, :, 1] - ts[:, :, 0])
row("barrier+aggregate+publish", ts[:, :, 2] - ts[:, :, 1])
row("look-back", ts[:, :, 3] - ts[:, :, 2])
row("flatten", ts[:, :, 4] - ts[:, :, 3])
row("tile total", ts[:, :, 4] - ts[:, :, 0])
# how long after its own aggregate did the slowest predecessor publish?  (tile-level, wave 0)
agg = ts[:, 0, 2]
run = np.maximum.accumulate(agg)
lag = np.concatenate([[0], run[:-1] - agg[1:]])
row("slowest predecessor's agg - own agg", np.maximum(lag, 0))
