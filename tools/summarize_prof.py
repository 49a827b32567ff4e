#!/usr/bin/env python3
"""Condense rocprofv3 (ROCm 7.2, rocpd sqlite output) result directories into one text summary:
per-kernel call count / average duration (view top_kernels) and per-kernel PMC counters
(view counters_collection: summed over hardware instances per dispatch, averaged over dispatches)."""
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def main(root, out):
    lines = []
    for db in sorted(glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True)):
        con = sqlite3.connect(db)
        cur = con.cursor()
        lines.append(f"== {os.path.relpath(db, root)}")
        try:
            for name, calls, total, avg, pct in cur.execute(
                    "select name,total_calls,total_duration,average,percentage from top_kernels"):
                lines.append(f"  kernel {name[:70]:70s} calls={calls} avg_us={avg:.3f} total_us={total:.1f} pct={pct:.1f}")
        except sqlite3.Error as e:
            lines.append(f"  (no top_kernels: {e})")
        # one kernel name can cover launches on documents of very different sizes (the bench runs stage 1 on configs[1]
        # and on a 1 GiB document): split a kernel's dispatches into duration groups when they fall apart by > 2x
        try:
            per_k = defaultdict(list)
            for name, dur in cur.execute("select name,(end-start)/1000.0 from kernels"):
                per_k[name].append(dur)
            for name, v in per_k.items():
                v.sort()
                groups, cur_g = [], [v[0]]
                for x in v[1:]:
                    if x > 2.0 * cur_g[0]:
                        groups.append(cur_g)
                        cur_g = [x]
                    else:
                        cur_g.append(x)
                groups.append(cur_g)
                if len(groups) > 1:
                    for g in groups:
                        lines.append(f"  kernel-group {name[:70]:70s} calls={len(g)} avg_us={sum(g)/len(g):.3f} "
                                     f"min_us={g[0]:.3f} max_us={g[-1]:.3f}")
        except sqlite3.Error as e:
            lines.append(f"  (no per-dispatch durations: {e})")
        try:
            per = defaultdict(lambda: defaultdict(float))
            for kname, cname, disp, val in cur.execute(
                    "select kernel_name,counter_name,dispatch_id,value from counters_collection"):
                per[(kname[:70], cname)][disp] += val
            for (kname, cname), d in sorted(per.items()):
                vals = list(d.values())
                lines.append(f"  pmc {kname:70s} {cname:24s} dispatches={len(vals)} mean={sum(vals)/len(vals):.1f} "
                             f"min={min(vals):.1f} max={max(vals):.1f}")
        except sqlite3.Error as e:
            lines.append(f"  (no counters: {e})")
    text = "\n".join(lines) + "\n"
    open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
