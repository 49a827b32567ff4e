#!/usr/bin/env python3
"""profiles/r02_parse_kernels.json (per-kernel average microseconds of the whole parse, per workload) and
profiles/stage2_pmc.json (HBM bytes per launch of every kernel: 2*FETCH_SIZE + WRITE_SIZE KiB, MI355X_MICROARCH.md)
from a tools/profile_r6.sh summary: python tools/make_parse_json.py <summary.txt> <kernels.json> <pmc.json>"""
import json
import re
import sys

NAMES = {"twitter": "twitter_x426", "parking": "parking_x1000_nd"}


def short(name):
    name = name.replace("void ", "").replace("sj::", "")
    name = re.sub(r"\(.*$", "", name)
    return name.strip()


def main(summary, out_k, out_p):
    keys = list(NAMES.values()) + [v + "_nocopy" for v in NAMES.values()]
    kern = {v: {} for v in keys}
    pmc = {v: {} for v in keys}
    sect, wl = "", None
    for line in open(summary):
        if line.startswith("=="):
            sect = line
            wl = next((v for k, v in NAMES.items() if f"_{k}" in line), None)
            if wl and "nocopy" in line:  # (WithCopyStrings(false): its own table)
                wl += "_nocopy"
            continue
        if wl is None:
            continue
        m = re.match(r"\s+kernel (.*?)\s+calls=(\d+) avg_us=([\d.]+) total_us=([\d.]+)", line)
        if m and "trace_" in sect:
            kern[wl][short(m.group(1))] = {"calls": int(m.group(2)), "avg_us": float(m.group(3))}
        m = re.match(r"\s+pmc (.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+dispatches=(\d+) mean=([\d.]+)", line)
        if m:
            pmc[wl].setdefault(short(m.group(1)), {})[m.group(2) + "_KB"] = float(m.group(4))
    for wl in list(kern):
        if not kern[wl] and not pmc[wl]:
            del kern[wl], pmc[wl]
            continue
        per_parse = {}
        calls = [v["calls"] for v in kern[wl].values()]
        parses = max(calls) if calls else 1  # one launch of each parse kernel per parse (a context's first-use fills run less often)
        for k, v in kern[wl].items():
            per_parse[k] = round(v["avg_us"] * v["calls"] / parses, 1)
        kern[wl] = {"parses_traced": parses, "us_per_parse": dict(sorted(per_parse.items(), key=lambda kv: -kv[1])),
                    "sum_us": round(sum(per_parse.values()), 1)}
        for k, v in pmc[wl].items():
            if "FETCH_SIZE_KB" in v and "WRITE_SIZE_KB" in v:
                v["hbm_bytes_per_launch"] = int((2 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024)
    kern["source"] = "rocprofv3 --kernel-trace --stats -- python tools/parse_loop.py <workload> (tools/profile_r6.sh)"
    pmc["source"] = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2"
    json.dump(kern, open(out_k, "w"), indent=1)
    json.dump(pmc, open(out_p, "w"), indent=1)
    print(json.dumps(kern, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
