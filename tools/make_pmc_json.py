#!/usr/bin/env python3
"""profiles/stage1_pmc.json from a tools/profile_round.sh summary: python tools/make_pmc_json.py <summary.txt> <out.json> <source-note>

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB (MI355X_MICROARCH.md, HBM / rocprofv3 section: on gfx950
FETCH_SIZE counts 64 B per 128-byte request)."""
import json
import re
import sys


def main(summary, out, source):
    vals, kernel, section, trace_avg, pmc_avgs = {}, None, "", None, []
    for line in open(summary):
        if line.startswith("=="):
            section = line
        m = re.match(r"\s+pmc (.*stage1_kernel<[^>]*>).*?\s(\w+)\s+dispatches=\d+ mean=([\d.]+)", line)
        if m:
            kernel = kernel or m.group(1).replace("void ", "")
            vals[m.group(2)] = float(m.group(3))
    trace_groups = []
    for line in open(summary):  # durations of the kernel the counters belong to
        if line.startswith("=="):
            section = line
        m = re.match(r"\s+kernel (.*stage1_kernel<[^>]*>).*avg_us=([\d.]+)", line)
        if m and m.group(1).replace("void ", "") == kernel:
            if "trace" in section:
                trace_avg = float(m.group(2))
            else:
                pmc_avgs.append(float(m.group(2)))
        m = re.match(r"\s+kernel-group (.*stage1_kernel<[^>]*>).*calls=(\d+) avg_us=([\d.]+)", line)
        if m and m.group(1).replace("void ", "") == kernel and "trace" in section:
            trace_groups.append({"calls": int(m.group(2)), "avg_us": float(m.group(3))})
    if trace_groups:  # the traced command also runs the kernel on the 1 GiB document: the short group is configs[1]
        trace_avg = trace_groups[0]["avg_us"]
    d = {"kernel": kernel, "workload": "bench.py --stage1-only (configs[1])",
         "rocprofv3_avg_us_kernel_trace": trace_avg,
         "rocprofv3_kernel_trace_duration_groups": trace_groups or None,
         "rocprofv3_avg_us_pmc_passes": round(sum(pmc_avgs) / len(pmc_avgs), 3) if pmc_avgs else None,
         "FETCH_SIZE_KB": vals.get("FETCH_SIZE"), "WRITE_SIZE_KB": vals.get("WRITE_SIZE"),
         "hbm_bytes_per_launch": int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024),
         "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2 (MI355X_MICROARCH.md, HBM section); "
                       "WRITE_SIZE as reported",
         "source": source}
    for k, v in sorted(vals.items()):
        if k.startswith("SQ_") or k.startswith("GRBM"):
            d[k] = v
    json.dump(d, open(out, "w"), indent=1)
    print(json.dumps(d, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
