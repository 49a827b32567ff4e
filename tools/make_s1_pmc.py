#!/usr/bin/env python3
"""profiles/stage1_pmc.json (configs[1]), stage1_pmc_64MiB.json (twitter x107) and stage1_pmc_1GiB.json (twitter x1700) from a tools/profile_r6.sh
summary: python tools/make_s1_pmc.py <summary.txt> <profiles dir> <source note>
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts 64 B per
128-byte request); read_frac = 2 * FETCH_SIZE / kernel time / 8 TB/s."""
import json
import re
import sys

PEAK = 8000.0  # GB/s


def main(summary, outdir, source):
    sect = ""
    data = {}
    for line in open(summary):
        if line.startswith("=="):
            sect = line.split()[1].split("/")[0]
            continue
        m = re.match(r"s1(fetch|write|trace)_(\d+)", sect)
        if not m:
            continue
        kind, copies = m.group(1), int(m.group(2))
        d = data.setdefault(copies, {})
        k = re.match(r"\s+kernel (void sj::stage1_kernel<[^>]*>).*calls=(\d+) avg_us=([\d.]+)", line)
        if k:
            d["kernel"] = k.group(1).replace("void ", "")
            d[f"avg_us_{kind}_pass"] = float(k.group(3))
            d[f"calls_{kind}_pass"] = int(k.group(2))
        p = re.match(r"\s+pmc (void sj::stage1_kernel<[^>]*>).*?\s(FETCH_SIZE|WRITE_SIZE)\s+dispatches=\d+ mean=([\d.]+)", line)
        if p:
            d[p.group(2) + "_KB"] = float(p.group(3))
    for copies, d in data.items():
        rd = 2 * d["FETCH_SIZE_KB"] * 1024
        wr = d["WRITE_SIZE_KB"] * 1024
        out = {"kernel": d["kernel"], "workload": f"tools/s1_time.py, twitter.json x{copies} in one array (stage 1 only)",
               "rocprofv3_avg_us_kernel_trace": d.get("avg_us_trace_pass"),
               "rocprofv3_avg_us_fetch_pass": d.get("avg_us_fetch_pass"), "rocprofv3_avg_us_write_pass": d.get("avg_us_write_pass"),
               "FETCH_SIZE_KB": d["FETCH_SIZE_KB"], "WRITE_SIZE_KB": d["WRITE_SIZE_KB"],
               "hbm_read_bytes_per_launch": int(rd), "hbm_write_bytes_per_launch": int(wr), "hbm_bytes_per_launch": int(rd + wr),
               "read_frac_at_kernel_trace_duration": round(rd / (d["avg_us_trace_pass"] * 1e-6) / 1e9 / PEAK, 4),
               "read_frac_at_fetch_pass_duration": round(rd / (d["avg_us_fetch_pass"] * 1e-6) / 1e9 / PEAK, 4),
               "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
               "source": source}
        name = {107: "stage1_pmc_64MiB.json", 426: "stage1_pmc.json"}.get(copies, "stage1_pmc_1GiB.json")
        json.dump(out, open(f"{outdir}/{name}", "w"), indent=1)
        print(name, json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
