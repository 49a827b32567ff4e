"""one-screen summary of a bench.py JSON line: python tools/bench_brief.py <file>"""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]
print("stage1   value", d["value"], "GB/s  ms/step", d["ms_per_step"], " kernel_ms", r["kernel_ms"], " frac", r["frac"], " read_frac", r.get("read_frac"))
for k in ("at_64MiB", "at_1GiB"):
    x = r.get(k)
    if x:
        print(f"  {k}: kernel_ms", x["kernel_ms"], "frac", x["frac"], "input_frac", x["input_frac"], "read_frac", x.get("read_frac"))
for k in ("full_parse", "full_parse_nocopy", "ndjson"):
    x = d.get(k)
    if x:
        print(f"{k:18s} ms {x['ms']}  GB/s {x['GBps']}  frac {x['roofline']['frac']}")
for k, v in (d.get("single_documents") or {}).items():
    print(f"{k:26s}", {a: b for a, b in v.items() if a.endswith("_us")})
for k in ("batch", "query", "serialize", "marshal_json", "stream"):
    x = d.get(k)
    if x:
        print(k, {a: b for a, b in x.items() if a in ("ms", "GBps", "count_ms", "filter_ms", "us_per_document", "variants_GBps")})
if d.get("extra_error"):
    print("EXTRA ERROR", d["extra_error"])
if d.get("cpu_baseline"):
    print("cpu", {k: v.get("value") for k, v in d["cpu_baseline"].items() if isinstance(v, dict) and "value" in v})
