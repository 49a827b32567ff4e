"""Latency of Parse() on the single documents BASELINE.json names (configs[0], [2], [3]) and parking-citations.json:
device-resident message -> result left on the device (sjhip_parse_device), host buffer -> result read in place
(sjhip_parse + sjhip_fetch_view), host buffer -> host arrays (sjhip_parse + sjhip_fetch); microseconds per call, best of 5 runs
of 200: python tools/small_time.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "simdjson-go_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import fixtures  # noqa: E402
import sjhip  # noqa: E402

ctx = sjhip.Context(0)
out = []
for name in ("twitter", "canada", "twitterescaped", "parking-citations"):
    raw = fixtures.load(name).strip(b" \t\r\n")
    nd = name.startswith("parking")
    arr = np.frombuffer(raw, dtype=np.uint8)
    d = torch.empty(len(raw) + 256, dtype=torch.uint8, device="cuda:0")
    d[:len(raw)].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
    torch.cuda.synchronize()
    pj = ctx.parse(arr, ndjson=nd)

    def best(fn, n=200, runs=5):
        for _ in range(20):
            fn()
        b = 1e9
        for _ in range(runs):
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            b = min(b, (time.perf_counter() - t0) / n)
        return b * 1e6
    dev = best(lambda: ctx.parse_device(d.data_ptr(), len(raw), ndjson=nd, copy_strings=True))
    dev_nc = best(lambda: ctx.parse_device(d.data_ptr(), len(raw), ndjson=nd, copy_strings=False))
    view = best(lambda: ctx.parse(arr, ndjson=nd, view=True))
    h2h = best(lambda: ctx.parse(arr, ndjson=nd, reuse=pj))
    out.append(f"{name} dev {dev:.1f} nocopy {dev_nc:.1f} view {view:.1f} h2h {h2h:.1f}")
print(" | ".join(out), flush=True)
