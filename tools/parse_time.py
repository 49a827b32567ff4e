"""Wall time of the whole parse (device-resident input) of the two BASELINE workloads and host->host latency of the
fixtures: python tools/parse_time.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "simdjson-go_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import fixtures  # noqa: E402
import sjhip  # noqa: E402
import workloads  # noqa: E402

ctx = sjhip.Context(0)
for which in ("twitter", "parking"):
    if which == "twitter":
        doc, nd = workloads.c2_twitter_array(426), False
    else:
        doc, nd = workloads.c5_parking_nd(1000).rstrip(b"\n"), True
    d = torch.empty(len(doc) + 256, dtype=torch.uint8, device="cuda:0")
    d[:len(doc)].copy_(torch.frombuffer(bytearray(doc), dtype=torch.uint8))
    torch.cuda.synchronize()
    for _ in range(3):
        tl, sl = ctx.parse_device(d.data_ptr(), len(doc), ndjson=nd, copy_strings=True)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            ctx.parse_device(d.data_ptr(), len(doc), ndjson=nd, copy_strings=True)
        best = min(best, (time.perf_counter() - t0) / 10)
    print(f"{which:8s} {len(doc)} B  tape {tl} strings {sl}  {best*1e3:.3f} ms  {len(doc)/best/1e9:.1f} GB/s", flush=True)
    del d
for name in ("twitter", "canada", "twitterescaped", "parking-citations"):
    d = fixtures.load(name)
    nd = name.startswith("parking")
    for _ in range(5):
        ctx.parse(d, ndjson=nd)
    t0 = time.perf_counter()
    N = 100
    for _ in range(N):
        ctx.parse(d, ndjson=nd)
    dt = (time.perf_counter() - t0) / N
    print(f"{name:20s} {len(d):9d} B  {dt*1e6:8.1f} us/parse  {len(d)/dt/1e9:6.2f} GB/s (host buffer -> tape on host)", flush=True)
