"""Whole parse of one BASELINE workload in a loop (for rocprofv3): python tools/parse_loop.py twitter|parking [iters] [nocopy]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "simdjson-go_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import sjhip  # noqa: E402
import workloads  # noqa: E402

which = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
copy = not (len(sys.argv) > 3 and sys.argv[3] == "nocopy")
if which == "twitter":
    doc, nd = workloads.c2_twitter_array(426), False
else:
    doc, nd = workloads.c5_parking_nd(1000).rstrip(b"\n"), True
d = torch.empty(len(doc) + 256, dtype=torch.uint8, device="cuda:0")
d[:len(doc)].copy_(torch.frombuffer(bytearray(doc), dtype=torch.uint8))
torch.cuda.synchronize()
ctx = sjhip.Context(0)
for _ in range(iters + 1):
    tl, sl = ctx.parse_device(d.data_ptr(), len(doc), ndjson=nd, copy_strings=copy)
print(which, len(doc), tl, sl)
