"""Parse one BASELINE workload once, then MarshalJSON / Serialize / Deserialize of the resident result in a loop (for
rocprofv3): python tools/marshal_loop.py twitter|parking|canada [iters] [kf]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "simdjson-go_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import sjhip  # noqa: E402
import workloads  # noqa: E402

which = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if which == "twitter":
    doc, nd = workloads.c2_twitter_array(426), False
elif which == "canada":  # floats: 20 copies of canada.json in one array (45 MB)
    import fixtures
    doc, nd = b"[" + b",".join([fixtures.load("canada")] * 20) + b"]", False
else:
    doc, nd = workloads.c5_parking_nd(1000).rstrip(b"\n"), True
d = torch.empty(len(doc) + 256, dtype=torch.uint8, device="cuda:0")
d[:len(doc)].copy_(torch.frombuffer(bytearray(doc), dtype=torch.uint8))
torch.cuda.synchronize()
ctx = sjhip.Context(0)
kf = len(sys.argv) > 3 and sys.argv[3] == "kf"
tl, sl = ctx.parse_device(d.data_ptr(), len(doc), ndjson=nd, copy_strings=True, key_flags=kf)
ctx.marshal_json(fetch=False)
t0 = time.perf_counter()
for _ in range(iters):
    n = ctx.marshal_json(fetch=False)
dt = (time.perf_counter() - t0) / iters
print(f"{which}: marshal_json {dt*1e3:.3f} ms  ({len(doc)/dt/1e9:.1f} GB/s of input, text {n} B)")
ctx.serialize(fetch=False)
t0 = time.perf_counter()
for _ in range(iters):
    ctx.serialize(fetch=False)
dt = (time.perf_counter() - t0) / iters
print(f"{which}: serialize {dt*1e3:.3f} ms")
