"""sjhip_parse_device of one fixture (message resident on the device, result left there) in a loop, for rocprofv3:
python tools/small_dev_loop.py twitter 20"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "simdjson-go_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import fixtures  # noqa: E402
import sjhip  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "twitter"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
raw = fixtures.load(name).strip(b" \t\r\n")
nd = name.startswith("parking")
d = torch.empty(len(raw) + 256, dtype=torch.uint8, device="cuda:0")
d[:len(raw)].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
torch.cuda.synchronize()
ctx = sjhip.Context(0)
for _ in range(5):
    ctx.parse_device(d.data_ptr(), len(raw), ndjson=nd, copy_strings=True)
t0 = time.perf_counter()
for _ in range(iters):
    ctx.parse_device(d.data_ptr(), len(raw), ndjson=nd, copy_strings=True)
print(f"{name} {len(raw)} B {(time.perf_counter() - t0) / iters * 1e6:.1f} us/parse (device-resident)")
