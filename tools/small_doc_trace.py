"""Host -> host Parse() of one fixture in a loop (for rocprofv3 --kernel-trace): python tools/small_doc_trace.py twitter 20"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "simdjson-go_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import fixtures  # noqa: E402
import sjhip  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "twitter"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
d = fixtures.load(name)
ctx = sjhip.Context(0)
pj = ctx.parse(d, ndjson=name.startswith("parking"))
for _ in range(5):
    ctx.parse(d, ndjson=name.startswith("parking"), reuse=pj)
t0 = time.perf_counter()
for _ in range(iters):
    ctx.parse(d, ndjson=name.startswith("parking"), reuse=pj)
dt = (time.perf_counter() - t0) / iters
print(f"{name} {len(d)} B {dt * 1e6:.1f} us/parse (SJHIP_SMALL_BYTES={os.environ.get('SJHIP_SMALL_BYTES', 'default')})")
