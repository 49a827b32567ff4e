#!/bin/bash
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
(timeout 900 python -m pytest tests/test_gpu_marshal.py tests/test_gpu_serialize.py tests/test_gpu_query.py -m gpu -x -q > gpurun_out/pytest_g.log 2>&1; echo "exit $?" >> gpurun_out/pytest_g.log)
python - > gpurun_out/marshal_time.log 2>&1 <<'PY'
import sys, os, time
sys.path.insert(0, "simdjson-go_amd"); sys.path.insert(0, "tests")
import torch, sjhip, workloads, fixtures
ctx = sjhip.Context(0)
for name, doc, nd in (("twitter_x426", workloads.c2_twitter_array(426), False), ("parking_x1000", workloads.c5_parking_nd(1000).rstrip(b"\n"), True), ("canada_x100", b"[" + b",".join([fixtures.load("canada").strip()] * 100) + b"]", False)):
    d = torch.empty(len(doc) + 256, dtype=torch.uint8, device="cuda:0"); d[:len(doc)].copy_(torch.frombuffer(bytearray(doc), dtype=torch.uint8)); torch.cuda.synchronize()
    ctx.parse_device(d.data_ptr(), len(doc), ndjson=nd, copy_strings=True)
    ctx.marshal_json(fetch=False)
    t = time.perf_counter()
    for _ in range(5): n = ctx.marshal_json(fetch=False)
    dt = (time.perf_counter() - t) / 5
    print(name, "marshal_json (device text)", round(dt * 1e3, 3), "ms", n, "bytes of text", round(len(doc) / dt / 1e9, 1), "GB/s of input")
PY
for f in gpurun_out/pytest_g.log gpurun_out/marshal_time.log; do echo "== $f"; tail -n 25 $f | cut -c1-900; done
