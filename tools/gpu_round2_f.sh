#!/bin/bash
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
(timeout 900 python -m pytest tests/test_gpu_serialize.py -m gpu -x -q > gpurun_out/pytest_f.log 2>&1; echo "exit $?" >> gpurun_out/pytest_f.log)
python - > gpurun_out/serialize_time.log 2>&1 <<'PY'
import sys, os, time
sys.path.insert(0, "simdjson-go_amd"); sys.path.insert(0, "tests")
import torch, sjhip, workloads
ctx = sjhip.Context(0)
for name, doc, nd in (("twitter_x426", workloads.c2_twitter_array(426), False), ("parking_x1000", workloads.c5_parking_nd(1000).rstrip(b"\n"), True)):
    d = torch.empty(len(doc) + 256, dtype=torch.uint8, device="cuda:0"); d[:len(doc)].copy_(torch.frombuffer(bytearray(doc), dtype=torch.uint8)); torch.cuda.synchronize()
    ctx.parse_device(d.data_ptr(), len(doc), ndjson=nd, copy_strings=True)
    ctx.serialize(fetch=False)
    t = time.perf_counter()
    for _ in range(5): sz = ctx.serialize(fetch=False)
    dt = (time.perf_counter() - t) / 5
    print(name, "serialize (device columns)", round(dt * 1e3, 3), "ms", sz, round(len(doc) / dt / 1e9, 1), "GB/s of input")
PY
for f in gpurun_out/pytest_f.log gpurun_out/serialize_time.log; do echo "== $f"; tail -n 12 $f | cut -c1-600; done
