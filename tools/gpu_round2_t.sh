#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parse.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python - <<'PY'
import sys, os, time
sys.path.insert(0, "simdjson-go_amd"); sys.path.insert(0, "tests")
import torch, sjhip, workloads
ctx = sjhip.Context(0)
for name, doc, nd in (("twitter x426", workloads.c2_twitter_array(426), False), ("parking x1000", workloads.c5_parking_nd(1000).rstrip(b"\n"), True)):
    d = torch.empty(len(doc) + 256, dtype=torch.uint8, device="cuda:0"); d[:len(doc)].copy_(torch.frombuffer(bytearray(doc), dtype=torch.uint8)); torch.cuda.synchronize()
    for _ in range(3): ctx.parse_device(d.data_ptr(), len(doc), ndjson=nd, copy_strings=True)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(10): ctx.parse_device(d.data_ptr(), len(doc), ndjson=nd, copy_strings=True)
        best = min(best, (time.perf_counter() - t0) / 10)
    print(name, "parse %.3f ms" % (best * 1e3))
PY
