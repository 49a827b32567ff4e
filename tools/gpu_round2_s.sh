#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_marshal.py tests/test_gpu_serialize.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python - <<'PY'
import sys, os, time
sys.path.insert(0, "simdjson-go_amd"); sys.path.insert(0, "tests")
import torch, sjhip, workloads
ctx = sjhip.Context(0)
for name, doc, nd in (("twitter x426", workloads.c2_twitter_array(426), False), ("parking x1000", workloads.c5_parking_nd(1000).rstrip(b"\n"), True)):
    d = torch.empty(len(doc) + 256, dtype=torch.uint8, device="cuda:0"); d[:len(doc)].copy_(torch.frombuffer(bytearray(doc), dtype=torch.uint8)); torch.cuda.synchronize()
    ctx.parse_device(d.data_ptr(), len(doc), ndjson=nd, copy_strings=True)
    for fn_name in ("marshal_json", "serialize"):
        fn = getattr(ctx, fn_name)
        fn(fetch=False); t0 = time.perf_counter()
        for _ in range(3): r = fn(fetch=False)
        print(name, fn_name, "%.3f ms" % ((time.perf_counter() - t0) / 3 * 1e3))
PY
