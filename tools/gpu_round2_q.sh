#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_query.py tests/test_gpu_parse.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python - <<'PY'
import sys, os, time
sys.path.insert(0, "simdjson-go_amd"); sys.path.insert(0, "tests")
import torch, sjhip, workloads
ctx = sjhip.Context(0)
doc = workloads.c5_parking_nd(1000).rstrip(b"\n")
d = torch.empty(len(doc) + 256, dtype=torch.uint8, device="cuda:0"); d[:len(doc)].copy_(torch.frombuffer(bytearray(doc), dtype=torch.uint8)); torch.cuda.synchronize()
ctx.parse_device(d.data_ptr(), len(doc), ndjson=True, copy_strings=True)
for name, fn in (("count", lambda: ctx.count_where(b"Make", b"HOND")), ("filter", lambda: ctx.filter_where(b"Make", b"HOND", fetch=False))):
    fn(); t0 = time.perf_counter()
    for _ in range(5): r = fn()
    print(name, r if name == "count" else r[0], "%.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
doc = workloads.c2_twitter_array(426)
d = torch.empty(len(doc) + 256, dtype=torch.uint8, device="cuda:0"); d[:len(doc)].copy_(torch.frombuffer(bytearray(doc), dtype=torch.uint8)); torch.cuda.synchronize()
ctx.parse_device(d.data_ptr(), len(doc), ndjson=False, copy_strings=True)
t0 = time.perf_counter()
for _ in range(10): ctx.parse_device(d.data_ptr(), len(doc), ndjson=False, copy_strings=True)
print("twitter x426 parse %.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
PY
