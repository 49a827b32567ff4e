#!/bin/bash
# A/B of whole-parse wall time between the libraries in build_ab/ (same box, interleaved twice)
for round in 1 2; do
for lib in $(ls build_ab/*.so); do
SJHIP_LIB=$PWD/$lib timeout 200 python - <<PY
import sys, os, time
sys.path.insert(0, "simdjson-go_amd"); sys.path.insert(0, "tests")
import torch, sjhip, workloads
ctx = sjhip.Context(0)
out = []
for name, doc, nd in (("twitter", workloads.c2_twitter_array(426), False), ("parking", workloads.c5_parking_nd(1000).rstrip(b"\n"), True)):
    d = torch.empty(len(doc) + 256, dtype=torch.uint8, device="cuda:0"); d[:len(doc)].copy_(torch.frombuffer(bytearray(doc), dtype=torch.uint8)); torch.cuda.synchronize()
    for _ in range(3): ctx.parse_device(d.data_ptr(), len(doc), ndjson=nd, copy_strings=True)
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        for _ in range(10): ctx.parse_device(d.data_ptr(), len(doc), ndjson=nd, copy_strings=True)
        best = min(best, (time.perf_counter() - t0) / 10)
    out.append("%s %.3f ms" % (name, best * 1e3))
print("$lib", *out)
PY
done; done
