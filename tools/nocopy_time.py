"""Wall time of the whole parse of the two BASELINE workloads (device-resident), WithCopyStrings(false) and (true): python tools/nocopy_time.py"""
import sys, time
sys.path.insert(0, "simdjson-go_amd"); sys.path.insert(0, "tests")
import torch, sjhip, workloads
ctx = sjhip.Context(0)
for name, doc, nd in (("twitter", workloads.c2_twitter_array(426), False), ("parking", workloads.c5_parking_nd(1000).rstrip(b"\n"), True)):
    d = torch.empty(len(doc) + 256, dtype=torch.uint8, device="cuda:0"); d[:len(doc)].copy_(torch.frombuffer(bytearray(doc), dtype=torch.uint8)); torch.cuda.synchronize()
    for copy in (False, True):
        for _ in range(3): ctx.parse_device(d.data_ptr(), len(doc), ndjson=nd, copy_strings=copy)
        best = 1e9
        for rep in range(4):
            t0 = time.perf_counter()
            for _ in range(10): ctx.parse_device(d.data_ptr(), len(doc), ndjson=nd, copy_strings=copy)
            best = min(best, (time.perf_counter() - t0) / 10)
        print(name, "copy" if copy else "nocopy", "%.3f ms" % (best * 1e3))
