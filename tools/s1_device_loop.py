"""sjhip_stage1_device in a loop on one document (for rocprofv3 --kernel-trace: the kernel as a real call runs it, host word and all): COPIES=426 python tools/s1_device_loop.py [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "simdjson-go_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, sjhip, workloads
copies = int(os.environ.get("COPIES", "426"))
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
doc = workloads.c2_twitter_array(copies)
n = len(doc)
d = torch.empty(n + 256, dtype=torch.uint8, device="cuda:0"); d[:n].copy_(torch.frombuffer(bytearray(doc), dtype=torch.uint8))
pos = torch.empty(workloads.c2_expected_structurals(copies) + 1024, dtype=torch.int32, device="cuda:0"); torch.cuda.synchronize()
ctx = sjhip.Context(0)
for _ in range(iters):
    ok, cnt = ctx.stage1_device(d.data_ptr(), n, pos.data_ptr(), pos.numel())
assert ok and cnt == workloads.c2_expected_structurals(copies)
print(copies, n, cnt)
