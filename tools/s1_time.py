"""kernel-only timing of stage 1 on configs[1] (no correctness assertions; for A/B experiments)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "simdjson-go_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, sjhip, workloads
copies = int(os.environ.get("COPIES", "426"))
doc = workloads.c2_twitter_array(copies)
n = len(doc)
d_msg = torch.empty(n + 256, dtype=torch.uint8, device="cuda:0")
d_msg[:n].copy_(torch.frombuffer(bytearray(doc), dtype=torch.uint8))
d_pos = torch.empty(workloads.c2_expected_structurals(copies) + 1024, dtype=torch.int32, device="cuda:0")
torch.cuda.synchronize()
ctx = sjhip.Context(0)
ctx.stage1_time(d_msg.data_ptr(), n, d_pos.data_ptr(), d_pos.numel(), 5)
ms = ctx.stage1_time(d_msg.data_ptr(), n, d_pos.data_ptr(), d_pos.numel(), 30)
print(f"{ms:.4f} ms  {n/ms/1e6:.1f} GB/s input")
