"""Host-side cost of one sjhip_stage1_device call as a function of the document size (wall clock per call)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "simdjson-go_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, sjhip, workloads
ctx = sjhip.Context(0)
for copies in (1, 8, 53, 426):
    doc = workloads.c2_twitter_array(copies)
    n = len(doc)
    d = torch.empty(n + 256, dtype=torch.uint8, device="cuda:0"); d[:n].copy_(torch.frombuffer(bytearray(doc), dtype=torch.uint8))
    pos = torch.empty(workloads.c2_expected_structurals(copies) + 1024, dtype=torch.int32, device="cuda:0"); torch.cuda.synchronize()
    for _ in range(5): ctx.stage1_device(d.data_ptr(), n, pos.data_ptr(), pos.numel())
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(50): ctx.stage1_device(d.data_ptr(), n, pos.data_ptr(), pos.numel())
        best = min(best, (time.perf_counter() - t0) / 50)
    km = ctx.stage1_time(d.data_ptr(), n, pos.data_ptr(), pos.numel(), 20)
    print(f"x{copies}: {n} B  call {best*1e6:.1f} us  kernel {km*1e3:.1f} us  overhead {best*1e6-km*1e3:.1f} us")
