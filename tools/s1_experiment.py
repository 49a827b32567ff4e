"""Stage-1 A/B on the GPU box: every kernel variant on configs[1] (twitter x426, 256.6 MiB) and on a >1 GiB
document (twitter x1700: outside the 256 MiB Infinity Cache), kernel-only time with hipEvents, output compared
between variants on the device, and the per-phase s_memtime timeline of the barrier (1) and barrier-free (3)
kernels.  Writes gpurun_out/s1_experiment.json and gpurun_out/s1_trace_v*.npz."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "simdjson-go_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import sjhip  # noqa: E402
import workloads  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
L = sjhip.lib()
ctx = sjhip.Context(0)
VARIANTS = [int(x) for x in os.environ.get("VARIANTS", "1 3 4").split()]
NAMES = {0: "512x2 barrier", 1: "1024x2 barrier", 2: "768x2 barrier", 3: "1024x2 barrier-free depth 2", 4: "1024x2 barrier-free depth 3"}
report = {"variants": NAMES, "runs": []}


def device_doc(copies):
    doc = workloads.c2_twitter_array(copies)
    n = len(doc)
    d = torch.empty(n + 256, dtype=torch.uint8, device="cuda:0")
    d[:n].copy_(torch.frombuffer(bytearray(doc), dtype=torch.uint8))
    torch.cuda.synchronize()
    return d, n


def trace(d_msg, n, d_pos, v):
    tiles, waves, words = C.c_uint(0), C.c_int(0), C.c_int(0)
    cap = (n // (512 * 2 * 64) + 2) * 16 * 8 + 1024
    buf = np.zeros(cap, dtype=np.uint64)
    rc = L.sjhip_stage1_trace(ctx._h, C.c_void_p(d_msg.data_ptr()), n, C.c_void_p(d_pos.data_ptr()), d_pos.numel(),
                              buf.ctypes.data, cap, C.byref(tiles), C.byref(waves), C.byref(words))
    assert rc == 0, (rc, ctx.last_error())
    t = buf[: tiles.value * waves.value * words.value].reshape(tiles.value, waves.value, words.value).astype(np.int64)
    return t


def summarize(path):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import s1_timeline
    return s1_timeline.analyse(path)


for copies in [int(x) for x in os.environ.get("COPIES", "426 1700").split()]:
    d_msg, n = device_doc(copies)
    expect = workloads.c2_expected_structurals(copies)
    d_pos = torch.empty(expect + 1024, dtype=torch.int32, device="cuda:0")
    ref_pos = None
    for v in VARIANTS:
        assert L.sjhip_stage1_set_variant(v) == v
        d_pos.zero_()
        ok, cnt = ctx.stage1_device(d_msg.data_ptr(), n, d_pos.data_ptr(), d_pos.numel())
        same = None
        if ref_pos is None:
            ref_pos = d_pos.clone()
        else:
            same = bool(torch.equal(ref_pos[:expect], d_pos[:expect]))
        ctx.stage1_time(d_msg.data_ptr(), n, d_pos.data_ptr(), d_pos.numel(), 5)
        ms = [ctx.stage1_time(d_msg.data_ptr(), n, d_pos.data_ptr(), d_pos.numel(), 20) for _ in range(3)]
        run = {"copies": copies, "bytes": n, "variant": v, "name": NAMES[v], "ok": bool(ok), "count_ok": cnt == expect,
               "same_positions_as_first_variant": same, "kernel_ms": [round(x, 4) for x in ms],
               "input_GBps": round(n / min(ms) / 1e6, 1), "algo_GBps": round((n + 4 * expect) / min(ms) / 1e6, 1)}
        if v in (1, 3, 4) and copies == 426:
            t = trace(d_msg, n, d_pos, v)
            path = os.path.join(OUT, f"s1_trace_v{v}.npz")
            np.savez_compressed(path, trace=t)
            run["timeline"] = summarize(path)
        print(json.dumps(run), flush=True)
        report["runs"].append(run)
    del d_msg, d_pos, ref_pos
    torch.cuda.empty_cache()
L.sjhip_stage1_set_variant(-1)
with open(os.path.join(OUT, "s1_experiment.json"), "w") as f:
    json.dump(report, f, indent=1)
