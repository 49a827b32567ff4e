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
VARIANTS = [int(x) for x in os.environ.get("VARIANTS", "1 3 4 0 2").split()]
NAMES = {0: "512 barrier", 1: "1024 barrier", 2: "768 barrier", 3: "1024 barrier-free", 4: "512 barrier-free"}
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


def summarize(t, kernel_ms):
    t0 = t[:, :, 0][t[:, :, 0] > 0].min()
    span = t[:, :, 4].max() - t0
    a = (t[:, :, 1] - t[:, :, 0])
    wait = (t[:, :, 3] - t[:, :, 1])          # arrival -> state known (barrier wait + serial section, or the flag poll)
    fl = (t[:, :, 4] - t[:, :, 3])
    ser = t[:, :, 2] - t[:, :, 1]
    ser = ser[t[:, :, 2] > 0]
    tiles, waves = t.shape[0], t.shape[1]
    # the wave timeline is covered by A(next) , wait, flatten(cur): busy = A + flatten of all tiles
    busy = float(a.sum() + fl.sum())
    wave_time = float(span) * 256 * waves if tiles >= 256 else float(span) * tiles * waves
    return {"ticks_per_us": round(float(span) / (kernel_ms * 1e3), 2), "span_ticks": int(span),
            "phaseA_mean": round(float(a.mean()), 1), "phaseA_p95": float(np.percentile(a, 95)),
            "wait_mean": round(float(wait.mean()), 1), "wait_p50": float(np.percentile(wait, 50)),
            "wait_p95": float(np.percentile(wait, 95)),
            "serial_mean": round(float(ser.mean()), 1) if ser.size else None,
            "serial_p95": float(np.percentile(ser, 95)) if ser.size else None,
            "flatten_mean": round(float(fl.mean()), 1), "flatten_p95": float(np.percentile(fl, 95)),
            "busy_fraction_of_resident_wave_time": round(busy / wave_time, 4)}


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
            np.savez_compressed(os.path.join(OUT, f"s1_trace_v{v}.npz"), trace=t)
            run["timeline"] = summarize(t, min(ms))
        print(json.dumps(run), flush=True)
        report["runs"].append(run)
    del d_msg, d_pos, ref_pos
    torch.cuda.empty_cache()
L.sjhip_stage1_set_variant(-1)
with open(os.path.join(OUT, "s1_experiment.json"), "w") as f:
    json.dump(report, f, indent=1)
