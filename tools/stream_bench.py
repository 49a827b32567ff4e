"""ParseNDStream through the library (sjhip_stream_*): host memory -> tapes in host memory, configs[4] sized input
(parking-citations x COPIES), 10 MiB blocks.  The input is read (memmove) straight into the pinned blocks and every
result is copied out of pinned memory into preallocated arrays -- the work a Go caller does with its reader and
its `reuse`d ParsedJson.  Prints one JSON line."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "simdjson-go_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import sjhip  # noqa: E402
import workloads  # noqa: E402
from sjhip import _lib  # noqa: E402


_POOL = None


def pmemmove(dst, src, n, threads):
    """memmove split over `threads` host threads (ctypes releases the GIL): one thread moves ~10 GB/s, the PCIe link
    five times that -- a reader that fills the pinned block is the bottleneck of the pipeline unless it is parallel."""
    global _POOL
    if threads <= 1 or n < (1 << 20):
        C.memmove(dst, src, n)
        return
    if _POOL is None:
        import concurrent.futures
        _POOL = concurrent.futures.ThreadPoolExecutor(max_workers=16)
    part = (n + threads - 1) // threads
    futs = [_POOL.submit(C.memmove, dst + o, src + o, min(part, n - o)) for o in range(0, n, part)]
    for f in futs:
        f.result()


def open_stream(block=10 << 20, slots=0, n_devices=1):
    """A stream for `run`: a long-lived object (its contexts' arenas and pinned buffers grow on first use)."""
    L = sjhip.lib()
    cap = block + block // 8 + (64 << 10)
    h = L.sjhip_stream_create(0, n_devices, cap, slots, 0)
    assert h
    return h


def run(data, block=10 << 20, slots=0, n_devices=1, copy_out=True, copy_threads=1, fill=True, stream=None):
    L = sjhip.lib()
    n = len(data)
    src = np.frombuffer(data, dtype=np.uint8)
    cap = block + block // 8 + (64 << 10)
    h = stream or open_stream(block, slots, n_devices)
    out_t = np.empty(cap // 2, dtype=np.uint64)   # a reused ParsedJson: capacity for the largest block
    out_s = np.empty(cap, dtype=np.uint8)
    res = _lib.StreamResult()
    off = 0
    blocks_sent = 0
    nslots = L.sjhip_stream_slots(h)
    held = {}
    tape_words = strings = blocks = records = 0
    t0 = time.perf_counter()

    def take():
        nonlocal tape_words, strings, blocks, records
        rc = L.sjhip_stream_next(h, C.byref(res))
        if rc in (7, 8):
            return False
        assert rc == 0, (rc, L.sjhip_stream_last_error(h))
        if copy_out:
            pmemmove(out_t.ctypes.data, res.tape, res.tape_len * 8, copy_threads)
            pmemmove(out_s.ctypes.data, res.strings, res.strings_len, copy_threads)
        tape_words += res.tape_len
        strings += res.strings_len
        records += res.records
        blocks += 1
        L.sjhip_stream_release(h)
        return True

    ptr, c = C.c_void_p(), C.c_size_t()
    while off < n:
        rc = L.sjhip_stream_acquire(h, C.byref(ptr), C.byref(c))
        if rc == 6:
            take()
            continue
        assert rc == 0
        end = min(n, off + block)
        if end < n and data[end - 1:end] != b"\n":
            nl = data.find(b"\n", end)
            end = n if nl < 0 else nl + 1
        slot = blocks_sent % nslots
        if fill or held.get(slot) != end - off:  # fill=False: identical blocks are filled once (ceiling of the pipeline)
            pmemmove(ptr.value, src.ctypes.data + off, end - off, copy_threads)
            held[slot] = end - off
        blocks_sent += 1
        assert L.sjhip_stream_submit(h, end - off) == 0
        off = end
    while take():
        pass
    dt = time.perf_counter() - t0
    if stream is None:
        L.sjhip_stream_destroy(h)
    return {"bytes": n, "blocks": blocks, "seconds": round(dt, 4), "GBps": round(n / dt / 1e9, 2),
            "tape_words": tape_words, "strings_bytes": strings, "matching_records": records, "slots": L.sjhip_stream_slots(h) if stream else slots,
            "devices": n_devices,
            "copy_out": copy_out, "copy_threads": copy_threads, "fill": fill,
            "output_bytes_per_input_byte": round((tape_words * 8 + strings) / n, 3)}


if __name__ == "__main__":
    copies = int(os.environ.get("COPIES", "1000"))
    data = workloads.c5_parking_nd(copies)
    L = sjhip.lib()
    out = []
    for slots in (3, 4, 6):
        h = open_stream(slots=slots)
        run(data, stream=h, copy_out=False)  # warm-up: every arena and pinned buffer at its final size
        for copy_out, threads, fill in ((True, 1, True), (True, 4, True), (False, 4, True), (False, 1, False)):
            blk = (10 << 20) if fill else 28 * (len(data) // copies)  # fill=False: every block the same 28 files
            if not fill:
                run(data, block=blk, stream=h, copy_out=False, fill=True)
            out.append(run(data, block=blk, copy_out=copy_out, copy_threads=threads, fill=fill, stream=h))
            print(json.dumps(out[-1]), flush=True)
        L.sjhip_stream_destroy(h)
    print(json.dumps({"stream": out}))
