import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (os.path.join(ROOT, "simdjson-go_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import sjhip, fixtures
c = sjhip.Context(0)
for name in ("twitter", "canada", "twitterescaped", "parking-citations"):
    d = fixtures.load(name)
    nd = name.startswith("parking")
    for _ in range(5): c.parse(d, ndjson=nd)
    t0 = time.perf_counter(); N = 100
    for _ in range(N): c.parse(d, ndjson=nd)
    dt = (time.perf_counter() - t0) / N
    print(f"{name:20s} {len(d):9d} B  {dt*1e6:8.1f} us/parse  {len(d)/dt/1e9:6.2f} GB/s (host buffer -> tape on host)")

# ParseNDStream shape: parking-citations x400 (149 MB) from memory, 10 MiB blocks, host buffer -> tapes on host
import io
import workloads
big = workloads.c5_parking_nd(400)
for inflight in (1, 2, 4):
    list(sjhip.parse_nd_stream(io.BytesIO(big[: big.rfind(b"\n", 0, 30 << 20) + 1]), inflight=inflight))  # warm-up
    t0 = time.perf_counter()
    nblk = sum(1 for _ in sjhip.parse_nd_stream(io.BytesIO(big), inflight=inflight))
    dt = time.perf_counter() - t0
    print(f"parse_nd_stream inflight={inflight}: {len(big)} B in {nblk} blocks, {dt*1e3:7.1f} ms, {len(big)/dt/1e9:5.2f} GB/s")
    import queue
    back = queue.Queue()
    for rep in range(2):  # results handed back through `reuse` (second pass: every buffer recycled)
        t0 = time.perf_counter()
        for pj in sjhip.parse_nd_stream(io.BytesIO(big), inflight=inflight, reuse=back):
            back.put(pj)
        dt = time.perf_counter() - t0
    print(f"   ... with reuse:          {dt*1e3:7.1f} ms, {len(big)/dt/1e9:5.2f} GB/s")
