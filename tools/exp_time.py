"""Same-process A/B of SJ_EXP bit masks on a correct-results basis (the sweep-order bits): the whole parse of the BASELINE
workloads (device-resident) under each mask in turn, ROUNDS times; tape and Strings.B of every mask are compared with mask 0.
SJHIP_LIB=build_ab/libsjhip_exp.so python tools/exp_time.py <mask> <mask> ...   (WORKLOADS=twitter,parking,twitter1g; MODE=copy|nocopy)"""
import hashlib, os, sys, time
sys.path.insert(0, "simdjson-go_amd"); sys.path.insert(0, "tests")
import torch, sjhip, workloads
masks = [int(a, 0) for a in sys.argv[1:]] or [0]
rounds = int(os.environ.get("ROUNDS", "3"))
copy = os.environ.get("MODE", "copy") == "copy"
ctx = sjhip.Context(0)
for name in os.environ.get("WORKLOADS", "twitter,parking").split(","):
    if name == "twitter": doc, nd = workloads.c2_twitter_array(426), False
    elif name == "twitter1g": doc, nd = workloads.c2_twitter_array(1700), False
    else: doc, nd = workloads.c5_parking_nd(1000).rstrip(b"\n"), True
    d = torch.empty(len(doc) + 256, dtype=torch.uint8, device="cuda:0"); d[:len(doc)].copy_(torch.frombuffer(bytearray(doc), dtype=torch.uint8)); torch.cuda.synchronize()
    ref = None
    for m in masks:  # results
        os.environ["SJHIP_EXP"] = str(m)
        tl, sl = ctx.parse_device(d.data_ptr(), len(doc), ndjson=nd, copy_strings=copy)
        tape, strings = ctx.fetch(tl, sl)
        h = hashlib.sha1(bytes(memoryview(tape).cast("B"))).hexdigest()[:12] + "/" + hashlib.sha1(bytes(strings)).hexdigest()[:12]
        if ref is None: ref = h
        print(f"{name} mask {m:#x}: tape {tl} strings {sl} {h} {'same' if h == ref else 'DIFFERENT'}", flush=True)
        del tape, strings
    best = {m: 1e9 for m in masks}
    for r in range(rounds):
        for m in masks:
            os.environ["SJHIP_EXP"] = str(m)
            for _ in range(2): ctx.parse_device(d.data_ptr(), len(doc), ndjson=nd, copy_strings=copy)
            t0 = time.perf_counter()
            for _ in range(10): ctx.parse_device(d.data_ptr(), len(doc), ndjson=nd, copy_strings=copy)
            best[m] = min(best[m], (time.perf_counter() - t0) / 10)
    print(name, "copy" if copy else "nocopy", "; ".join(f"{m:#x}: {best[m]*1e3:.3f} ms" for m in masks), flush=True)
    del d
