"""One mid-size parse against the oracle (diagnostics): python tools/parse_check.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "simdjson-go_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import sjhip, workloads, fixtures
import oracle_lib as O
ctx = sjhip.Context(0)
for name, doc, nd in (("twitter", fixtures.load("twitter"), False), ("twitter x9", workloads.c2_twitter_array(9), False),
                      ("parking x30", workloads.c5_parking_nd(30), True)):
    ref = O.parse(doc, ndjson=nd, copy_strings=True)
    try:
        pj = ctx.parse(doc, ndjson=nd, copy_strings=True)
        same = np.array_equal(pj.Tape, ref.tape) and np.array_equal(pj.Strings, ref.strings)
        print(name, "OK" if same else f"DIFFERENT tape {len(pj.Tape)} vs {len(ref.tape)}", flush=True)
    except sjhip.ParseError as e:
        print(name, "ParseError", e, flush=True)
