import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "simdjson-go_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, sjhip, oracle_lib as O
ctx = sjhip.Context(0)
docs = [b'["abc"]', b'["a\\nb"]', b'["\\u00e9"]', b'["x\\u00e9y","z"]', b'["\\ud83d\\ude00"]', b'{"k":"v\\/w","n":"\\u20ac"}',
        b'["' + b"a" * 70 + b'\\u00e9' + b"b" * 10 + b'"]', b'["' + b"a" * 58 + b'\\u00e9' + b"b" * 10 + b'","q"]',
        b'["' + b"a" * 61 + b'\\u00e9' + b"b" * 10 + b'","q"]', b'[ "a" ,\n   "\\u00e9\\u00e9" ]']
for d in docs:
    ref = O.parse(d, ndjson=False, copy_strings=True)
    try:
        pj = ctx.parse(d, ndjson=False, copy_strings=True)
        ok = np.array_equal(pj.Tape, ref.tape) and np.array_equal(pj.Strings, ref.strings)
        print(ok, d[:60], [hex(int(x)) for x in pj.Tape] if not ok else "", bytes(pj.Strings) if not ok else "", bytes(ref.strings) if not ok else "")
    except Exception as e:
        print("EXC", d[:60], e, ref.rc)
