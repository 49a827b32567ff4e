"""first difference between the device parse and the oracle for a fixture: python tools/debug_diff.py <fixture> [nd]"""
import sys
sys.path.insert(0, "simdjson-go_amd"); sys.path.insert(0, "tests")
import numpy as np
import fixtures, sjhip, oracle_lib as O
name = sys.argv[1]
nd = len(sys.argv) > 2
d = fixtures.load(name)
ctx = sjhip.Context(0)
for copy in (True, False):
    ref = O.parse(d, ndjson=nd, copy_strings=copy)
    try:
        pj = ctx.parse(d, ndjson=nd, copy_strings=copy)
    except Exception as e:
        print(name, "copy" if copy else "nocopy", "EXC", repr(e)[:200], "oracle rc", ref.rc)
        continue
    t_ok = np.array_equal(pj.Tape, ref.tape)
    s_ok = np.array_equal(pj.Strings, ref.strings)
    print(name, "copy" if copy else "nocopy", "tape", len(pj.Tape), len(ref.tape), t_ok, "strings", len(pj.Strings), len(ref.strings), s_ok)
    if not t_ok and len(pj.Tape) == len(ref.tape):
        i = int(np.nonzero(pj.Tape != ref.tape)[0][0])
        print("  first tape diff at", i, [hex(int(x)) for x in pj.Tape[i-2:i+3]], [hex(int(x)) for x in ref.tape[i-2:i+3]], "ndiff", int((pj.Tape != ref.tape).sum()))
    if not s_ok:
        m = min(len(pj.Strings), len(ref.strings))
        nz = np.nonzero(pj.Strings[:m] != ref.strings[:m])[0]
        if len(nz):
            j = int(nz[0])
            print("  first strings diff at", j, bytes(pj.Strings[max(0,j-20):j+20]), bytes(ref.strings[max(0,j-20):j+20]), "ndiff", len(nz))
