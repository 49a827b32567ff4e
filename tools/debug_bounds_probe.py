"""debug build: parse the given documents and print bounds-check reports: SJHIP_LIB=.../libsjhip_dbg.so python tools/debug_bounds_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "simdjson-go_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import sjhip
import golden_util as GU
ctx = sjhip.Context(0)
print("selftest", sjhip.lib().sjhip_debug_bounds_selftest())
docs = [b"[12a]", b'["a"]', b'[1]', b'{"a":"\\u00e9"}']
corp = GU.load("corpus")
docs += [bytes.fromhex(c["js_hex"]) for c in corp["fail_cases"] + corp["pass_cases"]]
seen = set()
for d in docs:
    for copy in (True, False):
        for nd in (False, True):
            try:
                ctx.parse(d, ndjson=nd, copy_strings=copy)
            except sjhip.ParseError as e:
                if e.code not in (1, 2):
                    key = str(e)[:160]
                    if key not in seen:
                        seen.add(key)
                        print(repr(d[:60]), "copy" if copy else "nocopy", "nd" if nd else "", "->", e.code, str(e))
