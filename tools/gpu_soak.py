"""Time-boxed randomized parity run on the GPU against the oracle (fresh seeds every run unless one is given):
    python tools/gpu_soak.py [seconds] [seed]
Generated records (tests/test_gpu_parse._random_value: nested containers, every escape form, numbers of all shapes) as
single documents, arrays and NDJSON, plus byte mutations of them (flipped, deleted, duplicated bytes: mostly invalid
documents -- the verdict must match), and every eighth round a document of 4-7 MB of a random token density (large documents are
laid out for the density the context has learned: the path depends on the order) -- each through Parse / ParseND in both copy
modes, the in-place view, the key flags,
MarshalJSON (one pass and two) and the serializer stream, compared with oracle/ bit for bit.  Prints a summary line;
exit code 1 and the offending document (hex) on the first difference."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "simdjson-go_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import oracle_lib as O  # noqa: E402
import sjhip  # noqa: E402
from test_gpu_parse import _random_value  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rnd = random.Random(seed)
ctx = sjhip.Context(0)
stats = {"docs": 0, "valid": 0, "invalid": 0, "bytes": 0, "marshal": 0, "serialize": 0}


def fail(what, doc, nd, copy):
    print("MISMATCH", what, "nd=%s copy=%s seed=%d len=%d" % (nd, copy, seed, len(doc)))
    print(bytes(doc[:4000]).hex())
    sys.exit(1)


def check(doc, nd):
    stats["docs"] += 1
    stats["bytes"] += len(doc)
    for copy in (True, False):
        ref = O.parse(doc, ndjson=nd, copy_strings=copy)
        kf = rnd.random() < 0.5
        view = rnd.random() < 0.5
        try:
            pj = ctx.parse(doc, ndjson=nd, copy_strings=copy, view=view, key_flags=kf)
            rc = 0
        except sjhip.ParseError as e:
            rc, pj = e.code, None
        if rc != ref.rc:
            fail("verdict %d vs %d" % (rc, ref.rc), doc, nd, copy)
        if rc != 0:
            stats["invalid"] += 1
            continue
        stats["valid"] += 1
        if not (np.array_equal(pj.Tape, ref.tape) and np.array_equal(pj.Strings, ref.strings)):
            fail("tape / strings (view=%s kf=%s)" % (view, kf), doc, nd, copy)
        msg = bytes(doc[ref.msg_off:ref.msg_off + ref.msg_len])
        if rnd.random() < 0.5:
            mrc, want = O.marshal_json(ref.tape, ref.strings, msg)
            try:
                got = ctx.marshal_json()
                grc = 0
            except sjhip.ParseError:
                got, grc = None, 1
            if (mrc != 0) != (grc != 0) or (mrc == 0 and got != want):
                fail("marshal (kf=%s)" % kf, doc, nd, copy)
            stats["marshal"] += 1
        elif copy:
            want = O.serialize(ref.tape, ref.strings, msg, dedup=False)[0]
            got = ctx.serialize()
            if len(got) != len(want) or not np.array_equal(got, want):
                fail("serialize", doc, nd, copy)
            stats["serialize"] += 1


def mutate(doc):
    b = bytearray(doc)
    for _ in range(rnd.randrange(1, 4)):
        if not b:
            break
        i = rnd.randrange(len(b))
        k = rnd.randrange(5)
        if k == 0:
            b[i] = rnd.choice(b'"\\{}[],: \n\t0-e.u')
        elif k == 1:
            del b[i]
        elif k == 2:
            b.insert(i, b[i])
        elif k == 3:
            b[i] ^= 1 << rnd.randrange(8)
        else:
            del b[i:i + rnd.randrange(1, 64)]
    return bytes(b)


def large_document():
    """A document beyond SJHIP_SMALL_BYTES (4 MiB): it is parsed without a host round trip between the stages, laid out for the
    token density the context learned from its earlier LARGE parses (parse_api.hip) -- so which path a document takes depends on
    the documents before it.  Densities from 0.002 to 1 token per byte, in random order on the one context of this run."""
    kind = rnd.randrange(5)
    target = rnd.randrange(4200 << 10, 7 << 20)
    if kind == 0:    # long strings: ~0.002 tokens per byte
        item, sep, nd = '"%s"' % ("s" * rnd.randrange(500, 1500)), ",", False
    elif kind == 1:  # one token per byte
        item, sep, nd = rnd.choice(["1", "[]", "0"]), ",", False
    elif kind == 2:  # generated records as NDJSON
        item, sep, nd = None, "\n", True
    elif kind == 3:  # generated values in one array
        item, sep, nd = None, ",", False
    else:            # escape-free records (WithCopyStrings(false) copies nothing: its own path, stage2.hip no_escapes)
        item, sep, nd = '{"k":"plain value %d","n":[1,2.5,true,null],"o":{"p":"q"}}' % rnd.randrange(1000), "\n", True
    parts, size = [], 0
    while size < target:
        v = item
        if v is None:
            v = _random_value(rnd, 0)
            if v[0] not in "[{":
                v = "[" + v + "]"
        parts.append(v)
        size += len(v) + 1
    body = sep.join(parts)
    return (body if nd else "[" + body + "]").encode("utf-8"), nd


stats["large"] = 0
t_end = time.time() + budget
rounds = 0
while time.time() < t_end:
    rounds += 1
    if rounds % 8 == 0:
        doc, nd = large_document()
        check(doc, nd)
        stats["large"] += 1
        continue
    n = rnd.choice([1, 1, 3, 20, 200, 3000])
    vals = []
    for _ in range(n):
        v = _random_value(rnd, 0)
        if v[0] not in "[{":
            v = "[" + v + "]"
        vals.append(v.replace(",", "," + rnd.choice(["", " ", "\t", "\r\n "])))
    shape = rnd.randrange(3)
    if shape == 0:
        doc, nd = ("[" + ",".join(vals) + "]").encode("utf-8"), False
    elif shape == 1:
        doc, nd = ("\n".join(vals) + rnd.choice(["", "\n", "\n \n"])).encode("utf-8"), True
    else:
        doc, nd = ('{"a":' + vals[0] + ',"b":[' + ",".join(vals[1:]) + "]}").encode("utf-8"), False
    doc = rnd.choice([b"", b" ", b"\n\t"]) + doc
    check(doc, nd)
    for _ in range(3):
        check(mutate(doc), nd)
print("soak ok: seed %d, %.0f s, %s" % (seed, budget, stats))
