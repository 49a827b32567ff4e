"""GPU: queries on the device-resident tape (sjhip_count_where / sjhip_filter_where) against
a host evaluation of the reference's countWhere (ndjson_test.go:421-471: Object.FindKey on every record's root object,
then a string compare) and against the oracle's ParseND of the matching lines."""
import json
import random
import struct

import numpy as np
import pytest

import fixtures
import golden_util as GU
import oracle_lib as O
from test_gpu_parse import ctx  # noqa: F401

pytestmark = pytest.mark.gpu


def host_matches(line, key, value):
    """FindKey(key) on the root object (first occurrence, top level), value must be a string equal to `value`."""
    try:
        pairs = json.loads(line, object_pairs_hook=lambda p: p)
    except ValueError:
        return False
    if not isinstance(pairs, list) or (pairs and not isinstance(pairs[0], tuple)):
        return False  # the root is an array (or an empty object parsed as [])
    for k, v in pairs:
        if k == key:
            return isinstance(v, str) and v == value
    return False


def check_query(ctx, doc, key, value, copy=True):
    lines = [l for l in doc.split(b"\n") if l.strip()]
    want = [l for l in lines if host_matches(l.decode("utf-8"), key.decode(), value.decode())]
    ctx.parse(doc, ndjson=True, copy_strings=copy)
    assert ctx.count_where(key, value) == len(want)
    if not copy:
        return
    n, pj = ctx.filter_where(key, value)
    assert n == len(want)
    if not want:
        assert len(pj.Tape) == 0 and len(pj.Strings) == 0
        return
    ref = O.parse(b"\n".join(want), ndjson=True, copy_strings=True)
    assert ref.rc == 0
    assert np.array_equal(pj.Tape, ref.tape), (key, value)
    assert np.array_equal(pj.Strings, ref.strings), (key, value)


def test_parking_citations_hond(ctx):  # ndjson_test.go:250-267: 116
    park = fixtures.load("parking-citations")
    ctx.parse(park, ndjson=True)
    assert ctx.count_where(b"Make", b"HOND") == GU.load("stage2")["parking_citations_hond"] == 116
    check_query(ctx, park, b"Make", b"HOND")
    check_query(ctx, park, b"Make", b"HOND", copy=False)
    check_query(ctx, park, b"Color", b"WH")
    check_query(ctx, park, b"Make", b"NO SUCH MAKE")
    check_query(ctx, park * 7, b"RP State Plate", b"CA")


def test_record_shapes(ctx):
    """first-occurrence semantics, nested keys that must not match, non-string values, array roots, escapes (the
    comparison sees unescaped bytes), and numbers whose raw word looks like a tag (the copy classifies tag / raw words
    by position parity)"""
    def dbl(bits):
        return repr(struct.unpack("<d", struct.pack("<Q", bits))[0])
    tricky = [dbl((ord(c) << 56) | 0x000123456789ab) for c in '"lud{[}]rtfn']
    recs = [
        b'{"k":"v"}',
        b'{"k":"x","k":"v"}',                       # first occurrence decides: no match
        b'{"k":"v","k":"x"}',                       # match
        b'{"a":{"k":"v"},"b":[{"k":"v"}]}',         # nested only: no match
        b'{"a":{"k":"x"},"k":"v","z":[1,2,{"k":3}]}',
        b'{"k":1}', b'{"k":null}', b'{"k":["v"]}', b'{"k":{"v":"v"}}',
        b'[{"k":"v"}]', b'[]', b'{}', b'[1,2,3]',
        b'{"k":"\\u0076"}',                          # "v" through an escape
        b'{"\\u006b":"v"}',                          # "k" through an escape
        b'{"kk":"v","k":"vv","k ":"v"}',
        ('{"n":[' + ",".join(tricky) + '],"k":"v","m":' + tricky[0] + "}").encode(),
        ('{"big":-9223372036854775808,"u":18446744073709551615,"k":"v","d":' + tricky[1] + "}").encode(),
        b'{"s":"' + b"x" * 300 + b'","k":"v"}',
        b'{"k":"v","t":true,"f":false,"n":null}',
    ]
    rnd = random.Random(3)
    for trial in range(6):
        rnd.shuffle(recs)
        doc = b"\n".join(recs * (1 + trial * 40)) + (b"\n" if trial & 1 else b"")
        check_query(ctx, doc, b"k", b"v")
        check_query(ctx, doc, b"k", b"vv")
        check_query(ctx, doc, b"s", b"x" * 300 if trial & 1 else b"x" * 299)
    # one record without a newline, and a plain (non-ND) document: one record
    check_query(ctx, b'{"k":"v"}', b"k", b"v")
    ctx.parse(b'{"k":"v","z":[1,2]}', ndjson=False)
    assert ctx.count_where(b"k", b"v") == 1 and ctx.count_where(b"z", b"v") == 0


def test_random_records(ctx):
    from test_gpu_parse import _random_records
    rnd, lines = _random_records(77, 3 << 20)
    keys = [b"k0", b"k1", b"k2", b"k4", b"zz"]
    doc = "\n".join(lines).encode("utf-8")
    for key in keys:
        vals = set()
        for l in lines[:400]:
            try:
                v = json.loads(l)
            except ValueError:
                continue
            if isinstance(v, dict) and isinstance(v.get(key.decode()), str):
                try:
                    v[key.decode()].encode("utf-8")
                except UnicodeEncodeError:
                    continue  # a lone surrogate (quirk Q2 bytes): not expressible as a query value here
                vals.add(v[key.decode()])
        for value in list(vals)[:3] + ["nope"]:
            check_query(ctx, doc, key, value.encode("utf-8"))


def test_full_size_count(ctx):  # 116 per file -> 116 000 on configs[4]
    import workloads
    nd = workloads.c5_parking_nd(1000)
    ctx.parse(nd, ndjson=True)
    assert ctx.count_where(b"Make", b"HOND") == 116_000
    n, pj = ctx.filter_where(b"Make", b"HOND", fetch=False)
    assert n == 116_000
    # the filtered (Tape, Strings.B) at full size (977 tiles of the record scans): the matching lines repeat with the
    # file, so the oracle parses the 116 of one file x 1000
    park = fixtures.load("parking-citations")
    want = [l for l in park.split(b"\n") if l.strip() and host_matches(l.decode("utf-8"), "Make", "HOND")]
    assert len(want) == 116
    ref = O.parse(b"\n".join(want * 1000), ndjson=True, copy_strings=True)
    assert ref.rc == 0
    n, pj = ctx.filter_where(b"Make", b"HOND")
    assert np.array_equal(pj.Tape, ref.tape)
    assert np.array_equal(pj.Strings, ref.strings)


def test_filtered_stream():
    """ParseNDStream composed with the filter (sjhip_stream_set_filter): every block delivers ParseND of its matching lines
    only; countWhere of the reference's test (ndjson_test.go:250-267: 116 matches per file) adds up over the blocks."""
    import io
    import sjhip
    park = fixtures.load("parking-citations")
    stream = park * 9
    bs = 1 << 20
    blocks = list(sjhip.cut_blocks(io.BytesIO(stream), bs))
    got = list(sjhip.parse_nd_stream(io.BytesIO(stream), block_size=bs, inflight=3, where=(b"Make", b"HOND")))
    assert len(got) == len(blocks)
    assert sum(pj.records for pj in got) == 116 * 9
    for pj, blk in zip(got, blocks):
        lines = [ln for ln in blk.split(b"\n") if b'"Make":"HOND"' in ln]
        assert pj.records == len(lines)
        if not lines:
            assert len(pj.Tape) == 0
            continue
        ref = O.parse(b"\n".join(lines), ndjson=True, copy_strings=True)
        assert ref.rc == 0
        assert np.array_equal(pj.Tape, ref.tape) and np.array_equal(pj.Strings, ref.strings)
    # without a match anywhere
    none = list(sjhip.parse_nd_stream(io.BytesIO(stream), block_size=bs, inflight=2, where=(b"Make", b"no such make")))
    assert len(none) == len(blocks) and all(pj.records == 0 and len(pj.Tape) == 0 for pj in none)


# ---- paths, typed values, key sets (sjhip_find_path / _count_where_path / _project_keys) against the restated walks -------
import query_walk as Q  # noqa: E402
import test_query_walk as TQ  # noqa: E402


def _walk_of(pj):
    return Q.Walk(pj.Tape, pj.Strings, pj.Message)


def check_paths(ctx, doc, nd, paths, copy=True):
    pj = ctx.parse(doc, ndjson=nd, copy_strings=copy)
    w = _walk_of(pj)
    roots = w.records()
    for path in paths:
        got = ctx.find_path(*path)
        want = np.array([w.find_path(r, list(path)) for r in roots], dtype=np.uint64)
        assert np.array_equal(got, want), (path, nd, copy, got[:5], want[:5])
    return w, roots


def test_find_path_tables_of_the_reference(ctx):
    # TestObject_FindPath (parsed_object_test.go:10-132) on the device, both copy modes
    for copy in (True, False):
        w, (root,) = check_paths(ctx, TQ.FINDPATH_INPUT, False, [tuple(p.encode() for p in path) for path, _ in TQ.FINDPATH_CASES], copy)
        for path, want in TQ.FINDPATH_CASES:
            (v,) = ctx.find_path(*[p.encode() for p in path])
            if want is None:
                assert v == Q.NOT_FOUND
            else:
                assert TQ.value_at(w, int(v)) == want
        assert ctx.find_path(b"Alt", b"x")[0] == Q.NOT_OBJECT
        assert ctx.find_path(b"Image", b"IDs", b"0")[0] == Q.NOT_OBJECT
    # a root that is not an object: "type ... found before object was found"
    ctx.parse(b"[1,2,3]")
    assert ctx.find_path(b"a")[0] == Q.NOT_OBJECT


def test_project_keys_table_of_the_reference(ctx):
    # TestObject_ForEach with onlyKeys (parsed_object_test.go:134-240)
    pj = ctx.parse(TQ.FOREACH_INPUT)
    w = _walk_of(pj)
    for keys, want in TQ.FOREACH_CASES:
        got = ctx.project_keys([k.encode() for k in keys])
        assert got.shape == (1, len(keys))
        found = {}
        for e in got[0]:
            e = int(e)
            if e != 0xFFFFFFFFFFFFFFFF:
                found[keys[e >> 56]] = TQ.value_at(w, e & Q.MASK)
        assert found == want, keys
    with pytest.raises(Exception):
        ctx.project_keys([b"key1", b"key1"])  # a set: equal keys are refused


def test_paths_on_records(ctx):
    """every record of an ND message: nested paths, duplicate keys (the first member wins), type errors, escaped keys"""
    rnd = random.Random(5)
    lines = []
    for i in range(3000):
        kind = rnd.randrange(8)
        if kind == 0:
            lines.append(b'{"a":{"b":{"c":%d}},"x":"y"}' % i)
        elif kind == 1:
            lines.append(b'{"a":{"b":[1,2,{"c":3}]},"a":{"b":{"c":7}}}')       # the first "a" wins; its "b" is an array
        elif kind == 2:
            lines.append(b'{"a":{"b":{"c":null,"c":5}}}')
        elif kind == 3:
            lines.append(b'[{"a":{"b":{"c":1}}}]')                                # the root is an array
        elif kind == 4:
            lines.append(b'{"a":"str","k\\u00e9y":{"c":true}}')
        elif kind == 5:
            lines.append(b'{"zz":[%s],"a":{"bb":1,"b":{"cc":2,"c":%d.5}}}' % (b",".join(b"{}" for _ in range(rnd.randrange(5))), i))
        elif kind == 6:
            lines.append(b'{}')
        else:
            lines.append(b'{"a":{"b":{"c":"' + bytes(rnd.choice(b"abc\\n") for _ in range(0)) + b'v%d"}}}' % (i % 7))
    doc = b"\n".join(lines)
    paths = [(b"a",), (b"a", b"b"), (b"a", b"b", b"c"), (b"x",), ("kéy".encode(), b"c"), (b"missing",), (b"a", b"b", b"c", b"d")]
    for copy in (True, False):
        w, roots = check_paths(ctx, doc, True, paths, copy)
        # typed comparisons at the end of a path, against the restated Iter conversions
        for op, val in ((Q.OP_EXISTS, None), (Q.OP_EQ_INT, 5), (Q.OP_EQ_INT, 7), (Q.OP_EQ_UINT, 7), (Q.OP_EQ_FLOAT, 7.0), (Q.OP_EQ_FLOAT, 12.5),
                        (Q.OP_EQ_STRING, b"v3"), (Q.OP_IS_NULL, None), (Q.OP_EQ_BOOL, True)):
            for path in ((b"a", b"b", b"c"), ("kéy".encode(), b"c")):
                want = 0
                for r in roots:
                    v = w.find_path(r, list(path))
                    want += v < Q.NOT_OBJECT and w.element_is(v, op, val)
                assert ctx.count_where_path(path, op, val) == want, (path, op, val, copy)
        keys = [b"a", b"x", b"zz"]
        got = ctx.project_keys(keys)
        assert got.shape == (len(roots), 3)
        for r, row in zip(roots, got):
            want = w.project_keys(r, keys)
            have = [(int(e) >> 56, int(e) & Q.MASK) for e in row if int(e) != 0xFFFFFFFFFFFFFFFF]
            assert have == want


def test_paths_on_fixtures(ctx):
    park = fixtures.load("parking-citations")
    check_paths(ctx, park, True, [(b"Make",), (b"Fine amount",), (b"nope",), (b"Make", b"x")])
    ctx.parse(park, ndjson=True)
    assert ctx.count_where_path((b"Make",), ctx.OP_EQ_STRING, b"HOND") == ctx.count_where(b"Make", b"HOND") == 116
    tw = fixtures.load("twitter")
    w, (root,) = check_paths(ctx, tw, False, [(b"search_metadata", b"count"), (b"search_metadata", b"max_id_str"), (b"statuses",), (b"statuses", b"0")])
    (v,) = ctx.find_path(b"search_metadata", b"count")
    assert TQ.value_at(w, int(v)) == 100
    assert ctx.count_where_path((b"search_metadata", b"count"), ctx.OP_EQ_INT, 100) == 1
    assert ctx.count_where_path((b"search_metadata", b"count"), ctx.OP_EQ_FLOAT, 100.0) == 1
    assert ctx.count_where_path((b"search_metadata", b"count"), ctx.OP_EQ_UINT, 99) == 0
    # floats at the edges of Iter.Int / Iter.Uint (exactly 2^63 / 2^64 pass the reference's range checks: amd64 conversion results)
    ctx.parse(b'{"a":9223372036854775808.0,"b":18446744073709551616.0,"c":18446744073709555000.0}')
    assert ctx.count_where_path((b"a",), ctx.OP_EQ_INT, -(2 ** 63)) == 1 and ctx.count_where_path((b"a",), ctx.OP_EQ_UINT, 2 ** 63) == 1
    assert ctx.count_where_path((b"b",), ctx.OP_EQ_UINT, 0) == 1 and ctx.count_where_path((b"b",), ctx.OP_EQ_INT, 0) == 0
    assert ctx.count_where_path((b"c",), ctx.OP_EQ_UINT, 0) == 0
