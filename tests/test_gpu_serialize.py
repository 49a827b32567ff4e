"""GPU: Serializer.Serialize on the device (sjhip_serialize).  The framed stream must be
byte-identical to the oracle's Serialize without de-duplication hits (oracle/sjo_serialize.c) and must deserialize
(oracle Deserialize = parsed_serialize.go:466-695) to the document -- the property the reference's tests pin
(parsed_serialize_test.go:220-340)."""
import random
import struct

import numpy as np
import pytest

import fixtures
import golden_util as GU
import oracle_lib as O
import tape_reader
from test_gpu_parse import ctx  # noqa: F401

pytestmark = pytest.mark.gpu


def check_serialize(ctx, data, nd, what):
    ref = O.parse(data, ndjson=nd, copy_strings=True)
    assert ref.rc == 0, what
    msg = bytes(data[ref.msg_off:ref.msg_off + ref.msg_len])
    ctx.parse(data, ndjson=nd, copy_strings=True)
    stream = ctx.serialize()
    want, tags, vals, sbuf = O.serialize(ref.tape, ref.strings, msg, dedup=False)
    assert len(stream) == len(want), (what, len(stream), len(want))
    assert np.array_equal(stream, want), (what, np.nonzero(stream != want)[0][:5])
    rc, tape2, strs2, msg2 = O.deserialize(stream)
    assert rc == 0, what
    if len(ref.tape) < 400000:
        a = tape_reader.to_python(ref.tape, ref.strings, msg)
        b = tape_reader.to_python(tape2, strs2, bytes(msg2))
        assert repr(a) == repr(b), what


@pytest.mark.parametrize("name", fixtures.ALL)
def test_fixtures(ctx, name):
    check_serialize(ctx, fixtures.load(name), name == "parking-citations", name)


def test_tables_flags_and_tag_lookalikes(ctx):
    corp = GU.load("corpus")
    for c in corp["pass_cases"]:
        check_serialize(ctx, bytes.fromhex(c["js_hex"]), False, c["name"])
    check_serialize(ctx, bytes.fromhex(GU.load("stage2")["demo_ndjson_hex"]), True, "demo_nd")
    # floats with the overflowed-integer flag ('e' entries), and values whose top byte is a tag: runs of such values
    # make the "last anchor" of the tag / raw classification lie far back, across 2048-word tiles
    def dbl(bits):
        return repr(struct.unpack("<d", struct.pack("<Q", bits))[0])
    rnd = random.Random(5)
    look = [dbl((ord(c) << 56) | rnd.getrandbits(52)) for c in '"lud{[}]rtfne']
    docs = [
        "[" + ",".join(["123456789012345678901234567890", "1.5", "-1", "18446744073709551615", '"s"', "true", "null", '{"a":[]}'] * 3) + "]",
        "[" + ",".join(look * 700) + "]",
        "[" + ",".join([look[3]] * 9000) + "]",                       # every raw word looks like 'd'
        "[" + ",".join(['"x"', look[0]] * 5000) + "]",
        "{" + ",".join('"k%d":[%s]' % (i, ",".join(look[:4] * (i % 7))) for i in range(800)) + "}",
    ]
    for i, d in enumerate(docs):
        check_serialize(ctx, d.encode(), False, f"doc{i}")
    nd = "\n".join('{"i":%d,"v":%s,"s":"%s"}' % (i, look[i % len(look)], "y" * (i % 50)) for i in range(6000))
    check_serialize(ctx, nd.encode(), True, "nd-lookalikes")


def test_random_documents(ctx):
    from test_gpu_parse import _random_records
    rnd, lines = _random_records(123, 2 << 20)
    check_serialize(ctx, ("[" + ",".join(lines) + "]").encode("utf-8"), False, "random-array")
    check_serialize(ctx, "\n".join(lines).encode("utf-8"), True, "random-nd")


def test_full_size(ctx):
    """configs[1] / configs[4] at full size: stream identical to the oracle's"""
    import workloads
    for doc, nd in ((workloads.c2_twitter_array(426), False), (workloads.c5_parking_nd(1000), True)):
        check_serialize(ctx, doc, nd, "full-size")


# ---- de-duplicated string column and Deserialize on the device ------------------------------------------------------------
def check_round_trip(ctx, data, nd, what, expect_smaller=False):
    """The property the reference's tests pin (parsed_serialize_test.go:220-340): Deserialize(Serialize(pj)) marshals to the
    same JSON.  Here: device Serialize with and without de-duplication -> oracle Deserialize and device Deserialize ->
    the oracle's MarshalJSON of the result equals the oracle's MarshalJSON of the parse."""
    ref = O.parse(data, ndjson=nd, copy_strings=True)
    assert ref.rc == 0, what
    msg = bytes(data[ref.msg_off:ref.msg_off + ref.msg_len])
    rc, want = O.marshal_json(ref.tape, ref.strings, msg)
    assert rc == 0, what
    streams, cols = {}, {}
    for dedup in (False, True):
        ctx.parse(data, ndjson=nd, copy_strings=True)
        cols[dedup] = ctx.serialize(fetch=False, dedup=dedup)["strings"]
        streams[dedup] = ctx.serialize(dedup=dedup)
    if expect_smaller:  # the string column holds every distinct string about once
        assert cols[True] < cols[False] * 0.2, (what, cols)
    assert cols[True] <= cols[False] and len(streams[True]) <= len(streams[False]), what
    for dedup, stream in streams.items():
        rc, tape2, strs2, msg2 = O.deserialize(stream)          # the oracle reads the device's stream
        assert rc == 0, (what, dedup)
        rc, got = O.marshal_json(tape2, strs2, bytes(msg2))
        assert rc == 0 and got == want, (what, dedup, "oracle Deserialize")
        pj = ctx.deserialize(stream)                            # the device reads it
        assert len(pj.Tape) == len(ref.tape), (what, dedup)
        assert np.array_equal(pj.Tape, tape2), (what, dedup, "device tape differs from the oracle's Deserialize")
        assert pj.Message == bytes(msg2) and len(pj.Strings) == len(strs2), (what, dedup)
        rc, got = O.marshal_json(pj.Tape, pj.Strings, pj.Message)
        assert rc == 0 and got == want, (what, dedup, "device Deserialize")


@pytest.mark.parametrize("name", fixtures.ALL)
def test_round_trip_fixtures(ctx, name):
    check_round_trip(ctx, fixtures.load(name), name == "parking-citations", name)


def test_round_trip_dedup_shrinks_repeated_records(ctx):
    park = fixtures.load("parking-citations")
    check_round_trip(ctx, park * 40, True, "parking x40", expect_smaller=True)
    doc = ("[" + ",".join('{"key":"value","tag":"%s","n":%d,"empty":""}' % ("abc" * (i % 5), i) for i in range(20000)) + "]").encode()
    check_round_trip(ctx, doc, False, "repeated keys", expect_smaller=True)


def test_round_trip_tables_and_lookalikes(ctx):
    corp = GU.load("corpus")
    for c in corp["pass_cases"]:
        check_round_trip(ctx, bytes.fromhex(c["js_hex"]), False, c["name"])
    from test_gpu_parse import _random_records
    rnd, lines = _random_records(77, 1 << 20)
    check_round_trip(ctx, "\n".join(lines).encode("utf-8"), True, "random-nd")


def test_deserialize_rejects_corrupt_streams(ctx):
    import sjhip
    ctx.parse(fixtures.load("twitter"))
    sizes = ctx.serialize(fetch=False, dedup=True)
    stream = ctx.serialize(dedup=True)

    def varint_len(v):
        n = 1
        while v >= 0x80:
            v >>= 7
            n += 1
        return n
    vl, tl = sizes["values"], sizes["tags"]
    tags_at = len(stream) - vl - (varint_len(vl) + varint_len(vl + 1) + 1) - tl  # first byte of the tag column
    assert stream[tags_at] == ord("r")
    bad_tag = stream.copy()
    bad_tag[tags_at + 5] = ord("N")                      # a TagNop entry
    swapped = stream.copy()
    swapped[tags_at + 1] = ord("t")                      # the root object became an atom: the columns no longer add up
    version = stream.copy()
    version[0] = 9
    for bad in (stream[:len(stream) // 2], stream[:3], bad_tag, swapped, version):
        with pytest.raises(sjhip.ParseError):
            ctx.deserialize(np.ascontiguousarray(bad))
    ctx.deserialize(stream)  # the context is still usable


def _frame(tags, values, strings=b"", tape_len=None):
    """a version-3 stream of uncompressed blocks around the given columns (parsed_serialize.go:376-431)"""
    def uv(v):
        out = bytearray()
        while v >= 0x80:
            out.append((v & 0x7f) | 0x80)
            v >>= 7
        out.append(v)
        return bytes(out)
    vals = b"".join((v & 0xFFFFFFFFFFFFFFFF).to_bytes(8, "little") for v in values)
    body = uv(tape_len) + b"\x00\x00" + uv(len(strings)) + uv(len(strings) + 1) + b"\x00" + strings + \
        uv(len(tags)) + uv(len(tags) + 1) + b"\x00" + tags + uv(len(vals)) + uv(len(vals) + 1) + b"\x00" + vals
    return np.frombuffer(b"\x03" + uv(len(body)) + body, dtype=np.uint8)


def test_deserialize_checks_closing_brackets(ctx):
    """parsed_serialize.go:666-671: a closing tag must meet the word its opener left.  The device scatters in parallel, so
    a slot no opener wrote would keep a word of the previous parse: streams with a closing tag without an opener, with
    the wrong kind of closing tag, with an opener whose distance does not reach past itself, and with two openers that
    claim one slot are all rejected; the well-formed stream next to each is read."""
    import sjhip
    good = _frame(b"r[[]]r", [6, 4, 2, -5], tape_len=6)
    for _ in range(2):
        pj = ctx.deserialize(good)
        assert [int(x) >> 56 for x in pj.Tape] == [ord(c) for c in "r[[]]r"]
        assert [int(x) & 0xFFFFFFFF for x in pj.Tape] == [6, 5, 4, 2, 1, 0]
        for tags, values, tl in (
                (b"r[]]]r", [6, 4, -5], 6),          # two closing tags no opener wrote (stale words of the run before)
                (b"rt]tr", [5, -4], 5),
                (b"r[[}]r", [6, 4, 2, -5], 6),       # wrong kind
                (b"r[tr", [4, 1, -3], 4),            # an opener whose closing slot is itself
                (b"r[[]tr", [6, 3, 2, -5], 6),       # two openers, one closing slot
                (b"r[[]]r", [6, 4, 3, -5], 6)):      # the inner opener points at the outer's slot: the outer lost it
            with pytest.raises(sjhip.ParseError):
                ctx.deserialize(_frame(tags, values, tape_len=tl))


def test_hand_derived_format_vectors(ctx):
    """tests/golden/serialize_v3_vectors.py (streams derived by hand from parsed_serialize.go:201-236, 283-341, 376-431):
    the device's de-duplicating serializer must produce exactly these bytes, the plain one where no string repeats,
    and the device's Deserialize must read them."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("serialize_v3_vectors", os.path.join(os.path.dirname(__file__), "golden", "serialize_v3_vectors.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for v in mod.VECTORS:
        ref = O.parse(v["doc"], ndjson=v["ndjson"], copy_strings=True)
        ctx.parse(v["doc"], ndjson=v["ndjson"], copy_strings=True)
        got = ctx.serialize(dedup=True)
        assert bytes(got) == v["stream"], (v["name"], bytes(got).hex(), v["stream"].hex())
        if not v["repeats"]:
            assert bytes(ctx.serialize(dedup=False)) == v["stream"], v["name"]
        pj = ctx.deserialize(v["stream"])
        assert len(pj.Tape) == len(ref.tape), v["name"]
        assert tape_reader.to_python(pj.Tape, pj.Strings, pj.Message) == tape_reader.to_python(ref.tape, ref.strings, v["doc"]), v["name"]
