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
