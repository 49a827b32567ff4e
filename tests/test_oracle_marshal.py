"""Oracle MarshalJSON (oracle/sjo_marshal.c restating parsed_json.go:401-556) pinned by the reference's own expected
texts: the `want` column of TestParsePassCases (simdjson_amd64_test.go:701-955) and TestParseND (:33-86); its float
formatting against Python's shortest digits on every binade; and a JSON round trip on the fixtures."""
import json
import math
import random
import struct

import pytest

import fixtures
import golden_util as GU
import oracle_lib as O
from test_host_ftoa import go_format

CORP = GU.load("corpus")


def marshal(doc, nd):
    p = O.parse(doc, ndjson=nd, copy_strings=True)
    assert p.rc == 0
    rc, text = O.marshal_json(p.tape, p.strings, doc[p.msg_off:p.msg_off + p.msg_len])
    assert rc == 0
    # the same text when the strings stay in the message
    q = O.parse(doc, ndjson=nd, copy_strings=False)
    rc2, text2 = O.marshal_json(q.tape, q.strings, doc[q.msg_off:q.msg_off + q.msg_len])
    assert rc2 == 0 and text2 == text
    return text


@pytest.mark.parametrize("case", [c for c in CORP["pass_cases"] if not c["want_err"]], ids=lambda c: c["name"])
def test_reference_pass_cases(case):  # `want` = the reference's MarshalJSON of the parsed document
    got = marshal(bytes.fromhex(case["js_hex"]), False)
    assert got == bytes.fromhex(case["want_hex"]), (case["name"], got[:200])


@pytest.mark.parametrize("case", [c for c in CORP["parse_nd"] if not c["want_err"]], ids=lambda c: c["name"])
def test_reference_nd_cases(case):
    got = marshal(bytes.fromhex(case["js_hex"]), True)
    assert got == bytes.fromhex(case["want_hex"]), (case["name"], got[:200])


def test_float_format_against_python_shortest_digits():
    rnd = random.Random(7)
    cases = [e << 52 | m for e in range(0, 2047) for m in (0, rnd.getrandbits(52), (1 << 52) - 1)]
    cases += [rnd.getrandbits(64) for _ in range(60000)]
    for bits in cases:
        x = struct.unpack("<d", struct.pack("<Q", bits))[0]
        if math.isinf(x) or math.isnan(x):
            assert O.format_float(bits) == ""
        else:
            assert O.format_float(bits) == go_format(x), hex(bits)


@pytest.mark.parametrize("name", ["twitter", "canada", "twitterescaped", "github_events", "mesh", "numbers", "random"])
def test_fixtures_round_trip_through_json(name):
    doc = fixtures.load(name)
    text = marshal(doc, False)
    assert json.loads(text) == json.loads(doc)
    assert b" " not in text.replace(b'" "', b"").split(b'"')[0]  # compact: no whitespace outside strings at the start


def test_parking_citations_lines():
    doc = fixtures.load("parking-citations")
    text = marshal(doc, True)
    lines = text.split(b"\n")
    assert len(lines) == 1000 and not text.endswith(b"\n")
    assert [json.loads(l) for l in lines] == [json.loads(l) for l in doc.split(b"\n") if l.strip()]
