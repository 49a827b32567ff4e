"""Pin the CPU oracle against the reference's own golden vectors (tests/golden/*.json,
extracted from the reference's *_test.go tables by tools/extract_goldens.py)."""
import ctypes as C
import json
import math
import struct

import numpy as np
import pytest

import fixtures
import golden_util as G
import oracle_lib as O

L = O.lib()
U = lambda s: int(s)  # noqa: E731


def u64(v=0):
    return C.c_uint64(v)


# ---------------------------------------------------------------- stage 1 KATs
S1 = G.load("stage1")


def test_finalize_structurals():  # find_subroutines_amd64_test.go:32-65
    for r in S1["finalize"]:
        pp = u64(0)
        got = L.sjo_finalize_structurals(U(r["structurals"]), U(r["whitespace"]), U(r["quote_mask"]),
                                         U(r["quote_bits"]), C.byref(pp))
        assert got == U(r["expected"])
        assert pp.value == U(r["expected_pseudo"])


def test_find_newline_delimiters():  # :69-93
    nd = bytes.fromhex(S1["demo_ndjson_hex"])
    want = [U(x) for x in S1["newline_demo_ndjson"]]
    for off in range(0, len(nd) - 64, 64):
        assert L.sjo_find_newline_delimiters(nd[off:off + 64], 0) == want[off >> 6]


def test_exclude_newline_within_quotes():  # :113-127
    t = S1["newline_in_quotes"]
    inp = bytearray(bytes.fromhex(t["input_hex"]))
    for p in t["set_0a_at"]:
        inp[p] = 0x0A
    piq, qb, em = u64(0), u64(0), u64(0)
    qm = L.sjo_find_quote_mask_and_bits(bytes(inp), 0, C.byref(piq), C.byref(qb), C.byref(em))
    assert L.sjo_find_newline_delimiters(bytes(inp), qm) == U(t["expected"])


def test_find_odd_backslash_sequences():  # :145-199
    for r in S1["odd_backslash"]:
        prev = u64(U(r["prev"]))
        got = L.sjo_find_odd_backslash_sequences(bytes.fromhex(r["input_hex"]), C.byref(prev))
        assert got == U(r["expected"])
        assert prev.value == U(r["ends_odd"])
    for i in range(1, 129):  # shifted positions across two chunks (:182-198)
        t = b" " * (i - 1) + b'\\"' + b" " * (62 + 64)
        prev = u64(0)
        lo = L.sjo_find_odd_backslash_sequences(t[:64], C.byref(prev))
        hi = L.sjo_find_odd_backslash_sequences(t[64:128], C.byref(prev))
        if i < 64:
            assert (lo, hi) == (1 << i, 0)
        else:
            assert (lo, hi) == (0, (1 << (i - 64)) & (2**64 - 1))  # Go: uint64 shift wraps at i=128


def test_find_quote_mask_and_bits():  # :215-303
    for r in S1["quote_mask"]:
        piq, qb, em = u64(0), u64(0), u64(0)
        got = L.sjo_find_quote_mask_and_bits(bytes.fromhex(r["input_hex"]), U(r["odd_ends"]), C.byref(piq),
                                             C.byref(qb), C.byref(em))
        assert got == U(r["expected"])
        assert qb.value == U(r["quote_bits"])
        assert piq.value == U(r["inside_quote"])
        assert em.value == U(r["error_mask"])
    for r in S1["quote_mask_carry"]:
        piq, qb, em = u64(U(r["inside_quote_in"])), u64(0), u64(0)
        L.sjo_find_quote_mask_and_bits(bytes.fromhex(r["input_hex"]), 0, C.byref(piq), C.byref(qb), C.byref(em))
        assert piq.value == U(r["inside_quote_out"])


def test_find_whitespace_and_structurals():  # :645-690
    for r in S1["whitespace_structurals"]:
        ws, st = u64(0), u64(0)
        inp = bytes.fromhex(r["input_hex"])[:64].ljust(64, b"\0")
        L.sjo_find_whitespace_and_structurals(inp, C.byref(ws), C.byref(st))
        assert ws.value == U(r["whitespace"])
        assert st.value == U(r["structurals"])


def test_flatten_bits_incremental():  # :706-770
    for r in S1["flatten"]:
        base = (C.c_uint32 * 1536)()
        idx = C.c_int(0)
        carried, position = u64(0), u64(2**64 - 1)
        for m in r["masks"]:
            L.sjo_flatten_bits_incremental(base, C.byref(idx), U(m), C.byref(carried), C.byref(position))
        assert list(base[: idx.value]) == r["expected"]


def test_fused_equals_multiple_calls():  # :321-360
    a = [u64(0), u64(0), u64(0), u64(1)]
    b = [u64(0), u64(0), u64(0), u64(1)]
    for hx in S1["fused_chunks"]:
        chunk = bytes.fromhex(hx)
        fused = L.sjo_find_structural_bits(chunk, *[C.byref(x) for x in a])
        qb, ws, st = u64(0), u64(0), u64(0)
        oe = L.sjo_find_odd_backslash_sequences(chunk, C.byref(b[0]))
        qm = L.sjo_find_quote_mask_and_bits(chunk, oe, C.byref(b[1]), C.byref(qb), C.byref(b[2]))
        L.sjo_find_whitespace_and_structurals(chunk, C.byref(ws), C.byref(st))
        mc = L.sjo_finalize_structurals(st.value, ws.value, qm, qb.value, C.byref(b[3]))
        assert fused == mc


def _slice(buf, ndjson=0, carried0=0):
    st = [u64(0), u64(0), u64(0), u64(1)]
    idx = (C.c_uint32 * 1536)()
    n = C.c_int(0)
    carried, position = u64(carried0), u64(2**64 - 1)
    processed = L.sjo_find_structural_bits_in_slice(buf, len(buf), *[C.byref(x) for x in st], idx, C.byref(n),
                                                    C.byref(carried), C.byref(position), ndjson)
    return processed, list(idx[: n.value]), carried.value


def test_whitespace_padding():  # :381-421
    msg = b":" * 64
    for l in range(64, -1, -1):
        processed, idx, carried = _slice(msg[:l], carried0=2**64 - 1)
        assert processed == l
        assert len(idx) == l
        # carried starts at -1, so the first delta is the absolute position and the deltas sum to l-1
        if l > 0:
            assert sum(idx) & 0xFFFFFFFF == l - 1
        else:
            assert carried == (2**64 - 1 + 64) & (2**64 - 1) or processed == 0


def test_twitter_structural_loop():  # :463-479
    msg = fixtures.load("twitter")
    ok, pos = O.stage1(msg)
    assert ok
    t = S1["twitter_loop"]
    assert len(pos) == t["expected_length"]
    assert bytes(msg[p] for p in pos[::-1][:5]).decode() == t["last_structurals_reversed"]


def test_demo_json_marks():  # stage1_find_marks_amd64_test.go:29-85
    dj = bytes.fromhex(S1["demo_json_hex"])[:64]
    m = S1["demo_json_marks"]
    prev = u64(0)
    assert L.sjo_find_odd_backslash_sequences(dj, C.byref(prev)) == 0
    piq, qb, em = u64(0), u64(0), u64(0)
    qm = L.sjo_find_quote_mask_and_bits(dj, 0, C.byref(piq), C.byref(qb), C.byref(em))
    assert qm == G.bits_lsb_first(m["quoted"])
    ws, st = u64(0), u64(0)
    L.sjo_find_whitespace_and_structurals(dj, C.byref(ws), C.byref(st))
    assert st.value == G.bits_lsb_first(m["structurals"])
    assert ws.value == G.bits_lsb_first(m["whitespace"])
    pp = u64(0)
    fin = L.sjo_finalize_structurals(st.value, ws.value, qm, qb.value, C.byref(pp))
    assert fin == G.bits_lsb_first(m["structurals_finalized"])


def test_demo_json_positions():  # stage1_find_marks_amd64_test.go:87-165
    ok, pos = O.stage1(bytes.fromhex(S1["demo_json_hex"]))
    assert ok
    assert list(pos) == S1["demo_json_positions"]


# ---------------------------------------------------------------- stage 2 / tapes
S2 = G.load("stage2")


def test_stage2_build_tape():  # stage2_build_tape_amd64_test.go:26-193
    for t in S2["tapes_nocopy"]:
        p = O.parse(bytes.fromhex(t["input_hex"]), copy_strings=False)
        assert p.rc == 0
        assert [int(x) for x in p.tape] == [U(x) for x in t["tape"]]


@pytest.mark.parametrize("atom", ["true", "false", "null"])
def test_atoms(atom):  # :195-262
    fn = getattr(L, f"sjo_is_valid_{atom}_atom")
    for r in S2["atom_" + atom]:
        b = bytes.fromhex(r["input_hex"])
        assert bool(fn(b, len(b))) == r["expected"]


def test_demo_ndjson_tape():  # ndjson_test.go:36-248 via parse_json_amd64_test.go:34
    p = O.parse(bytes.fromhex(S2["demo_ndjson_hex"]), ndjson=True, copy_strings=False)
    assert p.rc == 0
    assert [int(x) for x in p.tape] == [U(x) for x in S2["demo_ndjson_tape_nocopy"]]


def test_ndjson_empty_lines():  # parse_json_amd64_test.go:47-73
    for hx in S2["ndjson_empty_lines_hex"]:
        assert O.parse(bytes.fromhex(hx), ndjson=True).rc == 0


# ---------------------------------------------------------------- strings
def test_parse_string_table():  # parse_string_test.go:19-235, parse_json_amd64_test.go:540-593
    for r in G.load("strings"):
        body = bytes.fromhex(r["str_hex"]) + b'"'
        dst = C.create_string_buffer(len(body) + 64)
        n = u64(0)
        ok = L.sjo_parse_string(body, len(body), dst, C.byref(n))
        assert bool(ok) == r["success"], r["name"]
        sl, dl = u64(0), u64(0)
        ok2 = L.sjo_parse_string_validate_only(body, len(body), C.byref(sl), C.byref(dl))
        assert bool(ok2) == r["success"], r["name"]
        if r["success"]:
            want = bytes.fromhex(r["want_hex"])
            assert dst.raw[: n.value] == want, r["name"]
            assert dl.value == len(want)
            assert sl.value == len(body) - 1


# ---------------------------------------------------------------- numbers
NUM = G.load("numbers")


def parse_number(s: bytes):
    v = u64(0)
    tag = L.sjo_parse_number(s, len(s), C.byref(v))
    return (chr(tag >> 56) if tag else ""), tag & 0x00FFFFFFFFFFFFFF, v.value


def f64(bits):
    return struct.unpack("<d", struct.pack("<Q", bits))[0]


def test_parse_number_table():  # parse_json_amd64_test.go:222-277
    for r in NUM["parse_number"]:
        tag, flags, val = parse_number(r["input"].encode() + b":")
        assert tag == r["tag"], r
        assert flags == r["flags"], r
        if tag == "d":
            assert f64(val) == float(r["float_repr"]), r
        elif tag == "l":
            assert val == int(r["int"]) & (2**64 - 1)
        else:
            assert val == int(r["uint"])


def test_parse_int64_table():  # :287-339
    for r in NUM["parse_int64"]:
        tag, _, val = parse_number(r["input"].encode() + b":")
        assert tag == r["tag"], r
        if tag == "l":
            assert val == int(r["out"]) & (2**64 - 1), r


def test_parse_float64_table():  # :349-538 (Go's atoftests)
    for r in NUM["atof"]:
        tag, _, val = parse_number(r["input"].encode() + b":")
        if tag == "":
            assert r["err"] is not None, r
        elif tag == "d":
            want = float(r["out"].replace("+Inf", "inf").replace("-Inf", "-inf"))
            got = f64(val)
            assert struct.pack("<d", got) == struct.pack("<d", want), r
        elif tag == "l":
            v = val - 2**64 if val >= 2**63 else val
            assert str(v) == r["out"], r
        else:
            assert str(val) == r["out"], r


def test_number_is_valid():  # parse_number_test.go:30-129
    for s in NUM["valid"]:
        assert parse_number(s.encode())[0] != "", s
    for s in NUM["invalid"]:
        assert parse_number(s.encode())[0] == "", s


# ---------------------------------------------------------------- accept / reject corpora
CORP = G.load("corpus")


def _py_equiv(js: bytes, p, ndjson=False):
    """Differential check of an accepted document against Python's json module."""
    import tape_reader
    docs = tape_reader.to_python(p.tape, p.strings, js[p.msg_off:p.msg_off + p.msg_len])
    if ndjson:
        want = [json.loads(l) for l in js.decode("utf-8", "surrogateescape").split("\n") if l.strip()]
    else:
        want = [json.loads(js.decode("utf-8"))]
    assert docs == want


@pytest.mark.parametrize("case", CORP["fail_cases"], ids=lambda c: c["name"])
def test_parse_fail_cases(case):  # simdjson_amd64_test.go:162-693
    js = bytes.fromhex(case["js_hex"])
    p = O.parse(js)
    assert (p.rc != 0) == case["want_err"]
    if p.rc == 0:
        _py_equiv(js, p)


@pytest.mark.parametrize("case", CORP["pass_cases"], ids=lambda c: c["name"])
def test_parse_pass_cases(case):  # simdjson_amd64_test.go:695-1017
    js = bytes.fromhex(case["js_hex"])
    p = O.parse(js)
    assert (p.rc != 0) == case["want_err"]
    if p.rc == 0 and not case["only_precise"]:
        _py_equiv(js, p)


@pytest.mark.parametrize("case", CORP["parse_nd"], ids=lambda c: c["name"])
def test_parse_nd_cases(case):  # simdjson_amd64_test.go:29-160
    js = bytes.fromhex(case["js_hex"])
    p = O.parse(js, ndjson=True)
    assert (p.rc != 0) == case["want_err"]
    if p.rc == 0:
        _py_equiv(js, p, ndjson=True)


# ---------------------------------------------------------------- fixtures
@pytest.mark.parametrize("name", fixtures.ALL)
def test_every_fixture_parses(name):  # TestVerifyTape, parse_json_amd64_test.go:682-698
    data = fixtures.load(name)
    nd = name == "parking-citations"
    p = O.parse(data, ndjson=nd)
    assert p.rc == 0
    if name not in ("parking-citations",):
        _py_equiv(data, p)


def test_parking_citations_count_where():  # ndjson_test.go:250-267
    import tape_reader
    data = fixtures.load("parking-citations")
    p = O.parse(data, ndjson=True)
    docs = tape_reader.to_python(p.tape, p.strings, data[p.msg_off:p.msg_off + p.msg_len])
    assert sum(1 for d in docs if d.get("Make") == "HOND") == S2["parking_citations_hond"]


def test_twitterescaped_equals_twitter_in_copy_mode():
    a = O.parse(fixtures.load("twitter"))
    b = O.parse(fixtures.load("twitterescaped"))
    assert np.array_equal(a.tape, b.tape) and np.array_equal(a.strings, b.strings)


def test_baseline_configuration_sizes():
    """The sizes SURVEY.md section 8(d) derives for the BASELINE.json configurations (structurals, tape words, Strings.B
    bytes, floats) hold for the oracle: C1 twitter.json (55 263 structurals is the reference's own golden,
    find_subroutines_amd64_test.go:464), C3 canada.json, C4 twitterescaped.json in both copy modes, one file of C5."""
    tw = O.parse(fixtures.load("twitter"))
    assert O.stage1(fixtures.load("twitter"))[1].size == 55263
    assert (tw.tape.size, tw.strings.size) == (49783, 367917)
    ca = O.parse(fixtures.load("canada"))
    assert O.stage1(fixtures.load("canada").strip())[1].size == 334373
    assert (ca.tape.size, ca.strings.size) == (334376, 90)
    tags, i = {}, 0
    while i < ca.tape.size:  # entry by entry: a string / number entry is two words
        t = chr(int(ca.tape[i]) >> 56)
        tags[t] = tags.get(t, 0) + 1
        i += 2 if t in 'lud"' else 1
    assert tags["d"] + tags["l"] == 111126 and tags["l"] == 46  # 111 126 numbers, 46 of them written without a fraction
    te = O.parse(fixtures.load("twitterescaped"), copy_strings=False)
    assert te.strings.size == 113036 and te.tape.size == 49783
    pk = O.parse(fixtures.load("parking-citations"), ndjson=True)
    assert (pk.tape.size, pk.strings.size) == (80000, 256664)
