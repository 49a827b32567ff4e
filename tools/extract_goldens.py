#!/usr/bin/env python3
"""Extract the reference's golden vectors / known-answer tables into tests/golden/*.json.

Runs in the development container only (needs /root/reference).  It parses the Go
composite literals of the reference's *_test.go tables with tools/golit.py and stores them
as plain JSON (byte strings are stored hex-encoded under keys ending in `_hex`, 64-bit
integers as decimal strings) so that the test-suite can replay them on any box.

Sources (all relative to /root/reference):
  find_subroutines_amd64_test.go   per-routine stage-1 KATs
  stage1_find_marks_amd64_test.go  demo_json masks and structural positions
  stage2_build_tape_amd64_test.go  5 golden tapes + atom tables
  ndjson_test.go                   demo_ndjson + its golden tape
  parse_string_test.go             string unescape table
  parse_json_amd64_test.go         number tables (TestParseNumber/Int64/Float64), ND empty lines
  parse_number_test.go             valid / invalid number lists
  simdjson_amd64_test.go           TestParseND / TestParseFailCases / TestParsePassCases corpora
"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import golit  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(HERE, "..", "tests", "golden")


def src(name):
    return open(os.path.join(REF, name), encoding="utf-8").read()


def hx(b):
    return bytes(b).hex()


def u(x):
    return str(int(x) & 0xFFFFFFFFFFFFFFFF)


def dump(name, obj):
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(obj, f, indent=1)
    print("wrote", name)


def const_string(text, name):
    m = re.search(r"const\s+" + name + r"\s*=\s*`([^`]*)`", text)
    return m.group(1).encode("utf-8")


def main():
    os.makedirs(OUT, exist_ok=True)
    demo_json = const_string(src("parsed_json_test.go"), "demo_json")
    demo_ndjson = const_string(src("ndjson_test.go"), "demo_ndjson")
    env = {"demo_json": demo_json, "demo_ndjson": demo_ndjson, "nul": 0}

    # ---------------- stage-1 per-routine KATs ----------------
    s = src("find_subroutines_amd64_test.go")
    st1 = {"demo_json_hex": hx(demo_json), "demo_ndjson_hex": hx(demo_ndjson)}

    rows = golit.find_literal(s, r"func TestFinalizeStructurals[^\n]*\n(?:.*\n)*?\s*testCases := ", env)
    st1["finalize"] = [dict(structurals=u(r[0]), whitespace=u(r[1]), quote_mask=u(r[2]), quote_bits=u(r[3]),
                            expected=u(r[4]), expected_pseudo=u(r[5])) for r in rows]

    rows = golit.find_literal(s, r"func testFindNewlineDelimiters[^\n]*\n(?:.*\n)*?\s*want := ", env)
    st1["newline_demo_ndjson"] = [u(r) for r in rows]
    st1["newline_in_quotes"] = {"note": "find_subroutines_amd64_test.go:113-127", "input_hex": hx(
        bytearray(b'  "-------------------------------------"                       ')), "set_0a_at": [10, 50],
        "expected": u(1 << 50)}

    rows = golit.find_literal(s, r"func testFindOddBackslashSequences[^\n]*\n(?:.*\n)*?\s*testCases := ", env)
    st1["odd_backslash"] = [dict(prev=u(r[0]), input_hex=hx(r[1]), expected=u(r[2]), ends_odd=u(r[3])) for r in rows]

    m = re.search(r"func testFindQuoteMaskAndBits", s)
    s_q = s[m.start():]
    rows = golit.find_literal(s_q, r"testCases := ", env)
    st1["quote_mask"] = [dict(odd_ends=u(r[0]), input_hex=hx(r[1]), expected=u(r[2]), quote_bits=u(r[3]),
                              inside_quote=u(r[4]), error_mask=u(r[5])) for r in rows]
    rows = golit.find_literal(s_q, r"testCasesPIIQ := ", env)
    st1["quote_mask_carry"] = [dict(inside_quote_in=u(r[0]), input_hex=hx(r[1]), inside_quote_out=u(r[2])) for r in rows]

    m = re.search(r"func testFindStructuralBits\(", s)
    rows = golit.find_literal(s[m.start():], r"testCases := ", env)
    st1["fused_chunks"] = [hx(r[0]) for r in rows]

    m = re.search(r"func testFindWhitespaceAndStructurals", s)
    rows = golit.find_literal(s[m.start():], r"testCases := ", env)
    st1["whitespace_structurals"] = [dict(input_hex=hx(r[0]), whitespace=u(r[1]), structurals=u(r[2])) for r in rows]

    m = re.search(r"func TestFlattenBitsIncremental", s)
    rows = golit.find_literal(s[m.start():], r"testCases := ", env)
    st1["flatten"] = [dict(masks=[u(x) for x in r[0]], expected=[int(x) for x in r[1]]) for r in rows]

    st1["twitter_loop"] = {"note": "find_subroutines_amd64_test.go:463-464", "expected_length": 55263,
                           "last_structurals_reversed": '}}":"'}

    s = src("stage1_find_marks_amd64_test.go")
    rows = golit.find_literal(s, r"func TestStage1FindMarks[^\n]*\n(?:.*\n)*?\s*testCases := ", env)
    r = rows[0]
    st1["demo_json_marks"] = dict(quoted=r[0].decode(), structurals=r[1].decode(), whitespace=r[2].decode(),
                                  structurals_finalized=r[3].decode())
    rows = golit.find_literal(s, r"func TestFindStructuralIndices[^\n]*\n(?:.*\n)*?\s*parsed := ", env)
    # every row is demo_json with the prefix blanked: position = number of leading blanks (first row: 0)
    pos = []
    for row in rows:
        t = row.decode()
        pos.append(len(t) - len(t.lstrip(" ")) if t[0] == " " else 0)
    st1["demo_json_positions"] = pos
    dump("stage1.json", st1)

    # ---------------- stage-2 tapes ----------------
    s = src("stage2_build_tape_amd64_test.go")
    env2 = dict(env)
    env2["floatHexRepresentation1"] = 0x69066666666666
    env2["floatHexRepresentation2"] = 0x79066666666666
    rows = golit.find_literal(s, r"func TestStage2BuildTape[^\n]*\n(?:.*\n)*?\s*testCases := ", env2)
    tapes = []
    for inp, exp in rows:
        tapes.append(dict(input_hex=hx(inp), tape=[u((c << 56) | v) for c, v in exp]))
    st2 = {"tapes_nocopy": tapes}
    for atom in ("True", "False", "Null"):
        m = re.search(r"func TestIsValid%sAtom" % atom, s)
        rows = golit.find_literal(s[m.start():], r"testCases := ", env)
        st2["atom_" + atom.lower()] = [dict(input_hex=hx(r[0]), expected=bool(r[1])) for r in rows]

    s = src("ndjson_test.go")
    m = re.search(r"func verifyDemoNdjson", s)
    rows = golit.find_literal(s[m.start():], r"testCases := ", env)
    exp = rows[0][0]
    st2["demo_ndjson_tape_nocopy"] = [u((c << 56) | v) for c, v in exp]
    st2["demo_ndjson_hex"] = hx(demo_ndjson)
    st2["parking_citations_hond"] = 116  # ndjson_test.go:257-267
    s = src("parse_json_amd64_test.go")
    m = re.search(r"func TestNdjsonEmptyLines", s)
    rows = golit.find_literal(s[m.start():], r"ndjson_emptylines := ", env)
    st2["ndjson_empty_lines_hex"] = [hx(r) for r in rows]  # all must parse (parse_json_amd64_test.go:52-73)
    dump("stage2.json", st2)

    # ---------------- strings ----------------
    s = src("parse_string_test.go")
    rows = golit.find_literal(s, r"var tests = ", env)
    dump("strings.json", [dict(name=r["name"].decode(), str_hex=hx(r["str"]), success=bool(r["success"]),
                               want_hex=hx(r.get("want", b"") or b"")) for r in rows])

    # ---------------- numbers ----------------
    s = src("parse_json_amd64_test.go")
    tagmap = {"TagInteger": "l", "TagUint": "u", "TagFloat": "d", "TagEnd": ""}
    envn = dict(env)
    envn["FloatOverflowedInteger"] = 1
    for k, v in tagmap.items():
        envn[k] = v
    rows = golit.find_literal(s, r"func TestParseNumber[^\n]*\n\s*testCases := ", envn)
    nums = {"parse_number": []}
    for r in rows:
        e = dict(input=r["input"].decode(), tag=r["wantTag"], flags=int(r.get("flags", 0)))
        if "expectedD" in r:
            e["float_repr"] = repr(float(r["expectedD"]))
        if "expectedI" in r:
            e["int"] = str(r["expectedI"])
        if "expectedU" in r:
            e["uint"] = str(r["expectedU"])
        nums["parse_number"].append(e)
    rows = golit.find_literal(s, r"var parseInt64Tests = ", envn)
    nums["parse_int64"] = [dict(input=r[0].decode(), out=str(r[1]), tag=r[2]) for r in rows]
    envf = dict(envn)
    rows = golit.find_literal(s, r"var atoftests = ", envf)
    at = []
    for r in rows:
        err = r[2]
        if err is None:
            errs = None
        elif isinstance(err, tuple) and err[0] == "error":
            errs = "invalid json"
        else:
            errs = str(err[1] if isinstance(err, tuple) else err)
        at.append(dict(input=r[0].decode(), out=r[1].decode(), err=errs))
    nums["atof"] = at
    s = src("parse_number_test.go")
    rows = golit.find_literal(s, r"validTests := ", env)
    nums["valid"] = [r.decode() for r in rows]
    rows = golit.find_literal(s, r"invalidTests := ", env)
    nums["invalid"] = [r.decode() for r in rows]
    dump("numbers.json", nums)

    # ---------------- accept / reject corpora ----------------
    s = src("simdjson_amd64_test.go")
    corp = {}
    for fn, key in (("TestParseND", "parse_nd"), ("TestParseFailCases", "fail_cases"), ("TestParsePassCases", "pass_cases")):
        m = re.search(r"func %s\(" % fn, s)
        rows = golit.find_literal(s[m.start():], r"tests := ", env)
        out = []
        for r in rows:
            out.append(dict(name=r["name"].decode(), js_hex=hx(r.get("js", b"")), want_hex=hx(r.get("want", b"") or b""),
                            want_err=bool(r.get("wantErr", False)), skip_floats=bool(r.get("skipFloats", False)),
                            only_precise=bool(r.get("onlyPrecise", False))))
        corp[key] = out
    dump("corpus.json", corp)


if __name__ == "__main__":
    main()
