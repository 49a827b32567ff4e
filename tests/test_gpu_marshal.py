"""GPU: Iter.MarshalJSON on the device (sjhip_marshal_json) against the reference's expected
texts (the `want` column of TestParsePassCases / TestParseND) and byte for byte against the oracle's MarshalJSON."""
import random
import struct

import pytest

import fixtures
import golden_util as GU
import oracle_lib as O
from test_gpu_parse import ctx  # noqa: F401

pytestmark = pytest.mark.gpu
CORP = GU.load("corpus")


def check_marshal(ctx, doc, nd, what):
    for copy in (True, False):
        ref = O.parse(doc, ndjson=nd, copy_strings=copy)
        assert ref.rc == 0, what
        rc, want = O.marshal_json(ref.tape, ref.strings, doc[ref.msg_off:ref.msg_off + ref.msg_len])
        assert rc == 0, what
        # the key flags recovered from the token kinds (three launches of marshal.hip), then left by the parser
        # (SJHIP_FLAG_KEY_FLAGS); the parse itself must not notice the flag
        for kf in (False, True):
            pj = ctx.parse(doc, ndjson=nd, copy_strings=copy, key_flags=kf)
            assert pj.Tape.tolist() == ref.tape.tolist() and pj.Strings.tobytes() == ref.strings.tobytes(), (what, copy, kf)
            got = ctx.marshal_json()
            if got != want:
                k = next(i for i in range(min(len(got), len(want)) + 1) if got[i:i + 1] != want[i:i + 1])
                raise AssertionError((what, copy, kf, len(got), len(want), k, got[max(0, k - 30):k + 30], want[max(0, k - 30):k + 30]))
    return want


def test_reference_expected_texts(ctx):  # simdjson_amd64_test.go:701-955, :33-86
    for c in CORP["pass_cases"]:
        if not c["want_err"]:
            assert check_marshal(ctx, bytes.fromhex(c["js_hex"]), False, c["name"]) == bytes.fromhex(c["want_hex"])
    for c in CORP["parse_nd"]:
        if not c["want_err"]:
            assert check_marshal(ctx, bytes.fromhex(c["js_hex"]), True, c["name"]) == bytes.fromhex(c["want_hex"])


@pytest.mark.parametrize("name", fixtures.ALL)
def test_fixtures(ctx, name):
    check_marshal(ctx, fixtures.load(name), name == "parking-citations", name)


def test_numbers_strings_and_shapes(ctx):
    rnd = random.Random(9)
    floats = []
    for e in range(0, 2047, 3):  # across the binades, incl. their first and last values
        for m in (0, rnd.getrandbits(52), (1 << 52) - 1):
            x = struct.unpack("<d", struct.pack("<Q", (e << 52) | m))[0]
            floats += [repr(x), repr(-x)]
    floats += ["0.0", "-0.0", "1e-7", "1e-6", "1e21", "1e20", "123456789012345678901234567890", "5e-324", "1.7976931348623157e308",
               "0.1", "100", "1E+2", "2.5e-8", "4.35", "1e22", "9007199254740993"]
    ints = ["0", "-1", "9223372036854775807", "-9223372036854775808", "9223372036854775808", "18446744073709551615", "42"]
    check_marshal(ctx, ("[" + ",".join(floats + ints) + "]").encode(), False, "numbers")
    ctl = "".join("\\u%04x" % c for c in range(0x20)) + '\\"\\\\\\/\\b\\f\\n\\r\\t' + "é世\U0001f600 plain"
    docs = [
        '{"k":"' + ctl + '","' + ctl + '":[true,false,null,{},[],{"a":{"b":{"c":[1,[2,[3]]]}}}]}',
        '[[],[[]],{},{"a":{}},[{"b":[]},{}],"",{"":""}]',
        '{"a":"b","c":"d","e":["f","g",{"h":"i"}],"j":{"k":"l","m":["n"]}}',
        '["' + "x" * 5000 + '","' + '\\n' * 3000 + '",' + ",".join('"s%d"' % i for i in range(3000)) + "]",
    ]
    for i, d in enumerate(docs):
        check_marshal(ctx, d.encode("utf-8"), False, f"shape{i}")
    nd = "\n".join('{"i":%d,"f":%s,"s":"%s","a":[%s]}' % (i, floats[i % len(floats)], "q\\n" * (i % 7), ",".join(ints[: i % 5]))
                   for i in range(4000))
    check_marshal(ctx, nd.encode(), True, "nd")
    check_marshal(ctx, (nd + "\n\n").encode(), True, "nd-trailing-newlines")


def test_raw_words_that_look_like_tags(ctx):
    """Long runs of numbers whose VALUE word has a top byte equal to a tag (0x6c 'l', 0x75 'u', 0x64 'd', 0x22 '"'): a
    tile of the walk then finds no anchor among the 64 words in front of it and the library repeats the walk with the
    global anchor scan (marshal.hip); long strings (whole-wave copies), strings around the 64-byte threshold, escapes at
    8-byte boundaries."""
    looks = []
    for top in (0x6c, 0x75, 0x64, 0x22):
        x = struct.unpack("<d", struct.pack("<Q", (top << 56) | 0x0123456789abcd))[0]
        looks.append(repr(x))
    doc = "[" + ",".join(looks[i % 4] for i in range(9000)) + ',"tail",{"k":[1,2]}]'
    check_marshal(ctx, doc.encode(), False, "tag-like raw words")
    nd = "\n".join("[" + ",".join(looks[(i + j) % 4] for j in range(700)) + "]" for i in range(12))
    check_marshal(ctx, nd.encode(), True, "tag-like raw words, nd")
    strs = []
    for n in list(range(56, 76)) + [127, 128, 129, 511, 512, 513, 1023, 4097, 40000, 70001]:
        body = "".join(("\\n" if (i % 61) == 7 else '\\"' if (i % 97) == 11 else "\\u0001" if (i % 389) == 5 else chr(97 + i % 26))
                       for i in range(n))
        strs.append('"' + body + '"')
    check_marshal(ctx, ("[" + ",".join(strs) + "]").encode(), False, "long strings")
    check_marshal(ctx, ("{" + ",".join(s + ":" + s for s in strs) + "}").encode(), False, "long keys")


def test_random_documents(ctx):
    from test_gpu_parse import _random_records
    rnd, lines = _random_records(321, 2 << 20)
    check_marshal(ctx, ("[" + ",".join(lines) + "]").encode("utf-8"), False, "random-array")
    check_marshal(ctx, "\n".join(lines).encode("utf-8"), True, "random-nd")


def test_full_size_c5(ctx):
    import workloads
    check_marshal(ctx, workloads.c5_parking_nd(200), True, "parking x200")


def test_key_flags_on_every_parse_path(ctx):
    """SJHIP_FLAG_KEY_FLAGS on the paths a parse can take: the deferred small parse, the synchronous one (> 4 MiB), the
    per-string fallback behind a long surrogate run (k_s2_emit<false> without string masks), and a parse without the
    flag after one with it (stale flags must not be used)."""
    from test_host_stage2 import surrogate_run_docs
    fell_back = 0
    for doc, what in surrogate_run_docs():
        if O.parse(doc).rc == 0:
            check_marshal(ctx, doc, False, what)
            fell_back += doc.count(b"\\ud800") > 4096
    assert fell_back >= 3
    hi = b"\\ud800"
    keys = b'{"a' + hi * 4100 + b'":{"b":"c' + hi * 4098 + b'"},"d":["e","f"]}'  # (an even run pairs up: valid)
    assert O.parse(keys).rc == 0
    check_marshal(ctx, keys, False, "long surrogate runs in a key and in a value")
    big = b"[" + b",".join(b'{"id":%d,"name":"n%d","tags":["a","b",{"c":"d"}],"ok":true}' % (i, i) for i in range(120000)) + b"]"
    assert len(big) > (4 << 20)
    check_marshal(ctx, big, False, "synchronous path")
    # flags of an earlier parse are not picked up by a later one without the flag
    ctx.parse(b'{"a":"b","c":["d",{"e":"f"}]}', key_flags=True)
    ctx.parse(b'["a","b",{"c":"d"},"e"]')
    assert ctx.marshal_json() == b'["a","b",{"c":"d"},"e"]'


@pytest.mark.parametrize("env", [{"SJHIP_MS_TEST_BOUND": "4096"}, {"SJHIP_MS_ONEPASS": "0"}, {"SJHIP_MS_VARIANT": "0"},
                                 {"SJHIP_MS_VARIANT": "7"}, {"SJHIP_MS_VARIANT": "6"}, {"SJHIP_MS_VARIANT": "3"}])
def test_one_pass_fallbacks_and_variants(env):
    """The single-pass MarshalJSON (key flags from the parser, text buffer sized by a bound) must fall back to the two-pass
    form when a tile would write past the bound (forced here with a bound of 4 KiB) and give the same text with the
    single pass turned off and with the other shapes of the tile kernel; the variables are read once per process."""
    import os
    import subprocess
    import sys
    code = r"""
import sys
sys.path.insert(0, 'simdjson-go_amd'); sys.path.insert(0, 'tests')
import fixtures, sjhip, oracle_lib as O, workloads
ctx = sjhip.Context(0)
docs = [(fixtures.load(n), n == 'parking-citations') for n in ('twitter', 'twitterescaped', 'canada', 'parking-citations', 'mesh.pretty')]
docs.append((workloads.c5_parking_nd(30), True))
docs.append((b'[' + b','.join([b'1e20', b'-1e20', b'1E+20', b'123456789e12'] * 3000) + b']', False))  # numbers that print longer
for d, nd in docs:
    ref = O.parse(d, ndjson=nd)
    rc, want = O.marshal_json(ref.tape, ref.strings, d[ref.msg_off:ref.msg_off + ref.msg_len])
    assert ref.rc == 0 and rc == 0
    for kf in (True, False):
        ctx.parse(d, ndjson=nd, key_flags=kf)
        assert ctx.marshal_json() == want, (len(d), nd, kf)
print('ok')
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (env, r.stdout[-500:], r.stderr[-1500:])


def test_numbers_that_print_longer_than_their_source(ctx):
    """The bound of the single-pass text buffer rests on: only numbers print longer than they were written, by at most 17
    bytes ("1e20" -> 21 digits).  Documents made of nothing but such numbers stay inside it and give the oracle's text."""
    worst = ["1e20", "-1e20", "1E20", "9e20", "1e19", "-9.9e20", "1e-7", "2e-6", "1e300", "5e-324", "0e0", "-0.0", "1E+2", "1e0"]
    doc = ("[" + ",".join(worst * 6000) + "]").encode()
    check_marshal(ctx, doc, False, "expanding numbers")
    nd = "\n".join("[" + ",".join(worst[(i + j) % len(worst)] for j in range(9)) + "]" for i in range(20000)).encode()
    check_marshal(ctx, nd, True, "expanding numbers, nd")
