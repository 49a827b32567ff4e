"""GPU parity tests for the whole Parse()/ParseND() path (run with -m gpu on an MI355X):
Tape and Strings.B produced by the HIP kernels, fetched through the C ABI, must be bit-identical
to the oracle's; for rejected documents the error class (stage 1 / stage 2) must match."""
import random

import numpy as np
import pytest

import fixtures
import golden_util as GU
import oracle_lib as O
import workloads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import sjhip
    assert sjhip.supported(), "gfx950 device required"
    c = sjhip.Context(0)
    yield c
    c.close()


def gpu_parse(ctx, data, nd, copy):
    import sjhip
    try:
        pj = ctx.parse(data, ndjson=nd, copy_strings=copy)
        return 0, pj
    except sjhip.ParseError as e:
        return e.code, None


def check(ctx, data, nd=False, what=""):
    for copy in (True, False):
        ref = O.parse(data, ndjson=nd, copy_strings=copy)
        rc, pj = gpu_parse(ctx, data, nd, copy)
        assert rc == ref.rc, (what, nd, copy, rc, ref.rc, data[:80])
        if rc == 0:
            assert pj.Message == bytes(data[ref.msg_off:ref.msg_off + ref.msg_len]), what
            assert len(pj.Tape) == len(ref.tape), (what, nd, copy, len(pj.Tape), len(ref.tape))
            if not np.array_equal(pj.Tape, ref.tape):
                d = np.nonzero(pj.Tape != ref.tape)[0]
                raise AssertionError((what, nd, copy, "tape differs at", d[:5], [hex(int(x)) for x in pj.Tape[d[:3]]],
                                      [hex(int(x)) for x in ref.tape[d[:3]]]))
            assert np.array_equal(pj.Strings, ref.strings), (what, nd, copy, "strings differ")


S2 = GU.load("stage2")
CORP = GU.load("corpus")
NUM = GU.load("numbers")


def test_golden_tapes(ctx):  # stage2_build_tape_amd64_test.go:26-193, ndjson_test.go:36-248
    import sjhip
    for t in S2["tapes_nocopy"]:
        pj = ctx.parse(bytes.fromhex(t["input_hex"]), copy_strings=False)
        assert [int(x) for x in pj.Tape] == [int(x) for x in t["tape"]]
    pj = ctx.parse(bytes.fromhex(S2["demo_ndjson_hex"]), ndjson=True, copy_strings=False)
    assert [int(x) for x in pj.Tape] == [int(x) for x in S2["demo_ndjson_tape_nocopy"]]
    for hx in S2["ndjson_empty_lines_hex"]:
        ctx.parse(bytes.fromhex(hx), ndjson=True)


@pytest.mark.parametrize("name", fixtures.ALL)
def test_fixture_tapes_equal_oracle(ctx, name):
    data = fixtures.load(name)
    check(ctx, data, nd=(name == "parking-citations"), what=name)
    if name not in ("parking-citations",):
        check(ctx, data, nd=True, what=name + "/nd")


@pytest.mark.parametrize("name", ["twitter", "twitterescaped", "canada", "parking-citations"])
def test_fixture_mutations(ctx, name):
    """single-byte corruptions of the real documents (positions spread over the file, replacement bytes that
    matter to the grammar, the string scanner and the number parser): same verdict and tape as the oracle"""
    import os
    data = fixtures.load(name)
    nd = name == "parking-citations"
    import zlib
    rnd = random.Random(zlib.crc32(name.encode()))
    for _ in range(int(os.environ.get("SJ_MUTATIONS", "40"))):
        b = bytearray(data)
        at = rnd.randrange(len(b))
        b[at] = rnd.choice(b'{}[]:,"\\ \n\t0179-+.eEtfnu\x00\x1f\x80')
        if rnd.random() < 0.3:  # a second corruption nearby
            b[min(len(b) - 1, at + rnd.randrange(1, 70))] = rnd.choice(b'{}[]:,"\\ u')
        check(ctx, bytes(b), nd, "mutation@%d" % at)


def test_reference_corpora(ctx):  # simdjson_amd64_test.go fail / pass / ND tables
    for key in ("fail_cases", "pass_cases"):
        for c in CORP[key]:
            check(ctx, bytes.fromhex(c["js_hex"]), False, c["name"])
    for c in CORP["parse_nd"]:
        check(ctx, bytes.fromhex(c["js_hex"]), True, c["name"])


def test_string_table_through_parse(ctx):  # parse_string_test.go:19-235
    for r in GU.load("strings"):
        body = bytes.fromhex(r["str_hex"])
        check(ctx, b'["' + body + b'"]', False, r["name"])
        check(ctx, b'{"' + body + b'":"' + body + b'x"}', False, r["name"])


def test_number_tables_through_parse(ctx):  # parse_json_amd64_test.go:222-538, parse_number_test.go
    for r in NUM["atof"] + NUM["parse_int64"] + NUM["parse_number"]:
        s = r["input"].encode()
        check(ctx, b"[" + s + b"]", False, "num " + r["input"][:30])
        check(ctx, b'{"a":' + s + b"}", False, "num " + r["input"][:30])
    for s in NUM["valid"] + NUM["invalid"]:
        check(ctx, b"[1," + s.encode() + b",2]", False, "numv " + s)


def test_float_rounding_torture(ctx):
    import struct
    import decimal
    decimal.getcontext().prec = 1200
    rnd = random.Random(99)
    docs = []
    for i in range(3000):
        bits = rnd.getrandbits(64) & 0x7FFFFFFFFFFFFFFF
        if (bits >> 52) == 0x7FF:
            continue
        d = struct.unpack("<d", struct.pack("<Q", bits))[0]
        docs.append(repr(d))
        docs.append("%.25e" % d)
        nxt = struct.unpack("<d", struct.pack("<Q", bits + 1))[0]
        if nxt != float("inf"):
            mid = (decimal.Decimal(d) + decimal.Decimal(nxt)) / 2
            s = format(mid, "e")
            m, e = s.split("e")
            docs += [s, m + "1e" + e, m + "00000000000000000001e" + e]
    for i in range(0, len(docs), 500):
        check(ctx, ("[" + ",".join(docs[i:i + 500]) + "]").encode(), False, "floats")


def test_byte_soup_with_controls_and_high_bytes(ctx):
    """short random byte strings over an alphabet that includes control characters, DEL, bytes >= 0x80, unicode
    escape material and number characters: every verdict and every accepted tape must be the oracle's"""
    rng = np.random.default_rng(11)
    alpha = np.frombuffer(b'{}[]:,""\\\\u00dD8aAfF19 \n\t-+.eE0tn\x00\x1f\x7f\x80\xc3\xa9\xff', dtype=np.uint8)
    for trial in range(2500):
        body = bytes(alpha[rng.integers(0, alpha.size, int(rng.integers(1, 60)))])
        check(ctx, body, bool(trial & 1), "soup2")
        check(ctx, b'["' + body.replace(b'"', b"") + b'"]', False, "soup2-in-string")


def test_numbers_cut_by_the_32_byte_window(ctx):
    nums = workloads.window_cut_numbers()
    for n in nums[-7:]:  # malformed ones: each alone
        check(ctx, ("[" + n + "]").encode(), False, "cut-bad")
    good = nums[:-7]
    for i in range(0, len(good), 200):
        check(ctx, ("[" + ",".join(good[i:i + 200]) + "]").encode(), False, "cut")
        check(ctx, ("[ " + " , ".join(good[i:i + 200]) + " ]").encode(), False, "cut-spaced")


def test_random_documents(ctx):
    rnd = random.Random(5)
    rng = np.random.default_rng(7)
    alpha = np.frombuffer(b'{}[]:,"""  \n\\tfn0123-.e"a', dtype=np.uint8)
    for trial in range(600):
        body = bytes(alpha[rng.integers(0, alpha.size, int(rng.integers(1, 40)))])
        check(ctx, body, bool(trial & 1), "soup")

    def gen(depth=0):
        r = rnd.random()
        if depth > 6 or r < 0.3:
            return rnd.choice(['1', '-2.5e3', 'true', 'false', 'null', '"s"', '"a\\nb"', '"\\u00e9"', '[]', '{}',
                               '12345678901234567890', '0.1', '"\\ud83d\\ude00"'])
        if r < 0.65:
            return '[' + ','.join(gen(depth + 1) for _ in range(rnd.randint(0, 5))) + ']'
        return '{' + ','.join('"k%d":%s' % (i, gen(depth + 1)) for i in range(rnd.randint(0, 5))) + '}'

    for trial in range(300):
        doc = gen()
        if doc[0] not in '[{':
            doc = '[' + doc + ']'
        doc = doc.replace(',', ',' + rnd.choice(['', ' ', '\n ', '\t']))
        check(ctx, doc.encode(), False, 'gen')
        b = bytearray(doc.encode())
        if len(b) > 2:
            b[rnd.randrange(len(b))] = rnd.choice(b'{}[]:,"\\ 1tx')
            check(ctx, bytes(b), False, 'mut')
        lines = '\n'.join('{"a":%s}' % gen() for _ in range(rnd.randint(1, 5)))
        check(ctx, lines.encode(), True, 'gennd')


def test_deep_nesting_and_long_flat_containers(ctx):
    check(ctx, b"[" * 5000 + b"]" * 5000, False, "deep")
    check(ctx, b"[" * 5000 + b"]" * 4999, False, "deep-unbalanced")
    check(ctx, b'{"a":' * 3000 + b"1" + b"}" * 3000, False, "deep-obj")
    check(ctx, b"[" + b",".join(b"%d" % i for i in range(200000)) + b"]", False, "flat")
    check(ctx, b"[" + b",".join(b"[%d,%d]" % (i, i) for i in range(100000)) + b"]", False, "pairs")
    check(ctx, b"[" + b",".join(b'{"k":"v%d"}' % i for i in range(50000)) + b"]", False, "objs")
    check(ctx, b"\n".join(b'{"k":[%d,{"z":null}]}' % i for i in range(50000)), True, "nd-many")
    # sawtooth depth: partners and parents spread over several 64-bracket groups and min-tree levels
    saw = b"[" + b",".join(b"[" * (i % 70) + b"1" + b"]" * (i % 70) for i in range(1, 20000)) + b"]"
    check(ctx, saw, False, "sawtooth")
    check(ctx, saw[:-200] + b"}" + saw[-199:], False, "sawtooth-mismatch")
    mixed = b'{"r":' * 200 + b"[" + b",".join(b'{"a":[1,{"b":[]},[[2]]]}' for _ in range(30000)) + b"]" + b"}" * 200
    check(ctx, mixed, False, "mixed")
    check(ctx, mixed[:-150] + b"]" + mixed[-149:], False, "mixed-mismatch")


def test_trim_space_variants(ctx):
    for pre in (b"", b" \t\r\n", b"\x0b\x0c", "  ".encode(), b"\xc2\x85"):
        for post in (b"", b"\n\n", "　".encode(), b" \xe2\x80\xa8"):
            check(ctx, pre + b'{"a":[1,2,"x"]}' + post, False, "trim")
            check(ctx, pre + b'{"a":1}\n{"b":2}' + post, True, "trim-nd")
    check(ctx, b"   ", False, "blank")
    check(ctx, b"", False, "empty")


def test_multi_tile_documents(ctx):
    check(ctx, workloads.c2_twitter_array(12), False, "twitter x12")
    check(ctx, fixtures.load("parking-citations") * 12, True, "parking x12")
    check(ctx, b"[" + b",".join([fixtures.load("canada").strip()] * 4) + b"]", False, "canada x4")


def _random_value(rnd, depth):
    r = rnd.random()
    if depth > 7 or r < 0.35:
        k = rnd.randrange(12)
        if k == 0:
            return rnd.choice(["true", "false", "null"])
        if k == 1:
            return str(rnd.randrange(-10**18, 10**18))
        if k == 2:
            return repr(rnd.uniform(-1e6, 1e6))
        if k == 3:
            return "%d.%de%+d" % (rnd.randrange(10**rnd.randrange(1, 25)), rnd.randrange(10**6), rnd.randrange(-330, 310))
        if k == 4:
            return str(rnd.randrange(2**63 - 5, 2**64 + 5))
        # strings: plain runs of every length around the 64-byte / 4 KiB boundaries, escapes, unicode, surrogates
        parts = []
        for _ in range(rnd.randrange(0, 6)):
            c = rnd.randrange(9)
            if c < 4:
                parts.append("abcdefghijklmnopqrstuvwxyz0123456789 ,:{}[]"[rnd.randrange(43)] * rnd.choice([1, 3, 7, 31, 63, 64, 65, 200]))
            elif c == 4:
                parts.append(rnd.choice(['\\n', '\\t', '\\"', '\\\\', '\\/', '\\b', '\\f', '\\r']))
            elif c == 5:
                parts.append("\\u%04x" % rnd.choice([0x41, 0xe9, 0x20ac, 0x7ff, 0x800, 0xffff, 0x0]))
            elif c == 6:
                parts.append("\\ud83d\\ude00" if rnd.random() < 0.8 else "\\ud800\\u0041")
            elif c == 7:
                parts.append("\u00e9\u4e16\U0001f600")
            else:
                parts.append("\\" * (2 * rnd.randrange(1, 40)))
        return '"' + "".join(parts) + '"'
    if r < 0.65:
        return "[" + ",".join(_random_value(rnd, depth + 1) for _ in range(rnd.randrange(0, 6))) + "]"
    return "{" + ",".join('"k%d":%s' % (i, _random_value(rnd, depth + 1)) for i in range(rnd.randrange(0, 6))) + "}"


def _random_records(seed, nbytes):
    rnd = random.Random(seed)
    lines = []
    size = 0
    while size < nbytes:
        v = _random_value(rnd, 0)
        if v[0] not in "[{":
            v = "[" + v + "]"
        v = v.replace(",", "," + rnd.choice(["", " ", "\t", "  "]))
        if O.parse(v.encode("utf-8"), ndjson=False, copy_strings=True).rc != 0:
            continue  # e.g. a float that overflows: the document must be one the reference accepts
        lines.append(v)
        size += len(v) + 1
    return rnd, lines


def test_large_random_ndjson(ctx):
    """~20 MB of generated records in one ParseND: every tile / unit / chunk boundary falls somewhere inside
    strings, escapes, numbers and nested containers; whole and sharded parses must equal the oracle's."""
    from sjhip import ndshard
    rnd, lines = _random_records(20260922, 20 << 20)
    doc = ("\n".join(lines) + rnd.choice(["", "\n", "\n\n "])).encode("utf-8")
    check(ctx, doc, True, "random-nd")
    for copy_strings in (True, False):
        ref = O.parse(doc, ndjson=True, copy_strings=copy_strings)
        assert ref.rc == 0
        trim, begin, finish = ndshard.device_callbacks(ctx, copy_strings)
        world = 5
        sizes = []
        for a, b in ndshard.record_cuts(doc, world):
            off, ln = trim(doc[a:b])
            sizes.append((0, 0) if ln == 0 else begin(doc[a + off:a + off + ln]))
        tapes, strs = [], []
        for r in range(world):
            t, s, _, _ = ndshard.parse_shard(doc, r, world, trim, begin, finish, lambda s: sizes if len(s) == 3 else [(0,)] * world, copy_strings)
            tapes.append(t)
            strs.append(s)
        assert np.array_equal(np.concatenate(tapes), ref.tape)
        assert np.array_equal(np.concatenate(strs), ref.strings)
    # one corrupted byte somewhere in the middle: same verdict as the oracle
    for _ in range(12):
        b = bytearray(doc)
        b[rnd.randrange(len(b))] = rnd.choice(b'{}[]:,"\\ 1tx\n')
        check(ctx, bytes(b), True, "random-nd-mut")


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_documents_at_scale(ctx, seed):
    """the same generator, other seeds: one JSON document (records as the members of an array nested a few
    hundred levels deep) and the records as NDJSON with blank lines"""
    rnd, lines = _random_records(seed, 6 << 20)
    depth = rnd.randrange(1, 300)
    doc = ('{"r":' * depth + "[" + ",\n".join(lines) + "]" + "}" * depth).encode("utf-8")
    check(ctx, doc, False, "random-array")
    nd = ("\n\n".join(lines) + "\n").encode("utf-8")
    check(ctx, nd, True, "random-nd-blank-lines")
    for _ in range(6):
        b = bytearray(doc)
        b[rnd.randrange(len(b))] = rnd.choice(b'{}[]:,"\\ 1tx\n')
        check(ctx, bytes(b), False, "random-array-mut")


def test_parse_nd_stream():
    """ParseNDStream (simdjson_amd64.go:101-216): blocks cut at record ends, parsed concurrently, delivered in
    order; every block's ParsedJson is the oracle's ParseND of that block; the first bad block ends the stream."""
    import io
    import sjhip
    park = fixtures.load("parking-citations")
    stream = park * 24  # ~9 MB
    bs = 1 << 20
    blocks = list(sjhip.cut_blocks(io.BytesIO(stream), bs))
    assert b"".join(blocks) == stream and all(b.endswith(b"\n") and len(b) >= bs for b in blocks[:-1])
    for inflight in (1, 3):
        got = list(sjhip.parse_nd_stream(io.BytesIO(stream), block_size=bs, inflight=inflight))
        assert len(got) == len(blocks)
        for pj, blk in zip(got, blocks):
            ref = O.parse(blk, ndjson=True, copy_strings=True)
            assert ref.rc == 0
            assert np.array_equal(pj.Tape, ref.tape) and np.array_equal(pj.Strings, ref.strings)
            assert pj.Message == bytes(blk[ref.msg_off:ref.msg_off + ref.msg_len])
    # the results read in place (views of the stream's pinned blocks, valid until the generator is resumed): the same
    # blocks, nothing copied; with one slot, with a filter, and ended early by the consumer
    for inflight in (1, 3):
        n = 0
        for pj, blk in zip(sjhip.parse_nd_stream(io.BytesIO(stream), block_size=bs, inflight=inflight, view=True), blocks):
            ref = O.parse(blk, ndjson=True, copy_strings=True)
            assert not pj.Tape.flags.writeable
            assert np.array_equal(pj.Tape, ref.tape) and np.array_equal(pj.Strings, ref.strings)
            assert pj.Message == bytes(blk[ref.msg_off:ref.msg_off + ref.msg_len])
            n += 1
        assert n == len(blocks)
    it = sjhip.parse_nd_stream(io.BytesIO(stream), block_size=bs, inflight=2, view=True, where=(b"Make", b"HOND"))
    assert sum(pj.records for pj in it) == 116 * 24
    it = sjhip.parse_nd_stream(io.BytesIO(stream), block_size=bs, inflight=2, view=True)
    first = next(it)
    assert first.Tape.size > 0
    it.close()  # (a held block is released, the stream destroyed)
    # a broken record in the fourth block: three results, then the error
    bad = bytearray(stream)
    at = sum(len(b) for b in blocks[:3]) + 5000
    bad[at:at + 1] = b"\x01" if bad[at:at + 1] != b"\x01" else b"\x02"
    it = sjhip.parse_nd_stream(io.BytesIO(bytes(bad)), block_size=bs, inflight=2)
    n_ok = 0
    with pytest.raises(sjhip.ParseError):
        for _ in it:
            n_ok += 1
    ref_bad = O.parse(bytes(bad), ndjson=True, copy_strings=True)
    assert ref_bad.rc != 0 and n_ok == 3


def test_stream_long_records_reuse_and_whitespace_block():
    """sjhip_stream_*: a record longer than the pinned block (grow path), `reuse` recycling, every visible device,
    and a whitespace-only block, which the reference parses and rejects (simdjson_amd64.go:178, parseMessage)."""
    import io
    import queue
    import sjhip
    rnd = random.Random(11)
    recs = [('{"i":%d,"s":"%s"}' % (i, "x" * rnd.choice([3, 50, 700]))).encode() for i in range(3000)]
    recs[1500] = b'{"big":"' + b"y" * (3 << 20) + b'"}'          # 3 MiB record, blocks of 256 KiB
    stream = b"\n".join(recs) + b"\n"
    bs = 256 << 10
    blocks = list(sjhip.cut_blocks(io.BytesIO(stream), bs))
    back = queue.SimpleQueue()
    got = []
    for pj in sjhip.parse_nd_stream(io.BytesIO(stream), block_size=bs, inflight=4, reuse=back, n_devices=0):
        got.append((bytes(pj.Message), pj.Tape.copy(), pj.Strings.copy()))
        back.put(pj)
    assert len(got) == len(blocks)
    for (msg, tape, strs), blk in zip(got, blocks):
        ref = O.parse(blk, ndjson=True, copy_strings=True)
        assert ref.rc == 0 and np.array_equal(tape, ref.tape) and np.array_equal(strs, ref.strings)
        assert msg == bytes(blk[ref.msg_off:ref.msg_off + ref.msg_len])
    # blank lines for more than a block: the second block is whitespace only -> the stream ends with its error
    ws = b'{"a":1}\n' + b"\n" * (3 * bs) + b'{"b":2}\n'
    n_ok = 0
    with pytest.raises(sjhip.ParseError):
        for _ in sjhip.parse_nd_stream(io.BytesIO(ws), block_size=bs, inflight=2):
            n_ok += 1
    assert n_ok <= 1


def test_concurrent_contexts():
    """One context per concurrent parse (the reference's goroutine-per-parse model, benchmarks_test.go:60-75):
    four host threads parse different documents on the same GPU at the same time; every result must be the oracle's."""
    import threading
    import sjhip
    docs = [(workloads.c2_twitter_array(3), False), (fixtures.load("parking-citations") * 3, True),
            (b"[" + b",".join([fixtures.load("canada").strip()] * 2) + b"]", False),
            (b"[" + b",".join(b'{"k":[%d,{"z":"v%d"}]}' % (i, i) for i in range(40000)) + b"]", False)]
    refs = [O.parse(d, ndjson=nd, copy_strings=True) for d, nd in docs]
    errors = []

    def worker(k):
        try:
            c = sjhip.Context(0)
            d, nd = docs[k]
            for _ in range(6):
                pj = c.parse(d, ndjson=nd, copy_strings=True)
                if not (np.array_equal(pj.Tape, refs[k].tape) and np.array_equal(pj.Strings, refs[k].strings)):
                    errors.append((k, "mismatch"))
            c.close()
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(docs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("lead", [1, 17, 63])
def test_parse_device_unaligned_pointer(ctx, lead):
    """sjhip_parse_device takes any device pointer: the kernels work from the 64-byte aligned base and the string
    masks, token kinds and Strings.B offsets are all relative to it"""
    import sjhip
    import torch
    docs = [(workloads.c2_twitter_array(3), False), (fixtures.load("twitterescaped"), False),
            (fixtures.load("parking-citations") * 5, True), (b'["a\\u00e9b","' + b"x" * 200 + b'",-1.5e3,{"k":[true,null]}]', False)]
    for doc, nd in docs:
        doc = doc.strip()  # parse_device takes the TrimSpace'd message
        dev = torch.zeros(len(doc) + 512, dtype=torch.uint8, device="cuda:0")
        dev[lead:lead + len(doc)].copy_(torch.frombuffer(bytearray(doc), dtype=torch.uint8))
        torch.cuda.synchronize()
        for copy in (True, False):
            ref = O.parse(doc, ndjson=nd, copy_strings=copy)
            assert ref.rc == 0 and ref.msg_off == 0 and ref.msg_len == len(doc)
            tl, sl = ctx.parse_device(dev.data_ptr() + lead, len(doc), ndjson=nd, copy_strings=copy)
            tape, strings = ctx.fetch(tl, sl)
            assert np.array_equal(tape, ref.tape), (lead, nd, copy)
            assert np.array_equal(strings, ref.strings), (lead, nd, copy)


def _device_doc(doc):
    import torch
    dev = torch.empty(len(doc) + 256, dtype=torch.uint8, device="cuda:0")
    dev[:len(doc)].copy_(torch.frombuffer(bytearray(doc), dtype=torch.uint8))
    torch.cuda.synchronize()
    return dev


def _assert_same_words(got, want, what):
    assert len(got) == len(want), (what, len(got), len(want))
    if not np.array_equal(got, want):
        d = np.nonzero(got != want)[0]
        raise AssertionError((what, "first differences at", d[:5].tolist(), [hex(int(x)) for x in got[d[:3]]],
                              [hex(int(x)) for x in want[d[:3]]]))


def test_full_size_c2_bit_exact(ctx):
    """BASELINE configs[1] at full size (twitter.json x426 in one array, 269 025 391 B, 21.2 M tape words): the whole
    Tape and Strings.B against the oracle's parse of the same bytes, in both copy modes -- every container payload and
    every Strings.B / Message offset, i.e. where a 32-bit offset or a cross-tile prefix would break -- plus the
    closed forms of SURVEY.md section 8d."""
    doc = workloads.c2_twitter_array(426)
    dev = _device_doc(doc)
    one = O.parse(fixtures.load("twitter"))
    for copy in (True, False):
        ref = O.parse(doc, ndjson=False, copy_strings=copy)
        assert ref.rc == 0
        tl, sl = ctx.parse_device(dev.data_ptr(), len(doc), ndjson=False, copy_strings=copy)
        assert tl == 426 * (len(one.tape) - 2) + 4 == 21_206_710
        if copy:
            assert sl == 426 * len(one.strings)
        tape, strings = ctx.fetch(tl, sl)
        _assert_same_words(tape, ref.tape, f"C2 tape copy={copy}")
        _assert_same_words(strings, ref.strings, f"C2 strings copy={copy}")
        del ref, tape, strings


def test_full_size_c5_bit_exact(ctx):
    """BASELINE configs[4] at full size (parking-citations x1000 as one ParseND document: 372.7 MB, 80 M tape words,
    1 M records): whole Tape and Strings.B against the oracle, the closed forms, and the reference's functional
    golden Make == "HOND" (116 per file, ndjson_test.go:250-267) evaluated on the fetched tape."""
    nd = workloads.c5_parking_nd(1000).rstrip(b"\n")
    dev = _device_doc(nd)
    tl, sl = ctx.parse_device(dev.data_ptr(), len(nd), ndjson=True, copy_strings=True)
    assert tl == 80_000_000 and sl == 256_664_000
    tape, strings = ctx.fetch(tl, sl)
    ref = O.parse(nd, ndjson=True, copy_strings=True)
    assert ref.rc == 0
    _assert_same_words(tape, ref.tape, "C5 tape")
    _assert_same_words(strings, ref.strings, "C5 strings")
    del ref
    tags = tape >> np.uint64(56)
    assert int((tags == ord("r")).sum()) == 2 * 1_000_000
    sview = strings
    idx = np.nonzero(tags == ord('"'))[0]
    offs = (tape[idx] & np.uint64((1 << 55) - 1)).astype(np.int64)
    lens = tape[idx + 1].astype(np.int64)

    def eq4(o, word):
        w = np.frombuffer(word, dtype=np.uint8)
        o = np.minimum(o, len(sview) - 4)
        return (sview[o] == w[0]) & (sview[o + 1] == w[1]) & (sview[o + 2] == w[2]) & (sview[o + 3] == w[3])

    l4 = np.nonzero(lens == 4)[0]
    l4 = l4[l4 + 1 < len(idx)]
    keys = l4[eq4(offs[l4], b"Make")]
    vals = keys + 1                      # the value string follows its key in tape order
    hond = (lens[vals] == 4) & eq4(offs[vals], b"HOND")
    assert int(hond.sum()) == S2["parking_citations_hond"] * 1000
    # selective copy at full size: nothing is copied (no escapes in this file), offsets point into Message
    del tape, strings
    tl2, sl2 = ctx.parse_device(dev.data_ptr(), len(nd), ndjson=True, copy_strings=False)
    ref2 = O.parse(nd, ndjson=True, copy_strings=False)
    tape2, strings2 = ctx.fetch(tl2, sl2)
    _assert_same_words(tape2, ref2.tape, "C5 tape nocopy")
    _assert_same_words(strings2, ref2.strings, "C5 strings nocopy")


# ---- sharded ParseND (the multi-GPU path, here with the shards parsed one after the other on one GPU) ----
@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("copy_strings", [True, False])
def test_sharded_parse_nd_equals_oracle(ctx, world, copy_strings):
    from sjhip import ndshard
    park = fixtures.load("parking-citations")
    lines = park.split(b"\n")
    docs = [b"\n".join(lines[:200]) + b"\n", b"  \n" + b"\n\n".join(lines[:9]) + b"\n\n \n",
            b'{"a":"x\\ny","b":[1,2.5e3,{"c":null}]}\n[1,2]\n{"k":"\\u00e9\\ud83d\\ude00"}']
    trim, begin, finish = ndshard.device_callbacks(ctx, copy_strings)
    for doc in docs:
        # pass 1: every "rank" measures its shard (what the all_gather distributes)
        sizes = []
        for a, b in ndshard.record_cuts(doc, world):
            off, ln = trim(doc[a:b])
            sizes.append((0, 0) if ln == 0 else begin(doc[a + off:a + off + ln]))
        # pass 2: every rank parses its shard with the gathered sizes
        tapes, strs = [], []
        for r in range(world):
            t, s, _, _ = ndshard.parse_shard(doc, r, world, trim, begin, finish, lambda s: sizes if len(s) == 3 else [(0,)] * world, copy_strings)
            tapes.append(t)
            strs.append(s)
        ref = O.parse(doc, ndjson=True, copy_strings=copy_strings)
        assert np.array_equal(np.concatenate(tapes), ref.tape)
        assert np.array_equal(np.concatenate(strs), ref.strings)


def test_emit_sixteen_tokens_per_lane_equals_default():
    """SJHIP_S2_ITEMS=16 (stage2.hip stage2_launch_emit) runs the emit pass with sixteen tokens per lane (256 threads per
    4096-token tile) instead of eight -- the A/B shape of round 5.  Both copy modes must produce the oracle's tape; the
    variable is read once per process, so the variant runs in its own interpreter."""
    import os
    import subprocess
    import sys
    code = r"""
import sys
sys.path.insert(0, 'simdjson-go_amd'); sys.path.insert(0, 'tests')
import numpy as np
import fixtures, sjhip, oracle_lib as O, workloads
ctx = sjhip.Context(0)
docs = [(fixtures.load(n), n == 'parking-citations') for n in ('twitter', 'twitterescaped', 'canada', 'parking-citations', 'mesh.pretty')]
docs.append((workloads.c5_parking_nd(30), True))            # > SJHIP_SMALL_BYTES: two host synchronisations
docs.append((workloads.c2_twitter_array(9), False))
for copy in (True, False):
    for d, nd in docs:
        ref = O.parse(d, ndjson=nd, copy_strings=copy)
        pj = ctx.parse(d, ndjson=nd, copy_strings=copy)
        assert ref.rc == 0 and np.array_equal(pj.Tape, ref.tape) and np.array_equal(pj.Strings, ref.strings), (len(d), nd, copy)
print('ok')
"""
    env = dict(os.environ, SJHIP_S2_ITEMS="16")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-1500:])


def test_fetch_after_small_parse_paths(ctx):
    """sjhip_parse of a small document leaves its result in pinned host memory (k_pack) and sjhip_fetch copies from there;
    a stage-1-only call in between, a result larger than the pinned block, a parse that needs the bignum pass and a
    second fetch must all still deliver the oracle's tape."""
    import ctypes as C
    import sjhip
    from sjhip import _lib
    L = _lib.lib()
    big_tape = ("[" + ",".join(["1"] * 300000) + "]").encode()            # 600 KB in, 4.8 MB of tape out
    bignum = b'[1.00000000000000011102230246251565404236316680908203125, 123456789012345678901234567890e-10, "x"]'
    for doc, nd in ((fixtures.load("twitter"), False), (fixtures.load("parking-citations"), True), (big_tape, False), (bignum, False)):
        ref = O.parse(doc, ndjson=nd, copy_strings=True)
        assert ref.rc == 0
        a = np.frombuffer(doc, dtype=np.uint8)
        for stage1_between in (False, True):
            tl, sl, mo, ml = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
            rc = L.sjhip_parse(ctx._h, a.ctypes.data, a.size, (1 if nd else 0) | 2, C.byref(tl), C.byref(sl), C.byref(mo), C.byref(ml))
            assert rc == 0 and tl.value == len(ref.tape) and sl.value == len(ref.strings)
            if stage1_between:
                ok, pos = ctx.stage1(fixtures.load("canada").strip())
                assert ok
            for _ in range(2):
                tape = np.zeros(tl.value, dtype=np.uint64)
                strs = np.zeros(sl.value, dtype=np.uint8)
                assert L.sjhip_fetch(ctx._h, tape.ctypes.data, strs.ctypes.data) == 0
                assert np.array_equal(tape, ref.tape) and np.array_equal(strs, ref.strings), (len(doc), stage1_between)


def test_fetch_view_equals_fetch(ctx):
    """sjhip_fetch_view: Tape / Strings read in place from the context's pinned memory -- the packed block of a small
    parse, the grown view block of everything else (results beyond 2 MiB, ND, nocopy, a parse that needed the bignum
    pass) -- are the arrays sjhip_fetch copies out; the views are read-only and are overwritten by the next parse."""
    import sjhip
    docs = [(fixtures.load("twitter"), False), (fixtures.load("canada"), False), (fixtures.load("twitterescaped"), False),
            (fixtures.load("parking-citations") * 9, True), (b'{"a":1}', False), (b"[]", False),
            (b"[" + b"123456789012345678901234567890e-5," * 40 + b"1]", False),
            (b"\n".join(b'{"k":"%d\\n","v":[%d]}' % (i, i) for i in range(30000)), True)]
    for data, nd in docs:
        for copy in (True, False):
            want = ctx.parse(data, ndjson=nd, copy_strings=copy)
            wt, ws = want.Tape.copy(), want.Strings.copy()
            got = ctx.parse(data, ndjson=nd, copy_strings=copy, view=True)
            assert not got.Tape.flags.writeable and (got.Strings.size == 0 or not got.Strings.flags.writeable)
            assert np.array_equal(got.Tape, wt) and np.array_equal(got.Strings, ws), (len(data), nd, copy)
            assert got.Message == want.Message
    # the input read into the context's pinned block (sjhip_input_block): same results, the block grows with the request
    for data, nd in docs[:4]:
        blk = ctx.input_block(len(data))
        blk[:] = np.frombuffer(data, dtype=np.uint8)
        got = ctx.parse(blk, ndjson=nd, view=True)
        want = O.parse(data, ndjson=nd)
        assert np.array_equal(got.Tape, want.tape) and np.array_equal(got.Strings, want.strings), (len(data), nd)
    # the view of a large result, then a small one, then the large one again (the block only grows)
    big = ctx.parse(docs[3][0], ndjson=True, view=True)
    big_t = big.Tape.copy()
    small = ctx.parse(b'[1,"x"]', view=True)
    assert small.Tape.tolist() == O.parse(b'[1,"x"]').tape.tolist()
    again = ctx.parse(docs[3][0], ndjson=True, view=True)
    assert np.array_equal(again.Tape, big_t)
    # a failed parse leaves no view behind
    with pytest.raises(sjhip.ParseError):
        ctx.parse(b'{"a":', view=True)


def test_small_documents_denser_than_the_deferred_layout(ctx):
    """Documents up to SJHIP_SMALL_BYTES run with one host synchronisation on stage-2 arrays laid out for one token per
    four bytes (csrc/parse_api.hip); a denser document must come back through the synchronous path with the same
    result -- and an invalid dense one with its own error class, not with whatever the clamped run saw."""
    rnd = random.Random(11)
    docs = [
        b"[" + b",".join([b"1"] * 40000) + b"]",                       # one token per byte
        b"[" * 3000 + b"]" * 3000,
        b"[" + b",".join(b'{"a":[%d,"%s"]}' % (rnd.randrange(10), b"x" * rnd.randrange(3)) for _ in range(30000)) + b"]",
        b"\n".join(b"[[],{},[1]]" for _ in range(20000)),
        b"[" + b",".join([b"1"] * 16384) + b"]",                       # right at the layout's edge (len/4 + 4096)
    ]
    for i, d in enumerate(docs):
        check(ctx, d, nd=(i == 3), what=f"dense {i}")
        check(ctx, fixtures.load("payload-small"), what="sparse after dense")
    check(ctx, b"[" + b",".join([b"1"] * 40000) + b",]", what="dense, stage-2 error")
    check(ctx, b"[" + b",".join([b"1"] * 40000), what="dense, stage-1 error")


def test_arena_allocation_failure_is_an_error_not_a_verdict():
    """A context's arenas are grown with hipMalloc when a larger document comes along.  When the device has no room the
    call returns SJHIP_ERR_HIP with the allocation in its message -- not a parse verdict -- and the context works again
    once memory is there (the failed arena starts from scratch)."""
    import torch
    import sjhip
    c = sjhip.Context(0)
    doc = workloads.c2_twitter_array(40)                      # 25 MB: its arenas need a few hundred MB
    dev = _device_doc(doc)
    small = fixtures.load("payload-small")
    assert c.parse(small).Tape.size > 0                         # streams, scratch and small arenas exist before the squeeze
    hog = []
    try:
        free, _ = torch.cuda.mem_get_info()
        while free > (96 << 20):                                # leave less than the parse needs
            take = min(free - (64 << 20), 32 << 30)
            if take < (16 << 20):
                break
            hog.append(torch.empty(take, dtype=torch.uint8, device="cuda:0"))
            free, _ = torch.cuda.mem_get_info()
        try:
            c.parse_device(dev.data_ptr(), len(doc))
            failed = None
        except sjhip.ParseError as e:
            failed = e
    finally:
        del hog
        torch.cuda.empty_cache()
    if failed is None:  # (seen at the end of a long test session: the runtime still found room although < 96 MB were reported free)
        pytest.skip("device memory could not be exhausted from this process")
    assert failed.code == -1 and "hipMalloc" in str(failed), (failed.code, str(failed))
    ref = O.parse(doc)
    tl, sl = c.parse_device(dev.data_ptr(), len(doc))           # the same context, now with room
    tape, strings = c.fetch(tl, sl)
    assert np.array_equal(tape, ref.tape) and np.array_equal(strings, ref.strings)
    c.close()


def test_strings_on_chunk_and_unit_boundaries(ctx):
    """tests/workloads.py string_boundary_documents through the kernels, both copy modes: string ends, escapes and the
    blanks between a closing quote and the next token on 64-byte and 4 KiB seams (the selective copy finds the closing
    quote by walking back from the next token, possibly over whole chunks of blanks)."""
    docs = workloads.string_boundary_documents()
    for what, d in docs:
        check(ctx, d, False, what)
    # all of them as one ND message, and inside one large array (many tiles, unaligned starts)
    check(ctx, b"\n".join(d for _, d in docs), True, "boundaries/nd")
    check(ctx, b"[" + b",".join(d for _, d in docs) + b"]", False, "boundaries/array")


def test_trim_gives_the_arenas_back_and_the_context_keeps_working():
    """sjhip_ctx_device_bytes / sjhip_ctx_trim: a context that has parsed a large message holds arenas sized for it; trim
    frees them (and drops the resident result: queries and fetches of it are refused or empty), the next parses -- small
    and large, every entry point -- allocate again and give the oracle's result."""
    import sjhip
    c = sjhip.Context(0)
    assert c.device_bytes() == 0
    small = fixtures.load("twitter")
    big = fixtures.load("parking-citations") * 40  # 15 MB: the synchronous path
    ref_small, ref_big = O.parse(small), O.parse(big, ndjson=True)
    pj = c.parse(small)
    assert np.array_equal(pj.Tape, ref_small.tape)
    b_small = c.device_bytes()
    assert 0 < b_small < 100 << 20
    pj = c.parse(big, ndjson=True, view=True, key_flags=True)
    assert np.array_equal(pj.Tape, ref_big.tape) and np.array_equal(pj.Strings, ref_big.strings)
    b_big = c.device_bytes()
    assert b_big > 10 * len(big)
    c.trim()
    assert c.device_bytes() == 0
    with pytest.raises(sjhip.ParseError):
        c.marshal_json()  # no resident result any more
    for _ in range(2):
        pj = c.parse(small, view=True)
        assert np.array_equal(pj.Tape, ref_small.tape) and np.array_equal(pj.Strings, ref_small.strings)
        assert c.device_bytes() <= b_small
        c.trim()
    pj = c.parse(big, ndjson=True, copy_strings=False)
    ref_nc = O.parse(big, ndjson=True, copy_strings=False)
    assert np.array_equal(pj.Tape, ref_nc.tape) and np.array_equal(pj.Strings, ref_nc.strings)
    assert c.count_where(b"Make", b"HOND") == 116 * 40
    ok, pos = c.stage1(small)
    assert ok
    n1 = int(pos.size)
    c.trim()
    ok, pos = c.stage1(small)
    assert ok and int(pos.size) == n1
    c.close()


def test_units_without_tokens(ctx):
    """Strings and runs of blanks longer than a 4 KiB unit: units that hold no token and no quote at all.  The string path works
    unit by unit -- a string that is open at a unit's end is followed into the units behind it (WithCopyStrings(false): whether
    it holds an escape, where it closes), tiles of 4096 tokens end inside the runs.  Strings of 5-40 KB with and without
    escapes, blank runs between tokens, both copy modes, plain and ND."""
    rnd = random.Random(11)
    parts = []
    for i in range(60):
        n = rnd.choice([10, 3000, 5000, 9000, 20000, 40000])
        body = ("x" * n) if i % 3 else ("a\\n" * (n // 3))
        blanks = " " * rnd.choice([0, 1, 4097, 10000]) + "\n" * rnd.choice([0, 1, 5000])
        parts.append('{"k%d":%s"%s"%s,"n":%d,"t":true%s}' % (i, blanks, body, blanks, i, blanks))
    doc = ("[" + ",".join(parts) + "]").encode()
    check(ctx, doc, False, "long strings and blank runs")
    # the same records as ND (blank runs without newlines), and small tokens in between so that tiles end inside the runs
    lines = []
    for i in range(400):
        filler = ",".join('"f%d":%d' % (j, j) for j in range(rnd.randrange(0, 40)))
        n = rnd.choice([0, 4200, 8300, 12345])
        lines.append('{%s%s"s":"%s"%s}' % (filler, "," if filler else "", "y" * n, " " * rnd.choice([0, 4100, 9000])))
    check(ctx, "\n".join(lines).encode(), True, "nd long strings")


def test_very_long_strings_without_copy(ctx):
    """WithCopyStrings(false) decides per string whether it is copied; for a string that runs over many 4 KiB units the units'
    ends are resolved by walking to the unit that holds its quote -- 64 quote-free units per step on the unit flags of stage 1,
    at most 64 steps (16 MiB), beyond which the document takes the per-string path.  Strings of 300 KB, 3 MB and 17 MB whose
    only escape lies at the very beginning, the very end, or nowhere; both copy modes."""
    for n in (300_000, 3_000_000, 17_000_000):
        for where in ("none", "front", "back"):
            body = b"z" * n
            if where == "front":
                body = b"\\n" + body
            elif where == "back":
                body = body + b"\\t"
            doc = b'{"a":"short","big":"' + body + b'","b":[1,"x\\\\y",true],"c":"' + b"q" * 5000 + b'"}'
            check(ctx, doc, False, "long string %d %s" % (n, where))


def test_large_documents_of_changing_density():
    """A large document (beyond SJHIP_SMALL_BYTES) is parsed without the host round trip between the stages once its context has
    seen a parse: the stage-2 arrays are laid out for the token density of the context's last parse + 25 %.  A denser document
    than the one before must fall back to the synchronous path (and teach the context its density), a sparser one must not
    mind; every result against the oracle, own context so that the order of the densities is the test's."""
    import sjhip
    c = sjhip.Context(0)
    sparse = ('[' + ','.join('"%s"' % ('s' * 900) for _ in range(7000)) + ']').encode()        # 6.3 MB, 0.002 tokens per byte
    medium = workloads.c2_twitter_array(9)                                                      # 5.7 MB, 0.09
    dense = ('[' + ','.join('1' for _ in range(3_000_000)) + ']').encode()                      # 6 MB, 1 token per byte
    nd = (fixtures.load("parking-citations") * 14)                                              # 5.2 MB ND, 0.21
    for doc, is_nd in ((sparse, False), (medium, False), (sparse, False), (dense, False), (medium, False), (nd, True), (dense, False), (sparse, False)):
        for copy in (True, False):
            ref = O.parse(doc, ndjson=is_nd, copy_strings=copy)
            pj = c.parse(doc, ndjson=is_nd, copy_strings=copy)
            assert ref.rc == 0 and np.array_equal(pj.Tape, ref.tape) and np.array_equal(pj.Strings, ref.strings), (len(doc), copy)
    c.close()


def test_launches_follow_each_other_without_a_preparation_kernel():
    """Round 6: a stage-1 launch has nothing in front of it -- every launch zeroes the control slot and the tile descriptors
    the launch BEFORE it used (csrc/sj_device.h Stage1State), and the first / last 4 KiB unit are assembled by the wave that
    loads them.  One context, launches of very different sizes and outcomes in a row (more tiles than the one before, fewer,
    an error, stage 1 alone, the timing entry point, a growing workspace, odd alignments of both message ends): every result
    must be the oracle's whatever ran before it."""
    import sjhip
    import torch
    c = sjhip.Context(0)
    small = fixtures.load("twitter")
    big = workloads.c2_twitter_array(48)          # ~30 MB: several rounds of tiles
    mid = fixtures.load("parking-citations") * 9   # ND, a few MB
    bad = small[:40000] + b'"\x01"' + small[40000:]          # a control character inside a string: stage-1 error
    open_str = small[:len(small) - 3]                         # ends inside ... something: not a document
    seq = [(small, False), (big, False), (small, False), (bad, False), (small, False), (mid, True), (open_str, False),
           (b"[]", False), (big, False), (b'{"a":"b\\n"}', False), (mid, True)]
    refs = {}
    for rep in range(2):
        for data, nd in seq:
            for copy in (True, False):
                key = (id(data), nd, copy)
                if key not in refs:
                    refs[key] = O.parse(data, ndjson=nd, copy_strings=copy)
                ref = refs[key]
                rc, pj = gpu_parse(c, data, nd, copy)
                assert rc == ref.rc, (len(data), nd, copy, rc, ref.rc)
                if rc == 0:
                    assert np.array_equal(pj.Tape, ref.tape) and np.array_equal(pj.Strings, ref.strings), (len(data), nd, copy)
            # stage 1 alone in between (its own launch on the same workspace), then the timing entry point
            ok, pos = c.stage1(data, ndjson=nd)
            ok_ref, pos_ref = O.stage1(data, ndjson=nd)
            assert ok == ok_ref and (not ok or np.array_equal(pos, pos_ref)), (len(data), nd)
        d = torch.empty(len(big) + 4096, dtype=torch.uint8, device="cuda:0")
        p = torch.empty(len(big) // 4 + 4096, dtype=torch.int32, device="cuda:0")
        ok_ref, pos_ref = O.stage1(big)
        for lead in (0, 1, 63):  # the message's first byte anywhere in its 64-byte line; its last one wherever that puts it
            d[lead:lead + len(big)].copy_(torch.frombuffer(bytearray(big), dtype=torch.uint8))
            torch.cuda.synchronize()
            assert c.stage1_time(d.data_ptr() + lead, len(big), p.data_ptr(), p.numel(), 3) > 0
            ok, n = c.stage1_device(d.data_ptr() + lead, len(big), p.data_ptr(), p.numel())
            assert ok == ok_ref and n == len(pos_ref)
            assert np.array_equal(p[:n].cpu().numpy().view(np.uint32), pos_ref)
        if rep == 0:
            c.trim()  # a fresh workspace in the middle of the sequence
    # short messages at every alignment and length around the chunk and unit sizes: both ends of the message in one chunk, in
    # neighbouring chunks, the last byte on the last byte of a unit
    d = torch.empty(3 * 4096 + 256, dtype=torch.uint8, device="cuda:0")
    p = torch.empty(3 * 4096 + 256, dtype=torch.int32, device="cuda:0")
    for n_items in (1, 7, 9, 10, 400, 578, 579, 580, 1160):
        doc = b"[" + b",".join(b'"a\\"%d"' % (k % 10) for k in range(n_items)) + b"]"
        ok_ref, pos_ref = O.stage1(doc)
        for lead in (0, 3, 60, 63):
            d.fill_(0x22)  # quotes all around the message: a byte taken from outside it would change the result
            d[lead:lead + len(doc)].copy_(torch.frombuffer(bytearray(doc), dtype=torch.uint8))
            torch.cuda.synchronize()
            ok, n = c.stage1_device(d.data_ptr() + lead, len(doc), p.data_ptr(), p.numel())
            assert ok == ok_ref and n == len(pos_ref), (n_items, lead)
            assert np.array_equal(p[:n].cpu().numpy().view(np.uint32), pos_ref), (n_items, lead)
            ref = O.parse(doc)
            tl, sl = c.parse_device(d.data_ptr() + lead, len(doc))
            tape, strings = c.fetch(tl, sl)
            assert np.array_equal(tape, ref.tape) and np.array_equal(strings, ref.strings), (n_items, lead)
    c.close()
