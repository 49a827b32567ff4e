"""GPU: ND messages beyond one context's reach.  The reference parses "arbitrarily large" ND inputs (README.md:567-569;
its index stream is deltas, flatten_bits_amd64.s:38-41); a context here holds absolute uint32 positions, so
sjhip_parse / sjhip_parse_device cut an ND message longer than 4 GiB - 128 into shards at record boundaries inside the
library (csrc/multi_api.hip parse_nd_big) and the merged result must be the ParsedJson of the whole message.  The
threshold and the shard size can be moved with SJHIP_ND_LIMIT_BYTES / SJHIP_ND_SHARD_BYTES, which lets the same path run
on megabytes against the oracle; the real thing (4.8 GB) is checked through the closed form of its tape.  A single
document does not shard; since round 5 it parses all the same -- the 32-bit positions wrap and the token kernels rebuild them
tile by tile (test_single_document_beyond_4GiB)."""
import os

import numpy as np
import pytest

import fixtures
import oracle_lib as O

pytestmark = pytest.mark.gpu

TAG = np.uint64(56)
PAYLOAD = np.uint64((1 << 56) - 1)


@pytest.fixture
def small_limits():
    os.environ["SJHIP_ND_LIMIT_BYTES"] = str(2 << 20)
    os.environ["SJHIP_ND_SHARD_BYTES"] = str(1 << 20)
    yield
    del os.environ["SJHIP_ND_LIMIT_BYTES"], os.environ["SJHIP_ND_SHARD_BYTES"]


def _device_copy(data):
    import torch
    d = torch.empty(len(data) + 256, dtype=torch.uint8, device="cuda:0")
    d[:len(data)].copy_(torch.frombuffer(bytearray(data), dtype=torch.uint8))
    torch.cuda.synchronize()
    return d


def test_sharded_inside_parse_equals_oracle(small_limits):
    import sjhip
    ctx = sjhip.Context(0)
    park = fixtures.load("parking-citations")
    esc = b'{"k":"\\u00e9\\ud83d\\ude00 \\"q\\"","n":[1.5e3,-7,null,18446744073709551616]}\n'
    docs = [
        ("parking x16", park * 16, 0),
        ("blank lines and blanks at the cuts", b"\n\n" + (park + b"\n \r\n") * 9 + b"  ", 0),
        ("escapes and numbers", esc * 60000, 0),
        ("stage-2 error in a middle shard", park * 5 + b'{"a":[1,2}\n' + park * 5, 2),
        ("stage-1 error in the last shard", park * 9 + b'{"broken":"unterminated\n', 1),
        ("stage 1 wins", b'{"a":[1,2}\n' + park * 8 + b'{"broken":"unterminated', 1),
    ]
    for what, doc, want in docs:
        assert len(doc) > (2 << 20)
        for copy in (True, False):
            ref = O.parse(doc, ndjson=True, copy_strings=copy)
            assert ref.rc == want, what
            # from a host buffer (sjhip_parse + sjhip_fetch: the Go binding's ParseND)
            try:
                pj = ctx.parse(doc, ndjson=True, copy_strings=copy)
                rc = 0
            except sjhip.ParseError as e:
                rc = e.code
            assert rc == want, (what, copy, rc)
            if rc == 0:
                assert pj.Message == bytes(doc[ref.msg_off:ref.msg_off + ref.msg_len]), what
                assert np.array_equal(pj.Tape, ref.tape) and np.array_equal(pj.Strings, ref.strings), (what, copy)
            # device-resident (sjhip_parse_device takes the message as Parse() trims it)
            trimmed = doc[ref.msg_off:ref.msg_off + ref.msg_len] if ref.msg_len else doc.strip()
            dev = _device_copy(trimmed)
            try:
                tl, sl = ctx.parse_device(dev.data_ptr(), len(trimmed), ndjson=True, copy_strings=copy)
                rc = 0
            except sjhip.ParseError as e:
                rc = e.code
            assert rc == want, (what, copy, "device", rc)
            if rc == 0:
                tape, strings = ctx.fetch(tl, sl)
                assert np.array_equal(tape, ref.tape) and np.array_equal(strings, ref.strings), (what, copy, "device")
    # the merged result read in place (sjhip_fetch_view sizes its view from the context: the sharded path sets the totals)
    doc = park * 16
    ref = O.parse(doc, ndjson=True)
    pj = ctx.parse(doc, ndjson=True, view=True)
    assert len(pj.Tape) == len(ref.tape) and len(pj.Strings) == len(ref.strings)
    assert np.array_equal(pj.Tape, ref.tape) and np.array_equal(pj.Strings, ref.strings), "view of a sharded ND result"
    # below the threshold nothing changes, and the context goes back and forth between the two paths
    small = park * 2
    ref = O.parse(small, ndjson=True)
    pj = ctx.parse(small, ndjson=True)
    assert np.array_equal(pj.Tape, ref.tape)
    assert ctx.count_where(b"Make", b"HOND") == 232
    ctx.close()


def test_consumers_on_a_sharded_result():
    """Iter / ForEach / FindElement of the reference work on any ParsedJson (parsed_json.go:96,125,833): the count and path
    queries and MarshalJSON must give, on a result that was parsed shard by shard (the thresholds lowered so that a few
    megabytes take that path: 7-11 shards here), exactly what they give on the same document parsed by one context -- counts
    add up, per-record answers come in document order with tape indexes of the MERGED tape, the texts are joined with the
    newline between two records."""
    import sjhip
    import query_walk as Q
    park = fixtures.load("parking-citations")
    rec = b'{"a":{"b":{"c":%d}},"s":"x\\ny %d","k":[1,{"z":null}],"f":%d.5}\n'
    docs = [("parking x9", park * 9),
            ("nested paths, escapes, numbers", b"".join(rec % (i % 7, i, i) for i in range(60000))),
            ("blank lines at the cuts", b"\n\n" + (park + b"\n \r\n") * 8 + b" ")]
    one = sjhip.Context(0)      # parses the document whole
    for what, doc in docs:
        for copy in (True, False):
            os.environ.pop("SJHIP_ND_LIMIT_BYTES", None)
            os.environ.pop("SJHIP_ND_SHARD_BYTES", None)
            ref = O.parse(doc, ndjson=True, copy_strings=copy)
            assert ref.rc == 0
            pj1 = one.parse(doc, ndjson=True, copy_strings=copy, key_flags=True)
            assert np.array_equal(pj1.Tape, ref.tape)
            os.environ["SJHIP_ND_LIMIT_BYTES"] = str(2 << 20)
            os.environ["SJHIP_ND_SHARD_BYTES"] = str(1 << 20)
            try:
                many = sjhip.Context(0)
                pjm = many.parse(doc, ndjson=True, copy_strings=copy, key_flags=True)
            finally:
                del os.environ["SJHIP_ND_LIMIT_BYTES"], os.environ["SJHIP_ND_SHARD_BYTES"]
            assert np.array_equal(pjm.Tape, ref.tape) and np.array_equal(pjm.Strings, ref.strings), what
            assert many.count_where(b"Make", b"HOND") == one.count_where(b"Make", b"HOND")
            assert many.count_where(b"s", b"x\ny 17") == one.count_where(b"s", b"x\ny 17")
            for path in ((b"Make",), (b"a", b"b", b"c"), (b"a", b"b"), (b"k",), (b"nope",), (b"s", b"t")):
                a, b = many.find_path(*path), one.find_path(*path)
                assert np.array_equal(a, b), (what, path, copy)
                for op, val in ((many.OP_EXISTS, None), (many.OP_EQ_INT, 3), (many.OP_EQ_STRING, b"HOND"), (many.OP_EQ_FLOAT, 3.0)):
                    assert many.count_where_path(path, op, val) == one.count_where_path(path, op, val), (what, path, op, copy)
            keys = [b"Make", b"a", b"f", b"Color"]
            assert np.array_equal(many.project_keys(keys), one.project_keys(keys)), (what, copy)
            # the path answers are indexes of the merged tape: the words they point at are the oracle's
            idx = many.find_path(b"f")
            hit = idx < Q.NOT_OBJECT
            if hit.any():
                assert np.array_equal(pjm.Tape[idx[hit].astype(np.int64)], ref.tape[idx[hit].astype(np.int64)])
            # MarshalJSON: shard texts joined with the newline between two records == the text of the whole result
            rc, want = O.marshal_json(ref.tape, ref.strings, doc[ref.msg_off:ref.msg_off + ref.msg_len])
            assert rc == 0
            assert many.marshal_json() == want, (what, copy, "sharded MarshalJSON")
            assert one.marshal_json() == want
            # what needs one context's result says so instead of answering for a shard
            with pytest.raises(sjhip.ParseError):
                many.filter_where(b"Make", b"HOND")
            with pytest.raises(sjhip.ParseError):
                many.serialize()
            many.close()
    one.close()


def _closed_form(tape0, n_strings0, n_msg0, block, copy=True):
    """tape of block `block` of (document x N) parsed as ND, from the tape of the document itself: every word that
    stores a tape index moves by block * len(tape0), every string word by the block's Strings.B (or Message) offset"""
    tags = (tape0 >> TAG).astype(np.uint8)
    is_str = tags == ord('"')
    raw = np.zeros(len(tape0), dtype=bool)
    raw[1:] = is_str[:-1]  # the length word behind a string word (parking-citations holds no numbers)
    is_str &= ~raw
    is_idx = np.isin(tags, np.frombuffer(b"r{[}]", dtype=np.uint8)) & ~raw
    add = np.zeros(len(tape0), dtype=np.uint64)
    add[is_idx] = np.uint64(block * len(tape0))
    add[is_str] = np.uint64(block * (n_strings0 if copy else n_msg0))
    return tape0 + add


def test_nd_message_beyond_4GiB():
    import psutil
    import sjhip
    if psutil.virtual_memory().available < (64 << 30):
        pytest.skip("needs ~40 GB of host memory")
    park = fixtures.load("parking-citations")
    ref1 = O.parse(park, ndjson=True, copy_strings=True)
    assert ref1.rc == 0 and not np.any((ref1.tape >> TAG) == ord("l")) and not np.any((ref1.tape >> TAG) == ord("d"))
    # the closed form is the oracle's tape (checked where the oracle is quick)
    ref3 = O.parse(park * 3, ndjson=True, copy_strings=True)
    T, S = len(ref1.tape), len(ref1.strings)
    for b in range(3):
        assert np.array_equal(ref3.tape[b * T:(b + 1) * T], _closed_form(ref1.tape, S, len(park), b))
    assert bytes(ref3.strings) == bytes(ref1.strings) * 3
    copies = (4608 << 20) // len(park) + 1   # 4.5 GiB and a bit
    doc = np.frombuffer(park * copies, dtype=np.uint8)
    assert doc.size > (1 << 32) + (256 << 20)
    ctx = sjhip.Context(0)
    pj = ctx.parse(doc, ndjson=True, copy_strings=True)
    assert len(pj.Tape) == copies * T and len(pj.Strings) == copies * S
    tape = pj.Tape.reshape(copies, T)
    for b0 in range(0, copies, 256):
        for b in range(b0, min(copies, b0 + 256)):
            want = _closed_form(ref1.tape, S, len(park), b)
            if not np.array_equal(tape[b], want):
                d = np.nonzero(tape[b] != want)[0]
                raise AssertionError((b, d[:5], [hex(int(x)) for x in tape[b][d[:3]]], [hex(int(x)) for x in want[d[:3]]]))
    strings = pj.Strings.reshape(copies, S)
    assert (strings == ref1.strings[None, :]).all()
    ctx.close()


def _periodic_form(t2, t3):
    """A document  [ B , B , ... , B ]  of N equal blocks has a tape that is periodic with a linear drift: the words of block b are
    the words of block 0 plus b times a per-word step (indexes into the tape advance by the block's words, string offsets by its
    Strings.B bytes or message bytes, everything else stays).  Steps, head and tail are read off the oracle's tapes of the
    documents with two and three blocks and checked on the third block.  -> (words per block, block 0, step, head(N), tail(N))"""
    tb = len(t3) - len(t2)
    head = 2  # r [
    b0, b1, b2 = (t3[head + k * tb: head + (k + 1) * tb] for k in range(3))
    step = b1 - b0
    assert np.array_equal(b2, b0 + np.uint64(2) * step) and np.array_equal(t2[head:head + tb], b0) and np.array_equal(t2[head + tb:head + 2 * tb], b1)

    def ends(n):
        out = []
        for w2, w3 in ((t2[0], t3[0]), (t2[1], t3[1]), (t2[-2], t3[-2]), (t2[-1], t3[-1])):
            d = int(w3) - int(w2)
            assert d in (0, tb)
            out.append(np.uint64(int(w3) + (n - 3) * d))
        return out[:2], out[2:]
    return tb, b0, step, ends


def test_single_document_beyond_4GiB():
    """The reference parses "arbitrarily large" inputs because its index stream is deltas (README.md:567-569,
    flatten_bits_amd64.s:30-44).  Here the positions are 32 bits wide and simply wrap: every 4096-token tile of the token kernels
    rebuilds its true offsets from the unit its first token lies in.  A JSON array of 4.6 GiB (13 000 copies of parking-citations
    as arrays of records) against the closed form of its tape, every block, both copy modes; plain stage 1 -- whose contract is
    32-bit positions -- still refuses such a message."""
    import psutil
    import sjhip
    if psutil.virtual_memory().available < (64 << 30):
        pytest.skip("needs ~45 GB of host memory")
    park = fixtures.load("parking-citations")
    block = b"[" + b",".join(l for l in park.split(b"\n") if l) + b"]"
    mk = lambda n: b"[" + b",".join([block] * n) + b"]"
    copies = (4700 << 20) // (len(block) + 1) + 1
    doc = np.frombuffer(mk(copies), dtype=np.uint8)
    assert doc.size > (1 << 32) + (256 << 20)
    ctx = sjhip.Context(0)
    for copy in (True, False):
        r2, r3 = O.parse(mk(2), copy_strings=copy), O.parse(mk(3), copy_strings=copy)
        assert r2.rc == 0 and r3.rc == 0
        tb, b0, step, ends = _periodic_form(r2.tape, r3.tape)
        sb = len(r3.strings) - len(r2.strings)
        pj = ctx.parse(doc, ndjson=False, copy_strings=copy)
        assert len(pj.Tape) == 4 + copies * tb and len(pj.Strings) == copies * sb, (copy, len(pj.Tape), len(pj.Strings))
        head, tail = ends(copies)
        assert list(pj.Tape[:2]) == head and list(pj.Tape[-2:]) == tail
        tape = pj.Tape[2:-2].reshape(copies, tb)
        for b in range(copies):
            want = b0 + np.uint64(b) * step
            if not np.array_equal(tape[b], want):
                d = np.nonzero(tape[b] != want)[0]
                raise AssertionError((copy, b, d[:5], [hex(int(x)) for x in tape[b][d[:3]]], [hex(int(x)) for x in want[d[:3]]]))
        if sb:
            assert (pj.Strings.reshape(copies, sb) == r2.strings[None, :sb]).all()
        del pj, tape
    with pytest.raises(sjhip.ParseError) as e:
        ctx.stage1(doc)
    assert e.value.code == 4
    assert ctx.parse(b"[1]").Tape.size == 6   # the context is fine
    ctx.close()


def _affine(w2, w3, n):
    """a tape word of the documents with 2 and 3 blocks, extrapolated to n blocks (indexes, string offsets and everything else a
    tape word holds are affine in the number of blocks)"""
    return np.uint64((int(w3) + (n - 3) * (int(w3) - int(w2))) & 0xFFFFFFFFFFFFFFFF)


def test_single_document_beyond_4GiB_second_shape():
    """A second document shape across the 2^32 wrap of the positions: blocks with \\u escapes (incl. a surrogate pair), simple
    escapes, floats that take the slow number path, 64-bit integers, atoms and brackets nested 12 deep, inside an object nested five
    levels, with members BEHIND the 4.7 GB array (their strings lie beyond 4 GiB of the message / Strings.B).  Tape and Strings.B
    against the periodic closed form of the oracle's tapes (verified on a fourth block), both copy modes; then the consumers on the
    resident result: FindElement through the nesting, typed comparisons, and MarshalJSON (refused without a copy of the strings)."""
    import psutil
    import sjhip
    import query_walk as Q
    if psutil.virtual_memory().available < (64 << 30):
        pytest.skip("needs ~45 GB of host memory")
    deep = b"[" * 12 + b'{"k":[1,2,{"z":"\\u20ac"}]}' + b"]" * 12
    items = b",".join(b'{"i":%d,"s":"v\\/%d","f":%d.25e-3}' % (k, k, k) for k in range(400))
    block = (b'{"id":12345678901234567,"txt":"caf\\u00e9 \\ud83d\\ude00 line\\nbreak \\"q\\" \\\\ /","vals":[0.1,1.5e3,-7,1e-7,3.14159,'
             b'18446744073709551615,true,false,null],"deep":' + deep + b',"pad":"' + b"p" * 3000 + b'","items":[' + items + b"]}")
    pre = b'{"r":' * 5 + b'{"d":['
    post = b'],"tail":"caf\\u00e9 \\"end\\"","n":-1.5e3,"t":true}' + b"}" * 5
    mk = lambda n: pre + b",".join([block] * n) + post
    assert O.parse(mk(1)).rc == 0
    copies = (4700 << 20) // (len(block) + 1) + 1
    doc = np.frombuffer(mk(copies), dtype=np.uint8)
    assert doc.size > (1 << 32) + (256 << 20)
    path = [b"r"] * 5
    ctx = sjhip.Context(0)
    for copy in (True, False):
        r2, r3, r4 = (O.parse(mk(k), copy_strings=copy) for k in (2, 3, 4))
        assert r2.rc == 0 and r3.rc == 0 and r4.rc == 0
        tb = len(r3.tape) - len(r2.tape)
        sb = len(r3.strings) - len(r2.strings)
        head = int(np.nonzero((O.parse(pre + b"0" + post).tape >> TAG) == ord("l"))[0][0])   # tape words in front of the first block
        ntail = len(r2.tape) - head - 2 * tb
        b0, b1 = r3.tape[head:head + tb], r3.tape[head + tb:head + 2 * tb]
        step = b1 - b0
        assert np.array_equal(r3.tape[head + 2 * tb:head + 3 * tb], b0 + np.uint64(2) * step)
        form = lambda n: (np.array([_affine(a, b, n) for a, b in zip(r2.tape[:head], r3.tape[:head])], dtype=np.uint64),
                          np.array([_affine(a, b, n) for a, b in zip(r2.tape[-ntail:], r3.tape[-ntail:])], dtype=np.uint64))
        h4, t4 = form(4)   # the closed form is the oracle's tape where the oracle is quick
        assert np.array_equal(r4.tape[:head], h4) and np.array_equal(r4.tape[-ntail:], t4)
        assert np.array_equal(r4.tape[head + 3 * tb:head + 4 * tb], b0 + np.uint64(3) * step)
        pj = ctx.parse(doc, ndjson=False, copy_strings=copy)
        assert len(pj.Tape) == head + ntail + copies * tb, (copy, len(pj.Tape))
        hN, tN = form(copies)
        assert np.array_equal(pj.Tape[:head], hN) and np.array_equal(pj.Tape[-ntail:], tN), copy
        tape = pj.Tape[head:-ntail].reshape(copies, tb)
        for b in range(copies):
            want = b0 + np.uint64(b) * step
            if not np.array_equal(tape[b], want):
                d = np.nonzero(tape[b] != want)[0]
                raise AssertionError((copy, b, d[:5], [hex(int(x)) for x in tape[b][d[:3]]], [hex(int(x)) for x in want[d[:3]]]))
        # Strings.B: the strings in front of the array (the keys of the nesting, where they are copied), the blocks' bytes, the tail's
        s_rest = len(r2.strings) - 2 * sb
        assert len(pj.Strings) == copies * sb + s_rest
        s_head = 0
        if sb:  # the first string of block 0 that lies in Strings.B (bit 55 of its payload) lies at the end of the head's bytes
            is_str = ((b0 >> TAG) == ord('"')) & (((b0 >> np.uint64(55)) & np.uint64(1)) == 1)
            is_str[1:] &= ~(((b0[:-1] >> TAG) == ord('"')))      # (not the length word behind a string word)
            s_head = int(b0[np.nonzero(is_str)[0][0]] & np.uint64((1 << 55) - 1))
            assert (pj.Strings[s_head:s_head + copies * sb].reshape(copies, sb) == r2.strings[None, s_head:s_head + sb]).all()
        assert bytes(pj.Strings[:s_head]) == bytes(r2.strings[:s_head])
        assert bytes(pj.Strings[s_head + copies * sb:]) == bytes(r2.strings[s_head + 2 * sb:])
        # consumers on the resident result: the members behind the array, through five levels of nesting
        (v,) = ctx.find_path(*(path + [b"n"]))
        assert int(v) == len(pj.Tape) - ntail + int(np.nonzero((r2.tape[-ntail:] >> TAG) == ord("d"))[0][0])
        assert ctx.count_where_path(path + [b"n"], ctx.OP_EQ_FLOAT, -1500.0) == 1
        assert ctx.count_where_path(path + [b"tail"], ctx.OP_EQ_STRING, "caf\u00e9 \"end\"".encode()) == 1
        assert ctx.count_where_path(path + [b"t"], ctx.OP_EQ_BOOL, True) == 1
        assert int(ctx.find_path(*(path + [b"nope"]))[0]) == Q.NOT_FOUND
        if copy:
            txt2, txt3 = (O.marshal_json(r.tape, r.strings, b"")[1] for r in (r2, r3))
            unit = len(txt3) - len(txt2)                      # one block and its comma
            n_text = ctx.marshal_json(fetch=False)
            assert n_text == len(txt3) + (copies - 3) * unit
            text = np.frombuffer(ctx.marshal_json(), dtype=np.uint8)
            hl = txt3.index(b'{"id"')                         # text in front of the first block
            assert bytes(text[:hl]) == txt3[:hl] and bytes(text[hl + copies * unit - 1:]) == txt3[hl + 3 * unit - 1:]
            body = text[hl:hl + (copies - 1) * unit].reshape(copies - 1, unit)
            assert (body == np.frombuffer(txt3[hl:hl + unit], dtype=np.uint8)[None, :]).all()
            del text, body
        else:
            with pytest.raises(sjhip.ParseError) as e:   # message offsets beyond 32 bits: refused, not truncated
                ctx.marshal_json(fetch=False)
            assert e.value.code == 4
        del pj, tape
    ctx.close()
