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
