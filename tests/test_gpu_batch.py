"""GPU: many documents in one launch set (sjhip_parse_batch / sjhip_parse_batch_device, csrc/batch_api.hip).
The batch is defined as ParseND of the packed message (documents trimmed, '\\n' inside a document -> '\\r', '\\n'
between documents): the result must be bit for bit the oracle's ParseND of that message, document i must be root i, and
every document's piece of the tape must be the tape Parse() of the document alone produces, up to the rebased indices.
One invalid document fails the batch with the code Parse() of that document returns (stage 1 first): ParseND of the
packed message reports it, except for the end-of-message rule of stage 1 (the last structural must close a container,
stage1_find_marks_amd64.go:115-129), which inside a packed message only the last document would meet -- so every
document is held to it while the batch is packed (a scalar such as `1`, a truncated or an all-whitespace document fail
with the stage-1 code wherever they stand)."""
import numpy as np
import pytest

import fixtures
import oracle_lib as O

pytestmark = pytest.mark.gpu

WS = b" \t\n\r"


def _packed(docs):
    return b"\n".join(bytes(d).strip(WS).replace(b"\n", b"\r") for d in docs)


def _roots(tape):
    """(start, end) tape index pairs of the roots: 'r' word at start points behind the closing 'r' word."""
    out, i = [], 0
    while i < len(tape):
        w = int(tape[i])
        assert (w >> 56) == ord("r"), (i, hex(w))
        nxt = w & ((1 << 56) - 1)
        out.append((i, nxt))
        i = nxt
    return out


def _docs_ok():
    names = ["twitter", "canada", "twitterescaped", "apache_builds", "mesh.pretty", "numbers", "update-center", "random"]
    docs = [fixtures.load(n) for n in names]
    docs += [b'  {"a":[1,2.5,"x\\n\\u00e9"]}\n\n', b"[]", b"\r\n{\n \"k\" :\n null\n}\n", b'[{"deep":[[[]]]}]']
    return docs


def test_batch_equals_parse_nd_of_packed_message():
    import sjhip
    ctx = sjhip.Context(0)
    docs = _docs_ok()
    ref = O.parse(_packed(docs), ndjson=True, copy_strings=True)
    assert ref.rc == 0
    pj = ctx.parse_batch(docs)
    assert np.array_equal(pj.Tape, ref.tape)
    assert np.array_equal(pj.Strings, ref.strings)
    roots = _roots(pj.Tape)
    assert len(roots) == len(docs)
    # document i alone: the same words up to the rebased indices
    sbase = 0
    for (a, b), d in zip(roots, docs):
        one = O.parse(bytes(d), ndjson=False, copy_strings=True)
        assert one.rc == 0 and b - a == len(one.tape)
        piece = pj.Tape[a:b].copy()
        tag = (piece >> np.uint64(56)).astype(np.uint8)
        raw = np.zeros(len(piece), dtype=bool)  # second words of strings / numbers
        i = 0
        tl = one.tape
        while i < len(tl):
            t = int(tl[i]) >> 56
            if t in (ord('"'), ord("l"), ord("u"), ord("d")):
                raw[i + 1] = True
                i += 2
            else:
                i += 1
        idx = np.isin(tag, np.frombuffer(b"r{}[]", dtype=np.uint8)) & ~raw
        st = (tag == ord('"')) & ~raw
        piece[idx] -= np.uint64(a)
        piece[st] -= np.uint64(sbase)
        assert np.array_equal(piece, one.tape)
        assert bytes(pj.Strings[sbase:sbase + len(one.strings)]) == bytes(one.strings)
        sbase += len(one.strings)
    assert sbase == len(pj.Strings)


def test_batch_device_resident():
    import torch
    import sjhip
    ctx = sjhip.Context(0)
    docs = [bytes(d).strip(WS) for d in _docs_ok()] * 3
    blob, offs, lens = bytearray(), [], []
    for k, d in enumerate(docs):
        blob += b"\0" * (k % 5)          # arbitrary (also unaligned) placement
        offs.append(len(blob))
        lens.append(len(d))
        blob += d
    dev = torch.empty(len(blob) + 64, dtype=torch.uint8, device="cuda:0")
    dev[: len(blob)].copy_(torch.frombuffer(blob, dtype=torch.uint8))
    torch.cuda.synchronize()
    tl, sl = ctx.parse_batch_device(dev.data_ptr(), offs, lens)
    tape, strings = ctx.fetch(tl, sl)
    ref = O.parse(_packed(docs), ndjson=True, copy_strings=True)
    assert ref.rc == 0
    assert np.array_equal(tape, ref.tape) and np.array_equal(strings, ref.strings)
    assert ctx.count_where(b"no such key", b"x") == 0   # the queries work on the batch result


@pytest.mark.parametrize("bad,code", [
    (b'{"a":"unterminated', 1),
    (b'{"a":[1,2}', 2),
    (b"[1]\n[2]", 2),                   # two roots in ONE document stay an error
    (b'{"a":1}\n{"b":2}', 2),
    (b'{"a":"line\nbreak"}', 1),        # a raw newline inside a string
    (b"  \n ", 1),                      # empty after trimming
    (b"", 1),
])
def test_batch_one_bad_document_fails_the_batch(bad, code):
    import sjhip
    ctx = sjhip.Context(0)
    assert O.parse(bad, ndjson=False, copy_strings=True).rc == code
    good = [b'{"x":1}', fixtures.load("twitter"), b"[1,2,3]"]
    for pos in (0, 1, 3):
        docs = good[:pos] + [bad] + good[pos:]
        if bad.strip(WS):
            assert O.parse(_packed(docs), ndjson=True, copy_strings=True).rc == code
        with pytest.raises(sjhip.ParseError) as e:
            ctx.parse_batch(docs)
        assert e.value.code == code, (pos, e.value.code)
    pj = ctx.parse_batch(good)          # the context is fine afterwards
    assert len(_roots(pj.Tape)) == 3


def test_batch_stage1_wins_over_stage2():
    import sjhip
    ctx = sjhip.Context(0)
    with pytest.raises(sjhip.ParseError) as e:
        ctx.parse_batch([b'{"a":[1,2}', b"[1]", b'{"a":"unterminated'])
    assert e.value.code == 1


def test_batch_of_256_twitter():
    import sjhip
    ctx = sjhip.Context(0)
    d = fixtures.load("twitter")
    one = O.parse(d, ndjson=False, copy_strings=True)
    tl, sl = ctx.parse_batch([d] * 256, fetch=False)
    assert tl == 256 * len(one.tape) and sl == 256 * len(one.strings)
    tape, strings = ctx.fetch(tl, sl)
    assert bytes(strings) == bytes(one.strings) * 256
    n = len(one.tape)
    tags = (tape >> np.uint64(56)).reshape(256, n)
    assert (tags == (one.tape >> np.uint64(56))[None, :]).all()


def test_batch_of_many_tiny_documents():
    """More documents than one grid dimension holds (65 535): the packing kernels walk (y, z) block indices."""
    import sjhip
    ctx = sjhip.Context(0)
    n = 70001
    docs = [b'{"i":%d,"s":"v%d"}' % (k, k % 97) for k in range(n)]
    ref = O.parse(_packed(docs), ndjson=True, copy_strings=True)
    assert ref.rc == 0
    pj = ctx.parse_batch(docs)
    assert np.array_equal(pj.Tape, ref.tape) and np.array_equal(pj.Strings, ref.strings)
    # the same from one device buffer
    import torch
    blob = b"".join(docs)
    offs = np.concatenate(([0], np.cumsum([len(d) for d in docs])[:-1]))
    dev = torch.empty(len(blob) + 64, dtype=torch.uint8, device="cuda:0")
    dev[: len(blob)].copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
    torch.cuda.synchronize()
    tl, sl = ctx.parse_batch_device(dev.data_ptr(), offs.tolist(), [len(d) for d in docs])
    tape, strings = ctx.fetch(tl, sl)
    assert np.array_equal(tape, ref.tape) and np.array_equal(strings, ref.strings)


@pytest.mark.parametrize("bad", [b"1", b'"str"', b'{"a":1', b'{"a":1} x', b" ", b"\n", b" \t\r\n ", b"[1,2", b"true"])
def test_batch_documents_meet_parse_end_rule(bad):
    """A document that Parse() rejects in stage 1 because its last structural does not close a container fails the batch
    with the stage-1 code at every position, from host buffers and from one device buffer (where documents are taken
    untrimmed: an all-whitespace document would otherwise vanish as an empty ND line and shift every later root)."""
    import sjhip
    import torch
    ctx = sjhip.Context(0)
    assert O.parse(bad, ndjson=False, copy_strings=True).rc == 1
    good = [b'{"x":1}', b' [1,2,3]\n', fixtures.load("payload-small")]
    for pos in (0, 1, 3):
        docs = good[:pos] + [bad] + good[pos:]
        with pytest.raises(sjhip.ParseError) as e:
            ctx.parse_batch(docs)
        assert e.value.code == 1, (pos, e.value.code)
        blob = b"".join(docs)
        offs = np.concatenate(([0], np.cumsum([len(d) for d in docs])[:-1]))
        dev = torch.empty(len(blob) + 64, dtype=torch.uint8, device="cuda:0")
        dev[: len(blob)].copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
        torch.cuda.synchronize()
        with pytest.raises(sjhip.ParseError) as e:
            ctx.parse_batch_device(dev.data_ptr(), offs.tolist(), [len(d) for d in docs])
        assert e.value.code == 1, (pos, e.value.code)
    # the context is fine afterwards, and good documents with whitespace around them keep root i = document i
    blob = b"".join(good)
    offs = np.concatenate(([0], np.cumsum([len(d) for d in good])[:-1]))
    dev = torch.empty(len(blob) + 64, dtype=torch.uint8, device="cuda:0")
    dev[: len(blob)].copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
    torch.cuda.synchronize()
    tl, sl = ctx.parse_batch_device(dev.data_ptr(), offs.tolist(), [len(d) for d in good])
    tape, strings = ctx.fetch(tl, sl)
    ref = O.parse(b"\n".join(d.replace(b"\n", b"\r") for d in good), ndjson=True, copy_strings=True)
    assert ref.rc == 0 and np.array_equal(tape, ref.tape) and np.array_equal(strings, ref.strings)
    assert len(_roots(tape)) == 3


def test_batch_one_huge_document_among_many_small():
    """Sizes as skewed as they get: the packing grid is sized by the packed bytes, not by longest x count."""
    import sjhip
    ctx = sjhip.Context(0)
    big = b"[" + b",".join([fixtures.load("twitter")] * 12) + b"]"
    docs = [b'{"i":%d}' % k for k in range(20000)]
    docs.insert(7777, big)
    ref = O.parse(_packed(docs), ndjson=True, copy_strings=True)
    pj = ctx.parse_batch(docs)
    assert ref.rc == 0 and np.array_equal(pj.Tape, ref.tape) and np.array_equal(pj.Strings, ref.strings)
