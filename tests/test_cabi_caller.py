"""The C ABI driven by a non-Python caller: tests/cabi_caller.c performs the Go binding's exact call sequences
(simdjson-go_amd/go/simdjson_hip.go: parseMessageHip with a recycled ParsedJson, parseMessageMulti, ParseBatch, the
reader / deliverer protocol of ParseNDStream) through include/sjhip.h and dumps what it fetched; the results must be
the oracle's, bit for bit.  CPU half: the program compiles as C11 against the header and links every symbol it uses."""
import os
import subprocess

import numpy as np
import pytest

import fixtures

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "simdjson-go_amd")
BUILD = os.path.join(HERE, "_build")
EXE = os.path.join(BUILD, "cabi_caller")


def build_caller():
    import __graft_entry__ as G
    G.build_lib()
    os.makedirs(BUILD, exist_ok=True)
    src = os.path.join(HERE, "cabi_caller.c")
    deps = [src, os.path.join(ROOT, "include", "sjhip.h"), os.path.join(PKG, "libsjhip.so")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                               src, "-o", EXE, "-L", PKG, "-lsjhip", f"-Wl,-rpath,{PKG}"])
    return EXE


def test_caller_compiles_as_c_and_links():
    exe = build_caller()
    out = subprocess.run([exe, "symbols"], capture_output=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == b"0", (out.returncode, out.stdout, out.stderr)


def _words(path):
    return np.fromfile(path, dtype=np.uint64)


def _take_bytes(w, at, n):
    nw = (n + 7) // 8
    return w[at:at + nw].view(np.uint8)[:n], at + nw


def _read_result(path):
    w = _words(path)
    rc, tl, sl, off, ml = (int(x) for x in w[:5])
    rc = rc - (1 << 64) if rc >> 63 else rc
    tape = w[5:5 + tl]
    strings, at = _take_bytes(w, 5 + tl, sl)
    return rc, tape, strings, off, ml, w[at:]


def _run(args, tmp_path):
    out = subprocess.run([build_caller()] + [str(a) for a in args], capture_output=True, timeout=600)
    assert out.returncode == 0, (args, out.returncode, out.stderr[-2000:])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["twitter", "canada", "twitterescaped", "payload-small"])
@pytest.mark.parametrize("flags", [2, 0])
def test_parse_with_recycled_buffers(name, flags, tmp_path):
    import oracle_lib as O
    data = b" \n" + fixtures.load(name) + b"\t "
    src, dst = tmp_path / "in.json", tmp_path / "out.bin"
    src.write_bytes(data)
    for repeat in (1, 4):
        _run(["parse", src, dst, flags, repeat], tmp_path)
        rc, tape, strings, off, ml, _ = _read_result(dst)
        ref = O.parse(data, copy_strings=bool(flags & 2))
        assert rc == ref.rc == 0
        assert (off, ml) == (ref.msg_off, ref.msg_len)
        assert np.array_equal(tape, ref.tape) and np.array_equal(strings, ref.strings), (name, flags, repeat)


@pytest.mark.gpu
def test_parse_error_codes(tmp_path):
    import oracle_lib as O
    for doc in (b'{"a":[1,2}', b'{"a":"unterminated', b"", b"   ", b'{"a":1e400}'):
        src, dst = tmp_path / "in.json", tmp_path / "out.bin"
        src.write_bytes(doc)
        _run(["parse", src, dst, 2, 1], tmp_path)
        rc = _read_result(dst)[0]
        assert rc == O.parse(doc).rc != 0, doc


@pytest.mark.gpu
@pytest.mark.parametrize("shards", [0, 3])
def test_multi(shards, tmp_path):
    import oracle_lib as O
    park = fixtures.load("parking-citations")
    for doc, want in ((park * 12, 0), (park * 3 + b'{"broken":"unterminated\n' + park * 3, 1), (b'{"a":[1,2}\n' + park * 2, 2)):
        src, dst = tmp_path / "in.json", tmp_path / "out.bin"
        src.write_bytes(doc)
        for flags in (2, 0):
            _run(["multi", src, dst, flags, shards], tmp_path)
            rc, tape, strings, off, ml, rest = _read_result(dst)
            ref = O.parse(doc, ndjson=True, copy_strings=bool(flags & 2))
            assert rc == ref.rc == want
            if rc == 0:
                assert (off, ml) == (ref.msg_off, ref.msg_len)
                assert np.array_equal(tape, ref.tape) and np.array_equal(strings, ref.strings)
                assert int(rest[0]) >= 1


@pytest.mark.gpu
def test_batch(tmp_path):
    import oracle_lib as O
    docs = [fixtures.load("payload-small"), b'  {"a":"b\\n"}\n', fixtures.load("github_events"), b"[1,\n2]"]
    paths = []
    for i, d in enumerate(docs):
        p = tmp_path / f"d{i}.json"
        p.write_bytes(d)
        paths.append(p)
    dst = tmp_path / "out.bin"
    _run(["batch", dst] + paths, tmp_path)
    rc, tape, strings, _, _, _ = _read_result(dst)
    assert rc == 0
    packed = b"\n".join(d.strip(b" \t\r\n").replace(b"\n", b"\r") for d in docs)
    ref = O.parse(packed, ndjson=True, copy_strings=True)
    assert ref.rc == 0 and np.array_equal(tape, ref.tape) and np.array_equal(strings, ref.strings)


def _cut_blocks(data, block):
    """the reader of ParseNDStream (simdjson_amd64.go:155-176): block bytes, then up to the end of the record"""
    at, out = 0, []
    while at < len(data):
        n = min(block, len(data) - at)
        end = at + n
        if n == block:
            nl = data.find(b"\n", end)
            end = len(data) if nl < 0 else nl + 1
        out.append(data[at:end])
        at = end
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("slots", [1, 3])
def test_stream_protocol(slots, tmp_path):
    import oracle_lib as O
    park = fixtures.load("parking-citations")
    long_record = b'{"k":"' + b"x" * 300000 + b'"}\n'  # runs past the reserve of a block: sjhip_stream_grow
    good = park * 8 + long_record + park * 3
    bad = park * 4 + b'{"a":[1,2}\n' + park * 4
    for data, block in ((good, 1 << 20), (bad, 1 << 20), (b"", 1 << 20), (park[:-1], 1 << 16)):
        src, dst = tmp_path / "in.json", tmp_path / "out.bin"
        src.write_bytes(data)
        _run(["stream", src, dst, block, slots], tmp_path)
        w = _words(dst)
        at = 0
        for blk in _cut_blocks(data, block):
            ref = O.parse(blk, ndjson=True, copy_strings=True)
            rc = int(w[at])
            assert rc == ref.rc, (len(data), at)
            if rc != 0:  # the first error ends the stream
                at += 1
                break
            tl, sl, ml = (int(x) for x in w[at + 1:at + 4])
            at += 4
            assert np.array_equal(w[at:at + tl], ref.tape)
            at += tl
            s, at = _take_bytes(w, at, sl)
            assert np.array_equal(s, ref.strings)
            m, at = _take_bytes(w, at, ml)
            assert bytes(m) == blk[ref.msg_off:ref.msg_off + ref.msg_len]
        else:
            assert int(w[at]) == 7  # SJHIP_STREAM_EMPTY: the clean end (io.EOF)
            at += 1
        assert at == len(w)
