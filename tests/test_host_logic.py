"""CPU-only checks (run with -m "not gpu"): the C-ABI library loads and exports every symbol
declared in include/sjhip.h, and the lane-local device arithmetic (sj_chunk.h), replayed on the
CPU by csrc/host_selftest.cpp, agrees with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import __graft_entry__ as G
import fixtures
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    G.build_lib()
    return G.build_selftest()


def test_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "sjhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sjhip_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 10
    import sjhip
    assert declared == set(sjhip._lib.SYMBOLS), "sjhip/_lib.py must bind exactly the header's symbols"
    L = C.CDLL(os.path.join(ROOT, "simdjson-go_amd", "libsjhip.so"))
    for name in declared:
        assert hasattr(L, name), f"libsjhip.so does not export {name}"
    sjhip.lib()


def test_ctypes_bindings_match_the_prototypes():
    """sjhip/_lib.py: as many argtypes as the header's prototype has parameters, pointers where the header has pointers,
    and a pointer-sized result where the header returns a pointer (a c_int restype would truncate it)."""
    hdr = open(os.path.join(ROOT, "include", "sjhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    import sjhip
    protos = re.findall(r"([A-Za-z_][A-Za-z0-9_ \t]*?[\s\*]+)(sjhip_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr)
    assert len(protos) >= 60
    ptr_types = (C.c_void_p, C.c_char_p)
    for ret, name, params in protos:
        res, args = sjhip._lib.SYMBOLS[name]
        plist = [] if params.strip() in ("", "void") else [q.strip() for q in params.split(",")]
        assert len(args) == len(plist), (name, len(args), plist)
        for a, q in zip(args, plist):
            is_ptr_c = "*" in q or "[" in q  # (an array parameter is a pointer)
            is_ptr_py = a in ptr_types or hasattr(a, "contents") or (isinstance(a, type) and issubclass(a, C._Pointer))
            assert is_ptr_c == is_ptr_py, (name, q, a)
        if "*" in ret:
            assert res in ptr_types, (name, ret, res)
        elif ret.strip().endswith("void"):
            assert res is None, (name, res)


def _selftest():
    L = C.CDLL(G.build_selftest())
    L.sj_selftest_stage1.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t,
                                     C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    return L


def _st1(L, data, nd):
    a = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(a.size + 64, dtype=np.uint32)
    n, e, q = C.c_size_t(), C.c_uint32(), C.c_uint32()
    L.sj_selftest_stage1(a.ctypes.data, a.size, nd, out.ctypes.data, out.size, C.byref(n), C.byref(e), C.byref(q))
    return out[: n.value], e.value, q.value


@pytest.mark.parametrize("name", fixtures.ALL)
def test_chunk_math_matches_oracle_on_fixtures(built, name):
    L = _selftest()
    data = fixtures.load(name).strip()
    for nd in (0, 1):
        ok, pos = O.stage1(data, nd)
        got, err, inq = _st1(L, data, nd)
        assert np.array_equal(got, pos)
        assert err == 0 and inq == 0 and ok


def test_chunk_math_adversarial(built):
    L = _selftest()
    rng = np.random.default_rng(1234)
    alphabet = np.frombuffer(b'\\\\\\\\""""{}[]:,  \n\tabc019.-e\x01\x1f\x80\xff', dtype=np.uint8)
    for trial in range(300):
        n = int(rng.integers(0, 700))
        data = bytes(alphabet[rng.integers(0, alphabet.size, n)])
        for nd in (0, 1):
            _, pos_all = O.stage1(data, nd)
            got, err, inq = _st1(L, data, nd)
            # the oracle stops delivering indexes at its first failing buffer; compare the prefix
            assert np.array_equal(got[: len(pos_all)], pos_all) or len(pos_all) == 0
    # long backslash runs across chunk boundaries
    for k in range(0, 200):
        data = b'["' + b"\\" * k + b'\\"x"]'
        ok, pos = O.stage1(data, 0)
        got, err, inq = _st1(L, data, 0)
        if ok:
            assert np.array_equal(got, pos), k


def test_stream_block_cutter():
    """cut_blocks = the block cutter of ParseNDStream (simdjson_amd64.go:155-176); pure host logic"""
    import io
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "simdjson-go_amd"))
    from sjhip.stream import cut_blocks
    import random
    rnd = random.Random(4)
    for trial in range(50):
        lines = [b"x" * rnd.randrange(0, 300) for _ in range(rnd.randrange(1, 200))]
        data = b"\n".join(lines) + (b"\n" if trial & 1 else b"")
        bs = rnd.choice([1, 7, 64, 1000, 1 << 20])
        blocks = list(cut_blocks(io.BytesIO(data), bs))
        assert b"".join(blocks) == data
        assert all(len(b) >= bs and b.endswith(b"\n") for b in blocks[:-1])
        assert all(len(b) > 0 for b in blocks)

        class Raw(io.RawIOBase):  # a reader without readline and with short reads
            def __init__(self, d):
                self.d, self.p = d, 0

            def readable(self):
                return True

            def readinto(self, b):
                n = min(len(b), 13, len(self.d) - self.p)
                b[:n] = self.d[self.p:self.p + n]
                self.p += n
                return n
        assert b"".join(cut_blocks(Raw(data), bs)) == data


def test_classify_every_byte_value():
    """classify() of sj_chunk.h (bit planes + three-input boolean networks) against the byte definitions of the
    classes, every byte value at every position of a chunk; esc1 = the characters a simple escape may name
    (escape_map, parse_string_amd64.s:38-69)."""
    L = C.CDLL(G.build_selftest())
    L.sj_selftest_classify.argtypes = [C.c_char_p, C.POINTER(C.c_uint64)]
    want = {0: lambda b: b == 0x5c, 1: lambda b: b == 0x22, 2: lambda b: b in b"{}[]:,", 3: lambda b: b in b" \t\n\r",
            4: lambda b: b < 0x20, 5: lambda b: b == 0x0a, 6: lambda b: b in b'"\\/bfnrt'}
    # kind planes: the token kind a byte starts (sj_stage2.h Kind), '\n' left to the NDJSON kernel
    kinds = {ord("{"): 1, ord("["): 2, ord("}"): 3, ord("]"): 4, ord(":"): 5, ord(","): 6, ord('"'): 7, ord("-"): 8,
             ord("t"): 9, ord("f"): 10, ord("n"): 11}
    kinds.update({d: 8 for d in range(0x30, 0x3a)})
    out = (C.c_uint64 * 11)()
    for base in range(0, 256, 64):
        for rot in (0, 1, 17):
            chunk = bytes(((base + (j + rot) % 64) & 0xff) for j in range(64))
            L.sj_selftest_classify(chunk, out)
            for k, f in want.items():
                m = sum(1 << j for j in range(64) if f(chunk[j]))
                assert out[k] == m, (k, base, rot, hex(out[k]), hex(m))
            for j in range(64):
                kind = sum(((out[7 + b] >> j) & 1) << b for b in range(4))
                assert kind == kinds.get(chunk[j], 0), (chunk[j], kind)


def test_batch_newline_translation():
    """sj_chunk.h newlines_to_cr (the packing kernel of sjhip_parse_batch): four bytes at a time against the byte loop."""
    L = C.CDLL(G.build_selftest())
    L.sj_selftest_newlines_to_cr.restype = C.c_int
    assert L.sj_selftest_newlines_to_cr() == 0
