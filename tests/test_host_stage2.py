"""CPU-only checks of the data-parallel stage 2 (sj_stage2.h / sj_number.h / sj_bignum.h), replayed
lane by lane on the host by csrc/host_selftest.cpp, against the oracle."""
import ctypes as C
import random
import struct

import numpy as np
import pytest

import __graft_entry__ as G
import fixtures
import golden_util as GU
import oracle_lib as O

u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)


@pytest.fixture(scope="module")
def L():
    lib = C.CDLL(G.build_selftest())
    lib.sj_selftest_parse.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(u64p), C.POINTER(C.c_size_t),
                                      C.POINTER(u8p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                      C.POINTER(C.c_size_t)]
    lib.sj_selftest_free.argtypes = [C.c_void_p]
    lib.sj_selftest_parse_number.argtypes = [C.c_char_p, C.c_size_t, u64p, u64p, C.POINTER(C.c_int)]
    lib.sj_selftest_trim.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    return lib


def replay(L, data, nd, copy):
    a = np.frombuffer(data, dtype=np.uint8)
    tape, strs = u64p(), u8p()
    tl, sl, mo, ml = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
    rc = L.sj_selftest_parse(a.ctypes.data, a.size, (1 if nd else 0) | (2 if copy else 0), C.byref(tape), C.byref(tl),
                             C.byref(strs), C.byref(sl), C.byref(mo), C.byref(ml))
    if rc:
        return rc, None, None
    t = np.ctypeslib.as_array(tape, shape=(tl.value,)).copy()
    s = np.ctypeslib.as_array(strs, shape=(sl.value,)).copy() if sl.value else np.zeros(0, np.uint8)
    L.sj_selftest_free(tape)
    L.sj_selftest_free(strs)
    return 0, t, s


def check(L, data, nd=False, what=""):
    for copy in (True, False):
        ref = O.parse(data, ndjson=nd, copy_strings=copy)
        rc, t, s = replay(L, data, nd, copy)
        assert rc == ref.rc, (what, nd, copy, rc, ref.rc, data[:80])
        if rc == 0:
            assert np.array_equal(t, ref.tape) and np.array_equal(s, ref.strings), (what, nd, copy)


@pytest.mark.parametrize("name", fixtures.ALL)
def test_fixtures(L, name):
    check(L, fixtures.load(name), name == "parking-citations", name)


def test_reference_tables(L):
    corp = GU.load("corpus")
    for k in ("fail_cases", "pass_cases"):
        for c in corp[k]:
            check(L, bytes.fromhex(c["js_hex"]), False, c["name"])
    for c in corp["parse_nd"]:
        check(L, bytes.fromhex(c["js_hex"]), True, c["name"])
    s2 = GU.load("stage2")
    for t in s2["tapes_nocopy"]:
        check(L, bytes.fromhex(t["input_hex"]), False, "tape")
    check(L, bytes.fromhex(s2["demo_ndjson_hex"]), True, "demo_nd")
    for h in s2["ndjson_empty_lines_hex"]:
        check(L, bytes.fromhex(h), True, "emptylines")
    for r in GU.load("strings"):
        body = bytes.fromhex(r["str_hex"])
        check(L, b'["' + body + b'"]', False, r["name"])


def test_random_documents(L):
    rng = np.random.default_rng(7)
    alpha = np.frombuffer(b'{}[]:,"""  \n\\tfn0123-.e"a', dtype=np.uint8)
    for trial in range(4000):
        body = bytes(alpha[rng.integers(0, alpha.size, int(rng.integers(1, 40)))])
        check(L, body, bool(trial & 1), "soup")
    rnd = random.Random(5)

    def gen(depth=0):
        r = rnd.random()
        if depth > 6 or r < 0.3:
            return rnd.choice(['1', '-2.5e3', 'true', 'false', 'null', '"s"', '"a\\nb"', '"\\u00e9"', '[]', '{}',
                               '12345678901234567890', '0.1', '"\\ud83d\\ude00"'])
        if r < 0.65:
            return '[' + ','.join(gen(depth + 1) for _ in range(rnd.randint(0, 5))) + ']'
        return '{' + ','.join('"k%d":%s' % (i, gen(depth + 1)) for i in range(rnd.randint(0, 5))) + '}'

    for trial in range(600):
        doc = gen()
        if doc[0] not in '[{':
            doc = '[' + doc + ']'
        check(L, doc.encode(), False, 'gen')
        b = bytearray(doc.encode())
        if len(b) > 2:
            b[rnd.randrange(len(b))] = rnd.choice(b'{}[]:,"\\ 1tx')
            check(L, bytes(b), False, 'mut')
        lines = '\n'.join('{"a":%s}' % gen() for _ in range(rnd.randint(1, 5)))
        check(L, lines.encode(), True, 'gennd')


def test_number_parsing_against_strtod(L):
    OL = O.lib()

    def mine(s):
        t, v, b = C.c_uint64(), C.c_uint64(), C.c_int()
        st = L.sj_selftest_parse_number(s, len(s), C.byref(t), C.byref(v), C.byref(b))
        return ((t.value, v.value) if st else (0, 0)), b.value

    def ref(s):
        v = C.c_uint64()
        t = OL.sjo_parse_number(s, len(s), C.byref(v))
        return (t, v.value) if t else (0, 0)

    nbig = 0
    g = GU.load("numbers")
    cases = [r["input"] for r in g["parse_number"] + g["parse_int64"] + g["atof"]] + g["valid"] + g["invalid"]
    rnd = random.Random(42)
    for i in range(30000):
        d = struct.unpack("<d", struct.pack("<Q", rnd.getrandbits(64)))[0]
        if d == d and abs(d) != float("inf"):
            cases += [repr(d), "%.17e" % d, "%.25e" % d]
        cases.append("%de%d" % (rnd.getrandbits(rnd.choice([10, 30, 53, 60, 64])), rnd.randint(-345, 310)))
    import decimal
    decimal.getcontext().prec = 1200
    for i in range(3000):
        bits = (rnd.choice([rnd.randint(1, 2045), rnd.randint(0, 3)]) << 52) | rnd.getrandbits(52)
        d = struct.unpack("<d", struct.pack("<Q", bits))[0]
        nxt = struct.unpack("<d", struct.pack("<Q", bits + 1))[0]
        mid = (decimal.Decimal(d) + decimal.Decimal(nxt)) / 2
        s = format(mid, "e")
        m, e = s.split("e")
        cases += [s, m + "1e" + e, m + "0000000000000000000001e" + e]
    for s in cases:
        b = s.encode() + b","
        got, used = mine(b)
        nbig += used
        assert got == ref(b), s[:60]
    assert nbig > 1000  # the big-integer tie-break path was exercised
    # the integer fast path (parse_int_fast) and its hand-over to the general routine: every digit count around the
    # 9 / 18 / 19 / 20 digit limits, signs, leading zeros, every kind of byte behind the number, the 32-byte window
    ends = [b",", b"}", b"]", b" ", b"\t", b"\r", b"\n", b":", b".5,", b"e3,", b"E+2,", b"+,", b"-,", b"a,", b"\x00,", b"", b"0,",
            b".,", b"e,"]
    for nd in range(0, 23):
        for trial in range(40):
            digits = "".join(rnd.choice("0123456789") for _ in range(nd))
            if trial % 4 == 0 and nd:
                digits = rnd.choice("123456789") + digits[1:]
            if trial % 8 == 1 and nd:
                digits = "9" * nd
            for sign in (b"", b"-", b"+"):
                for e in ends:
                    b = sign + digits.encode() + e
                    got, _ = mine(b)
                    assert got == ref(b), b
                    padded = b + b" " * 40
                    got, _ = mine(padded)
                    assert got == ref(padded), padded
    for v in (2**63 - 1, 2**63, 2**63 + 1, 2**64 - 1, 2**64, 10**18 - 1, 10**18, 10**9 - 1, 10**9, 10**17, 999999999999999999):
        for sgn in ("", "-"):
            b = (sgn + str(v)).encode() + b","
            got, _ = mine(b)
            assert got == ref(b), b


def test_numbers_cut_by_the_32_byte_window(L):
    import workloads
    nums = workloads.window_cut_numbers()
    for n in nums[-7:]:
        check(L, ("[" + n + "]").encode(), False, "cut-bad")
    good = nums[:-7]
    for i in range(0, len(good), 200):
        check(L, ("[" + ",".join(good[i:i + 200]) + "]").encode(), False, "cut")


def test_trim_space(L):
    OL = O.lib()
    samples = [b"", b"  x ", b"\x0b\x0cx\x0b", "  x　".encode(), b"\xc2\x85x\xc2\xa0", b"\xe2\x80\xa8{}\xe2\x80\xa9",
               b"\xff x \xff", b" \xc2", b"\xe3\x80", b"x\xe2\x80", b"\xe1\x9a\x80\xe1\x9a\x80", b" \n\t\r "]
    for s in samples:
        a, b, c, d = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        L.sj_selftest_trim(s, len(s), C.byref(a), C.byref(b))
        OL.sjo_trim_space(s, len(s), C.byref(c), C.byref(d))
        assert (b.value == d.value) and (b.value == 0 or a.value == c.value), s


def surrogate_run_docs():
    """Runs of adjacent high-surrogate escapes: the byte-parallel string path pairs them by walking back to the start
    of the run; beyond SURROGATE_WALK_CAP (4096) escapes it hands the document to the per-string walks."""
    hi, lo = b"\\ud800", b"\\udc00"
    for n in (1, 2, 3, 7, 64, 4095, 4096, 4097, 4098, 9001):
        yield b'["' + hi * n + b'"]', f"highs{n}"                   # odd n: the last one is a lone high surrogate -> fail
        yield b'["' + hi * n + lo + b'"]', f"highs{n}+low"
        yield b'{"k":"x' + hi * n + b'","after":"\\u00e9' + hi * 2 + b'"}', f"highs{n}-then-more"
    yield b'["' + (b"\\ud800" * 5000 + b"x") * 3 + b'"]', "three-long-runs"


def test_long_surrogate_runs(L):
    for doc, what in surrogate_run_docs():
        check(L, doc, False, what)


def test_integer_fast_path_decides_like_the_general_routine(L):
    """sj_number.h parse_int_fast (what k_s2_emit runs on number tokens before it queues them for k_numbers): whenever it
    takes a number, the oracle's parseNumber gives an int64 of that value; it never takes floats, integers of more than
    18 digits, leading zeros or numbers with a stray byte behind them; and it does take every plain integer of up to 18
    digits that is followed by a byte that ends a value."""
    OL = O.lib()
    L.sj_selftest_int_fast.argtypes = [C.c_char_p, C.c_size_t, u64p]
    rnd = random.Random(7)
    ends = [b",", b"}", b"]", b" ", b"\t", b"\r", b"\n", b":"]
    others = [b".", b"e", b"E", b"-", b"+", b"a", b"\x00", b'"', b"{", b"[", b"/", b"9" * 3 + b"."]
    cases = []
    for nd in range(1, 22):
        for _ in range(40):
            digits = str(rnd.randrange(10 ** (nd - 1), 10 ** nd)) if nd > 1 else str(rnd.randrange(10))
            for sign in ("", "-"):
                for tail in (rnd.choice(ends), rnd.choice(ends) + b"123", rnd.choice(others) + b"5]", b""):
                    cases.append((sign + digits).encode() + tail)
        cases += [b"0" * nd + b",", b"-" + b"0" * nd + b"]", b"0" + b"7" * nd + b" ", b"9" * nd + b",", b"-" + b"9" * nd + b"}"]
    cases += [b"-,", b"-", b"", b",", b"--1,", b"1-2,", b"9223372036854775807,", b"-9223372036854775808,", b"999999999999999999,",
              b"1000000000000000000,", b"-999999999999999999:", b"12345678,", b"123456789,", b"1234567890123456,", b"12345678901234567,"]
    took = 0
    for s in cases:
        v = C.c_uint64()
        fast = L.sj_selftest_int_fast(s, len(s), C.byref(v))
        rv = C.c_uint64()
        tag = OL.sjo_parse_number(s, len(s), C.byref(rv))
        body = s.lstrip(b"-")
        ndig = len(body) - len(body.lstrip(b"0123456789"))
        plain = 1 <= ndig <= 18 and len(s) - len(body) <= 1 and not (ndig > 1 and body[:1] == b"0") and body[ndig:ndig + 1] in ends
        assert bool(fast) == plain, (s, fast, plain)
        if fast:
            took += 1
            assert tag == (ord("l") << 56) and rv.value == v.value, (s, hex(tag), rv.value, v.value)
            assert v.value == int(s[:len(s) - len(body) + ndig]) % (1 << 64), s
    assert took > 1000


def test_strings_on_chunk_and_unit_boundaries(L):
    """tests/workloads.py string_boundary_documents: the host replay (which also runs the emit-mask form of the selective
    copy beside the per-string walks, host_selftest.cpp) against the oracle in both copy modes."""
    import workloads
    docs = workloads.string_boundary_documents()
    assert len(docs) > 1000
    for what, d in docs[::3]:
        check(L, d, False, what)
    nd = b"\n".join(d for _, d in docs[1::40])
    check(L, nd, True, "boundaries/nd")
