"""CPU: the number formatting of the tape -> JSON text path (csrc/sj_ftoa.h, a restatement of the reference's
appendFloat + its copy of Go's Ryu, ftoaryu.go) against an independent source of shortest round-trip digits --
Python's repr (David Gay's algorithm) -- laid out by Go's rules (parsed_json.go:1250-1272, appendfloat_f.go, strconv %e),
and against the floats in the reference's expected MarshalJSON texts."""
import ctypes as C
import decimal
import math
import random
import struct

import pytest

import __graft_entry__ as G


@pytest.fixture(scope="module")
def L():
    lib = C.CDLL(G.build_selftest())
    lib.sj_selftest_format_float.argtypes = [C.c_uint64, C.c_char_p]
    lib.sj_selftest_format_float.restype = C.c_uint
    lib.sj_selftest_format_int.argtypes = [C.c_uint64, C.c_int, C.c_char_p]
    lib.sj_selftest_format_int.restype = C.c_uint
    return lib


def go_format(x):
    """appendFloat(x) from Python's shortest digits."""
    if x == 0:
        return "-0" if math.copysign(1, x) < 0 else "0"
    sign, digits, exp = decimal.Decimal(repr(x)).as_tuple()
    digits = list(digits)
    while len(digits) > 1 and digits[-1] == 0:
        digits.pop()
        exp += 1
    nd, dp = len(digits), len(digits) + exp
    ds = "".join(map(str, digits))
    out = "-" if sign else ""
    a = abs(x)
    if 1e-6 <= a < 1e21:
        if dp > 0:
            out += ds[:min(nd, dp)] + "0" * max(0, dp - nd)
        else:
            out += "0"
        prec = max(nd - dp, 0)
        if prec:
            out += "." + "".join(ds[dp + i] if 0 <= dp + i < nd else "0" for i in range(prec))
        return out
    out += ds[0] + ("." + ds[1:] if nd > 1 else "")
    e = dp - 1
    es = "%s%02d" % ("-" if e < 0 else "+", abs(e))
    if es[0] == "-" and es[1] == "0":
        es = "-" + es[2:]
    return out + "e" + es


def fmt(L, bits):
    buf = C.create_string_buffer(40)
    n = L.sj_selftest_format_float(bits, buf)
    assert n < 0x80000000, hex(bits)   # (bytes behind the text were written, or the length-only form disagrees)
    return buf.raw[:n].decode()


def test_against_python_shortest_digits(L):
    rnd = random.Random(2026)
    cases = [0, 1 << 63, 1, 2, (1 << 52) - 1, 1 << 52, (1 << 52) + 1, 0x7fefffffffffffff, 0x0010000000000000, 0x000fffffffffffff]
    for v in (1.0, 0.1, 0.5, 1e-6, 9.999999999999999e-7, 1e21, 9.999999999999999e20, 1e22, 1e23, 5e-324, 1.7976931348623157e308,
              123456789.0, 1e15, 1e16, 1e17, 123456789012345680.0, 0.3, 2.5e-8, 1e-7, 1e-10, 4.35, 0.000001, 100.0, 1e20,
              9007199254740993.0, 2.2250738585072014e-308, 2.225073858507201e-308, 8.41e21, 6.02214076e23, 299792458.0):
        cases += [struct.unpack("<Q", struct.pack("<d", v))[0], struct.unpack("<Q", struct.pack("<d", -v))[0]]
    for e in range(0, 2047):                                   # every binade: its first, a middle and its last value
        cases += [e << 52, (e << 52) | rnd.getrandbits(52), (e << 52) | ((1 << 52) - 1)]
    for _ in range(200000):
        cases.append(rnd.getrandbits(64))
    for _ in range(50000):                                     # short decimals (what documents contain)
        v = round(rnd.uniform(-1e6, 1e6), rnd.randrange(0, 8)) * 10.0 ** rnd.randrange(-30, 30)
        cases.append(struct.unpack("<Q", struct.pack("<d", v))[0])
    n = 0
    for bits in cases:
        x = struct.unpack("<d", struct.pack("<Q", bits))[0]
        if math.isinf(x) or math.isnan(x):
            assert fmt(L, bits) == ""                          # "INF or NaN number found"
            continue
        got, want = fmt(L, bits), go_format(x)
        assert got == want, (hex(bits), got, want)
        assert float(got) == x or (x == 0 and float(got) == 0)  # round trip
        n += 1
    assert n > 250000


def test_integers(L):
    buf = C.create_string_buffer(40)
    for v in (0, 1, 9, 10, 99, 12345678901234567890, 2**63, 2**64 - 1):
        n = L.sj_selftest_format_int(v, 1, buf)
        assert buf.raw[:n].decode() == str(v)
    for v in (0, 1, -1, 42, -42, 2**63 - 1, -2**63, -9223372036854775807):
        n = L.sj_selftest_format_int(v & (2**64 - 1), 0, buf)
        assert buf.raw[:n].decode() == str(v)
