"""The reference's fuzz corpora (testdata/fuzz/{corpus,go-corpus}.tar.zst: 9 036 inputs, 8 966 distinct) replayed on
the CPU: oracle vs Python's json (the role encoding/json plays in FuzzCorrect, fuzz_test.go:94-300) and the host
replay of the kernels' SJ_HD code vs the oracle.  The GPU replay is tests/test_gpu_fuzz.py."""
import ctypes as C
import json
import math

import numpy as np
import pytest

import __graft_entry__ as G
import fuzz_corpus
import oracle_lib as O
import tape_reader

SMALL = 64 << 10  # inputs up to this size are also compared value by value (pure-Python tape walk)


class _Reject(Exception):
    pass


def _no_const(_):
    raise _Reject("NaN / Infinity literals are not JSON")


def _finite_float(s):
    v = float(s)
    if math.isinf(v):
        raise _Reject("float overflow: strconv.ParseFloat returns ErrRange, the parse fails (quirk Q5)")
    return v


def _finite_int(s):
    v = int(s)
    try:  # integers beyond uint64 are parsed as floats (parse_number.go:96-134): the same overflow rule applies
        float(v)
    except OverflowError:
        raise _Reject("integer beyond float64") from None
    return v


def _stdlib(data):
    """What FuzzCorrect asks of encoding/json: valid UTF-8, valid JSON, top level object / array."""
    try:
        text = data.decode("utf-8")
        v = json.loads(text, parse_constant=_no_const, parse_float=_finite_float, parse_int=_finite_int)
    except (UnicodeDecodeError, ValueError, _Reject, RecursionError):
        return None
    if not isinstance(v, (dict, list)):
        return None
    try:  # a lone surrogate escape decodes to U+FFFD in Go and is rejected by the reference: not comparable
        json.dumps(v, ensure_ascii=False).encode("utf-8")
    except (UnicodeEncodeError, RecursionError):
        return None
    return v


def _same(a, b):
    if isinstance(a, dict):
        return isinstance(b, dict) and a.keys() == b.keys() and all(_same(a[k], b[k]) for k in a)
    if isinstance(a, list):
        return isinstance(b, list) and len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, bool) or isinstance(b, bool) or a is None or b is None:
        return a is b
    if isinstance(a, (int, float)) and isinstance(b, (int, float)):
        if isinstance(a, int) and isinstance(b, int):
            return a == b
        return float(a) == float(b)  # integers beyond uint64 become float64 in the reference (parse_number.go:96-118)
    return a == b


def test_oracle_accepts_and_agrees_with_stdlib_json():
    """FuzzCorrect's property: what the standard library accepts (UTF-8, object / array at the top) the parser
    accepts, with equal values.  Duplicate keys: the last one wins in both (tape_reader builds a dict in order)."""
    corpus = fuzz_corpus.load()
    accepted = compared = 0
    for i, data in enumerate(corpus):
        want = _stdlib(data)
        if want is None:
            continue
        accepted += 1
        p = O.parse(data, ndjson=False, copy_strings=True)
        assert p.rc == 0, (i, p.rc, data[:80])
        if len(data) <= SMALL:
            got = tape_reader.to_python(p.tape, p.strings, data[p.msg_off:p.msg_off + p.msg_len])
            assert len(got) == 1 and _same(want, got[0]), (i, data[:80])
            compared += 1
    assert accepted > 2000 and compared > 1500, (accepted, compared)


@pytest.fixture(scope="module")
def replay_lib():
    lib = C.CDLL(G.build_selftest())
    u64p, u8p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)
    lib.sj_selftest_parse.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(u64p), C.POINTER(C.c_size_t),
                                      C.POINTER(u8p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                      C.POINTER(C.c_size_t)]
    lib.sj_selftest_free.argtypes = [C.c_void_p]
    return lib


def test_host_replay_equals_oracle_on_fuzz_corpus(replay_lib):
    """The SJ_HD functions the kernels run (csrc/host_selftest.cpp replays them in launch order) against the
    oracle: verdict class, Tape and Strings.B, Parse and ParseND, both copy modes.  The 536 inputs above 64 KiB
    (268 MB, mutated copies of the fixtures) are replayed in one mode each, rotating."""
    from test_host_stage2 import replay
    corpus = fuzz_corpus.load()
    modes = [(nd, cp) for nd in (False, True) for cp in (True, False)]
    big = 0
    for i, data in enumerate(corpus):
        if len(data) <= SMALL:
            todo = modes
        else:
            todo = [modes[big % 4]]
            big += 1
        for nd, cp in todo:
            ref = O.parse(data, ndjson=nd, copy_strings=cp)
            rc, t, s = replay(replay_lib, data, nd, cp)
            assert rc == ref.rc, (i, nd, cp, rc, ref.rc, data[:80])
            if rc == 0:
                assert np.array_equal(t, ref.tape) and np.array_equal(s, ref.strings), (i, nd, cp)
