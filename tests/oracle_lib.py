"""ctypes binding of the CPU oracle (oracle/libsjoracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)

FLAG_NDJSON = 1
FLAG_COPY_STRINGS = 2


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ORACLE_DIR, "libsjoracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.sjo_find_odd_backslash_sequences.restype = C.c_uint64
        L.sjo_find_odd_backslash_sequences.argtypes = [C.c_char_p, u64p]
        L.sjo_find_quote_mask_and_bits.restype = C.c_uint64
        L.sjo_find_quote_mask_and_bits.argtypes = [C.c_char_p, C.c_uint64, u64p, u64p, u64p]
        L.sjo_find_whitespace_and_structurals.restype = None
        L.sjo_find_whitespace_and_structurals.argtypes = [C.c_char_p, u64p, u64p]
        L.sjo_finalize_structurals.restype = C.c_uint64
        L.sjo_finalize_structurals.argtypes = [C.c_uint64] * 4 + [u64p]
        L.sjo_find_newline_delimiters.restype = C.c_uint64
        L.sjo_find_newline_delimiters.argtypes = [C.c_char_p, C.c_uint64]
        L.sjo_flatten_bits_incremental.restype = None
        L.sjo_flatten_bits_incremental.argtypes = [u32p, C.POINTER(C.c_int), C.c_uint64, u64p, u64p]
        L.sjo_find_structural_bits.restype = C.c_uint64
        L.sjo_find_structural_bits.argtypes = [C.c_char_p, u64p, u64p, u64p, u64p]
        L.sjo_find_structural_bits_in_slice.restype = C.c_uint64
        L.sjo_find_structural_bits_in_slice.argtypes = [C.c_char_p, C.c_uint64, u64p, u64p, u64p, u64p, u32p,
                                                        C.POINTER(C.c_int), u64p, u64p, C.c_uint64]
        L.sjo_find_structural_indices.restype = C.c_int
        L.sjo_find_structural_indices.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t,
                                                  C.POINTER(C.c_size_t)]
        L.sjo_parse_string_validate_only.restype = C.c_int
        L.sjo_parse_string_validate_only.argtypes = [C.c_char_p, C.c_size_t, u64p, u64p]
        L.sjo_parse_string.restype = C.c_int
        L.sjo_parse_string.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, u64p]
        L.sjo_parse_number.restype = C.c_uint64
        L.sjo_parse_number.argtypes = [C.c_char_p, C.c_size_t, u64p]
        for f in ("true", "false", "null"):
            fn = getattr(L, f"sjo_is_valid_{f}_atom")
            fn.restype = C.c_int
            fn.argtypes = [C.c_char_p, C.c_size_t]
        L.sjo_trim_space.restype = None
        L.sjo_trim_space.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.sjo_parse.restype = C.c_int
        L.sjo_parse.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(u64p), C.POINTER(C.c_size_t),
                                C.POINTER(u8p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                C.POINTER(C.c_size_t)]
        L.sjo_free.restype = None
        L.sjo_free.argtypes = [C.c_void_p]
        # sjo_fast.c: AVX2 / PCLMULQDQ shapes for the CPU baseline
        szp = C.POINTER(C.c_size_t)
        L.sjo_avx2_available.restype = C.c_int
        L.sjo_find_structural_indices_avx2.restype = C.c_int
        L.sjo_find_structural_indices_avx2.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, szp]
        L.sjo_fast_create.restype = C.c_void_p
        L.sjo_fast_destroy.argtypes = [C.c_void_p]
        L.sjo_fast_parse.restype = C.c_int
        L.sjo_fast_parse.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.POINTER(u64p), szp,
                                     C.POINTER(u8p), szp]
        L.sjo_bench_stage1.restype = C.c_double
        L.sjo_bench_stage1.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, szp]
        L.sjo_bench_parse.restype = C.c_double
        L.sjo_bench_parse.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_int), szp]
        L.sjo_bench_nd_blocks.restype = C.c_double
        L.sjo_bench_nd_blocks.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_int)]
        # sjo_serialize.c
        bpp = C.POINTER(u8p)
        L.sjo_serialize.restype = C.c_int
        L.sjo_serialize.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int,
                                    bpp, szp, bpp, szp, bpp, szp, bpp, szp]
        L.sjo_deserialize.restype = C.c_int
        L.sjo_deserialize.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(u64p), szp, bpp, szp, bpp, szp]
        # sjo_marshal.c
        L.sjo_format_float.restype = C.c_int
        L.sjo_format_float.argtypes = [C.c_uint64, C.c_char_p]
        L.sjo_marshal_json.restype = C.c_int
        L.sjo_marshal_json.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, bpp, szp]
        _LIB = L
    return _LIB


def _as_np_u8(data):
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8)


def stage1(data, ndjson=False):
    """-> (ok, positions ndarray[uint32]) following findStructuralIndices."""
    a = _as_np_u8(data)
    cap = a.size + 64
    pos = np.empty(cap, dtype=np.uint32)
    n = C.c_size_t(0)
    ok = lib().sjo_find_structural_indices(a.ctypes.data, a.size, int(ndjson), pos.ctypes.data, cap, C.byref(n))
    return bool(ok), pos[: n.value].copy()


class Parsed:
    __slots__ = ("rc", "tape", "strings", "msg_off", "msg_len")


def parse(data, ndjson=False, copy_strings=True):
    """Whole parse (parseMessage).  -> Parsed(rc, tape u64 ndarray, strings u8 ndarray, msg_off, msg_len)."""
    a = _as_np_u8(data)
    tape = u64p()
    strs = u8p()
    tl, sl, mo, ml = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    flags = (FLAG_NDJSON if ndjson else 0) | (FLAG_COPY_STRINGS if copy_strings else 0)
    rc = lib().sjo_parse(a.ctypes.data, a.size, flags, C.byref(tape), C.byref(tl), C.byref(strs), C.byref(sl),
                         C.byref(mo), C.byref(ml))
    p = Parsed()
    p.rc = rc
    p.msg_off, p.msg_len = mo.value, ml.value
    if rc == 0:
        p.tape = np.ctypeslib.as_array(tape, shape=(tl.value,)).copy() if tl.value else np.zeros(0, np.uint64)
        p.strings = np.ctypeslib.as_array(strs, shape=(sl.value,)).copy() if sl.value else np.zeros(0, np.uint8)
        lib().sjo_free(tape)
        lib().sjo_free(strs)
    else:
        p.tape = np.zeros(0, np.uint64)
        p.strings = np.zeros(0, np.uint8)
    return p


def stage1_avx2(data, ndjson=False):
    """sjo_fast.c: findStructuralIndices with the reference's AVX2 / PCLMULQDQ shapes."""
    a = _as_np_u8(data)
    cap = a.size + 64
    pos = np.empty(cap, dtype=np.uint32)
    n = C.c_size_t(0)
    ok = lib().sjo_find_structural_indices_avx2(a.ctypes.data, a.size, int(ndjson), pos.ctypes.data, cap, C.byref(n))
    return bool(ok), pos[: n.value].copy()


class FastParser:
    """sjo_fast.c: whole parse with recycled buffers, 1 thread or the reference's 2-thread stage1 || stage2 shape."""

    def __init__(self):
        self._w = lib().sjo_fast_create()

    def close(self):
        if self._w:
            lib().sjo_fast_destroy(self._w)
            self._w = None

    def parse(self, data, ndjson=False, copy_strings=True, threads=1):
        a = _as_np_u8(data)
        tape, strs = u64p(), u8p()
        tl, sl = C.c_size_t(0), C.c_size_t(0)
        flags = (FLAG_NDJSON if ndjson else 0) | (FLAG_COPY_STRINGS if copy_strings else 0)
        rc = lib().sjo_fast_parse(self._w, a.ctypes.data, a.size, flags, threads, C.byref(tape), C.byref(tl),
                                  C.byref(strs), C.byref(sl))
        p = Parsed()
        p.rc = rc
        p.msg_off = p.msg_len = 0
        p.tape = np.ctypeslib.as_array(tape, shape=(tl.value,)).copy() if rc == 0 and tl.value else np.zeros(0, np.uint64)
        p.strings = np.ctypeslib.as_array(strs, shape=(sl.value,)).copy() if rc == 0 and sl.value else np.zeros(0, np.uint8)
        return p


def _take(ptr, n, dtype=np.uint8):
    a = np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].copy() if n else np.zeros(0, dtype)
    lib().sjo_free(ptr)
    return a


def serialize(tape, strings, message, dedup=True):
    """Serializer.Serialize (format v3, CompressNone).  -> (stream bytes, tags, values, string buffer)"""
    t = np.ascontiguousarray(tape, dtype=np.uint64)
    s = np.ascontiguousarray(strings, dtype=np.uint8)
    m = _as_np_u8(message)
    outs = [u8p() for _ in range(4)]
    lens = [C.c_size_t(0) for _ in range(4)]
    rc = lib().sjo_serialize(t.ctypes.data, t.size, s.ctypes.data if s.size else None, s.size, m.ctypes.data if m.size else None,
                             m.size, int(dedup), C.byref(outs[0]), C.byref(lens[0]), C.byref(outs[1]), C.byref(lens[1]),
                             C.byref(outs[2]), C.byref(lens[2]), C.byref(outs[3]), C.byref(lens[3]))
    assert rc == 0, rc
    return tuple(_take(o, l.value) for o, l in zip(outs, lens))


def deserialize(stream):
    """Serializer.Deserialize.  -> (rc, tape, strings, message)"""
    a = _as_np_u8(stream)
    tape, strs, msg = u64p(), u8p(), u8p()
    tl, sl, ml = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    rc = lib().sjo_deserialize(a.ctypes.data, a.size, C.byref(tape), C.byref(tl), C.byref(strs), C.byref(sl), C.byref(msg),
                               C.byref(ml))
    if rc:
        return rc, None, None, None
    return 0, _take(tape, tl.value, np.uint64), _take(strs, sl.value), _take(msg, ml.value)


def marshal_json(tape, strings, message):
    """Iter.MarshalJSON of the whole ParsedJson.  -> (rc, bytes)"""
    t = np.ascontiguousarray(tape, dtype=np.uint64)
    s = np.ascontiguousarray(strings, dtype=np.uint8)
    m = _as_np_u8(message)
    out, n = u8p(), C.c_size_t(0)
    rc = lib().sjo_marshal_json(t.ctypes.data, t.size, s.ctypes.data if s.size else None, m.ctypes.data if m.size else None,
                                C.byref(out), C.byref(n))
    if rc:
        return rc, b""
    return 0, _take(out, n.value).tobytes()


def format_float(bits):
    buf = C.create_string_buffer(48)
    n = lib().sjo_format_float(bits, buf)
    return buf.raw[:n].decode()
