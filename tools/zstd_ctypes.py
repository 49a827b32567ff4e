"""Minimal streaming zstd decompressor over the system libzstd.so.1 (ctypes).

Only used by the tools/ scripts that run in the development container (where
/root/reference exists) to read the reference's `testdata/*.zst` fixtures.
"""
import ctypes
import ctypes.util


class _Buf(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p), ("size", ctypes.c_size_t), ("pos", ctypes.c_size_t)]


def _lib():
    name = ctypes.util.find_library("zstd") or "libzstd.so.1"
    lib = ctypes.CDLL(name)
    lib.ZSTD_createDStream.restype = ctypes.c_void_p
    lib.ZSTD_freeDStream.argtypes = [ctypes.c_void_p]
    lib.ZSTD_initDStream.argtypes = [ctypes.c_void_p]
    lib.ZSTD_initDStream.restype = ctypes.c_size_t
    lib.ZSTD_decompressStream.argtypes = [ctypes.c_void_p, ctypes.POINTER(_Buf), ctypes.POINTER(_Buf)]
    lib.ZSTD_decompressStream.restype = ctypes.c_size_t
    lib.ZSTD_isError.argtypes = [ctypes.c_size_t]
    return lib


def decompress(data: bytes) -> bytes:
    lib = _lib()
    ds = lib.ZSTD_createDStream()
    lib.ZSTD_initDStream(ds)
    src = ctypes.create_string_buffer(data, len(data))
    inb = _Buf(ctypes.cast(src, ctypes.c_void_p), len(data), 0)
    chunk = 1 << 20
    dst = ctypes.create_string_buffer(chunk)
    out = bytearray()
    while True:
        outb = _Buf(ctypes.cast(dst, ctypes.c_void_p), chunk, 0)
        rc = lib.ZSTD_decompressStream(ds, ctypes.byref(outb), ctypes.byref(inb))
        if lib.ZSTD_isError(rc):
            raise RuntimeError("zstd error")
        out += dst.raw[: outb.pos]
        if inb.pos >= inb.size and outb.pos < chunk:
            break
    lib.ZSTD_freeDStream(ds)
    return bytes(out)
