"""The reference's fuzz corpora (testdata/fuzz/*.tar.zst, loaded like fuzz_test.go:308-408) as packed by
tools/make_fuzz_fixture.py into tests/data/fuzz.bin.xz."""
import functools
import lzma
import os
import struct

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "fuzz.bin.xz")


@functools.lru_cache(maxsize=1)
def load():
    """-> list of bytes (8 966 unique inputs of the 9 036 the two archives hold)."""
    with open(PATH, "rb") as f:
        blob = lzma.decompress(f.read())
    (n,) = struct.unpack_from("<I", blob, 0)
    out, o = [], 4
    for _ in range(n):
        (ln,) = struct.unpack_from("<I", blob, o)
        out.append(blob[o + 4:o + 4 + ln])
        o += 4 + ln
    assert o == len(blob)
    return out
