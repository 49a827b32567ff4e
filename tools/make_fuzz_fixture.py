#!/usr/bin/env python3
"""Materialise the reference's fuzz corpora inside the repo (last row).

Reads /root/reference/testdata/fuzz/{corpus,go-corpus}.tar.zst exactly like the reference's loader
(fuzz_test.go:308-347): every tar entry is one input; entries that start with "go test fuzz" are Go corpus files
whose lines are `[]byte("...")` literals (unmarshalCorpusFile / parseCorpusValue, fuzz_test.go:349-408).  The inputs
(data, not code) are packed as tests/data/fuzz.bin.xz: u32 count, then per input u32 length + bytes, duplicates
removed, order kept.  The GPU box has no /root/reference and no zstd module: it reads this file with lzma.

Run in the development container:  python tools/make_fuzz_fixture.py
"""
import hashlib
import io
import json
import lzma
import os
import struct
import sys
import tarfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from golit import unquote  # noqa: E402
from zstd_ctypes import decompress  # noqa: E402

REF = "/root/reference/testdata/fuzz"
OUT = os.path.join(HERE, "..", "tests", "data")


def corpus_values(b: bytes):
    lines = b.split(b"\n")
    for line in lines[1:]:
        line = line.strip()
        if not line:
            continue
        s = line.decode("utf-8")
        if not (s.startswith('[]byte("') and s.endswith('")')):
            raise SyntaxError("unexpected corpus line: " + s[:60])
        yield unquote(s[len('[]byte("'):-2])


def inputs_of(tar_zst):
    raw = decompress(open(tar_zst, "rb").read())
    n_entries = 0
    with tarfile.open(fileobj=io.BytesIO(raw)) as tf:
        for m in tf:
            if not m.isfile():
                continue
            n_entries += 1
            b = tf.extractfile(m).read()
            if b.startswith(b"go test fuzz"):
                yield from corpus_values(b)
            else:
                yield b
    print(f"{os.path.basename(tar_zst)}: {n_entries} tar entries", file=sys.stderr)


def main():
    seen, out, total = set(), [], 0
    per = {}
    for fn in ("corpus.tar.zst", "go-corpus.tar.zst"):
        k = 0
        for v in inputs_of(os.path.join(REF, fn)):
            k += 1
            h = hashlib.sha1(v).digest()
            if h in seen:
                continue
            seen.add(h)
            out.append(v)
            total += len(v)
        per[fn] = k
    blob = struct.pack("<I", len(out)) + b"".join(struct.pack("<I", len(v)) + v for v in out)
    with open(os.path.join(OUT, "fuzz.bin.xz"), "wb") as f:
        f.write(lzma.compress(blob, preset=9 | lzma.PRESET_EXTREME))
    meta = {"inputs_per_archive": per, "unique_inputs": len(out), "bytes": total,
            "sha1": hashlib.sha1(blob).hexdigest(), "largest": max(len(v) for v in out)}
    with open(os.path.join(OUT, "fuzz.MANIFEST.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(meta)


if __name__ == "__main__":
    main()
