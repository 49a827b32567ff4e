#!/usr/bin/env python3
"""Materialise the reference's JSON fixtures inside the repo.

Reads /root/reference/testdata/*.json.zst (public JSON corpora used by the
reference's tests and benchmarks; data, not code), decompresses them with the
system libzstd and re-packs them as tests/data/<name>.json.xz so that tests,
smoke() and bench.py can run on the GPU box, where /root/reference does not
exist and only the Python standard library (lzma) is needed to read them.

Run in the development container:  python tools/make_fixtures.py
"""
import hashlib
import json
import lzma
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from zstd_ctypes import decompress  # noqa: E402

REF = "/root/reference/testdata"
OUT = os.path.join(HERE, "..", "tests", "data")


def main():
    os.makedirs(OUT, exist_ok=True)
    manifest = {}
    for fn in sorted(os.listdir(REF)):
        if not fn.endswith(".json.zst"):
            continue
        name = fn[: -len(".json.zst")]
        raw = decompress(open(os.path.join(REF, fn), "rb").read())
        with open(os.path.join(OUT, name + ".json.xz"), "wb") as f:
            f.write(lzma.compress(raw, preset=9 | lzma.PRESET_EXTREME))
        manifest[name] = {"bytes": len(raw), "sha1": hashlib.sha1(raw).hexdigest()}
        print(f"{name:20s} {len(raw):10d} {manifest[name]['sha1'][:12]}")
    with open(os.path.join(OUT, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
