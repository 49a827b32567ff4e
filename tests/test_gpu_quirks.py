"""GPU: the named quirks of SURVEY.md A.6 (Q2..Q7, ND) and the reference's atom tables
(stage2_build_tape_amd64_test.go:195-262) through the HIP kernels: the verdict the reference's code implies
(tests/quirk_cases.py, pinned on the CPU by test_quirks_oracle.py) and bit-equality with the oracle."""
import numpy as np
import pytest

import oracle_lib as O
import quirk_cases as Q
from test_gpu_parse import check, ctx, gpu_parse  # noqa: F401  (ctx is the module fixture)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", Q.QUIRKS, ids=[c[0] for c in Q.QUIRKS])
def test_quirk(ctx, case):
    name, doc, nd, accepted, strings = case
    for copy in (True, False):
        rc, pj = gpu_parse(ctx, doc, nd, copy)
        assert (rc == 0) == accepted, (name, copy, rc)
        if copy and strings is not None:
            assert bytes(pj.Strings) == strings, (name, bytes(pj.Strings))
    check(ctx, doc, nd, name)
    # the same case in the middle of a larger document (other chunk / unit / tile offsets): still equal to the oracle
    if not nd and doc[:1] in b"[{" and doc[-1:] in b"]}":
        for pad in (61, 4093, 131069):
            check(ctx, b'["' + b"p" * pad + b'",' + doc + b"]", False, name + f" @+{pad}")


def test_atom_tables_through_parse(ctx):
    for name, doc, expected in Q.atom_documents():
        for copy in (True, False):
            rc, _ = gpu_parse(ctx, doc, False, copy)
            assert (rc == 0) == expected, (name, rc)
        check(ctx, doc, False, name)
    # every atom at every offset relative to a 64-byte chunk and to the end of the message
    for atom in (b"true", b"false", b"null", b"tru", b"nul", b"fals", b"trux", b"nullx", b"false1"):
        for pad in list(range(0, 70)) + [4090, 4091, 4092, 4093, 4094, 4095]:
            check(ctx, b"[" + b" " * pad + atom + b"]", False, f"atom {atom!r} @{pad}")
            check(ctx, b"[" + b" " * pad + atom, False, f"cut atom {atom!r} @{pad}")


def test_long_surrogate_runs(ctx):
    """runs of adjacent high-surrogate escapes up to and beyond SURROGATE_WALK_CAP (the byte-parallel string path
    hands the document to the per-string walks: parse_api.hip, S2_ERR_SERIAL_STRINGS)"""
    from test_host_stage2 import surrogate_run_docs
    for doc, what in surrogate_run_docs():
        check(ctx, doc, False, what)
    big = b'{"k":"' + b"\\ud800" * 100001 + b'\\udc00","t":[1,2,3]}\n'
    check(ctx, big * 3, True, "nd-long-runs")
