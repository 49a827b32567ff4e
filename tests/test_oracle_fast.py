"""The AVX2 / PCLMULQDQ restatement (oracle/sjo_fast.c, the CPU baseline of bench.py) against the scalar oracle:
identical positions, verdicts, tapes and Strings.B -- fixtures, the reference's corpora, the fuzz corpus, 1 and 2 threads."""
import numpy as np
import pytest

import fixtures
import fuzz_corpus
import golden_util as GU
import oracle_lib as O

pytestmark = pytest.mark.skipif(not O.lib().sjo_avx2_available(), reason="host CPU without AVX2 / PCLMULQDQ")


def same_parse(fp, data, nd, copy, threads, what):
    ref = O.parse(data, ndjson=nd, copy_strings=copy)
    got = fp.parse(data, ndjson=nd, copy_strings=copy, threads=threads)
    assert got.rc == ref.rc, (what, nd, copy, threads, got.rc, ref.rc)
    if ref.rc == 0:
        assert np.array_equal(got.tape, ref.tape) and np.array_equal(got.strings, ref.strings), (what, nd, copy, threads)


@pytest.mark.parametrize("name", fixtures.ALL)
def test_fixtures(name):
    data = fixtures.load(name)
    fp = O.FastParser()
    for nd in (False, True):
        ok, pos = O.stage1(data, nd)
        ok2, pos2 = O.stage1_avx2(data, nd)
        assert ok == ok2 and (not ok or np.array_equal(pos, pos2)), (name, nd)  # a failing scalar run stops early
        for copy in (True, False):
            for threads in (1, 2):
                same_parse(fp, data, nd, copy, threads, name)
    fp.close()


def test_reference_corpora_and_fuzz_inputs():
    fp = O.FastParser()
    corp = GU.load("corpus")
    docs = [bytes.fromhex(c["js_hex"]) for k in ("fail_cases", "pass_cases", "parse_nd") for c in corp[k]]
    docs += [b'["' + bytes.fromhex(r["str_hex"]) + b'"]' for r in GU.load("strings")]
    for i, d in enumerate(docs):
        for nd in (False, True):
            same_parse(fp, d, nd, True, 1, f"corpus{i}")
            same_parse(fp, d, nd, False, 2, f"corpus{i}")
    for i, d in enumerate(fuzz_corpus.load()):
        if len(d) > (64 << 10) and i % 7:
            continue
        nd, copy, threads = bool(i & 1), bool(i & 2), 1 + ((i >> 2) & 1)
        ok, pos = O.stage1(d, nd)
        ok2, pos2 = O.stage1_avx2(d, nd)
        assert ok == ok2 and (not ok or np.array_equal(pos, pos2)), ("fuzz", i)
        same_parse(fp, d, nd, copy, threads, f"fuzz{i}")
    fp.close()
