"""GPU: the reference's fuzz corpora (testdata/fuzz/{corpus,go-corpus}.tar.zst, 8 966 distinct inputs; loader
fuzz_test.go:308-408, used by FuzzParse :40 and FuzzCorrect :94) through the HIP kernels: error class, Tape and
Strings.B equal to the oracle's for Parse and ParseND in both copy modes."""
import numpy as np
import pytest

import fuzz_corpus
import oracle_lib as O
from test_gpu_parse import ctx, gpu_parse  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("part", range(8))
def test_fuzz_corpus_equals_oracle(ctx, part):
    corpus = fuzz_corpus.load()
    n_ok = n_bad = 0
    for i in range(part, len(corpus), 8):
        data = corpus[i]
        for nd in (False, True):
            for copy in (True, False):
                ref = O.parse(data, ndjson=nd, copy_strings=copy)
                rc, pj = gpu_parse(ctx, data, nd, copy)
                assert rc == ref.rc, (i, nd, copy, rc, ref.rc, data[:80])
                if rc == 0:
                    n_ok += 1
                    assert pj.Message == bytes(data[ref.msg_off:ref.msg_off + ref.msg_len]), i
                    assert np.array_equal(pj.Tape, ref.tape), (i, nd, copy, "tape")
                    assert np.array_equal(pj.Strings, ref.strings), (i, nd, copy, "strings")
                else:
                    n_bad += 1
    assert n_ok > 1000 and n_bad > 1000, (n_ok, n_bad)  # both verdicts are well represented in every slice
