"""The named quirks and the reference's atom tables through the oracle's whole parse."""
import numpy as np
import pytest

import oracle_lib as O
import quirk_cases as Q


@pytest.mark.parametrize("case", Q.QUIRKS, ids=[c[0] for c in Q.QUIRKS])
def test_quirk(case):
    name, doc, nd, accepted, strings = case
    for copy in (True, False):
        p = O.parse(doc, ndjson=nd, copy_strings=copy)
        assert (p.rc == 0) == accepted, (name, p.rc)
    if strings is not None:
        p = O.parse(doc, ndjson=nd, copy_strings=True)
        assert bytes(p.strings) == strings, (name, bytes(p.strings))


def test_atom_tables_through_parse():  # stage2_build_tape_amd64_test.go:195-262
    for name, doc, expected in Q.atom_documents():
        assert (O.parse(doc).rc == 0) == expected, name
