"""GPU: ParseND over several shards inside the library (sjhip_multi_*, csrc/multi_api.hip).  The one-GPU box lists
device 0 several times -- three contexts, three host threads, the same record cuts, prefix sums and rebased emits a
multi-GPU node runs -- and the merged ParsedJson must be bit for bit the oracle's ParseND of the whole message; an
invalid shard fails the whole parse with the reference's precedence (stage 1 before stage 2)."""
import numpy as np
import pytest

import fixtures
import oracle_lib as O

pytestmark = pytest.mark.gpu


def _docs():
    park = fixtures.load("parking-citations")
    good = park * 12                                               # ~4.5 MB, 12 000 records
    yield "parking x12", good, 0
    yield "stage-1 error in the middle shard", good[: len(good) // 2 + 5000] + b'{"broken":"unterminated\n' + good[len(good) // 2 + 5000:], 1
    yield "stage-2 error in the first shard", b'{"a":[1,2}\n' + good, 2
    yield "stage 1 wins over stage 2", b'{"a":[1,2}\n' + good + b'{"broken":"unterminated\n', 1
    yield "escapes, blank lines", b'\n\n' + b'{"k":"\\u00e9\\ud83d\\ude00","n":[1.5e3,-7,null]}\n' * 3000 + b' \n', 0
    yield "fewer records than shards", b'{"a":1}\n{"b":[true,"x"]}', 0
    yield "one record", b'  {"a":{"b":[1,2,3]}}  ', 0
    yield "blank", b" \n \n", 1


@pytest.mark.parametrize("devices", [[0], [0, 0, 0], [0] * 7])
def test_parse_nd_multi_equals_oracle(devices):
    import sjhip
    from sjhip.api import MultiContext
    m = MultiContext(devices)
    assert m.shards == len(devices)
    try:
        for what, doc, want in _docs():
            for copy in (True, False):
                ref = O.parse(doc, ndjson=True, copy_strings=copy)
                assert ref.rc == want, (what, ref.rc)
                try:
                    pj = m.parse_nd(doc, copy_strings=copy)
                    rc = 0
                except sjhip.ParseError as e:
                    rc = e.code
                assert rc == want, (what, devices, copy, rc)
                if rc == 0:
                    assert pj.Message == bytes(doc[ref.msg_off:ref.msg_off + ref.msg_len]), what
                    assert np.array_equal(pj.Tape, ref.tape), (what, devices, copy)
                    assert np.array_equal(pj.Strings, ref.strings), (what, devices, copy)
    finally:
        m.close()


def test_multi_all_devices_and_reuse():
    """devices = None: one shard per visible device; the handle is reused across documents of different sizes."""
    import sjhip
    from sjhip.api import MultiContext
    m = MultiContext(None)
    try:
        assert m.shards == sjhip.lib().sjhip_device_count()
        for copies in (1, 5, 2):
            doc = fixtures.load("parking-citations") * copies
            ref = O.parse(doc, ndjson=True, copy_strings=True)
            pj = m.parse_nd(doc)
            assert np.array_equal(pj.Tape, ref.tape) and np.array_equal(pj.Strings, ref.strings), copies
    finally:
        m.close()
