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


def test_shards_and_stream_blocks_sit_on_the_devices_asked_for():
    """Placement, as the HIP runtime reports it.  On the one-GPU development boxes everything is device 0; on a node
    with several GPUs this is the test that sees the multi-device code on real devices: one shard per GPU with its tape
    resident there, the merged result still bit for bit the oracle's, and the blocks of a stream spread round robin."""
    import io
    import sjhip
    from sjhip.api import MultiContext
    L = sjhip.lib()
    count = L.sjhip_device_count()
    assert count >= 1
    doc = fixtures.load("parking-citations") * (12 * count)
    ref = O.parse(doc, ndjson=True, copy_strings=True)
    for devices in (None, list(range(count)), list(range(count)) * 2, [count - 1]):
        m = MultiContext(devices)
        try:
            want = list(range(count)) if devices is None else devices
            assert m.shards == len(want)
            assert all(L.sjhip_multi_shard_device(m._h, k) == -1 for k in range(m.shards))  # nothing parsed yet
            pj = m.parse_nd(doc)
            assert np.array_equal(pj.Tape, ref.tape) and np.array_equal(pj.Strings, ref.strings), devices
            assert [L.sjhip_multi_shard_device(m._h, k) for k in range(m.shards)] == want
        finally:
            m.close()
    # ParseNDStream over every visible GPU: blocks go round robin over the devices, results arrive in input order
    bs = 1 << 20
    blocks = list(sjhip.cut_blocks(io.BytesIO(doc), bs))
    got = list(sjhip.parse_nd_stream(io.BytesIO(doc), block_size=bs, inflight=2 * count, n_devices=0))
    assert len(got) == len(blocks)
    for pj, blk in zip(got, blocks):
        r = O.parse(blk, ndjson=True, copy_strings=True)
        assert np.array_equal(pj.Tape, r.tape) and np.array_equal(pj.Strings, r.strings)
    assert {pj.device for pj in got} == set(range(min(count, len(blocks))))
    # one context per device through the plain API (what the Go binding's context pool does)
    for d in range(count):
        c = sjhip.Context(d)
        pj = c.parse(doc, ndjson=True)
        assert np.array_equal(pj.Tape, ref.tape)
        c.close()
