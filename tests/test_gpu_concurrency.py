"""GPU: stage 1's forward progress with several large parses in flight on one device.

The stage-1 kernel resolves the in-string state and the output offset of a tile with a decoupled look-back over the tiles
in front of it, which needs every tile in front to belong to a block that is running or through.  The reference gets the
order for free from its sequential loop (stage1_find_marks_amd64.go:41-148).  Here a message of more than one round of
tiles (> 256 tiles: > 32 MiB) draws every tile from a ticket counter; round 4 had given each block a static first tile,
which is only safe while the whole grid is resident -- with other kernels holding compute units a resident block could
look back on tiles of blocks that had not been dispatched, spin to its bound and fail the parse with an internal error.
The library creates exactly that load itself (shards of a big ND message on one device, stream slots, a pool of contexts
under goroutines), so: several host threads, one context each, documents of 64 MiB and more, back to back."""
import threading

import numpy as np
import pytest

import fixtures
import workloads

pytestmark = pytest.mark.gpu


def _device_copy(data):
    import torch
    d = torch.empty(len(data) + 256, dtype=torch.uint8, device="cuda:0")
    d[:len(data)].copy_(torch.frombuffer(bytearray(data), dtype=torch.uint8))
    torch.cuda.synchronize()
    return d


def test_large_parses_from_four_threads_keep_their_order():
    import sjhip
    tw = fixtures.load("twitter")
    park = fixtures.load("parking-citations")
    # (document, ndjson, expected structurals, expected tape words, expected Strings.B bytes): closed forms of SURVEY.md 8d
    docs = []
    for copies in (107, 130):  # 64.4 MiB / 78 MiB arrays of twitter.json: 516 / 627 tiles for 256 blocks
        d = workloads.c2_twitter_array(copies)
        docs.append((d, False, workloads.c2_expected_structurals(copies), copies * 49781 + 4, copies * 367917))
    for copies in (190, 215):  # 70.8 MB / 80 MB of parking-citations ND
        d = (park * copies).rstrip(b"\n")
        docs.append((d, True, 78 * 1000 * copies - 1, 80000 * copies, 256664 * copies))
    assert all(len(d[0]) > (64 << 20) for d in docs)
    dev = [_device_copy(d[0]) for d in docs]
    # one reference result per document (tape and Strings.B fetched once, single-threaded), then everything must repeat it
    ref = []
    c0 = sjhip.Context(0)
    for (d, nd, s, t, b), buf in zip(docs, dev):
        tl, sl = c0.parse_device(buf.data_ptr(), len(d), ndjson=nd, copy_strings=True)
        assert (tl, sl) == (t, b), (tl, sl, t, b)
        tape, strings = c0.fetch(tl, sl)
        ref.append((tape.copy(), strings.copy()))
    c0.close()
    errors = []
    ROUNDS = 130  # x 4 documents per thread = 520 parses per thread, 2080 in all

    def worker(k):
        try:
            c = sjhip.Context(0)
            import torch
            pos = torch.empty(30_000_000, dtype=torch.int32, device="cuda:0")
            for r in range(ROUNDS):
                for j in range(len(docs)):
                    i = (j + k) % len(docs)
                    d, nd, s, t, b = docs[i]
                    if (r + j) % 3 == 0:  # plain stage 1 between the whole parses (its own kernel instantiation)
                        ok, n = c.stage1_device(dev[i].data_ptr(), len(d), pos.data_ptr(), pos.numel(), ndjson=nd)
                        if not ok or n != s:
                            errors.append((k, r, i, "stage 1", ok, n, s))
                    tl, sl = c.parse_device(dev[i].data_ptr(), len(d), ndjson=nd, copy_strings=True)
                    if (tl, sl) != (t, b):
                        errors.append((k, r, i, "sizes", tl, sl))
                    if r % 26 == k:  # bit for bit, a few times per thread (the fetch is 0.7 GB each)
                        tape, strings = c.fetch(tl, sl)
                        if not (np.array_equal(tape, ref[i][0]) and np.array_equal(strings, ref[i][1])):
                            errors.append((k, r, i, "tape / Strings.B differ"))
                if errors:
                    break
            c.close()
        except Exception as e:  # noqa: BLE001  (sjhip.ParseError "internal synchronisation timeout" would land here)
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]
