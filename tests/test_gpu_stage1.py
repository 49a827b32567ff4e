"""GPU parity tests for stage 1 (run with -m gpu on an MI355X): the HIP path, called through the
C ABI, against the oracle and the reference's golden vectors."""
import ctypes as C

import numpy as np
import pytest

import fixtures
import golden_util as GU
import oracle_lib as O
import workloads

pytestmark = pytest.mark.gpu
U = int


@pytest.fixture(scope="module")
def ctx():
    import sjhip
    assert sjhip.supported(), "gfx950 device required"
    c = sjhip.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def L():
    import sjhip
    return sjhip.lib()


S1 = GU.load("stage1")


def u64(v=0):
    return C.c_uint64(v)


# ---- the reference's per-routine KATs replayed on the device (find_subroutines_amd64_test.go) ----
def test_kat_finalize(ctx, L):
    for r in S1["finalize"]:
        pp, out = u64(0), u64(0)
        assert L.sjhip_finalize_structurals(ctx._h, U(r["structurals"]), U(r["whitespace"]), U(r["quote_mask"]),
                                            U(r["quote_bits"]), C.byref(pp), C.byref(out)) == 0
        assert out.value == U(r["expected"]) and pp.value == U(r["expected_pseudo"])


def test_kat_odd_backslash(ctx, L):
    for r in S1["odd_backslash"]:
        prev, out = u64(U(r["prev"])), u64(0)
        assert L.sjhip_find_odd_backslash_sequences(ctx._h, bytes.fromhex(r["input_hex"]), C.byref(prev),
                                                    C.byref(out)) == 0
        assert out.value == U(r["expected"]) and prev.value == U(r["ends_odd"])
    for i in (1, 2, 31, 32, 33, 62, 63, 64, 65, 100, 127, 128):
        t = b" " * (i - 1) + b'\\"' + b" " * (62 + 64)
        prev, lo, hi = u64(0), u64(0), u64(0)
        L.sjhip_find_odd_backslash_sequences(ctx._h, t[:64], C.byref(prev), C.byref(lo))
        L.sjhip_find_odd_backslash_sequences(ctx._h, t[64:128], C.byref(prev), C.byref(hi))
        want = (1 << i, 0) if i < 64 else (0, (1 << (i - 64)) & (2**64 - 1))
        assert (lo.value, hi.value) == want


def test_kat_quote_mask(ctx, L):
    for r in S1["quote_mask"]:
        piq, qb, em, qm = u64(0), u64(0), u64(0), u64(0)
        assert L.sjhip_find_quote_mask_and_bits(ctx._h, bytes.fromhex(r["input_hex"]), U(r["odd_ends"]),
                                                C.byref(piq), C.byref(qb), C.byref(em), C.byref(qm)) == 0
        assert (qm.value, qb.value, piq.value, em.value) == (U(r["expected"]), U(r["quote_bits"]),
                                                             U(r["inside_quote"]), U(r["error_mask"]))
    for r in S1["quote_mask_carry"]:
        piq, qb, em, qm = u64(U(r["inside_quote_in"])), u64(0), u64(0), u64(0)
        L.sjhip_find_quote_mask_and_bits(ctx._h, bytes.fromhex(r["input_hex"]), 0, C.byref(piq), C.byref(qb),
                                         C.byref(em), C.byref(qm))
        assert piq.value == U(r["inside_quote_out"])


def test_kat_whitespace_structurals(ctx, L):
    for r in S1["whitespace_structurals"]:
        ws, st = u64(0), u64(0)
        inp = bytes.fromhex(r["input_hex"])[:64].ljust(64, b"\0")
        assert L.sjhip_find_whitespace_and_structurals(ctx._h, inp, C.byref(ws), C.byref(st)) == 0
        assert (ws.value, st.value) == (U(r["whitespace"]), U(r["structurals"]))


def test_kat_newline(ctx, L):
    nd = bytes.fromhex(S1["demo_ndjson_hex"])
    want = [U(x) for x in S1["newline_demo_ndjson"]]
    for off in range(0, len(nd) - 64, 64):
        m = u64(0)
        L.sjhip_find_newline_delimiters(ctx._h, nd[off:off + 64], 0, C.byref(m))
        assert m.value == want[off >> 6]


def test_kat_flatten(ctx, L):
    for r in S1["flatten"]:
        base = (C.c_uint32 * 1536)()
        idx = C.c_int(0)
        carried, position = u64(0), u64(2**64 - 1)
        for m in r["masks"]:
            assert L.sjhip_flatten_bits_incremental(ctx._h, base, C.byref(idx), U(m), C.byref(carried),
                                                    C.byref(position)) == 0
        assert list(base[: idx.value]) == r["expected"]


def test_kat_odd_backslash_random_chunks(ctx, L):
    """Chunks dense in backslashes and quotes, both carries: the entry point evaluates the reference form
    (odd_backslash_ends) and the form the kernel runs (escaped_mask + trailing-run carry) on the device and fails
    if they disagree; the result is compared with the oracle."""
    OL = O.lib()
    rng = np.random.default_rng(20260922)
    alphabets = [b'\\\\"a', b'\\"', b'\\', b'\\a"  ']
    for trial in range(400):
        al = np.frombuffer(alphabets[trial % 4], dtype=np.uint8)
        chunk = bytes(al[rng.integers(0, al.size, 64)])
        if trial % 50 == 0:
            chunk = b"\\" * 64
        for carry in (0, 1):
            prev, out = u64(carry), u64(0)
            rc = L.sjhip_find_odd_backslash_sequences(ctx._h, chunk, C.byref(prev), C.byref(out))
            assert rc == 0, (rc, L.sjhip_last_error(ctx._h), chunk)
            rprev = u64(carry)
            want = OL.sjo_find_odd_backslash_sequences(chunk, C.byref(rprev))
            assert out.value == want and prev.value == rprev.value, (chunk, carry)


# ---- whole stage 1 ----
def test_demo_json_positions(ctx):
    ok, pos = ctx.stage1(bytes.fromhex(S1["demo_json_hex"]))
    assert ok and list(pos) == S1["demo_json_positions"]


def test_twitter_loop_golden(ctx):
    msg = fixtures.load("twitter")
    ok, pos = ctx.stage1(msg)
    assert ok and len(pos) == S1["twitter_loop"]["expected_length"]
    assert bytes(msg[p] for p in pos[::-1][:5]).decode() == S1["twitter_loop"]["last_structurals_reversed"]


@pytest.mark.parametrize("name", fixtures.ALL)
def test_fixture_positions_equal_oracle(ctx, name):
    data = fixtures.load(name).strip()
    for nd in (False, True):
        ok_ref, pos_ref = O.stage1(data, nd)
        ok, pos = ctx.stage1(data, nd)
        assert ok == ok_ref
        assert np.array_equal(pos, pos_ref)


def test_whitespace_padding_lengths(ctx):  # find_subroutines_amd64_test.go:381-421
    for l in range(0, 65):
        ok, pos = ctx.stage1(b":" * l)
        assert list(pos) == list(range(l))


def test_small_and_edge_inputs(ctx):
    cases = [b"", b"{}", b"[]", b'"', b'{"a":"b"}', b'{"a":"\\""}', b"[1,2,3]", b'["\x01"]', b'{"a":1}x', b"\\",
             b'"abc', b"[" + b" " * 200 + b"]", b'["' + b"x" * 63 + b'"]', b'["' + b"x" * 62 + b'\\"' + b'"]']
    for data in cases:
        for nd in (False, True):
            ok_ref, pos_ref = O.stage1(data, nd)
            ok, pos = ctx.stage1(data, nd)
            assert ok == ok_ref, data
            if ok_ref:
                assert np.array_equal(pos, pos_ref), data


def test_backslash_runs_across_boundaries(ctx):
    # runs of backslashes straddling chunk (64), wave-unit (4096), pass (32768) and tile (65536, 131072) boundaries
    for boundary in (64, 4096, 32768, 65536, 131072):
        for k in list(range(0, 9)) + [63, 64, 65, 127, 128, 129, 200]:
            for shift in (-3, -1, 0, 1):
                pre = boundary - 2 - k // 2 + shift
                if pre < 0:
                    continue
                data = b'["' + b"a" * pre + b"\\" * k + b'\\"x","y"]'
                ok_ref, pos_ref = O.stage1(data, False)
                ok, pos = ctx.stage1(data, False)
                assert ok == ok_ref, (boundary, k, shift)
                if ok_ref:
                    assert np.array_equal(pos, pos_ref), (boundary, k, shift)


def test_random_structural_soup(ctx):
    rng = np.random.default_rng(20240922)
    alphabet = np.frombuffer(b'\\\\\\""""{}[]:,  \n\tabc019.-e', dtype=np.uint8)
    for trial in range(60):
        n = int(rng.integers(1, 100000))
        body = bytes(alphabet[rng.integers(0, alphabet.size, n)])
        data = b"[" + body + b"]"
        for nd in (False, True):
            ok_ref, pos_ref = O.stage1(data, nd)
            ok, pos = ctx.stage1(data, nd)
            assert ok == ok_ref
            if ok_ref:
                assert np.array_equal(pos, pos_ref)
            else:
                # the reference stops handing over index buffers at the first failure; the prefix must agree
                assert np.array_equal(pos[: len(pos_ref)], pos_ref) or len(pos_ref) == 0


def test_quotes_and_escapes_across_tile_boundaries(ctx):
    # strings that open in one tile and close in a later one; escaped quotes right at the boundaries
    for tile in (32768, 65536, 131072):
        for delta in (-2, -1, 0, 1, 2):
            n = tile + delta
            for body in (b"x" * n, b"x" * (n - 2) + b'\\"' + b"y" * 70000, b"x" * (n - 1) + b"\\\\" + b"z" * 5):
                data = b'{"k":"' + body + b'","n":[1,2,{"a":null}]}'
                for nd in (False, True):
                    ok_ref, pos_ref = O.stage1(data, nd)
                    ok, pos = ctx.stage1(data, nd)
                    assert ok == ok_ref and np.array_equal(pos, pos_ref), (tile, delta, len(body), nd)


def test_multi_tile_twitter_replicated(ctx):
    data = workloads.c2_twitter_array(40)  # ~25 MB, ~770 tiles
    ok_ref, pos_ref = O.stage1(data, False)
    ok, pos = ctx.stage1(data, False)
    assert ok and ok_ref
    assert np.array_equal(pos, pos_ref)


def test_unaligned_device_pointer_and_full_size_property(ctx):
    import torch
    data = workloads.c2_twitter_array(426)
    n_expect = workloads.c2_expected_structurals(426)
    host = torch.frombuffer(bytearray(data), dtype=torch.uint8)
    for lead in (0, 1, 17, 63):
        dev = torch.empty(len(data) + 256, dtype=torch.uint8, device="cuda:0")
        dev[lead:lead + len(data)].copy_(host)
        pos = torch.empty(n_expect + 64, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        ok, n = ctx.stage1_device(dev.data_ptr() + lead, len(data), pos.data_ptr(), pos.numel())
        assert ok and n == n_expect
        p = pos[:n].cpu().numpy().view(np.uint32)
        assert np.all(np.diff(p.astype(np.int64)) > 0)          # strictly increasing
        tw = len(fixtures.load("twitter")) + 1
        # periodicity: copy k's structurals are copy 0's shifted by k*(len(twitter)+1)
        per = 55263 + 1
        first = p[1:1 + 55263].astype(np.int64)
        k = 300
        assert np.array_equal(p[1 + k * per: 1 + k * per + 55263].astype(np.int64), first + k * tw)
        del dev, pos


def test_queued_launches_keep_their_own_results(ctx):
    """sjhip_stage1_device_queue / _wait / _result: launches of different messages (valid, rejected by stage 1, ending
    inside a string, empty) behind one another without a synchronisation in between; every slot's count and verdict
    are the oracle's and the positions of every message are intact afterwards (each launch has its own buffers)."""
    import torch
    tw = fixtures.load("twitter")
    docs = [workloads.c2_twitter_array(3), b'{"a":"\x01"}', tw, b'{"open":"never closed', b"", b'[1,2,3]',
            workloads.c5_parking_nd(40).rstrip(b"\n"), b'{"k":[true,false,null]}', tw * 2]  # (TrimSpace'd, like every message of the API)
    nd = [False, False, False, False, False, False, True, False, False]
    bufs = []
    for d in docs:
        dev = torch.empty(len(d) + 256, dtype=torch.uint8, device="cuda:0")
        if d:
            dev[:len(d)].copy_(torch.frombuffer(bytearray(d), dtype=torch.uint8))
        bufs.append((dev, torch.empty(len(d) + 64, dtype=torch.int32, device="cuda:0")))
    torch.cuda.synchronize()
    for rounds in range(2):  # (a slot is free again once its result has been taken)
        order = list(range(len(docs))) if rounds == 0 else list(reversed(range(len(docs))))
        for slot, i in enumerate(order):
            ctx.stage1_queue(bufs[i][0].data_ptr(), len(docs[i]), bufs[i][1].data_ptr(), bufs[i][1].numel(), slot, ndjson=nd[i])
        ctx.stage1_wait()
        for slot, i in enumerate(order):
            ok, n = ctx.stage1_result(slot, len(docs[i]))
            ok_ref, pos_ref = O.stage1(docs[i], nd[i])
            assert ok == ok_ref, (rounds, i)
            if ok_ref:
                assert n == len(pos_ref), (rounds, i)
                got = bufs[i][1][:n].cpu().numpy().view(np.uint32)
                assert np.array_equal(got, pos_ref), (rounds, i)
    with pytest.raises(Exception):
        ctx.stage1_queue(bufs[0][0].data_ptr(), len(docs[0]), bufs[0][1].data_ptr(), bufs[0][1].numel(), 64)
