"""The N>1 path of ParseND on CPU: two gloo ranks parse the two shards of an NDJSON document with the host
replay of the stage-2 functions (csrc/host_selftest.cpp -- the same SJ_HD code the kernels run), exchange
their (tape_len, strings_len) with all_gather, and the concatenation must equal the oracle's tape of the
whole document bit for bit.  Also unit-tests the cut placement."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

import fixtures
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simdjson-go_amd"))


def _selftest():
    import __graft_entry__ as G
    lib = C.CDLL(G.build_selftest())
    szp, u64pp, u8pp = C.POINTER(C.c_size_t), C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(C.c_uint8))
    lib.sj_selftest_parse_shard.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, u64pp,
                                            szp, u8pp, szp, szp, szp]
    lib.sj_selftest_parse_shard.restype = C.c_int
    lib.sj_selftest_trim.argtypes = [C.c_char_p, C.c_size_t, szp, szp]
    lib.sj_selftest_free.argtypes = [C.c_void_p]
    return lib


def _host_callbacks(lib, copy_strings):
    flags = 1 | (2 if copy_strings else 0)
    state = {}

    def trim(b):
        off, ln = C.c_size_t(0), C.c_size_t(0)
        lib.sj_selftest_trim(b, len(b), C.byref(off), C.byref(ln))
        return off.value, ln.value

    def run(window, tb, sb, mb):
        tape, strs = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint8)()
        tl, sl, mo, ml = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        rc = lib.sj_selftest_parse_shard(window, len(window), flags, tb, sb, mb, C.byref(tape), C.byref(tl), C.byref(strs),
                                         C.byref(sl), C.byref(mo), C.byref(ml))
        if rc != 0:
            from sjhip.api import ParseError
            raise ParseError(f"host replay rc {rc}", rc)
        t = np.ctypeslib.as_array(tape, shape=(tl.value,)).copy()
        s = np.ctypeslib.as_array(strs, shape=(max(sl.value, 1),))[: sl.value].copy()
        lib.sj_selftest_free(tape)
        lib.sj_selftest_free(strs)
        return t, s

    def begin(window):
        state["w"] = window
        t, s = run(window, 0, 0, 0)
        return len(t), len(s)

    def finish(tb, sb, mb):
        return run(state["w"], tb, sb, mb)

    return trim, begin, finish


def _nd_docs():
    park = fixtures.load("parking-citations")
    lines = park.split(b"\n")
    yield b"\n".join(lines[:40]) + b"\n"
    yield b"  \n" + b"\n\n".join(lines[:7]) + b"\n\n \n"              # blank lines, leading / trailing space
    yield b'{"a":"x\\ny","b":[1,2.5e3,{"c":null}]}\n[1,2]\n{"k":"\\u00e9\\ud83d\\ude00"}'
    yield b'{"only":"one record"}'


def _bad_docs():
    """(document, expected code): one shard of two is invalid; the other rank must not hang in the collective."""
    good = b'{"a":[1,2,{"b":"c"}]}\n' * 6
    yield good + b'{"a":"unterminated\n' + good[:-1], 1       # stage 1 (rank 1: the cut falls behind the 3rd record)
    yield b'{"x":"ctrl \x01 char"}\n' + good * 2, 1           # stage 1 on rank 0
    yield good + b'{"a":[1,2}\n{"b":}\n' + good, 2            # stage 2
    yield b'{"a":tru}\n' + good * 2 + b'"unterminated', 1     # stage 2 on rank 0 and stage 1 on rank 1: stage 1 wins


def _worker(rank, world, port, copy_strings, q, exchange="gloo"):
    import torch.distributed as dist
    from sjhip import ndshard
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    lib = _selftest()
    trim, begin, finish = _host_callbacks(lib, copy_strings)
    out = []

    def gather(vals):
        box = [None] * world
        dist.all_gather_object(box, tuple(int(x) for x in vals))
        return box

    mb = None
    if exchange == "shm":  # the node-local mailbox (ndshard.ShmMailbox) in place of the collective: same control flow, same results
        mb = ndshard.open_mailbox(rank, world, barrier=dist.barrier, tag=str(port))
        assert mb is not None
        gather = mb.gather  # noqa: F811

    # an invalid shard on one rank: every rank raises the same ShardError, nobody stays behind in a collective
    for doc, want in _bad_docs():
        try:
            ndshard.parse_shard(doc, rank, world, trim, begin, finish, gather, copy_strings)
            got = 0
        except ndshard.ShardError as e:
            got = e.code
        codes = [None] * world
        dist.all_gather_object(codes, got)
        assert codes == [want] * world, (doc[:40], codes, want)
    for doc in _nd_docs():
        tape, strings, tb, sb = ndshard.parse_shard(doc, rank, world, trim, begin, finish, gather, copy_strings)
        pieces = [None] * world
        dist.all_gather_object(pieces, (tape.tobytes(), strings.tobytes()))
        out.append(pieces)
    if rank == 0:
        q.put(out)
    dist.barrier()
    if mb:
        mb.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("copy_strings,exchange", [(True, "gloo"), (False, "gloo"), (True, "shm")])
def test_two_rank_gloo_merge_equals_oracle(copy_strings, exchange):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, copy_strings, q, exchange)) for r in range(2)]
    for p in procs:
        p.start()
    results = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for doc, pieces in zip(_nd_docs(), results):
        tape = np.frombuffer(b"".join(t for t, _ in pieces), dtype=np.uint64)
        strings = np.frombuffer(b"".join(s for _, s in pieces), dtype=np.uint8)
        ref = O.parse(doc, ndjson=True, copy_strings=copy_strings)
        assert np.array_equal(tape, ref.tape)
        assert np.array_equal(strings, ref.strings)


def test_record_cuts():
    from sjhip import ndshard
    doc = b'{"a":1}\n{"b":2}\n{"c":3}\n'
    for n in (1, 2, 3, 5, 8):
        cuts = ndshard.record_cuts(doc, n)
        assert len(cuts) == n and cuts[0][0] == 0 and cuts[-1][1] == len(doc)
        assert b"".join(doc[a:b] for a, b in cuts) == doc
        for a, b in cuts[:-1]:
            assert b == len(doc) or doc[b - 1:b] == b"\n"           # every cut follows a newline
    assert ndshard.bases_from_sizes([(5, 2), (0, 0), (7, 1)]) == [(0, 0), (5, 2), (5, 2)]


def test_mailbox_exchange_order_and_timeout():
    """ShmMailbox alone: values (negative ones too) arrive in rank order over many exchanges of alternating parity, a rank that
    never arrives fails the waiting rank with a ShardError instead of a hang, and a stale segment of the same name is replaced."""
    import threading
    from sjhip import ndshard
    tag = "t%d" % os.getpid()
    a = ndshard.ShmMailbox("sjhip_mb_%s_2" % tag, 0, 2, create=True, timeout=20.0)
    b = ndshard.ShmMailbox("sjhip_mb_%s_2" % tag, 1, 2, create=False, timeout=20.0)
    got = {}

    def run(mb, r):
        got[r] = [mb.gather((r * 100 + k, -k, 1 << 40)) for k in range(200)]
    ts = [threading.Thread(target=run, args=(m, r)) for r, m in enumerate((a, b))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for r in (0, 1):
        assert got[r] == [[(k, -k, 1 << 40), (100 + k, -k, 1 << 40)] for k in range(200)]
    a.timeout = 0.2
    with pytest.raises(ndshard.ShardError):
        a.gather((1,))  # rank 1 does not come
    b.close()
    a.close()
    c = ndshard.ShmMailbox("sjhip_mb_%s_2" % tag, 0, 2, create=True)  # (the name is free again / a leftover is replaced)
    c.close()
