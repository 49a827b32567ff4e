"""GPU: the distributed control flow of the sharded ParseND on the one-GPU box: two ranks (gloo) share cuda:0 and run the
real sjhip_parse_shard_begin / all_gather / sjhip_parse_shard_finish sequence that bench.py's `ndjson` leg runs with
one rank per GPU over RCCL; the concatenation must equal the oracle's ParseND of the whole document, and an invalid
shard must fail the parse on both ranks (no rank left behind in a collective)."""
import os
import socket
import sys

import numpy as np
import pytest

import fixtures
import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _docs():
    park = fixtures.load("parking-citations")
    good = park * 12                                               # ~4.5 MB, 12 000 records
    yield good, 0
    yield good[: len(good) // 2 + 5000] + b'{"broken":"unterminated\n' + good[len(good) // 2 + 5000:], 1   # stage 1, rank 1's shard
    yield b'{"a":[1,2}\n' + good, 2                                 # stage 2, rank 0's shard
    yield b'\n\n' + b'{"k":"\\u00e9\\ud83d\\ude00","n":[1.5e3,-7,null]}\n' * 3000 + b' \n', 0


def _worker(rank, world, port, q, exchange="gloo"):
    for p in (os.path.join(ROOT, "simdjson-go_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    torch.cuda.init()
    import sjhip
    from sjhip import ndshard
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ctx = sjhip.Context(0)
    out = []

    def gather(vals):
        box = [None] * world
        dist.all_gather_object(box, tuple(int(x) for x in vals))
        return box

    mb = None
    if exchange == "shm":  # the node-local mailbox instead of the collective (what bench.py uses when it can)
        mb = ndshard.open_mailbox(rank, world, barrier=dist.barrier, tag=str(port))
        assert mb is not None
        gather = mb.gather  # noqa: F811

    for rep in range(2):  # (the second round of a context runs phase 1 without a synchronisation of its own for stage 1)
      for doc, want in _docs():
        if rep == 1 and want != 0:
            continue
        for copy in (True, False):
            trim, begin, finish = ndshard.device_callbacks(ctx, copy)
            try:
                tape, strings, tb, sb = ndshard.parse_shard(doc, rank, world, trim, begin, finish, gather, copy)
                code = 0
            except ndshard.ShardError as e:
                tape, strings, code = np.zeros(0, np.uint64), np.zeros(0, np.uint8), e.code
            pieces = [None] * world
            dist.all_gather_object(pieces, (code, tape.tobytes(), strings.tobytes()))
            out.append(pieces)
    if rank == 0:
        q.put(out)
    dist.barrier()
    if mb:
        mb.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["gloo", "shm"])
def test_two_ranks_on_one_gpu(exchange):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    procs = [mpctx.Process(target=_worker, args=(r, 2, port, q, exchange)) for r in range(2)]
    for p in procs:
        p.start()
    results = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    k = 0
    for rep in range(2):
      for doc, want in _docs():
        if rep == 1 and want != 0:
            continue
        for copy in (True, False):
            pieces = results[k]
            k += 1
            codes = [c for c, _, _ in pieces]
            assert codes == [want, want], (want, codes)
            if want == 0:
                ref = O.parse(doc, ndjson=True, copy_strings=copy)
                assert ref.rc == 0
                tape = np.frombuffer(b"".join(t for _, t, _ in pieces), dtype=np.uint64)
                strings = np.frombuffer(b"".join(s for _, _, s in pieces), dtype=np.uint8)
                assert np.array_equal(tape, ref.tape) and np.array_equal(strings, ref.strings), copy
