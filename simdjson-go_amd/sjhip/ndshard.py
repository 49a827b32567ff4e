"""ParseND over several GPUs: the host-side logic of the sharded path.

An NDJSON document is cut at record boundaries into one shard per rank.  Every rank parses its shard
as an ordinary ND document (exactly what the reference's ParseNDStream does with its 10 MiB blocks,
simdjson_amd64.go:156-192); the merged ParsedJson is the concatenation of the shard tapes and
Strings.B once every index stored in a shard's tape is rebased by where the shard begins in the
merged Tape / Strings.B / Message.  These three offsets are exclusive prefix sums over the preceding
shards: the only data exchanged is one (tape_len, strings_len) pair per rank (an all_gather of 16
bytes, RCCL over xGMI on the GPUs, gloo in the CPU tests) plus each rank's return code, so that an invalid
shard fails the parse on every rank instead of leaving the others in a collective.

The functions here are pure host logic (no device code) and are exercised on the CPU by
tests/test_ndshard_gloo.py with the host replay standing in for the kernels.
"""
from typing import Callable, List, Sequence, Tuple

import numpy as np

WS = b" \t\n\v\f\r"


def record_cuts(data: bytes, n_shards: int) -> List[Tuple[int, int]]:
    """Byte ranges [start, end) of the shards.  A cut is placed right after the first raw newline at or
    after k*len/n: a raw newline never lies inside a string of a valid document (it is a stage-1 error
    there, find_quote_mask_and_bits_amd64.s:67-80) and inside a record it is a stage-2 error
    (startContinue only accepts it at root level, stage2_build_tape_amd64.go:196-221), so every raw
    newline of a valid ND document separates records.  Ranges may be empty."""
    n = len(data)
    cuts = [0]
    for k in range(1, n_shards):
        target = max(cuts[-1], (n * k) // n_shards)
        j = data.find(b"\n", target)
        cuts.append(n if j < 0 else j + 1)
    cuts.append(n)
    return [(cuts[i], cuts[i + 1]) for i in range(n_shards)]


def bases_from_sizes(sizes: Sequence[Tuple[int, int]]) -> List[Tuple[int, int]]:
    """Exclusive prefix sums of the gathered (tape_len, strings_len) pairs."""
    out, t, s = [], 0, 0
    for tl, sl in sizes:
        out.append((t, s))
        t += tl
        s += sl
    return out


class ShardError(Exception):
    """Raised on EVERY rank when any shard of a sharded ParseND fails.  `code` follows the C ABI (1 = stage 1,
    2 = stage 2, ...); `ranks` lists the failing ranks.  Stage 1 takes precedence over stage 2 like in
    parseMessage (parse_json_amd64.go:97-105,123-126)."""

    def __init__(self, code, ranks):
        from .api import ERR_STAGE1, ERR_STAGE2
        msg = {1: ERR_STAGE1, 2: ERR_STAGE2}.get(code, f"sjhip error {code}")
        super().__init__(f"{msg} (shard(s) {ranks})")
        self.code = code
        self.ranks = ranks


def _agree(codes):
    """The verdict all ranks reach from the gathered per-rank return codes (0 = ok)."""
    bad = [r for r, c in enumerate(codes) if c != 0]
    if not bad:
        return
    code = 1 if 1 in codes else codes[bad[0]]
    raise ShardError(code, bad)


def _code_of(exc):
    c = getattr(exc, "code", None)
    return c if isinstance(c, int) and c != 0 else -1


def parse_shard(data: bytes, rank: int, world: int, trim: Callable, begin: Callable, finish: Callable,
                all_gather: Callable, copy_strings: bool = True):
    """Runs one rank's part of a sharded ParseND.

    trim(bytes) -> (off, len)                              bytes.TrimSpace
    begin(shard_bytes) -> (tape_len, strings_len)          stage 1 + measure (0, 0 for an empty shard)
    all_gather(tuple of ints) -> list of tuples            one per rank, in rank order (called TWICE per parse,
                                                           by every rank, whatever happens locally)
    finish(tape_base, strings_base, msg_base) -> (tape, strings)

    A failure of `begin` or `finish` on one rank (ParseError: invalid shard, too big, HIP error) never leaves
    the other ranks blocked in a collective: the local return code travels with the sizes (first exchange) and
    alone (second exchange), and every rank raises the same ShardError after the exchange.

    Returns (tape, strings, tape_base, strings_base); concatenating the ranks' tapes / strings in rank
    order gives the merged ParsedJson, whose Message is TrimSpace(data)."""
    g_off, _ = trim(data)
    start, end = record_cuts(data, world)[rank]
    shard = data[start:end]
    off, ln = trim(shard)
    window = shard[off:off + ln]
    return run_shard(rank, world, ln == 0, lambda: begin(window), finish, all_gather, start + off - g_off)


def run_shard(rank: int, world: int, empty: bool, begin: Callable, finish: Callable, all_gather: Callable, msg_base: int):
    """The control flow of one rank once its shard is known (parse_shard above; bench.py calls it with the shard
    already resident on the device): phase 1, exchange of (tape_len, strings_len, return code), bases, phase 2,
    exchange of the return codes.  begin() -> (tape_len, strings_len); finish(tape_base, strings_base, msg_base)."""
    sizes, rc = (0, 0), 0
    if not empty:
        try:
            sizes = begin()
        except Exception as e:  # noqa: BLE001 -- the code is exchanged, the error re-raised on all ranks
            rc = _code_of(e)
    gathered = [tuple(g) for g in all_gather((int(sizes[0]), int(sizes[1]), int(rc)))]
    _agree([g[2] if len(g) > 2 else 0 for g in gathered])
    tape_base, strings_base = bases_from_sizes([g[:2] for g in gathered])[rank]
    tape, strings, rc2 = np.empty(0, np.uint64), np.empty(0, np.uint8), 0
    if not empty:
        try:
            tape, strings = finish(tape_base, strings_base, msg_base)
        except Exception as e:  # noqa: BLE001
            rc2 = _code_of(e)
    _agree([(tuple(g) + (0,))[0] for g in all_gather((int(rc2),))])
    return tape, strings, tape_base, strings_base


class ShmMailbox:
    """The exchange of a sharded ParseND between the ranks of ONE node without a collective launch: a POSIX shared-memory
    segment with one 128-byte slot per rank and exchange parity (two buffers: a rank can be at most one exchange ahead of a
    rank that is still reading, see gather()).  gather(vals) has the signature parse_shard / run_shard expect of all_gather.

    Why: the data the ranks exchange is 24 bytes per rank; through torch.distributed it costs a tensor, two launches and a
    synchronisation per call (RCCL) or a TCP round (gloo) -- tens of microseconds next to a shard parse of ~150 us.  Here a
    rank stores its values and then its sequence number (x86 total store order; the readers poll the sequence numbers), about a
    microsecond when the ranks arrive together.  RCCL / gloo stay the fallback (across nodes, or where /dev/shm is not shared)
    and bench.py reports which one it used.  A rank that does not arrive within `timeout` seconds fails the exchange with a
    ShardError(-1) on the ranks that waited (nobody spins forever)."""
    SLOT_WORDS = 16  # 128 bytes: [0] sequence number, [1] count, [2 ..] values
    MAX_VALS = SLOT_WORDS - 2

    def __init__(self, name: str, rank: int, world: int, create: bool, timeout: float = 30.0):
        from multiprocessing import shared_memory
        self.rank, self.world, self.timeout, self.name = rank, world, timeout, name
        size = 2 * world * self.SLOT_WORDS * 8
        if create:
            try:  # a segment a crashed run left behind
                old = shared_memory.SharedMemory(name=name)
                old.close()
                old.unlink()
            except FileNotFoundError:
                pass
            self.shm = shared_memory.SharedMemory(name=name, create=True, size=size)
            self.shm.buf[:size] = bytes(size)
        else:
            self.shm = shared_memory.SharedMemory(name=name)
        self.owner = create
        self.words = self.shm.buf.cast("Q")  # (a flat view of 64-bit words: plain Python indexing, ~50 ns per access)
        self.seq = 0

    def gather(self, vals):
        """all_gather of a short tuple of ints (each below 2^63, may be negative: stored as two's complement)."""
        import time
        vals = [int(v) & 0xFFFFFFFFFFFFFFFF for v in vals]
        assert len(vals) <= self.MAX_VALS
        self.seq += 1
        seq, par = self.seq, self.seq & 1
        # Two buffers are enough: a rank enters exchange k+2 (which reuses the buffer of exchange k) only after it has seen
        # EVERY rank's sequence number k+1, and a rank publishes k+1 only after it has finished reading exchange k.
        w, sw = self.words, self.SLOT_WORDS
        base = (par * self.world + self.rank) * sw
        w[base + 1] = len(vals)
        for k, v in enumerate(vals):
            w[base + 2 + k] = v
        w[base] = seq  # published last
        out = [None] * self.world
        t0 = None
        for r in range(self.world):
            b = (par * self.world + r) * sw
            spins = 0
            while w[b] != seq:
                spins += 1
                if spins & 0x3FF == 0:
                    now = time.perf_counter()
                    t0 = t0 or now
                    if now - t0 > self.timeout:
                        raise ShardError(-1, [r])
            out[r] = tuple(v - (1 << 64) if v >= 1 << 63 else v for v in w[b + 2:b + 2 + w[b + 1]])
        return out

    def close(self):
        try:
            self.words.release()
            self.words = None
            self.shm.close()
            if self.owner:
                self.shm.unlink()
        except Exception:  # noqa: BLE001
            pass


def open_mailbox(rank: int, world: int, barrier: Callable = None, tag: str = None, timeout: float = 30.0):
    """One ShmMailbox per job: rank 0 creates the segment, `barrier()` (any collective of the job, called once at set-up),
    the others attach.  The name is derived from the rendezvous (MASTER_PORT) so that every rank of a torchrun job computes
    the same one.  Returns None where shared memory is not available (the caller keeps its all_gather)."""
    import os
    name = "sjhip_mb_%s_%d" % (tag or os.environ.get("MASTER_PORT", "0"), world)
    try:
        mb = ShmMailbox(name, rank, world, create=True, timeout=timeout) if rank == 0 else None
        if barrier:
            barrier()
        if rank != 0:
            mb = ShmMailbox(name, rank, world, create=False, timeout=timeout)
        if barrier:
            barrier()
        return mb
    except Exception:  # noqa: BLE001 -- no /dev/shm, name clash, ...: the collective is the fallback
        return None


def device_callbacks(ctx, copy_strings=True):
    """begin / finish / trim bound to a sjhip.Context (the shard is uploaded with torch)."""
    import ctypes as C

    import torch

    from . import _lib
    L = _lib.lib()
    flags = 1 | (2 if copy_strings else 0)
    keep = {}

    def trim(b):
        a = np.frombuffer(b, dtype=np.uint8)
        off, ln = C.c_size_t(0), C.c_size_t(0)
        L.sjhip_trim_space(a.ctypes.data if a.size else None, a.size, C.byref(off), C.byref(ln))
        return off.value, ln.value

    def begin(window):
        dev = torch.device("cuda", ctx.device)
        d = torch.empty(len(window) + 256, dtype=torch.uint8, device=dev)
        d[:len(window)].copy_(torch.frombuffer(bytearray(window), dtype=torch.uint8))
        torch.cuda.synchronize(dev)
        keep["d"] = d
        tl, sl = C.c_size_t(0), C.c_size_t(0)
        ctx._check(L.sjhip_parse_shard_begin(ctx._h, C.c_void_p(d.data_ptr()), len(window), flags, C.byref(tl), C.byref(sl)))
        keep["sizes"] = (tl.value, sl.value)
        return tl.value, sl.value

    def finish(tape_base, strings_base, msg_base):
        ctx._check(L.sjhip_parse_shard_finish(ctx._h, tape_base, strings_base, msg_base))
        return ctx.fetch(*keep["sizes"])

    return trim, begin, finish
