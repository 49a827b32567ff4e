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
