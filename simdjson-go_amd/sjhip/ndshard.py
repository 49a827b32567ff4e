"""ParseND over several GPUs: the host-side logic of the sharded path.

An NDJSON document is cut at record boundaries into one shard per rank.  Every rank parses its shard
as an ordinary ND document (exactly what the reference's ParseNDStream does with its 10 MiB blocks,
simdjson_amd64.go:156-192); the merged ParsedJson is the concatenation of the shard tapes and
Strings.B once every index stored in a shard's tape is rebased by where the shard begins in the
merged Tape / Strings.B / Message.  These three offsets are exclusive prefix sums over the preceding
shards: the only data exchanged is one (tape_len, strings_len) pair per rank (an all_gather of 16
bytes, RCCL over xGMI on the GPUs, gloo in the CPU tests).

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


def parse_shard(data: bytes, rank: int, world: int, trim: Callable, begin: Callable, finish: Callable,
                all_gather_sizes: Callable, copy_strings: bool = True):
    """Runs one rank's part of a sharded ParseND.

    trim(bytes) -> (off, len)                              bytes.TrimSpace
    begin(shard_bytes) -> (tape_len, strings_len)          stage 1 + measure (0, 0 for an empty shard)
    all_gather_sizes((tape_len, strings_len)) -> list     one pair per rank, in rank order
    finish(tape_base, strings_base, msg_base) -> (tape, strings)
    Returns (tape, strings, tape_base, strings_base); concatenating the ranks' tapes / strings in rank
    order gives the merged ParsedJson, whose Message is TrimSpace(data)."""
    g_off, _ = trim(data)
    start, end = record_cuts(data, world)[rank]
    shard = data[start:end]
    off, ln = trim(shard)
    empty = ln == 0
    window = shard[off:off + ln]
    sizes = (0, 0) if empty else begin(window)
    allsizes = all_gather_sizes(sizes)
    tape_base, strings_base = bases_from_sizes(allsizes)[rank]
    if empty:
        return np.empty(0, np.uint64), np.empty(0, np.uint8), tape_base, strings_base
    msg_base = start + off - g_off
    tape, strings = finish(tape_base, strings_base, msg_base)
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
