"""ParseNDStream (simdjson_amd64.go:101-216) on top of libsjhip.

The reference reads the stream in 10 MiB blocks, extends every block to the end of its last record, parses the
blocks concurrently ((GOMAXPROCS+1)/2 at a time), each as an independent NDJSON document with every string copied,
and delivers the results in stream order; the first error ends the stream (a clean end is reported as io.EOF).

Here every block in flight has its own `sjhip_ctx` (its own HIP stream and device arenas), so the host-to-device
copy of one block, the kernels of another and the tape read-back of a third overlap; blocks are handed to a small
thread pool (ctypes releases the GIL inside the library) and yielded in submission order.
"""
import collections
import concurrent.futures
import io
import queue

from .api import Context, ParseError

BLOCK_SIZE = 10 << 20  # tmpSize, simdjson_amd64.go:127


def cut_blocks(reader, block_size=BLOCK_SIZE):
    """The block cutter of ParseNDStream (simdjson_amd64.go:155-176): `block_size` bytes, then on to the end of the
    current line; the last block is whatever is left.  Yields non-empty bytes objects whose concatenation is the
    stream."""
    if isinstance(reader, io.RawIOBase) or not hasattr(reader, "readline"):
        reader = io.BufferedReader(reader, buffer_size=max(block_size, 1 << 16))

    def read_full(n):  # like bufio: short reads of the underlying stream are not the end of it
        parts = []
        while n > 0:
            c = reader.read(n)
            if not c:
                break
            parts.append(c)
            n -= len(c)
        return b"".join(parts)

    while True:
        block = read_full(block_size)
        if not block:
            return
        if len(block) == block_size:  # a full block: finish the record it ends in
            block += reader.readline()
        yield block
        if len(block) < block_size:
            return


def parse_nd_stream(reader, block_size=BLOCK_SIZE, inflight=2, device=0, reuse=None):
    """Generator over the ParsedJson of every block, in stream order.

    Mirrors `ParseNDStream(r, res, reuse)`: a block that fails to parse raises `ParseError` after all earlier
    blocks have been delivered and ends the stream (the reference sends `Stream{Error: ...}` and closes `res`);
    normal exhaustion of the generator stands for the final `Stream{Error: io.EOF}`.  `reuse` is accepted for
    signature parity: the device arenas of the contexts are what is recycled here.
    """
    del reuse
    inflight = max(1, int(inflight))
    contexts = queue.SimpleQueue()
    made = []
    for _ in range(inflight):
        c = Context(device)
        made.append(c)
        contexts.put(c)

    def work(block):
        c = contexts.get()
        try:
            return c.parse(block, ndjson=True, copy_strings=True)  # pj.copyStrings = true, simdjson_amd64.go:180
        finally:
            contexts.put(c)

    pending = collections.deque()
    try:
        with concurrent.futures.ThreadPoolExecutor(max_workers=inflight) as pool:
            for block in cut_blocks(reader, block_size):
                pending.append(pool.submit(work, block))
                while len(pending) >= inflight + 1:  # one block cut ahead of the ones being parsed
                    yield pending.popleft().result()
            while pending:
                yield pending.popleft().result()
    except ParseError as e:
        for f in pending:
            f.cancel()
        raise ParseError("parsing input: %s" % e, e.code) from None
    finally:
        for c in made:
            c.close()
