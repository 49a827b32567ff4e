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


def cut_blocks(reader, block_size=BLOCK_SIZE, pool=None):
    """The block cutter of ParseNDStream (simdjson_amd64.go:155-176): `block_size` bytes, then on to the end of the
    current line; the last block is whatever is left.  Yields non-empty blocks whose concatenation is the stream:
    bytes objects, or -- with `pool`, a queue of recycled bytearrays like the reference's tmpPool -- bytearrays
    the consumer puts back into the pool when it is done with them."""
    if isinstance(reader, io.RawIOBase) or not hasattr(reader, "readline") or not hasattr(reader, "readinto"):
        reader = io.BufferedReader(reader, buffer_size=max(block_size, 1 << 16))

    while True:
        buf = None
        if pool is not None:
            try:
                buf = pool.get_nowait()
            except queue.Empty:
                buf = None
        if buf is None:
            buf = bytearray(block_size)
        elif len(buf) != block_size:
            del buf[block_size:]
            buf.extend(bytes(block_size - len(buf)))
        view = memoryview(buf)
        got = 0
        while got < block_size:  # like bufio: short reads of the underlying stream are not the end of it
            n = reader.readinto(view[got:])
            if not n:
                break
            got += n
        view.release()
        if got == 0:
            return
        if got == block_size:  # a full block: finish the record it ends in
            buf.extend(reader.readline())
        else:
            del buf[got:]
        yield buf if pool is not None else bytes(buf)
        if got < block_size:
            return


def parse_nd_stream(reader, block_size=BLOCK_SIZE, inflight=2, device=0, reuse=None):
    """Generator over the ParsedJson of every block, in stream order.

    Mirrors `ParseNDStream(r, res, reuse)`: a block that fails to parse raises `ParseError` after all earlier
    blocks have been delivered and ends the stream (the reference sends `Stream{Error: ...}` and closes `res`);
    normal exhaustion of the generator stands for the final `Stream{Error: io.EOF}`.  `reuse`: an optional
    `queue.Queue`-like object the consumer puts finished ParsedJson values into; like the reference's `reuse`
    channel it is polled without blocking and the Tape / Strings capacity of what it returns is recycled.
    """
    inflight = max(1, int(inflight))
    contexts = queue.SimpleQueue()
    made = []
    for _ in range(inflight):
        c = Context(device)
        made.append(c)
        contexts.put(c)

    blocks = queue.SimpleQueue()  # recycled input buffers (tmpPool, simdjson_amd64.go:129-131)

    def work(block):
        c = contexts.get()
        old = None
        if reuse is not None:
            try:
                old = reuse.get_nowait()  # `select { case v := <-reuse: ... default: }`, simdjson_amd64.go:181-190
            except queue.Empty:
                old = None
        try:
            return c.parse(block, ndjson=True, copy_strings=True, reuse=old)  # pj.copyStrings = true, :180
        finally:
            contexts.put(c)
            blocks.put(block)  # Message was copied out of it

    pending = collections.deque()
    try:
        with concurrent.futures.ThreadPoolExecutor(max_workers=inflight) as pool:
            for block in cut_blocks(reader, block_size, pool=blocks):
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
