"""ParseNDStream (simdjson_amd64.go:101-216) on top of libsjhip's sjhip_stream_* (csrc/stream_api.hip).

The reference reads the stream in 10 MiB blocks, extends every block to the end of its last record, parses the
blocks concurrently ((GOMAXPROCS+1)/2 at a time), each as an independent NDJSON document with every string copied,
and delivers the results in stream order; the first error ends the stream (a clean end is reported as io.EOF).

The pipeline itself lives in the library: pinned input blocks (the reader reads straight into them), one context and
HIP stream per block in flight, pinned result buffers, ordered delivery, first-error cut-off, devices used round
robin.  This module is the thin host side: the block cutter (`ReadFull` + `ReadBytes('\\n')`, :155-176) writing into
the acquired pinned block, and the copy of each result into the caller's arrays (`reuse` recycling).
"""
import ctypes as C
import io
import queue

import numpy as np

from . import _lib
from .api import ERR_STAGE1, ERR_STAGE2, ParsedJson, ParseError

BLOCK_SIZE = 10 << 20  # tmpSize, simdjson_amd64.go:127
STREAM_FULL, STREAM_EMPTY, STREAM_CLOSED = 6, 7, 8


def _buffered(reader, block_size):
    if isinstance(reader, io.RawIOBase) or not hasattr(reader, "readline") or not hasattr(reader, "readinto"):
        return io.BufferedReader(reader, buffer_size=max(block_size, 1 << 16))
    return reader


def read_block(reader, view, block_size):
    """One block of the cutter (simdjson_amd64.go:155-176) into `view` (a writable buffer of >= block_size bytes):
    `block_size` bytes (short reads of the underlying stream are not its end), then -- if the block is full -- the
    rest of the current line, returned separately.  -> (bytes in view, rest of the line or b'')."""
    got = 0
    while got < block_size:
        n = reader.readinto(view[got:block_size])
        if not n:
            break
        got += n
    tail = reader.readline() if got == block_size else b""
    return got, tail


def cut_blocks(reader, block_size=BLOCK_SIZE):
    """The block cutter alone, as bytes objects (tests; the stream itself cuts straight into pinned blocks)."""
    reader = _buffered(reader, block_size)
    buf = bytearray(block_size)
    while True:
        got, tail = read_block(reader, memoryview(buf), block_size)
        if got == 0:
            return
        yield bytes(buf[:got]) + tail
        if got < block_size:
            return


class Stream:
    """sjhip_stream: acquire / submit / next / release (include/sjhip.h)."""

    def __init__(self, block_size=BLOCK_SIZE, slots=0, first_device=0, n_devices=0, reserve=None):
        self._L = _lib.lib()
        self.block_size = block_size
        # room for the record a block ends in, so that the usual case needs no second allocation
        cap = block_size + (block_size // 8 + (64 << 10) if reserve is None else reserve)
        self._h = self._L.sjhip_stream_create(first_device, n_devices, cap, slots, 0)
        if not self._h:
            raise ParseError("sjhip_stream_create failed (no usable device?)", 3)
        self.slots = self._L.sjhip_stream_slots(self._h)
        self._held = False  # take(view=True): the delivered block has not been released yet

    def release_held(self):
        """Hands the block of the last take(view=True) back to the stream (its views must not be used any more)."""
        if self._held and self._h:
            self._L.sjhip_stream_release(self._h)
        self._held = False

    def close(self):
        if self._h:
            self.release_held()
            self._L.sjhip_stream_destroy(self._h)
            self._h = None

    def in_flight(self):
        return self._L.sjhip_stream_in_flight(self._h)

    def set_filter(self, key, value):
        """From now on every block delivers only the records whose root object has `key` with the string value `value`
        (sjhip_filter_where on the device-resident tape); ParsedJson.records counts them."""
        k, v = bytes(key), bytes(value)
        rc = self._L.sjhip_stream_set_filter(self._h, k, len(k), v, len(v))
        if rc:
            raise ParseError(f"sjhip_stream_set_filter: {self._L.sjhip_stream_last_error(self._h).decode()}", rc)

    def ready(self):
        """True if the oldest outstanding block has finished (take() would not wait)."""
        return bool(self._L.sjhip_stream_ready(self._h))

    def feed(self, reader):
        """Reads one block from `reader` into a pinned block and submits it.  -> 'full' (take a result first),
        'eof' (nothing left; nothing submitted), 'last' (submitted, the reader is exhausted) or 'more'."""
        L = self._L
        ptr, cap = C.c_void_p(), C.c_size_t()
        rc = L.sjhip_stream_acquire(self._h, C.byref(ptr), C.byref(cap))
        if rc == STREAM_FULL:
            return "full"
        if rc:
            raise ParseError(f"sjhip_stream_acquire: {rc}", rc)
        view = memoryview((C.c_uint8 * cap.value).from_address(ptr.value)).cast("B")
        got, tail = read_block(reader, view, self.block_size)
        if got == 0:  # `if len(tmp) > 0 { ... } else { tmpPool.Put(tmp) }`: an exhausted reader submits nothing
            L.sjhip_stream_cancel(self._h)
            return "eof"
        if got + len(tail) > cap.value:  # a record longer than the reserve: a larger pinned block for this slot
            rc = L.sjhip_stream_grow(self._h, got, got + len(tail), C.byref(ptr))
            if rc:
                raise ParseError(f"sjhip_stream_grow: {L.sjhip_stream_last_error(self._h).decode()}", rc)
            view = memoryview((C.c_uint8 * (got + len(tail))).from_address(ptr.value)).cast("B")
        if tail:
            view[got:got + len(tail)] = tail
        rc = L.sjhip_stream_submit(self._h, got + len(tail))
        if rc:
            raise ParseError(f"sjhip_stream_submit: {rc}", rc)
        return "more" if got == self.block_size else "last"

    def take(self, reuse=None, view=False):
        """The oldest outstanding result as a ParsedJson (arrays owned by the caller; `reuse` recycles capacity), or
        None when nothing is outstanding.  Raises ParseError for the block that ends the stream.
        `view=True`: no copy out of the stream's pinned memory -- Tape / Strings / Message are read-only views of the
        block, valid until the next take() (or release_held() / close()), which hands the block back."""
        L = self._L
        self.release_held()
        r = _lib.StreamResult()
        rc = L.sjhip_stream_next(self._h, C.byref(r))
        if rc == STREAM_EMPTY or rc == STREAM_CLOSED:
            return None
        if rc:
            msg = {1: ERR_STAGE1, 2: ERR_STAGE2}.get(rc, L.sjhip_stream_last_error(self._h).decode())
            raise ParseError("parsing input: %s" % msg, rc)
        if view:
            from .api import _pinned_view
            self._held = True
            pj = ParsedJson(_pinned_view(r.message, r.message_len, C.c_uint8, np.uint8), _pinned_view(r.tape, r.tape_len, C.c_uint64, np.uint64),
                            _pinned_view(r.strings, r.strings_len, C.c_uint8, np.uint8))
            pj.records = int(r.records)
            pj.device = int(r.device)
            pj._owner = self
            return pj
        try:
            tl, sl = r.tape_len, r.strings_len
            tape_buf = reuse._tape_buf if reuse is not None and reuse._tape_buf.size >= tl else np.empty(tl, np.uint64)
            str_buf = reuse._str_buf if reuse is not None and reuse._str_buf.size >= sl else np.empty(sl, np.uint8)
            if tl:
                C.memmove(tape_buf.ctypes.data, r.tape, tl * 8)
            if sl:
                C.memmove(str_buf.ctypes.data, r.strings, sl)
            msg = C.string_at(r.message, r.message_len) if r.message_len else b""
        finally:
            L.sjhip_stream_release(self._h)
        pj = ParsedJson(msg, tape_buf[:tl], str_buf[:sl], tape_buf, str_buf)
        pj.records = int(r.records)
        pj.device = int(r.device)  # the GPU that parsed the block
        return pj


def parse_nd_stream(reader, block_size=BLOCK_SIZE, inflight=0, device=0, reuse=None, n_devices=1, where=None, view=False):
    """Generator over the ParsedJson of every block, in stream order.

    Mirrors `ParseNDStream(r, res, reuse)`: a block that fails to parse raises `ParseError` after all earlier
    blocks have been delivered and ends the stream (the reference sends `Stream{Error: ...}` and closes `res`);
    normal exhaustion of the generator stands for the final `Stream{Error: io.EOF}`.  `reuse`: an optional
    `queue.Queue`-like object the consumer puts finished ParsedJson values into; like the reference's `reuse`
    channel it is polled without blocking and the Tape / Strings capacity of what it returns is recycled.
    `inflight`: blocks in flight (0 = three per device); `n_devices` = 0 uses every visible GPU, round robin.
    `view=True`: every ParsedJson is a set of read-only views of the stream's pinned memory, valid until the generator
    is resumed (the Go shim's ParseNDStreamInPlace): no copy of the 2.4 result bytes per input byte.
    """
    reader = _buffered(reader, block_size)
    st = Stream(block_size, slots=max(0, int(inflight)), first_device=device, n_devices=n_devices)
    if where is not None:  # (key, value): only the matching records of every block cross PCIe
        st.set_filter(*where)

    def old():
        if reuse is None:
            return None
        try:
            return reuse.get_nowait()  # `select { case v := <-reuse: ... default: }`, simdjson_amd64.go:181-190
        except queue.Empty:
            return None

    try:
        more = True
        while more:
            # Finished blocks are delivered before the next (possibly blocking) read of the input: the reference sends
            # every Stream value as soon as its block is parsed (simdjson_amd64.go:193-203), so on a slow or unbounded
            # reader results must not wait for the slots to fill up.
            while st.ready():
                pj = st.take(old(), view=view)
                if pj is None:
                    return
                yield pj
                st.release_held()  # (view=True: the consumer is done with the block)
            state = st.feed(reader)
            if state == "full":  # every slot holds a block: deliver the oldest one
                pj = st.take(old(), view=view)
                if pj is None:
                    return
                yield pj
                st.release_held()
                continue
            more = state == "more"
        while True:
            pj = st.take(old(), view=view)
            if pj is None:
                return
            yield pj
            st.release_held()
    finally:
        st.close()
