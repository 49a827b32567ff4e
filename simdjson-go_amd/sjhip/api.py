"""Host-side mirror of the reference API for the Parse()/ParseND() path.

Reference interface mirrored here:
  simdjson_amd64.go:37  SupportedCPU()        -> supported()
  simdjson_amd64.go:66  Parse(b, reuse, opts) -> parse(b, reuse=None, copy_strings=True)
  simdjson_amd64.go:82  ParseND(...)          -> parse_nd(...)
  options.go:13         WithCopyStrings(bool) -> copy_strings keyword
  parsed_json.go:64     ParsedJson{Message, Tape, Strings} -> ParsedJson
Errors follow parse_json_amd64.go:81,93 and simdjson_amd64.go:43 (same messages).
"""
import ctypes as C

import numpy as np

from . import _lib

FLAG_NDJSON = 1
FLAG_COPY_STRINGS = 2
FLAG_KEY_FLAGS = 4  # the parse leaves the key flags MarshalJSON needs (include/sjhip.h)

ERR_STAGE1 = "Failed to find all structural indices for stage 1"
ERR_STAGE2 = "Bad parsing while executing stage 2"
ERR_NODEVICE = "Host CPU does not meet target specs"  # kept verbatim from the reference


class ParseError(Exception):
    def __init__(self, msg, code):
        super().__init__(msg)
        self.code = code


def supported() -> bool:
    return bool(_lib.lib().sjhip_supported())


def _pinned_view(ptr, count, ctype, dtype):
    """numpy array over `count` elements of library-owned host memory (read-only)"""
    if not count or not ptr:
        return np.empty(0, dtype=dtype)
    a = np.frombuffer((ctype * count).from_address(ptr), dtype=dtype)
    a.flags.writeable = False
    return a


class Context:
    """One per concurrent parse: owns a HIP stream and recycled device arenas
    (the role of `reuse *ParsedJson`, simdjson_amd64.go:46-51)."""

    def __init__(self, device: int = 0):
        L = _lib.lib()
        self._h = L.sjhip_ctx_create(device)
        if not self._h:
            raise ParseError(ERR_NODEVICE, 3)
        self.device = device

    def close(self):
        if self._h:
            _lib.lib().sjhip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self) -> str:
        return _lib.lib().sjhip_last_error(self._h).decode()

    def device_bytes(self) -> int:
        """bytes of device memory the context's arenas hold right now (they only grow; see trim)"""
        return int(_lib.lib().sjhip_ctx_device_bytes(self._h))

    def trim(self):
        """give every arena back (after an unusually large message); the next parse allocates what it needs"""
        self._check(_lib.lib().sjhip_ctx_trim(self._h))

    def input_block(self, nbytes):
        """A pinned host block of the context as a writable uint8 array of `nbytes` bytes: read the input straight into
        it (file.readinto) and hand it to parse() -- the copy to the device then runs at the pinned rate."""
        p = _lib.lib().sjhip_input_block(self._h, nbytes)
        if not p:
            raise ParseError(f"sjhip_input_block: {self.last_error()}", -1)
        return np.frombuffer((C.c_uint8 * nbytes).from_address(p), dtype=np.uint8)

    def set_stream(self, stream_ptr):
        _lib.lib().sjhip_ctx_set_stream(self._h, C.c_void_p(stream_ptr or 0))

    def _check(self, rc):
        if rc == 0:
            return
        if rc == 1:
            raise ParseError(ERR_STAGE1, rc)
        if rc == 2:
            raise ParseError(ERR_STAGE2, rc)
        if rc == 3:
            raise ParseError(ERR_NODEVICE, rc)
        raise ParseError(f"sjhip error {rc}: {self.last_error()}", rc)

    # ---- stage 1 ---------------------------------------------------------------------------
    def stage1(self, data, ndjson=False):
        """findStructuralIndices on a host buffer -> (ok, uint32 positions)."""
        a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        cap = a.size + 64
        pos = np.empty(cap, dtype=np.uint32)
        n = C.c_size_t(0)
        ok = C.c_int(0)
        rc = _lib.lib().sjhip_stage1(self._h, a.ctypes.data if a.size else None, a.size, int(ndjson),
                                     pos.ctypes.data, cap, C.byref(n), C.byref(ok))
        self._check(rc)
        return bool(ok.value), pos[: n.value].copy()

    def stage1_device(self, d_msg_ptr, length, d_pos_ptr, pos_cap, ndjson=False):
        n = C.c_size_t(0)
        ok = C.c_int(0)
        rc = _lib.lib().sjhip_stage1_device(self._h, C.c_void_p(d_msg_ptr), length, int(ndjson),
                                            C.c_void_p(d_pos_ptr), pos_cap, C.byref(n), C.byref(ok))
        self._check(rc)
        return bool(ok.value), n.value

    STAGE1_QUEUE_SLOTS = 64

    def stage1_queue(self, d_msg_ptr, length, d_pos_ptr, pos_cap, slot, ndjson=False):
        """sjhip_stage1_device without the synchronisation: the launch goes behind what the stream holds; its result is
        taken with stage1_result(slot, length) after stage1_wait()."""
        self._check(_lib.lib().sjhip_stage1_device_queue(self._h, C.c_void_p(d_msg_ptr), length, int(ndjson),
                                                         C.c_void_p(d_pos_ptr), pos_cap, int(slot)))

    def stage1_wait(self):
        self._check(_lib.lib().sjhip_stage1_device_wait(self._h))

    def stage1_result(self, slot, length):
        n = C.c_size_t(0)
        ok = C.c_int(0)
        self._check(_lib.lib().sjhip_stage1_device_result(self._h, int(slot), length, C.byref(n), C.byref(ok)))
        return bool(ok.value), n.value

    def stage1_time(self, d_msg_ptr, length, d_pos_ptr, pos_cap, iters, ndjson=False):
        ms = C.c_float(0)
        rc = _lib.lib().sjhip_stage1_time(self._h, C.c_void_p(d_msg_ptr), length, int(ndjson), C.c_void_p(d_pos_ptr),
                                          pos_cap, iters, C.byref(ms))
        self._check(rc)
        return ms.value

    # ---- whole parse -----------------------------------------------------------------------
    def parse(self, data, ndjson=False, copy_strings=True, reuse=None, view=False, key_flags=False):
        """Parse / ParseND.  `reuse`: a ParsedJson whose Tape / Strings capacity is recycled (the reference's
        `reuse *ParsedJson`, simdjson_amd64.go:46-51): its arrays are overwritten.
        `view=True`: Tape / Strings are read-only views of the context's pinned result block (sjhip_fetch_view) --
        no copy into Python-owned arrays; like a recycled ParsedJson they are overwritten by the next parse on this
        context.  `key_flags=True`: marshal_json() of this result is going to be called (SJHIP_FLAG_KEY_FLAGS)."""
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        tl, sl, mo, ml = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        flags = (FLAG_NDJSON if ndjson else 0) | (FLAG_COPY_STRINGS if copy_strings else 0) | (FLAG_KEY_FLAGS if key_flags else 0)
        L = _lib.lib()
        rc = L.sjhip_parse(self._h, a.ctypes.data if a.size else None, a.size, flags, C.byref(tl), C.byref(sl),
                           C.byref(mo), C.byref(ml))
        self._check(rc)
        msg = a[mo.value: mo.value + ml.value]  # (a view: no copy of the message on the parse path)
        if view:
            tp, sp = C.c_void_p(), C.c_void_p()
            self._check(L.sjhip_fetch_view(self._h, C.byref(tp), C.byref(sp)))
            tape = _pinned_view(tp.value, tl.value, C.c_uint64, np.uint64)
            strings = _pinned_view(sp.value, sl.value, C.c_uint8, np.uint8)
            pj = ParsedJson(msg, tape, strings)
            pj._owner = self  # the views live in this context's pinned memory
            return pj
        tape_buf = reuse._tape_buf if reuse is not None and reuse._tape_buf.size >= tl.value else \
            np.empty(tl.value, dtype=np.uint64)
        str_buf = reuse._str_buf if reuse is not None and reuse._str_buf.size >= sl.value else \
            np.empty(sl.value, dtype=np.uint8)
        rc = L.sjhip_fetch(self._h, tape_buf.ctypes.data, str_buf.ctypes.data)
        self._check(rc)
        return ParsedJson(msg, tape_buf[:tl.value], str_buf[:sl.value], tape_buf, str_buf)

    def parse_device(self, d_msg_ptr, length, ndjson=False, copy_strings=True, key_flags=False):
        tl, sl = C.c_size_t(0), C.c_size_t(0)
        flags = (FLAG_NDJSON if ndjson else 0) | (FLAG_COPY_STRINGS if copy_strings else 0) | (FLAG_KEY_FLAGS if key_flags else 0)
        rc = _lib.lib().sjhip_parse_device(self._h, C.c_void_p(d_msg_ptr), length, flags, C.byref(tl), C.byref(sl))
        self._check(rc)
        return tl.value, sl.value

    # ---- many documents, one launch set (include/sjhip.h: sjhip_parse_batch) -----------------------------------
    def parse_batch(self, docs, fetch=True):
        """Parses the documents of `docs` (bytes-like, host memory) as ONE packed ND message: document i is root i of
        the returned ParsedJson (Message = b"": every string is copied).  One invalid document fails the batch."""
        arrs = [np.frombuffer(d, dtype=np.uint8) if not isinstance(d, np.ndarray) else d for d in docs]
        n = len(arrs)
        ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data if a.size else None for a in arrs])
        lens = (C.c_size_t * max(n, 1))(*[a.size for a in arrs])
        tl, sl = C.c_size_t(0), C.c_size_t(0)
        self._check(_lib.lib().sjhip_parse_batch(self._h, ptrs, lens, n, FLAG_COPY_STRINGS, C.byref(tl), C.byref(sl)))
        if not fetch:
            return tl.value, sl.value
        tape, strings = self.fetch(tl.value, sl.value)
        return ParsedJson(b"", tape, strings)

    def parse_batch_device(self, d_buf_ptr, offs, lens):
        """The same with the documents resident in one device buffer (offs[i], lens[i]); the result stays on the device."""
        n = len(offs)
        o = (C.c_size_t * max(n, 1))(*[int(x) for x in offs])
        ln = (C.c_size_t * max(n, 1))(*[int(x) for x in lens])
        tl, sl = C.c_size_t(0), C.c_size_t(0)
        self._check(_lib.lib().sjhip_parse_batch_device(self._h, C.c_void_p(d_buf_ptr), o, ln, n, FLAG_COPY_STRINGS,
                                                        C.byref(tl), C.byref(sl)))
        return tl.value, sl.value

    # ---- queries on the device-resident result (include/sjhip.h: sjhip_count_where / sjhip_filter_where) --------
    def count_where(self, key, value):
        """countWhere(key, value, pj) of the reference's tests (ndjson_test.go:421-471) on the device: records whose
        root object has `key` (first occurrence, top level) with the string value `value`."""
        k, v = bytes(key), bytes(value)
        n = C.c_uint64(0)
        self._check(_lib.lib().sjhip_count_where(self._h, k, len(k), v, len(v), C.byref(n)))
        return n.value

    def filter_where(self, key, value, fetch=True):
        """Compacts the matching records into a new (Tape, Strings.B) on the device -- what ParseND returns for the
        document made of the matching lines -- and fetches it.  -> (n_records, ParsedJson or None)"""
        k, v = bytes(key), bytes(value)
        n, tl, sl = C.c_uint64(0), C.c_size_t(0), C.c_size_t(0)
        L = _lib.lib()
        self._check(L.sjhip_filter_where(self._h, k, len(k), v, len(v), C.byref(n), C.byref(tl), C.byref(sl)))
        if not fetch:
            return n.value, None
        tape = np.empty(tl.value, dtype=np.uint64)
        strings = np.empty(sl.value, dtype=np.uint8)
        self._check(L.sjhip_fetch_filtered(self._h, tape.ctypes.data, strings.ctypes.data))
        return n.value, ParsedJson(b"", tape, strings)

    # ---- paths, typed values, key sets (include/sjhip.h: sjhip_find_path / _count_where_path / _project_keys) --------
    PATH_NOT_FOUND = 0xFFFFFFFFFFFFFFFF
    PATH_NOT_OBJECT = 0xFFFFFFFFFFFFFFFE
    OP_EXISTS, OP_EQ_STRING, OP_EQ_INT, OP_EQ_UINT, OP_EQ_FLOAT, OP_EQ_BOOL, OP_IS_NULL = range(7)

    @staticmethod
    def _keys(keys):
        ks = [bytes(k) for k in keys]
        lens = (C.c_uint32 * max(len(ks), 1))(*[len(k) for k in ks])
        return b"".join(ks), lens, len(ks)

    def find_path(self, *path):
        """Iter.FindElement(path...) (parsed_json.go:833-865) on the root of every record of the last parse, on the device.
        -> uint64 array, one entry per record: the tape index of the element's value, PATH_NOT_FOUND or PATH_NOT_OBJECT"""
        blob, lens, n = self._keys(path)
        L = _lib.lib()
        cnt = C.c_size_t(0)
        probe = np.empty(1, dtype=np.uint64)
        L.sjhip_find_path(self._h, blob, lens, n, probe.ctypes.data, 0, C.byref(cnt))  # (no room: only the record count is set)
        out = np.empty(max(cnt.value, 1), dtype=np.uint64)
        self._check(L.sjhip_find_path(self._h, blob, lens, n, out.ctypes.data, out.size, C.byref(cnt)))
        return out[: cnt.value]

    def count_where_path(self, path, op, value=None):
        """records whose element at `path` exists and satisfies op (OP_*): value = bytes for OP_EQ_STRING, an int for
        OP_EQ_INT / OP_EQ_UINT, a float for OP_EQ_FLOAT, a bool for OP_EQ_BOOL"""
        import struct
        blob, lens, n = self._keys(path)
        if op == self.OP_EQ_STRING:
            v = bytes(value)
        elif op == self.OP_EQ_INT:
            v = struct.pack("<q", int(value))
        elif op == self.OP_EQ_UINT:
            v = struct.pack("<Q", int(value))
        elif op == self.OP_EQ_FLOAT:
            v = struct.pack("<d", float(value))
        elif op == self.OP_EQ_BOOL:
            v = b"\x01" if value else b"\x00"
        else:
            v = b""
        buf = C.create_string_buffer(v, max(len(v), 1))
        cnt = C.c_uint64(0)
        self._check(_lib.lib().sjhip_count_where_path(self._h, blob, lens, n, int(op), buf, len(v), C.byref(cnt)))
        return cnt.value

    def project_keys(self, keys):
        """Object.ForEach(fn, onlyKeys) (parsed_object.go:142-196) on the root object of every record, on the device.
        -> uint64 array [records, len(keys)]: key number << 56 | tape index of the value of the j-th delivered member,
        2^64 - 1 where there is none"""
        blob, lens, n = self._keys(keys)
        L = _lib.lib()
        cnt = C.c_size_t(0)
        probe = np.empty(1, dtype=np.uint64)
        L.sjhip_project_keys(self._h, blob, lens, n, probe.ctypes.data, 0, C.byref(cnt))  # (no room: only the record count is set)
        out = np.empty((max(cnt.value, 1), n), dtype=np.uint64)
        self._check(L.sjhip_project_keys(self._h, blob, lens, n, out.ctypes.data, out.shape[0], C.byref(cnt)))
        return out[: cnt.value]

    def serialize(self, fetch=True, dedup=False):
        """Serializer.Serialize (format v3, CompressNone) of the device-resident result of the last parse.
        -> the framed stream as a uint8 array (what the reference's Deserialize reads), or its sizes with fetch=False.
        dedup: de-duplicate the strings like the reference's indexString (the plain form is byte-identical to the
        oracle's stream without de-duplication)."""
        L = _lib.lib()
        tl, vl, sl, n = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        self._check(L.sjhip_serialize_ex(self._h, 1 if dedup else 0, C.byref(tl), C.byref(vl), C.byref(sl), C.byref(n)))
        if not fetch:
            return {"tags": tl.value, "values": vl.value, "strings": sl.value, "stream": n.value}
        out = np.empty(n.value, dtype=np.uint8)
        got = C.c_size_t(0)
        self._check(L.sjhip_fetch_serialized(self._h, out.ctypes.data, out.size, C.byref(got)))
        return out[: got.value]

    def deserialize(self, stream):
        """Serializer.Deserialize of a stream with uncompressed blocks, on the device -> ParsedJson (strings point into
        Message = the string column, like the reference's result)."""
        L = _lib.lib()
        a = np.frombuffer(stream, dtype=np.uint8) if not isinstance(stream, np.ndarray) else stream
        tl, sl, ml = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        self._check(L.sjhip_deserialize(self._h, a.ctypes.data, a.size, C.byref(tl), C.byref(sl), C.byref(ml)))
        tape, strings = self.fetch(tl.value, sl.value)
        msg = np.empty(ml.value, dtype=np.uint8)
        self._check(L.sjhip_fetch_message(self._h, msg.ctypes.data))
        return ParsedJson(msg.tobytes(), tape, strings)

    def marshal_json(self, fetch=True):
        """pj.Iter().MarshalJSON() of the device-resident result of the last parse: compact JSON text, records
        separated by newlines.  -> bytes (or the length with fetch=False)"""
        L = _lib.lib()
        n = C.c_size_t(0)
        self._check(L.sjhip_marshal_json(self._h, C.byref(n)))
        if not fetch:
            return n.value
        out = np.empty(n.value, dtype=np.uint8)
        self._check(L.sjhip_fetch_marshaled(self._h, out.ctypes.data))
        return out.tobytes()

    def fetch(self, tape_len, strings_len):
        tape = np.empty(tape_len, dtype=np.uint64)
        strings = np.empty(strings_len, dtype=np.uint8)
        self._check(_lib.lib().sjhip_fetch(self._h, tape.ctypes.data, strings.ctypes.data))
        return tape, strings


class MultiContext:
    """ParseND over several GPUs in one call (include/sjhip.h: sjhip_multi_*): one shard per entry of `devices`
    (None = every visible device; a device may be listed more than once)."""

    def __init__(self, devices=None):
        L = _lib.lib()
        if devices is None:
            self._h = L.sjhip_multi_create(None, 0)
        else:
            arr = (C.c_int * len(devices))(*devices)
            self._h = L.sjhip_multi_create(arr, len(devices))
        if not self._h:
            raise ParseError(ERR_NODEVICE, 3)
        self.shards = L.sjhip_multi_shards(self._h)

    def close(self):
        if self._h:
            _lib.lib().sjhip_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def parse_nd(self, data, copy_strings=True):
        """ParseND(data): the merged ParsedJson of all shards (bit for bit what one context returns)."""
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        tl, sl, mo, ml = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        L = _lib.lib()
        rc = L.sjhip_parse_nd_multi(self._h, a.ctypes.data if a.size else None, a.size, FLAG_NDJSON | (FLAG_COPY_STRINGS if copy_strings else 0),
                                    C.byref(tl), C.byref(sl), C.byref(mo), C.byref(ml))
        if rc == 1:
            raise ParseError(ERR_STAGE1, rc)
        if rc == 2:
            raise ParseError(ERR_STAGE2, rc)
        if rc:
            raise ParseError(f"sjhip error {rc}: {L.sjhip_multi_last_error(self._h).decode()}", rc)
        tape = np.empty(tl.value, dtype=np.uint64)
        strings = np.empty(sl.value, dtype=np.uint8)
        rc = L.sjhip_fetch_multi(self._h, tape.ctypes.data, strings.ctypes.data)
        if rc:
            raise ParseError(f"sjhip error {rc}: {L.sjhip_multi_last_error(self._h).decode()}", rc)
        return ParsedJson(a[mo.value: mo.value + ml.value].tobytes(), tape, strings)


class ParsedJson:
    """parsed_json.go:64-71: Message / Tape / Strings."""

    __slots__ = ("_msg", "Tape", "Strings", "_tape_buf", "_str_buf", "records", "device", "_owner")

    def __init__(self, message, tape, strings, tape_buf=None, str_buf=None):
        # `message`: bytes, or a uint8 view of the caller's buffer -- the reference's pj.Message ALIASES the input
        # (bytes.TrimSpace, parse_json_amd64.go:55); the bytes object is only made when somebody asks for it
        self._msg = message
        self.Tape = tape
        self.Strings = strings
        self._tape_buf = tape if tape_buf is None else tape_buf  # capacity behind Tape / Strings (reuse)
        self._str_buf = strings if str_buf is None else str_buf
        self.records = 0  # filtered streams: matching records of the block
        self.device = -1  # streams: the GPU that parsed the block
        self._owner = None  # view=True: the Context whose pinned block Tape / Strings alias

    @property
    def Message(self):
        if not isinstance(self._msg, bytes):
            self._msg = self._msg.tobytes()
        return self._msg


_DEFAULT = {}


def _default_ctx(device=0):
    c = _DEFAULT.get(device)
    if c is None:
        c = _DEFAULT[device] = Context(device)
    return c


def parse(b, reuse=None, copy_strings=True, ctx=None, view=False):
    """Parse(b, reuse, WithCopyStrings(copy_strings)) -- simdjson_amd64.go:66."""
    return (ctx or _default_ctx()).parse(b, ndjson=False, copy_strings=copy_strings, reuse=reuse, view=view)


def parse_nd(b, reuse=None, copy_strings=True, ctx=None, view=False):
    """ParseND(b, reuse, ...) -- simdjson_amd64.go:82."""
    return (ctx or _default_ctx()).parse(b, ndjson=True, copy_strings=copy_strings, reuse=reuse, view=view)


def stage1(b, ndjson=False, ctx=None):
    return (ctx or _default_ctx()).stage1(b, ndjson=ndjson)
