"""The reference's lookups on a finished tape, restated on (Tape, Strings.B, Message) arrays -- the checker of the device
queries sjhip_find_path / sjhip_count_where_path / sjhip_project_keys (test infrastructure, like oracle/).

  find_path      Iter.FindElement (parsed_json.go:833-865) -> Object.FindPath (parsed_object.go:256-313)
  project_keys   Object.ForEach(fn, onlyKeys) (parsed_object.go:142-196)
  element_is     what Iter.StringBytes / Int / Uint / Float / Bool return (parsed_json.go:560-749) compared with a value

Pinned by the tables of the reference's own tests (tests/test_query_walk.py: TestObject_FindPath
parsed_object_test.go:10-132, TestObject_ForEach :134-240)."""
import struct

MASK = 0x00FFFFFFFFFFFFFF
STRINGBUFBIT = 0x0080000000000000
NOT_FOUND = 0xFFFFFFFFFFFFFFFF
NOT_OBJECT = 0xFFFFFFFFFFFFFFFE
OP_EXISTS, OP_EQ_STRING, OP_EQ_INT, OP_EQ_UINT, OP_EQ_FLOAT, OP_EQ_BOOL, OP_IS_NULL = range(7)


class Walk:
    def __init__(self, tape, strings, message):
        self.t = [int(x) for x in tape]
        self.s = bytes(strings)
        self.m = bytes(message)

    def string_at(self, i):  # stringByteAt
        off, ln = self.t[i] & MASK, self.t[i + 1]
        if off & STRINGBUFBIT:
            off &= STRINGBUFBIT - 1
            return self.s[off:off + ln]
        return self.m[off:off + ln]

    def records(self):
        """index of the open root word of every record"""
        out, i = [], 0
        while i < len(self.t):
            assert chr(self.t[i] >> 56) == "r"
            out.append(i)
            i = self.t[i] & MASK
        return out

    def skip(self, v):
        tag = chr(self.t[v] >> 56)
        if tag in "{[":
            return self.t[v] & MASK
        return v + 2 if tag in '"lud' else v + 1

    def find_path(self, root, path):
        """root: index of a record's open root word -> tape index of the element's value / NOT_FOUND / NOT_OBJECT"""
        v = root + 1
        if chr(self.t[v] >> 56) != "{":
            return NOT_OBJECT  # "type %q found before object was found"
        seg = 0
        end = (self.t[v] & MASK) - 1
        i = v + 1
        while i < end:
            val = i + 2
            if self.t[i + 1] == len(path[seg]) and self.string_at(i) == path[seg]:
                if seg + 1 == len(path):
                    return val
                if chr(self.t[val] >> 56) != "{":
                    return NOT_OBJECT  # "value of key %v is not an object"
                end = (self.t[val] & MASK) - 1
                i = val + 1
                seg += 1
                continue
            i = self.skip(val)
        return NOT_FOUND

    def project_keys(self, root, keys):
        """-> [(key number, tape index of the value)] in document order, at most len(keys) entries"""
        v = root + 1
        out = []
        if chr(self.t[v] >> 56) != "{":
            return out
        end = (self.t[v] & MASK) - 1
        i = v + 1
        while i < end and len(out) < len(keys):
            val = i + 2
            name = self.string_at(i)
            if name in keys:
                out.append((keys.index(name), val))
            i = self.skip(val)
        return out

    def element_is(self, v, op, want=None):
        tag = chr(self.t[v] >> 56)
        raw = self.t[v + 1] if tag in "lud" else 0
        as_f = lambda: struct.unpack("<d", struct.pack("<Q", raw))[0]
        as_i = lambda: raw - (1 << 64) if raw >= 1 << 63 else raw
        if op == OP_EXISTS:
            return True
        if op == OP_EQ_STRING:
            return tag == '"' and self.string_at(v) == want
        if op == OP_EQ_BOOL:
            return (tag == "t" and bool(want)) or (tag == "f" and not want)
        if op == OP_IS_NULL:
            return tag == "n"
        if op == OP_EQ_INT:  # Iter.Int
            if tag == "l":
                return as_i() == want
            if tag == "u":
                return raw <= (1 << 63) - 1 and raw == want
            if tag == "d":
                d = as_f()
                if d > 2.0 ** 63 or d < -(2.0 ** 63):
                    return False
                return (-(1 << 63) if d >= 2.0 ** 63 else int(d)) == want
            return False
        if op == OP_EQ_UINT:  # Iter.Uint
            if tag == "u":
                return raw == want
            if tag == "l":
                return as_i() >= 0 and raw == want
            if tag == "d":
                # `if v > math.MaxUint64` -- the constant converts to the float64 2^64, so exactly 2^64 passes and uint64(v)
                # is the amd64 conversion's result for it: 0 (parsed_json.go:685-692)
                d = as_f()
                if d != d or d < 0.0 or d > 2.0 ** 64:
                    return False
                return (0 if d >= 2.0 ** 64 else int(d)) == want
            return False
        if op == OP_EQ_FLOAT:  # Iter.Float
            if tag == "d":
                return as_f() == want
            if tag == "l":
                return float(as_i()) == want
            if tag == "u":
                return float(raw) == want
            return False
        raise ValueError(op)
