"""Minimal tape consumer (the role of the reference's Iter/Interface(), parsed_json.go:196-347)
used by tests to turn a (Tape, Strings, Message) triple into Python objects."""
import struct

MASK = 0x00FFFFFFFFFFFFFF
STRINGBUFBIT = 0x0080000000000000


def to_python(tape, strings, message):
    """Returns the list of root values."""
    tape = [int(x) for x in tape]
    sbuf = bytes(strings)
    msg = bytes(message)

    def string_at(i):
        w, ln = tape[i], tape[i + 1]
        off = w & MASK
        if off & STRINGBUFBIT:
            off &= STRINGBUFBIT - 1
            b = sbuf[off:off + ln]
        else:
            b = msg[off:off + ln]
        return b.decode("utf-8", "surrogatepass")

    def value(i):
        w = tape[i]
        tag = chr(w >> 56)
        if tag == '"':
            return string_at(i), i + 2
        if tag == "l":
            v = tape[i + 1]
            return (v - (1 << 64) if v >= 1 << 63 else v), i + 2
        if tag == "u":
            return tape[i + 1], i + 2
        if tag == "d":
            return struct.unpack("<d", struct.pack("<Q", tape[i + 1]))[0], i + 2
        if tag == "t":
            return True, i + 1
        if tag == "f":
            return False, i + 1
        if tag == "n":
            return None, i + 1
        if tag == "{":
            end = (w & MASK) - 1
            i += 1
            obj = {}
            while i < end:
                k = string_at(i)
                v, i = value(i + 2)
                obj[k] = v
            assert chr(tape[end] >> 56) == "}" and i == end
            return obj, end + 1
        if tag == "[":
            end = (w & MASK) - 1
            i += 1
            arr = []
            while i < end:
                v, i = value(i)
                arr.append(v)
            assert chr(tape[end] >> 56) == "]" and i == end
            return arr, end + 1
        raise ValueError(f"bad tag {tag!r} at {i}")

    out = []
    i = 0
    n = len(tape)
    while i < n:
        w = tape[i]
        assert chr(w >> 56) == "r", f"expected root at {i}"
        nxt = w & MASK
        v, j = value(i + 1)
        assert chr(tape[j] >> 56) == "r" and (tape[j] & MASK) == i and j + 1 == nxt
        out.append(v)
        i = nxt
    return out
