"""CPU: the Python restatement of the reference's lookups (tests/query_walk.py, the checker of the device queries) against the
tables of the reference's own tests: TestObject_FindPath (parsed_object_test.go:10-132) and TestObject_ForEach (:134-240),
on the oracle's tape of the same inputs."""
import oracle_lib as O
import query_walk as Q
import tape_reader

FINDPATH_INPUT = b"""{
    "Image":
    {
        "Animated": false,
        "Height": 600,
        "IDs":
        [
            116,
            943,
            234,
            38793
        ],
        "Thumbnail":
        {
            "Height": 125,
            "Url": "http://www.example.com/image/481989943",
            "Width": 100
        },
        "Title": "View from 15th Floor",
        "Width": 800
    },
	"Alt": "Image of city" 
}"""
# (path, want: the value as Python data, or None for wantErr)  parsed_object_test.go:24-73
FINDPATH_CASES = [
    (["Alt"], "Image of city"),
    (["Image", "Animated"], False),
    (["Image", "Thumbnail", "Url"], "http://www.example.com/image/481989943"),
    (["Image", "Height"], 600),
    (["Image", "Thumbnail"], {"Height": 125, "Url": "http://www.example.com/image/481989943", "Width": 100}),
    (["Image", "IDs"], [116, 943, 234, 38793]),
    (["Image", "NonEx"], None),
]
FOREACH_INPUT = b"""{
		"key1": "value1",
		"key2": "value2",
		"key3": "value3",
		"key4": "value4",
		"key5": "value5",
		"key6": "value6",
		"key7": "value7",
		"key8": "value8",
		"key9": "value9",
		"key10": "value10"
	}"""
# (onlyKeys, want)  parsed_object_test.go:155-211 (the all-keys case has no key set: not a projection)
FOREACH_CASES = [
    (["key1", "key3"], {"key1": "value1", "key3": "value3"}),
    (["key1", "key3", "key5", "key7", "key9"], {"key1": "value1", "key3": "value3", "key5": "value5", "key7": "value7", "key9": "value9"}),
    (["key1", "key2", "key3", "key9", "key10"], {"key1": "value1", "key2": "value2", "key3": "value3", "key9": "value9", "key10": "value10"}),
    (["key20"], {}),
]


def value_at(w, v):
    """the element whose value starts at tape index v, as Python data (through the tests' tape reader)"""
    def dec(i):
        x = w.t[i]
        tag = chr(x >> 56)
        if tag == '"':
            return w.string_at(i).decode(), i + 2
        if tag == "l":
            r = w.t[i + 1]
            return (r - (1 << 64) if r >= 1 << 63 else r), i + 2
        if tag == "u":
            return w.t[i + 1], i + 2
        if tag == "d":
            import struct
            return struct.unpack("<d", struct.pack("<Q", w.t[i + 1]))[0], i + 2
        if tag in "tfn":
            return {"t": True, "f": False, "n": None}[tag], i + 1
        e = (x & Q.MASK) - 1
        i += 1
        if tag == "{":
            o = {}
            while i < e:
                k = w.string_at(i).decode()
                o[k], i = dec(i + 2)
            return o, e + 1
        a = []
        while i < e:
            val, i = dec(i)
            a.append(val)
        return a, e + 1
    return dec(v)[0]


def test_find_path_table_of_the_reference():
    for copy in (True, False):
        ref = O.parse(FINDPATH_INPUT, copy_strings=copy)
        assert ref.rc == 0
        w = Q.Walk(ref.tape, ref.strings, FINDPATH_INPUT[ref.msg_off:ref.msg_off + ref.msg_len])
        (root,) = w.records()
        for path, want in FINDPATH_CASES:
            v = w.find_path(root, [p.encode() for p in path])
            if want is None:
                assert v == Q.NOT_FOUND, path
            else:
                assert v < Q.NOT_OBJECT and value_at(w, v) == want, (path, v)
        # the type errors of FindPath / FindElement
        assert w.find_path(root, [b"Alt", b"x"]) == Q.NOT_OBJECT      # "value of key Alt is not an object"
        assert w.find_path(root, [b"Image", b"IDs", b"0"]) == Q.NOT_OBJECT  # arrays are not entered


def test_for_each_only_keys_table_of_the_reference():
    ref = O.parse(FOREACH_INPUT, copy_strings=True)
    assert ref.rc == 0
    w = Q.Walk(ref.tape, ref.strings, FOREACH_INPUT[ref.msg_off:ref.msg_off + ref.msg_len])
    (root,) = w.records()
    for keys, want in FOREACH_CASES:
        kb = [k.encode() for k in keys]
        got = {keys[j]: value_at(w, v) for j, v in w.project_keys(root, kb)}
        assert got == want, keys


def test_typed_comparisons_follow_the_iter_conversions():
    doc = b'{"i":-5,"u":18446744073709551615,"f":2.5,"g":3.0,"s":"a\\\\b","t":true,"n":null,"big":9223372036854775808}'
    ref = O.parse(doc, copy_strings=True)
    assert ref.rc == 0
    w = Q.Walk(ref.tape, ref.strings, doc)
    (root,) = w.records()
    at = lambda k: w.find_path(root, [k])
    assert w.element_is(at(b"i"), Q.OP_EQ_INT, -5) and not w.element_is(at(b"i"), Q.OP_EQ_UINT, 5)
    assert w.element_is(at(b"i"), Q.OP_EQ_FLOAT, -5.0)
    assert w.element_is(at(b"u"), Q.OP_EQ_UINT, 2 ** 64 - 1) and not w.element_is(at(b"u"), Q.OP_EQ_INT, -1)
    assert w.element_is(at(b"g"), Q.OP_EQ_INT, 3) and w.element_is(at(b"f"), Q.OP_EQ_INT, 2)  # int64(2.5) truncates
    assert w.element_is(at(b"s"), Q.OP_EQ_STRING, b"a\\b") and not w.element_is(at(b"s"), Q.OP_EQ_STRING, b"a\\\\b")
    assert w.element_is(at(b"t"), Q.OP_EQ_BOOL, True) and not w.element_is(at(b"t"), Q.OP_EQ_BOOL, False)
    assert w.element_is(at(b"n"), Q.OP_IS_NULL) and not w.element_is(at(b"t"), Q.OP_IS_NULL)
    assert w.element_is(at(b"big"), Q.OP_EQ_UINT, 2 ** 63) and not w.element_is(at(b"big"), Q.OP_EQ_INT, 2 ** 63 - 1)
    # the two float edges the reference's range checks let through (`v > math.MaxInt64` / `v > math.MaxUint64` compare with the
    # float64 constants 2^63 / 2^64): int64(2^63) and uint64(2^64) are the amd64 conversions' results, MinInt64 and 0
    ref = O.parse(b'{"a":9223372036854775808.0,"b":18446744073709551616.0,"c":18446744073709555000.0}', copy_strings=True)
    w = Q.Walk(ref.tape, ref.strings, b"")
    (root,) = w.records()
    assert w.element_is(w.find_path(root, [b"a"]), Q.OP_EQ_INT, -(2 ** 63)) and w.element_is(w.find_path(root, [b"a"]), Q.OP_EQ_UINT, 2 ** 63)
    assert w.element_is(w.find_path(root, [b"b"]), Q.OP_EQ_UINT, 0) and not w.element_is(w.find_path(root, [b"b"]), Q.OP_EQ_INT, 0)
    assert not w.element_is(w.find_path(root, [b"c"]), Q.OP_EQ_UINT, 0)
