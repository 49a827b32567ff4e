"""Hand-derived byte vectors for Serializer.Serialize, format version 3, CompressNone (parsed_serialize.go:200-431).

The reference's tests hold no serialized bytes (they pin the round trip only, parsed_serialize_test.go:220-340) and its
string de-duplication is keyed by Go's per-process random memhash (:836-869), so the bytes of an arbitrary document are
not reproducible.  For the documents below they ARE determined by the format alone, whatever the hash seed:
  * distinct strings are never merged (indexString compares the bytes, :845-850) and always appended (:852-856);
  * a string equal to the string indexed immediately before it is always merged (same bytes -> same slot, and nothing
    has replaced the slot in between).
Every vector was written out by hand from the format comment (:201-236) and the encoding loop (:283-341); the derivation
is next to it.  tests/test_oracle_serialize.py replays them against the oracle (de-duplicating form),
tests/test_gpu_serialize.py against the device (sjhip_serialize_ex with SJHIP_SER_DEDUP, and without it where no string
repeats).

Framing (:376-431), every block = one mode byte (0 = uncompressed, encBlock :791-798) + the raw bytes:
    03                      serializedVersion (:39)
    uvarint(n)              n = 1 + len(msg block) + len(tags block) + len(values block) + bytes of the 8 varints below it
    uvarint(len(Tape))
    00 00                   strings: uncompressed size 0, an empty block (:398-401; v3 keeps all strings in the message)
    uvarint(len(stringBuf)) uvarint(len(msg block)) msg block      = 00 + the de-duplicated strings
    uvarint(rawTags)        uvarint(len(tags block)) tags block    = 00 + one tag byte per tape ENTRY (not per word)
    uvarint(rawValues)      uvarint(len(values block)) values block = 00 + little-endian u64 values
Values per entry (:283-341): root / '{' / '[' -> payload - own index (wraps for the closing root, :325-328); string ->
offset in stringBuf, length; 'l' 'u' 'd' -> the value word; 'd' with a non-zero flag payload -> tag 'e', then the tag
word itself and the value word (:313-320); '}' ']' 't' 'f' 'n' -> nothing.
"""


def _le(*words):
    return b"".join((w & 0xFFFFFFFFFFFFFFFF).to_bytes(8, "little") for w in words)


VECTORS = []

# ---- 1. [1] -----------------------------------------------------------------------------------------------------------
# tape (6 words): 0 r|6   1 [|5   2 l|0   3 1   4 ]|1   5 r|0        (stage2_build_tape_amd64_test.go encoding)
# entries: r -> 6-0 = 6;  [ -> 5-1 = 4;  l -> 1;  ] -> nothing;  r -> 0-5 = 0xff..fb
# tags "r[l]r" (5) -> block 00 + 5 bytes (6);  values 4 x 8 = 32 (0x20) -> block 33 (0x21);  msg block = 00 (1), stringBuf 0
# n = 1 + 1 + 6 + 33 + 8 one-byte varints (0, 1, 5, 6, 32, 33, 0, 6) = 49 = 0x31
VECTORS.append(dict(
    name="[1]", doc=b"[1]", ndjson=False, repeats=False,
    stream=bytes.fromhex("03 31 06 00 00 00 01 00 05 06 00") + b"r[l]r" + bytes.fromhex("20 21 00") + _le(6, 4, 1, -5)))

# ---- 2. {"a":"b"} -------------------------------------------------------------------------------------------------------
# tape (8): 0 r|8  1 {|7  2 "|off  3 1  4 "|off  5 1  6 }|1  7 r|0     (the string payloads do not matter: the bytes do)
# entries: r -> 8;  { -> 7-1 = 6;  "a" -> new: offset 0, len 1;  "b" -> differs from "a": offset 1, len 1;  } -> -;  r -> 0-7
# tags r{""}r (6) -> block 7;  values 7 x 8 = 56 (0x38) -> block 57 (0x39);  stringBuf "ab" (2) -> msg block 00 61 62 (3)
# n = 1 + 3 + 7 + 57 + 8 = 76 = 0x4c
VECTORS.append(dict(
    name='{"a":"b"}', doc=b'{"a":"b"}', ndjson=False, repeats=False,
    stream=bytes.fromhex("03 4c 08 00 00 02 03 00") + b"ab" + bytes.fromhex("06 07 00") + b'r{""}r' + bytes.fromhex("38 39 00") +
    _le(8, 6, 0, 1, 1, 1, -7)))

# ---- 3. a float that overflowed an integer carries FloatOverflowedInteger (parse_number.go:36-135) -> tag 'e' -------------
# [18446744073709551616,-1.5]: 2^64 does not fit uint64 -> 'd' | flag 1, value 0x43f0000000000000;  -1.5 -> 'd' | 0, 0xbff8...
# tape (8): 0 r|8  1 [|7  2 d|1  3 43f0..  4 d|0  5 bff8..  6 ]|1  7 r|0
# entries: r -> 8;  [ -> 6;  d|1 -> tag 'e', the tag word 0x6400000000000001, then 0x43f0000000000000;  d|0 -> 0xbff8..;  ] -> -;  r -> -7
# tags r[ed]r (6) -> block 7;  values 6 x 8 = 48 (0x30) -> block 49 (0x31);  no strings: msg block 00
# n = 1 + 1 + 7 + 49 + 8 = 66 = 0x42
VECTORS.append(dict(
    name="float with flag", doc=b"[18446744073709551616,-1.5]", ndjson=False, repeats=False,
    stream=bytes.fromhex("03 42 08 00 00 00 01 00 06 07 00") + b"r[ed]r" + bytes.fromhex("30 31 00") +
    _le(8, 6, 0x6400000000000001, 0x43F0000000000000, 0xBFF8000000000000, -7)))

# ---- 4. a two-record ND message; the second "a" directly follows the first in indexing order: always merged ---------------
# {"a":1}\n{"a":[true,null]}   tape (18): 0 r|8  1 {|7  2 "  3 1  4 l  5 1  6 }|1  7 r|0
#                                         8 r|18  9 {|17  10 "  11 1  12 [|16  13 t  14 n  15 ]|12  16 }|9  17 r|8   (ndjson_test.go:47-209 root chain)
# entries: r 8; { 6; "a" -> 0, 1; l -> 1; } -; r -> 0-7;   r -> 18-8 = 10; { -> 17-9 = 8; "a" -> merged: 0, 1; [ -> 16-12 = 4; t n ] } -; r -> 8-17 = -9
# tags r{"l}rr{"[tn]}r (15 = 0x0f) -> block 16 (0x10);  values 12 x 8 = 96 (0x60) -> block 97 (0x61);  stringBuf "a" -> msg block 00 61 (2)
# n = 1 + 2 + 16 + 97 + 8 = 124 = 0x7c
VECTORS.append(dict(
    name="two ND records, repeated key", doc=b'{"a":1}\n{"a":[true,null]}', ndjson=True, repeats=True,
    stream=bytes.fromhex("03 7c 12 00 00 01 02 00") + b"a" + bytes.fromhex("0f 10 00") + b'r{"l}rr{"[tn]}r' + bytes.fromhex("60 61 00") +
    _le(8, 6, 0, 1, 1, -7, 10, 8, 0, 1, 4, -9)))

# ---- 5. two-byte varints: [true x 200] ------------------------------------------------------------------------------------
# tape (204): 0 r|204  1 [|203  2..201 t  202 ]|1  203 r|0
# entries: r -> 204;  [ -> 203-1 = 202;  r -> 0-203
# tags r[ t*200 ]r = 204 -> uvarint cc 01, block 205 -> cd 01;  values 3 x 8 = 24 (0x18), block 25 (0x19);  msg block 00
# varints: 0 (1) + len(msg block)=1 (1) + 204 (2) + 205 (2) + 24 (1) + 25 (1) + len(stringBuf)=0 (1) + len(Tape)=204 (2) = 11
# n = 1 + 1 + 205 + 25 + 11 = 243 -> f3 01
VECTORS.append(dict(
    name="[true x200]: two-byte varints", doc=b"[" + b",".join([b"true"] * 200) + b"]", ndjson=False, repeats=False,
    stream=bytes.fromhex("03 f3 01 cc 01 00 00 00 01 00 cc 01 cd 01 00") + b"r[" + b"t" * 200 + b"]r" + bytes.fromhex("18 19 00") +
    _le(204, 202, -203)))
