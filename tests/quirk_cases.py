"""Named cases for the quirks of SURVEY.md A.6 (each with the verdict / bytes the reference's code implies) and the
reference's atom tables (stage2_build_tape_amd64_test.go:195-262) embedded in documents whose grammar is valid
exactly when the atom is.  Shared by the CPU test (oracle) and the GPU test (HIP kernels through the C ABI)."""
import golden_util as GU

# (name, document bytes, ndjson, accepted, expected Strings.B in copy mode or None)
QUIRKS = [
    # Q2: surrogate-pair arithmetic wraps mod 2^32, the low half is not range-checked (parse_string_amd64.s:203-210)
    ("Q2 golden \\udbff\\u1234 -> EF B8 B4 (parse_string_test.go:76-80)", b'["\\udbff\\u1234"]', False, True, b"\xef\xb8\xb4"),
    ("Q2 high + non-surrogate low half", b'["\\ud800\\u0041"]', False, True, None),
    ("Q2 proper pair", b'["\\ud83d\\ude00"]', False, True, "\U0001F600".encode()),
    ("Q2 lone low surrogate is a 3-byte sequence", b'["\\udc00"]', False, True, b"\xed\xb0\x80"),
    ("Q2 lone high surrogate fails", b'["\\ud800"]', False, False, None),
    ("Q2 high surrogate followed by a plain escape fails", b'["\\ud800\\n"]', False, False, None),
    ("Q2 high surrogate, low half cut by the closing quote", b'["\\ud800\\u00"]', False, False, None),
    # Q3: bytes 0x20..0x2f count as hex digit 0 inside \\uXXXX (DATA layout of parse_string_amd64.s:11-37; unpinned)
    ("Q3 '\\u 041' == 'A'", b'["\\u 041"]', False, True, b"A"),
    ("Q3 '\\u//41' == 'A'", b'["\\u//41"]', False, True, b"A"),
    ("Q3 a letter beyond f is not a digit", b'["\\u00g1"]', False, False, None),
    ("Q3 ':' (0x3a) is not a digit", b'["\\u00:1"]', False, False, None),
    # Q4: only buf[0] is checked for a leading zero (parse_number.go:126)
    ("Q4 -00.5 accepted", b"[-00.5]", False, True, None),
    ("Q4 -00e1 accepted", b"[-00e1]", False, True, None),
    ("Q4 00.5 rejected", b"[00.5]", False, False, None),
    ("Q4 -01 rejected (integer path)", b"[-01]", False, False, None),
    ("Q4 '+' inside a number reaches strconv and fails", b"[1+2]", False, False, None),
    ("Q4 exponent sign is fine", b"[1e+2,1E-2]", False, True, None),
    # Q5: float overflow fails the parse, underflow is 0
    ("Q5 1e400 fails (fail60)", b"[1e400]", False, False, None),
    ("Q5 1e+1111 fails", b"[1e+1111]", False, False, None),
    ("Q5 1e-400 underflows to 0", b"[1e-400]", False, True, None),
    ("Q5 largest finite", b"[1.7976931348623157e308]", False, True, None),
    ("Q5 rounds to infinity", b"[1.7976931348623159e308]", False, False, None),
    # Q6: bytes.TrimSpace strips \\v \\f and Unicode spaces at the ends only
    ("Q6 \\v ... \\f around the document", b'\x0b{"a":1}\x0c', False, True, None),
    ("Q6 NBSP / U+2028 / U+3000 around the document", "\u00a0[1]\u2028\u3000".encode(), False, True, None),
    ("Q6 U+0085 in front", b'\xc2\x85[1]', False, True, None),
    ("Q6 \\v inside the document is not whitespace", b'[1,\x0b2]', False, False, None),
    ("Q6 NBSP inside the document is not whitespace", "[1,\u00a02]".encode(), False, False, None),
    # Q7: NUL terminates an atom but not a number (stage2_build_tape_amd64.go:456)
    ("Q7 true\\x00 accepted", b"[true\x00]", False, True, None),
    ("Q7 false\\x00 / null\\x00 accepted", b"[false\x00,null\x00]", False, True, None),
    ("Q7 1\\x00 rejected", b"[1\x00]", False, False, None),
    # ND: one record per line
    ("ND newline inside a container fails", b'{"a":\n1}', True, False, None),
    ("ND the same bytes as a plain document", b'{"a":\n1}', False, True, None),
    ("ND blank lines between records", b'{"a":1}\n\n\n[2]\n', True, True, None),
    ("ND two roots without newline fail", b'{"a":1}[2]', True, False, None),
    ("plain document: a second root fails", b'{"a":1}\n[2]', False, False, None),
]


def atom_documents():
    """(name, document, expected) from the three atom tables: the 8-byte table input is followed by what makes the
    grammar around it valid, so the document is accepted iff the atom check passes."""
    s2 = GU.load("stage2")
    out = []
    for atom in ("true", "false", "null"):
        for r in s2["atom_" + atom]:
            inp = bytes.fromhex(r["input_hex"])
            term = inp[len(atom):len(atom) + 1]
            if term == b",":
                doc = b"[" + inp + b"1]"
            elif term == b"}":
                doc = b'{"a":' + inp
            elif term == b"]":
                doc = b"[" + inp
            else:
                doc = b"[" + inp + b"]"
            out.append((f"{atom} {inp!r}", doc, bool(r["expected"])))
    return out
