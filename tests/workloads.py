"""Deterministic construction of the BASELINE.json configurations."""
import fixtures


def c2_twitter_array(copies=426) -> bytes:
    """configs[1]: '[' + ','.join(copies x twitter.json) + ']' (426 copies = 269 025 391 B = 256.56 MiB)."""
    tw = fixtures.load("twitter")
    return b"[" + b",".join([tw] * copies) + b"]"


def c2_expected_structurals(copies=426) -> int:
    return copies * 55263 + (copies - 1) + 2


def c5_parking_nd(copies=1000) -> bytes:
    """configs[4]: parking-citations.json (1000 records, ends with \\n) concatenated `copies` times."""
    return fixtures.load("parking-citations") * copies


def window_cut_numbers():
    """Numbers whose '.', 'e', exponent sign and last digits fall on either side of byte 32: the kernels parse
    from a 32-byte copy of the head of a number and must notice every way the copy can cut it."""
    out = []
    for a in range(18, 36):          # digits in front of the dot
        for b in (1, 2, 3, 9):        # digits behind it
            for ex in ("", "e5", "e-5", "e+5", "E-185", "e-3"):
                out.append("9" * a + "." + "4" * b + ex)
                out.append("-" + "1" * a + "." + "0" * b + ex)
        for ex in ("e5", "e-5", "e+5", "e-185"):
            out.append("7" * a + ex)
    out += ["1" * 31 + ".", "1" * 31 + "-", "1" * 30 + "e-", "1" * 31 + "e", "1" * 32 + "e", "1" * 31 + "..5", "1" * 30 + ".5.5"]
    return out


def string_boundary_documents():
    """Documents that put the ends of strings, their escapes and the blanks behind their closing quotes on 64-byte chunk
    and 4 KiB unit boundaries: the string kernels work per chunk and per unit, and WithCopyStrings(false) finds a string's
    closing quote by walking back from the next token.  (name, document)"""
    import random
    rnd = random.Random(2024)
    docs = []
    blanks = [b" ", b"\n", b"\t", b"\r", b" \n \t"]
    for pad in list(range(50, 72)) + list(range(4080, 4104, 3)) + [8190, 8192, 8193]:
        for esc in (b"", b"\\n", b"\\u00e9", b"\\ud83d\\ude00", b"\\\\"):
            for gap in (0, 1, 62, 63, 64, 65, 130, 4097):
                # the string's content ends near byte `pad`, `gap` blanks follow its closing quote
                head = b'{"k":"'
                body = b"x" * max(0, pad - len(head) - len(esc) - 1) + esc
                docs.append((f"end@{pad} esc={esc!r} gap={gap}", head + body + b'"' + rnd.choice(blanks)[:1] * gap + b',"n":1}'))
    # long strings across several units, escapes at the unit seams, short strings in between
    for n in (4090, 4096, 9000, 20000):
        for esc in (b"", b"\\t", b"\\u4e2d"):
            s = (b"a" * 37 + esc) * (n // (37 + len(esc)) + 1)
            docs.append((f"long {n} esc={esc!r}", b'["' + s[:n] + b'", "y", "' + esc + b'", "' + b"b" * 70 + b'" ' + b" " * 200 + b', "z"]'))
    # an escaped quote / backslash right in front of the closing quote, at chunk ends
    for pad in (60, 61, 62, 63, 64, 65, 127, 128):
        docs.append((f"escaped quote @{pad}", b'["' + b"q" * pad + b'\\""' + b" " * 70 + b',"' + b"r" * pad + b'\\\\" ]'))
    return docs
