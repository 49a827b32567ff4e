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
