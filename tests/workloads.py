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
