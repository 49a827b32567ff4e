import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


def bits_lsb_first(s):
    """'0111…' strings in the reference are bits.Reverse64()'d: char j <-> bit j."""
    v = 0
    for j, c in enumerate(s):
        if c == "1":
            v |= 1 << j
    return v
