"""sjhip -- Python mirror of the simdjson-go backend API on top of libsjhip (MI355X / gfx950).

Mirrors the reference's backend symbol set (simdjson_other.go:29-76):
    SupportedCPU() -> supported()          Parse(b, reuse, opts...) -> parse(...)
    ParseND(b, reuse, opts...) -> parse_nd(...)    WithCopyStrings(bool) -> copy_strings=...
    ParseNDStream(r, res, reuse) -> parse_nd_stream(reader, ...)   (a generator instead of a channel)
"""
from ._lib import SjhipMissing, lib  # noqa: F401
from .api import (Context, MultiContext, ParsedJson, ParseError, parse, parse_nd, stage1, supported)  # noqa: F401
from .stream import cut_blocks, parse_nd_stream  # noqa: F401
