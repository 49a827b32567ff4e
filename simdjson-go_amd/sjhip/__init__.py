"""sjhip -- Python mirror of the simdjson-go backend API on top of libsjhip (MI355X / gfx950).

Mirrors the reference's backend symbol set (simdjson_other.go:29-76):
    SupportedCPU() -> supported()          Parse(b, reuse, opts...) -> parse(...)
    ParseND(b, reuse, opts...) -> parse_nd(...)    WithCopyStrings(bool) -> copy_strings=...
"""
from ._lib import SjhipMissing, lib  # noqa: F401
from .api import (Context, ParsedJson, ParseError, parse, parse_nd, stage1, supported)  # noqa: F401
