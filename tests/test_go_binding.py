"""The Go binding (simdjson-go_amd/go/simdjson_hip.go) cannot be compiled here (no Go toolchain).  What can be checked
without one: that it declares no identifier the reference's package already declares under `-tags hip` (the round-3 shim
re-declared `internalParsedJson`), that it uses only C symbols and constants include/sjhip.h declares, and that
INTEGRATION.md lists the tag edits the check assumes.  The reference's identifier inventory is a committed fixture
(tests/golden/go_reference_symbols.json, written by tools/check_go_collisions.py --write-inventory)."""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_go_collisions as G  # noqa: E402


def _inventory():
    with open(G.INVENTORY) as f:
        return json.load(f)["files"]


def _shim():
    with open(G.SHIM) as f:
        return f.read()


def test_shim_fits_into_the_reference_package():
    assert G.check(_shim(), _inventory()) == []


def test_checker_catches_redeclarations():
    bad = _shim() + "\ntype internalParsedJson struct {\n\tParsedJson\n\tcopyStrings bool\n}\n" \
                    "func (pj *internalParsedJson) parseMessage(msg []byte, ndjson bool) error { return nil }\n"
    problems = G.check(bad, _inventory())
    assert any("`internalParsedJson` is declared in simdjson_hip.go and parsed_json.go" in p for p in problems)
    assert any("internalParsedJson.parseMessage" in p and "parse_json_amd64.go" in p for p in problems)
    # without the tag edits both stock backends collide with the shim
    problems = G.check(_shim(), _inventory(), edits={})
    assert any("`Parse` is declared in simdjson_hip.go and simdjson_amd64.go" in p for p in problems)
    assert any("simdjson_other.go" in p for p in problems)


def test_inventory_is_current():
    if not os.path.isdir(G.REFERENCE):  # (the GPU box has no reference checkout)
        return
    fresh = G.make_inventory()
    assert fresh == _inventory(), "run tools/check_go_collisions.py --write-inventory"


def test_scanner_on_go_constructs():
    src = '''//go:build hip && cgo
package p
// type NotADecl struct
import "C"
const (
	A = iota // comment with ( paren
	B, C2 = 1, 2
)
var x, y int
var (
	z = map[string]int{"type Foo": 1}
)
type (
	T1 struct {
		a, b int
		*Emb
	}
	T2 = int
)
type S struct { inline int }
func F() { type local int; var q = `func Raw()`; _ = q }
func (s *S) M(a int) (int, error) { return 0, nil }
func (S) N() {}
func G[T any](v T) {}
'''
    got = G.scan_go(src)
    assert got["constraint"] == "hip && cgo"
    assert got["decls"] == ["A", "B", "C2", "x", "y", "z", "T1", "T2", "S", "F", "S.M", "S.N", "G"]
    assert got["structs"]["T1"] == ["a", "b", "Emb"]
    assert G.eval_constraint("(!amd64 || appengine || !gc || noasm) && !hip", {"amd64", "gc", "hip"}) is False
    assert G.eval_constraint("(!amd64 || appengine || !gc || noasm) && !hip", {"arm64", "gc"}) is True
    assert G.file_active("x_amd64.go", "", {"arm64"}) is False


def test_shim_uses_only_declared_c_symbols():
    with open(os.path.join(ROOT, "include", "sjhip.h")) as f:
        header = f.read()
    shim = G.strip_go(_shim())
    funcs = set(re.findall(r"\b(sjhip_[a-z0-9_]+)\s*\(", header))
    consts = set(re.findall(r"#define\s+(SJHIP_[A-Z0-9_]+)", header))
    types = set(re.findall(r"typedef struct (sjhip_[a-z_]+)", header)) | set(re.findall(r"\}\s*(sjhip_[a-z_]+);", header))
    used = set(re.findall(r"\bC\.(sjhip_[a-z0-9_]+|SJHIP_[A-Z0-9_]+)", shim))
    assert used, "no cgo references found"
    unknown = {u for u in used if u not in funcs | consts | types}
    assert not unknown, unknown


def _top_level_args(text, open_at):
    """number of top-level arguments of the call whose '(' is at text[open_at]"""
    depth, args, seen = 0, 0, False
    for ch in text[open_at:]:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
            if depth == 0:
                return args + (1 if seen else 0)
        elif ch == "," and depth == 1:
            args += 1
            seen = False
            continue
        if depth >= 1 and not ch.isspace() and not (depth == 1 and ch == "("):
            seen = True
    raise AssertionError("unbalanced call")


def test_cgo_calls_pass_as_many_arguments_as_the_header_declares():
    """Every C.sjhip_*(...) call of the shim against the prototype in include/sjhip.h (cgo rejects a wrong count; without a
    Go toolchain this is the part of that check that can be done here)."""
    with open(os.path.join(ROOT, "include", "sjhip.h")) as f:
        header = re.sub(r"/\*.*?\*/", " ", f.read(), flags=re.S)
    arity = {}
    for name, params in re.findall(r"\b(sjhip_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", header):
        params = params.strip()
        arity[name] = 0 if params in ("", "void") else params.count(",") + 1
    assert len(arity) > 50
    shim = G.strip_go(_shim())
    calls = [(m.group(1), m.end() - 1) for m in re.finditer(r"\bC\.(sjhip_[a-z0-9_]+)\s*\(", shim)]
    assert len(calls) > 30
    wrong = [(n, _top_level_args(shim, at), arity[n]) for n, at in calls if _top_level_args(shim, at) != arity[n]]
    assert not wrong, wrong
    assert _top_level_args("f(a, g(b, c), []int{1, 2})", 1) == 3 and _top_level_args("f()", 1) == 0 and _top_level_args("f( x )", 1) == 1


def test_shim_brackets_balance():
    """(no gofmt here) with comments, strings and runes stripped, every kind of bracket of the shim closes"""
    src = G.strip_go(_shim())
    for a, b in ("{}", "()", "[]"):
        depth = 0
        for ch in src:
            if ch == a:
                depth += 1
            elif ch == b:
                depth -= 1
                assert depth >= 0, (a, b)
        assert depth == 0, (a, b, depth)


def test_integration_md_lists_the_tag_edits():
    with open(os.path.join(ROOT, "INTEGRATION.md")) as f:
        text = f.read()
    for name, constraint in G.TAG_EDITS.items():
        assert name in text and f"//go:build {constraint}" in text, name
    assert "check_go_collisions.py" in text
