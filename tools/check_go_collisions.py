#!/usr/bin/env python3
"""Does simdjson-go_amd/go/simdjson_hip.go fit into the reference's Go package?  (There is no Go toolchain in this
image, so nobody can compile the binding; this is the part of `go build -tags hip` that can be checked without one.)

    python tools/check_go_collisions.py                      check the shim against the committed inventory
    python tools/check_go_collisions.py --write-inventory    regenerate tests/golden/go_reference_symbols.json from
                                                             /root/reference (top-level identifiers, build
                                                             constraints and struct fields of every .go file)

Checks, for linux/amd64 and linux/arm64, both with `-tags hip` and cgo, after the tag edits INTEGRATION.md lists:
  1. no top-level identifier (func, type, var, const, method as Recv.Name) is declared twice among the reference's
     non-test files whose build constraint holds and the shim  -- the failure of the round-3 shim, which re-declared
     `type internalParsedJson` although parsed_json.go:83 (no build tag) declares it;
  2. the backend symbol set of simdjson_other.go:29-76 is declared exactly once;
  3. every package-level identifier and struct field of the reference that the shim uses exists in a file that is
     still in the build (ParsedJson{Message, Tape, Strings, internal}, TStrings{B}, internalParsedJson{copyStrings},
     ParserOption);
  4. an identifier that only excluded files declare and files still in the build use is declared by the shim.
The scanner is a tokenizer (comments and literals stripped, bracket depth tracked), not a Go parser: it reads
declarations at depth 0 and the entries of grouped `type ( )` / `var ( )` / `const ( )` declarations."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
INVENTORY = os.path.join(ROOT, "tests", "golden", "go_reference_symbols.json")
SHIM = os.path.join(ROOT, "simdjson-go_amd", "go", "simdjson_hip.go")

# the complete tag edits (file -> new //go:build line)
TAG_EDITS = {
    "simdjson_amd64.go": "!appengine && !noasm && gc && !hip",
    "simdjson_other.go": "(!amd64 || appengine || !gc || noasm) && !hip",
}
BACKEND_SYMBOLS = ["SupportedCPU", "Parse", "ParseND", "Stream", "ParseNDStream"]  # simdjson_other.go:29-76
SCENARIOS = {
    "linux/amd64 -tags hip": {"linux", "amd64", "gc", "cgo", "hip", "go1.18", "go1.20", "go1.21", "unix"},
    "linux/arm64 -tags hip": {"linux", "arm64", "gc", "cgo", "hip", "go1.18", "go1.20", "go1.21", "unix"},
}
KNOWN_ARCH = {"386", "amd64", "arm", "arm64", "ppc64le", "riscv64", "s390x", "wasm", "mips64", "loong64"}
KNOWN_OS = {"linux", "darwin", "windows", "freebsd", "js", "plan9"}


def strip_go(text):
    """Comments -> spaces, string / rune literals -> "" (newlines kept: Go ends statements at line ends)."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            j = n if j < 0 else j + 2
            out.append("\n" * text.count("\n", i, j) or " ")
            i = j
        elif c == '"':
            j = i + 1
            while j < n and text[j] != '"':
                j += 2 if text[j] == "\\" else 1
            out.append('""')
            i = j + 1
        elif c == "`":
            j = text.find("`", i + 1)
            j = n if j < 0 else j
            out.append('""' + "\n" * text.count("\n", i, j))
            i = j + 1
        elif c == "'":
            j = i + 1
            while j < n and text[j] != "'":
                j += 2 if text[j] == "\\" else 1
            out.append("0")
            i = j + 1
        else:
            out.append(c)
            i += 1
    return "".join(out)


def build_constraint(text):
    m = re.search(r"^//go:build (.+)$", text, re.M)
    if m and text[:m.start()].strip("\n").replace("\n", "").startswith("//") or (m and m.start() == 0):
        return m.group(1).strip()
    return m.group(1).strip() if m and "package " not in text[:m.start()] else ""


def eval_constraint(expr, tags):
    toks = re.findall(r"&&|\|\||[!()]|[A-Za-z0-9_.]+", expr)
    pos = 0

    def atom():
        nonlocal pos
        t = toks[pos]
        pos += 1
        if t == "!":
            return not atom()
        if t == "(":
            v = or_()
            pos += 1  # ')'
            return v
        return t in tags

    def and_():
        nonlocal pos
        v = atom()
        while pos < len(toks) and toks[pos] == "&&":
            pos += 1
            v = atom() and v
        return v

    def or_():
        nonlocal pos
        v = and_()
        while pos < len(toks) and toks[pos] == "||":
            pos += 1
            v = and_() or v
        return v

    return or_() if toks else True


def file_active(name, constraint, tags):
    stem = name[:-3]
    if stem.endswith("_test"):
        stem = stem[:-5]
    parts = stem.split("_")
    if parts[-1] in KNOWN_ARCH:
        if parts[-1] not in tags:
            return False
        parts = parts[:-1]
    if len(parts) > 1 and parts[-1] in KNOWN_OS and parts[-1] not in tags:
        return False
    return eval_constraint(constraint, tags)


IDENT = r"[A-Za-z_][A-Za-z0-9_]*"


def scan_go(text):
    """-> {"constraint", "decls": [name | Recv.Name], "structs": {type: [fields]}, "idents": sorted set of all identifiers}"""
    constraint = build_constraint(text)
    src = strip_go(text)
    decls, structs = [], {}
    depth = {"(": 0, "{": 0, "[": 0}
    close = {")": "(", "}": "{", "]": "["}
    lines = src.split("\n")
    group = None  # 'type' / 'var' / 'const' while inside a top-level grouped declaration
    struct_of = None  # (type name, brace depth at which its fields live)
    for line in lines:
        at0 = depth["("] == 0 and depth["{"] == 0 and depth["["] == 0
        in_group = group is not None and depth["("] == 1 and depth["{"] == 0 and depth["["] == 0
        s = line.strip()
        if at0:
            group = None
            m = re.match(r"func\s*\(\s*(?:%s\s+)?\*?\s*(%s)(?:\[[^\]]*\])?\s*\)\s*(%s)" % (IDENT, IDENT, IDENT), s)
            if m:
                decls.append(f"{m.group(1)}.{m.group(2)}")
            else:
                m = re.match(r"func\s+(%s)" % IDENT, s)
                if m:
                    decls.append(m.group(1))
            m = re.match(r"(type|var|const)\s*\(", s)
            if m:
                group = m.group(1)
            else:
                m = re.match(r"type\s+(%s)" % IDENT, s)
                if m:
                    decls.append(m.group(1))
                    if re.search(r"\bstruct\s*\{", s):
                        struct_of = (m.group(1), 1)
                        structs[m.group(1)] = []
                m = re.match(r"(?:var|const)\s+((?:%s\s*,\s*)*%s)" % (IDENT, IDENT), s)
                if m:
                    decls += [x.strip() for x in m.group(1).split(",")]
        elif in_group and s:
            if group == "type":
                m = re.match(r"(%s)\b" % IDENT, s)
                if m:
                    decls.append(m.group(1))
                    if re.search(r"\bstruct\s*\{", s):
                        struct_of = (m.group(1), 1)
                        structs[m.group(1)] = []
            else:
                m = re.match(r"((?:%s\s*,\s*)*%s)" % (IDENT, IDENT), s)
                if m:
                    decls += [x.strip() for x in m.group(1).split(",")]
        elif struct_of and depth["{"] == struct_of[1] and depth["["] == 0 and s and not s.startswith("}"):
            # a field line: `a, b T`, `name T`, or an embedded `T` / `*T`
            m = re.match(r"\*?((?:%s\s*,\s*)*%s)" % (IDENT, IDENT), s)
            if m:
                structs[struct_of[0]] += [x.strip() for x in m.group(1).split(",")]
        for ch in line:
            if ch in depth:
                depth[ch] += 1
            elif ch in close:
                depth[close[ch]] -= 1
        if struct_of and depth["{"] < struct_of[1]:
            struct_of = None
        if group is not None and depth["("] == 0:
            group = None
    decls = [d for d in decls if d != "_"]
    idents = sorted(set(re.findall(IDENT, src)))
    return {"constraint": constraint, "decls": decls, "structs": structs, "idents": idents}


def make_inventory(ref_dir=REFERENCE):
    inv = {}
    for name in sorted(os.listdir(ref_dir)):
        if name.endswith(".go"):
            with open(os.path.join(ref_dir, name), encoding="utf-8", errors="replace") as f:
                inv[name] = scan_go(f.read())
    return inv


def check(shim_text, inventory, edits=TAG_EDITS, scenarios=SCENARIOS):
    """-> list of problems (empty: the shim fits)"""
    problems = []
    shim = scan_go(shim_text)
    for label, tags in scenarios.items():
        if not eval_constraint(shim["constraint"], tags):
            problems.append(f"{label}: the shim's own constraint `{shim['constraint']}` is false")
            continue
        owners = {}
        for d in shim["decls"]:
            owners.setdefault(d, []).append("simdjson_hip.go")
        active, excluded = [], []
        for name, info in inventory.items():
            if name.endswith("_test.go"):
                continue
            (active if file_active(name, edits.get(name, info["constraint"]), tags) else excluded).append(name)
        for name in active:
            for d in inventory[name]["decls"]:
                if d != "init":
                    owners.setdefault(d, []).append(name)
        for d, files in sorted(owners.items()):
            if len(files) > 1:
                problems.append(f"{label}: `{d}` is declared in {' and '.join(files)}")
        for sym in BACKEND_SYMBOLS:
            if len(owners.get(sym, [])) != 1:
                problems.append(f"{label}: backend symbol `{sym}` is declared {len(owners.get(sym, []))} times")
        # what the shim needs from the reference
        structs = {}
        for name in active:
            structs.update(inventory[name]["structs"])
        needs = {"ParsedJson": ["Message", "Tape", "Strings", "internal"], "TStrings": ["B"],
                 "internalParsedJson": ["ParsedJson", "copyStrings"], "ParserOption": []}
        for typ, fields in needs.items():
            if typ not in owners or owners[typ] == ["simdjson_hip.go"]:
                problems.append(f"{label}: the shim uses the reference's `{typ}`, which no file in the build declares")
                continue
            for fld in fields:
                if fld not in structs.get(typ, []):
                    problems.append(f"{label}: `{typ}` has no field `{fld}` (fields: {structs.get(typ)})")
        # identifiers that only excluded files declare but files in the build still use
        used = set()
        for name in active:
            used.update(inventory[name]["idents"])
        for name in excluded:
            for d in inventory[name]["decls"]:
                if "." in d or d == "init" or d in owners:
                    continue
                if d in used:
                    problems.append(f"{label}: `{d}` is declared only by {name} (not in this build) but files in the build use it")
    return problems


def main(argv):
    if "--write-inventory" in argv:
        inv = make_inventory()
        # the identifier lists are only needed for files that stay in some build (check 4): keep the fixture small
        with open(INVENTORY, "w") as f:
            json.dump({"source": "tools/check_go_collisions.py --write-inventory over /root/reference/*.go "
                                 "(minio/simdjson-go): build constraint, top-level identifiers, struct fields and the "
                                 "set of identifiers used, per file", "files": inv}, f, indent=0, sort_keys=True)
        print(f"wrote {INVENTORY}: {len(inv)} files")
        return 0
    with open(INVENTORY) as f:
        inv = json.load(f)["files"]
    with open(SHIM) as f:
        problems = check(f.read(), inv)
    for p in problems:
        print("PROBLEM:", p)
    if not problems:
        print("simdjson_hip.go fits: no identifier collisions, backend symbols declared once, reference types present "
              f"({', '.join(SCENARIOS)})")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
