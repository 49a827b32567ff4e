"""A small evaluator for the Go composite literals found in the reference's *_test.go tables.

Only what those tables use: raw / interpreted string literals, rune and integer / float
literals, []byte / []uintN / struct composite literals (positional or keyed), conversions
(string(), []byte(), uint64() ...), strings.Repeat, unary ^ - +, binary + - * | & ^ << >>,
identifiers resolved through an environment.  Strings evaluate to `bytes`.
"""
import re

TOKEN_RE = re.compile(
    r"""
    (?P<ws>\s+)
  | (?P<lc>//[^\n]*)
  | (?P<bc>/\*.*?\*/)
  | (?P<raw>`[^`]*`)
  | (?P<str>"(?:\\.|[^"\\])*")
  | (?P<rune>'(?:\\.[^']*|[^'\\])')
  | (?P<float>(?:\d[\d_]*\.\d*(?:[eE][+-]?\d+)?|\d[\d_]*[eE][+-]?\d+|\.\d+(?:[eE][+-]?\d+)?))
  | (?P<int>0[xX][0-9a-fA-F_]+|0[bB][01_]+|0[oO][0-7_]+|\d[\d_]*)
  | (?P<id>[A-Za-z_][A-Za-z_0-9]*)
  | (?P<op><<|>>|&\^|:=|==|!=|<=|>=|&&|\|\||[-+*/%&|^<>=!(){}\[\],.:;])
    """,
    re.X | re.S,
)


def tokenize(src):
    out = []
    pos = 0
    while pos < len(src):
        m = TOKEN_RE.match(src, pos)
        if not m:
            raise SyntaxError(f"cannot tokenize at {pos}: {src[pos:pos+40]!r}")
        pos = m.end()
        k = m.lastgroup
        if k in ("ws", "lc", "bc"):
            continue
        out.append((k, m.group()))
    return out


_ESC = {"a": 7, "b": 8, "f": 12, "n": 10, "r": 13, "t": 9, "v": 11, "\\": 92, "'": 39, '"': 34}


def unquote(s):
    """Go interpreted string literal body -> bytes."""
    out = bytearray()
    i = 0
    while i < len(s):
        c = s[i]
        if c != "\\":
            out += c.encode("utf-8")
            i += 1
            continue
        e = s[i + 1]
        if e in _ESC:
            out.append(_ESC[e])
            i += 2
        elif e == "x":
            out.append(int(s[i + 2 : i + 4], 16))
            i += 4
        elif e == "u":
            out += chr(int(s[i + 2 : i + 6], 16)).encode("utf-8")
            i += 6
        elif e == "U":
            out += chr(int(s[i + 2 : i + 10], 16)).encode("utf-8")
            i += 10
        elif e in "01234567":
            out.append(int(s[i + 1 : i + 4], 8))
            i += 4
        else:
            raise SyntaxError("bad escape \\" + e)
    return bytes(out)


class Struct(dict):
    """Keyed composite literal."""


class Parser:
    def __init__(self, toks, env=None):
        self.t = toks
        self.i = 0
        self.env = env or {}

    def peek(self, k=0):
        return self.t[self.i + k] if self.i + k < len(self.t) else ("eof", "")

    def next(self):
        tok = self.peek()
        self.i += 1
        return tok

    def expect(self, v):
        tok = self.next()
        if tok[1] != v:
            raise SyntaxError(f"expected {v!r} got {tok!r} at token {self.i}")

    # ---- types (skipped) ----
    def skip_type(self):
        """Skip a Go type expression; returns a short description."""
        k, v = self.peek()
        if v == "[":
            self.next()
            while self.peek()[1] != "]":
                self.next()
            self.next()
            return "[]" + self.skip_type()
        if v == "struct":
            self.next()
            self.skip_braces()
            return "struct"
        if v == "*":
            self.next()
            return self.skip_type()
        if k == "id":
            self.next()
            name = v
            while self.peek()[1] == ".":
                self.next()
                name += "." + self.next()[1]
            return name
        raise SyntaxError(f"type? {self.peek()}")

    def skip_braces(self):
        self.expect("{")
        depth = 1
        while depth:
            v = self.next()[1]
            if v == "{":
                depth += 1
            elif v == "}":
                depth -= 1

    # ---- expressions ----
    PREC = {"*": 5, "/": 5, "%": 5, "<<": 5, ">>": 5, "&": 5, "&^": 5, "+": 4, "-": 4, "|": 4, "^": 4}

    def expr(self, minprec=1):
        lhs = self.unary()
        while True:
            op = self.peek()[1]
            p = self.PREC.get(op)
            if self.peek()[0] != "op" or p is None or p < minprec:
                return lhs
            self.next()
            rhs = self.expr(p + 1)
            lhs = self.binop(op, lhs, rhs)

    @staticmethod
    def binop(op, a, b):
        if op == "+":
            return a + b
        if op == "-":
            return a - b
        if op == "*":
            return a * b
        if op == "<<":
            return a << b
        if op == ">>":
            return a >> b
        if op == "|":
            return a | b
        if op == "&":
            return a & b
        if op == "^":
            return a ^ b
        raise SyntaxError(op)

    def unary(self):
        k, v = self.peek()
        if k == "op" and v in ("^", "-", "+", "&"):
            self.next()
            x = self.unary()
            if v == "^":
                return ~x & 0xFFFFFFFFFFFFFFFF
            if v == "-":
                return -x
            return x
        return self.postfix(self.primary())

    def postfix(self, x):
        while True:
            v = self.peek()[1]
            if v == "." and self.peek(1)[0] == "id":
                name = self.peek(1)[1]
                self.i += 2
                if self.peek()[1] == "(":
                    args = self.args()
                    x = self.method(x, name, args)
                else:
                    x = (x, name) if not isinstance(x, dict) else x[name]
            elif v == "[" and isinstance(x, (bytes, list)):
                self.next()
                lo = None if self.peek()[1] == ":" else self.expr()
                if self.peek()[1] == ":":
                    self.next()
                    hi = None if self.peek()[1] == "]" else self.expr()
                    self.expect("]")
                    x = x[lo:hi]
                else:
                    self.expect("]")
                    x = x[lo]
            else:
                return x

    def method(self, x, name, args):
        if name == "Flags":
            return x
        raise SyntaxError(f"method {name}")

    def args(self):
        self.expect("(")
        out = []
        while self.peek()[1] != ")":
            out.append(self.expr())
            if self.peek()[1] == ",":
                self.next()
        self.expect(")")
        return out

    def primary(self):
        k, v = self.next()
        if k == "raw":
            return v[1:-1].encode("utf-8")
        if k == "str":
            return unquote(v[1:-1])
        if k == "rune":
            b = unquote(v[1:-1])
            return ord(b.decode("utf-8")) if len(b) > 1 else b[0]
        if k == "int":
            return int(v.replace("_", ""), 0) if not re.fullmatch(r"0\d+", v) else int(v, 8)
        if k == "float":
            return float(v.replace("_", ""))
        if v == "(":
            x = self.expr()
            self.expect(")")
            return x
        if v == "[":
            self.i -= 1
            ty = self.skip_type()
            if self.peek()[1] == "(":  # conversion, e.g. []byte("..")
                (x,) = self.args()
                return self.convert(ty, x)
            return self.composite(ty)
        if v == "struct":
            self.i -= 1
            self.skip_type()
            return self.composite("struct")
        if v == "{":  # elided type
            self.i -= 1
            return self.composite("")
        if k == "id":
            if v in ("true", "false"):
                return v == "true"
            if v == "nil":
                return None
            name = v
            # qualified identifier / call
            if self.peek()[1] == "." and self.peek(1)[0] == "id" and name in ("strings", "strconv", "errors", "math", "cpuid"):
                self.next()
                name += "." + self.next()[1]
            if self.peek()[1] == "(":
                args = self.args()
                return self.call(name, args)
            if self.peek()[1] == "{" and name[:1].isupper() is False and name in self.env.get("__types__", ()):
                return self.composite(name)
            if name in self.env:
                return self.env[name]
            return ("ident", name)
        raise SyntaxError(f"primary? {(k, v)}")

    def convert(self, ty, x):
        if ty in ("[]byte", "string"):
            if isinstance(x, list):
                return bytes(x)
            if isinstance(x, int):
                return chr(x).encode("utf-8")
            return bytes(x)
        if ty.startswith("uint") or ty.startswith("int") or ty == "byte":
            bits = {"uint64": 64, "uint32": 32, "uint8": 8, "byte": 8, "uint": 64}.get(ty)
            return x & ((1 << bits) - 1) if bits and isinstance(x, int) else x
        return x

    def call(self, name, args):
        if name == "strings.Repeat":
            return args[0] * args[1]
        if name in ("string", "uint64", "uint32", "uint8", "uint", "int", "int64", "byte", "float64"):
            return self.convert(name, args[0])
        if name == "errors.New":
            return ("error", args[0])
        return ("call", name, args)

    def composite(self, ty):
        self.expect("{")
        elems = []
        keyed = None
        while self.peek()[1] != "}":
            if self.peek()[0] == "id" and self.peek(1)[1] == ":":
                key = self.next()[1]
                self.next()
                val = self.expr()
                if keyed is None:
                    keyed = Struct()
                keyed[key] = val
            else:
                elems.append(self.expr())
            if self.peek()[1] == ",":
                self.next()
        self.expect("}")
        if keyed is not None:
            return keyed
        if ty in ("[]byte", "[]uint8"):
            return bytes(elems)
        return elems


def find_literal(src, anchor, env=None, after_type=True):
    """Locate `anchor` (regex) in Go source and evaluate the composite literal that follows.

    The text after the anchor must start with a type (`[]struct {...}` / `[]uint64` / `[]T`) and
    then the `{ ... }` literal."""
    m = re.search(anchor, src)
    if not m:
        raise KeyError(anchor)
    toks = tokenize(src[m.end():])
    p = Parser(toks, env)
    ty = p.skip_type()
    return p.composite(ty)
