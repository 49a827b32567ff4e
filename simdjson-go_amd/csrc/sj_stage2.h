// sj_stage2.h -- stage 2 (tape build) as data-parallel per-token functions, host+device.
//
// The reference builds the tape with a sequential goto state machine over the structural
// indexes (unifiedMachine, stage2_build_tape_amd64.go:160-446).  Here every structural index
// ("token") is handled by an independent lane; the state the machine carries is recovered from
// prefix scans and nearest-smaller-value queries over the token array:
//
//   depth[i]     = (# '{' '[' - # '}' ']') over tokens 0..i            (+ scan)
//   tape_off[i]  = 1 + sum of tape words of tokens < i                 (+ scan; word 0 is the root)
//   str_off[i]   = sum of unescaped lengths of copied strings < i      (+ scan) -> Strings.B offsets
//   last_br[i]   = index+1 of the last bracket token <= i              (max scan)
//   match / parent of a bracket = "previous smaller value" of depth[]  (64-ary min tree)
//
// and the grammar is checked per token against its predecessor(s) and the type of the innermost
// open container ("context"); DESIGN.md lists the rule table and why it accepts exactly the
// documents the machine accepts.  Any violation sets one global error flag; like the reference,
// a failed parse returns no tape, so only the first-violation-free prefix needs exact bookkeeping.
#pragma once
#include <stdint.h>

#include "sj_chunk.h"

namespace sj {

typedef int32_t i32;
typedef int64_t i64;

enum Kind : u8 {
    K_BAD = 0,
    K_OPEN_OBJ = 1,
    K_OPEN_ARR = 2,
    K_CLOSE_OBJ = 3,
    K_CLOSE_ARR = 4,
    K_COLON = 5,
    K_COMMA = 6,
    K_STRING = 7,
    K_NUM = 8,
    K_TRUE = 9,
    K_FALSE = 10,
    K_NULL = 11,
    K_NL = 12,
};
enum Ctx : u8 { CTX_ROOT = 0, CTX_OBJ = 1, CTX_ARR = 2 };

static constexpr u64 STRINGBUFBIT = 0x0080000000000000ull;  // parsed_json.go:29

SJ_HD u8 token_kind(u8 c, bool ndjson) {
    switch (c) {
    case '{': return K_OPEN_OBJ;
    case '[': return K_OPEN_ARR;
    case '}': return K_CLOSE_OBJ;
    case ']': return K_CLOSE_ARR;
    case ':': return K_COLON;
    case ',': return K_COMMA;
    case '"': return K_STRING;
    case 't': return K_TRUE;
    case 'f': return K_FALSE;
    case 'n': return K_NULL;
    case '-': return K_NUM;
    case '\n': return ndjson ? K_NL : K_BAD;
    default: return (c >= '0' && c <= '9') ? K_NUM : K_BAD;
    }
}
SJ_HD bool is_open(u8 k) { return k == K_OPEN_OBJ || k == K_OPEN_ARR; }
SJ_HD bool is_close(u8 k) { return k == K_CLOSE_OBJ || k == K_CLOSE_ARR; }
SJ_HD bool is_bracket(u8 k) { return k >= K_OPEN_OBJ && k <= K_CLOSE_ARR; }
SJ_HD i32 depth_delta(u8 k) { return is_open(k) ? 1 : (is_close(k) ? -1 : 0); }

// number of tape words a token writes (stage2_build_tape_amd64.go: write_tape call sites).
// A newline token writes the "close root / open root" pair (:213-218) iff it is the last one of
// its run and another token follows.
SJ_HD u32 tape_words(u8 k, u8 next_kind, bool is_last) {
    switch (k) {
    case K_OPEN_OBJ:
    case K_OPEN_ARR:
    case K_CLOSE_OBJ:
    case K_CLOSE_ARR:
    case K_TRUE:
    case K_FALSE:
    case K_NULL: return 1;
    case K_STRING:
    case K_NUM: return 2;
    case K_NL: return (!is_last && next_kind != K_NL) ? 2 : 0;
    default: return 0;
    }
}

// ---- strings: parse_string_amd64.s restated byte-serially (one string per lane) ----------------
// digittoval with the DATA-section hole (bytes < 0x30 -> 0), see oracle/sjo_parse_string.c (Q3).
SJ_HD i32 hex_digit(u8 b) {
    if (b < 0x30) return 0;
    if (b <= '9') return b - '0';
    if (b >= 'A' && b <= 'F') return b - 'A' + 10;
    if (b >= 'a' && b <= 'f') return b - 'a' + 10;
    return -1;
}
SJ_HD u8 escape_value(u8 b) {  // escape_map, parse_string_amd64.s:38-69
    switch (b) {
    case '"': return 0x22;
    case '/': return 0x2f;
    case '\\': return 0x5c;
    case 'b': return 0x08;
    case 'f': return 0x0c;
    case 'n': return 0x0a;
    case 'r': return 0x0d;
    case 't': return 0x09;
    default: return 0;
    }
}

struct MsgView {
    const u8 *p;
    u64 len;
    SJ_HD u8 at(u64 i) const { return i < len ? p[i] : (u8)0; }  // the Go caller zero-pads (stage2…:75-86)
};

SJ_HD u32 hex4(const MsgView &m, u64 p) {
    const u32 d0 = (u32)hex_digit(m.at(p)), d1 = (u32)hex_digit(m.at(p + 1)), d2 = (u32)hex_digit(m.at(p + 2)),
              d3 = (u32)hex_digit(m.at(p + 3));
    return (d0 << 12) | (d1 << 8) | (d2 << 4) | d3;  // sign-extended -1 poisons the high bits
}

// ---- unaligned 8-byte access and the exact zero-byte detector used by the plain-byte fast path ------
SJ_HD u64 load_u64(const u8 *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *reinterpret_cast<const u64 *>(p);
#else
    u64 v;
    __builtin_memcpy(&v, p, 8);
    return v;
#endif
}
SJ_HD void store_u64(u8 *p, u64 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    *reinterpret_cast<u64 *>(p) = v;
#else
    __builtin_memcpy(p, &v, 8);
#endif
}
// the low n (< 8) bytes of v
SJ_HD void store_bytes(u8 *p, u64 v, u32 n) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (n & 4u) {
        *reinterpret_cast<u32 *>(p) = (u32)v;
        p += 4;
        v >>= 32;
    }
    if (n & 2u) {
        *reinterpret_cast<uint16_t *>(p) = (uint16_t)v;
        p += 2;
        v >>= 16;
    }
    if (n & 1u) *p = (u8)v;
#else
    for (u32 i = 0; i < n; i++) p[i] = (u8)(v >> (8 * i));
#endif
}
// 0x80 in every byte of x that is zero, 0 elsewhere (exact, no borrow artefacts)
SJ_HD u64 zero_bytes(u64 x) {
    const u64 k = 0x7f7f7f7f7f7f7f7full;
    return ~(((x & k) + k) | x | k);
}

// Walks the string whose opening quote is at `q`.  If dst != nullptr the unescaped bytes are
// written there.  Returns false if the reference's _parse_string_validate_only fails.
SJ_HD bool string_walk(const MsgView &m, u64 q, u8 *dst, u32 *src_len, u32 *dst_len) {
    u64 pos = q + 1;
    u32 out = 0;
    for (;;) {
        // plain bytes, eight at a time (unaligned 8-byte loads and stores are fine on gfx950 global memory)
        while (pos + 8 <= m.len) {
            const u64 x = load_u64(m.p + pos);
            const u64 z = zero_bytes(x ^ 0x2222222222222222ull) | zero_bytes(x ^ 0x5c5c5c5c5c5c5c5cull);
            if (z == 0) {
                if (dst) store_u64(dst + out, x);
                pos += 8;
                out += 8;
                continue;
            }
            const u32 nplain = (u32)ctz64(z) >> 3;  // bytes in front of the first quote / backslash
            if (dst) store_bytes(dst + out, x, nplain);
            pos += nplain;
            out += nplain;
            break;
        }
        if (pos >= m.len) return false;  // unterminated: unreachable once stage 1 has accepted the document
        const u8 c = m.p[pos];
        if (c == '"') {
            *src_len = (u32)(pos - (q + 1));
            *dst_len = out;
            return true;
        }
        if (c != '\\') {
            if (dst) dst[out] = c;
            out++;
            pos++;
            continue;
        }
        const u8 e = m.at(pos + 1);
        if (e != 'u') {
            const u8 v = escape_value(e);
            if (v == 0) return false;
            if (dst) dst[out] = v;
            out++;
            pos += 2;
            continue;
        }
        // \uXXXX: the next raw quote must be at least 6 (12 for a surrogate pair) bytes away
        u32 d = 12;
        for (u32 j = 1; j < 12; j++)
            if (m.at(pos + j) == '"') {
                d = j;
                break;
            }
        if (d < 6) return false;
        u32 cp = hex4(m, pos + 2);
        u64 next = pos + 6;
        if ((cp & 0xfffffc00u) == 0xd800u) {
            if (d < 12) return false;
            if (m.at(pos + 6) != '\\' || m.at(pos + 7) != 'u') return false;
            const u32 cp2 = hex4(m, pos + 8);
            if ((cp | cp2) > 0xffffu) return false;
            cp = (((cp << 10) + 0xfca00000u) | (cp2 + 0xffff2400u)) + 0x10000u;  // 32-bit wrap-around, low half unchecked
            next = pos + 12;
        }
        if (cp < 0x80u) {
            if (dst) dst[out] = (u8)cp;
            out += 1;
        } else if (cp < 0x800u) {
            if (dst) {
                dst[out] = (u8)((cp >> 6) + 192);
                dst[out + 1] = (u8)((cp & 63) | 128);
            }
            out += 2;
        } else if (cp < 0x10000u) {
            if (dst) {
                dst[out] = (u8)((cp >> 12) + 224);
                dst[out + 1] = (u8)(((cp >> 6) & 63) | 128);
                dst[out + 2] = (u8)((cp & 63) | 128);
            }
            out += 3;
        } else if (cp <= 0x10ffffu) {
            if (dst) {
                dst[out] = (u8)((cp >> 18) + 240);
                dst[out + 1] = (u8)(((cp >> 12) & 63) | 128);
                dst[out + 2] = (u8)(((cp >> 6) & 63) | 128);
                dst[out + 3] = (u8)((cp & 63) | 128);
            }
            out += 4;
        } else {
            return false;
        }
        pos = next;
    }
}

// ---- atoms (stage2_build_tape_amd64.go:124-158, table :455-476) ----------------------------------
SJ_HD bool atom_terminator(u8 c) {
    switch (c) {
    case 0: case '\t': case '\n': case '\r': case ' ': case ',': case ':': case '[': case ']': case '{': case '}':
        return true;
    default: return false;
    }
}
SJ_HD bool atom_valid(const MsgView &m, u64 p, u8 kind) {
    const u64 rem = m.len - p;
    if (kind == K_TRUE)
        return rem >= 5 && m.p[p + 1] == 'r' && m.p[p + 2] == 'u' && m.p[p + 3] == 'e' && atom_terminator(m.p[p + 4]);
    if (kind == K_NULL)
        return rem >= 5 && m.p[p + 1] == 'u' && m.p[p + 2] == 'l' && m.p[p + 3] == 'l' && atom_terminator(m.p[p + 4]);
    return rem >= 6 && m.p[p + 1] == 'a' && m.p[p + 2] == 'l' && m.p[p + 3] == 's' && m.p[p + 4] == 'e' &&
           atom_terminator(m.p[p + 5]);
}

// ---- previous-smaller-value over depth[] with a 64-ary min tree ------------------------------------
struct MinTree {
    static constexpr int MAXLEV = 7;
    const i32 *lev[MAXLEV];  // lev[0] = depth[], lev[k][g] = min of lev[k-1][64g .. 64g+63]
    u64 size[MAXLEV];
    int nlev;
};

// last k < i with depth[k] < tau, or -1
SJ_HD i64 psv(const MinTree &t, i64 i, i32 tau) {
    if (i <= 0) return -1;
    i64 k = i - 1;
    int L = 0;
    for (;;) {
        const i64 gstart = (k >> 6) << 6;
        for (; k >= gstart; k--)
            if (t.lev[L][k] < tau) goto found;
        if (gstart == 0 || L + 1 >= t.nlev) return -1;
        k = (gstart >> 6) - 1;
        L++;
    }
found:
    while (L > 0) {
        L--;
        const i64 base = k << 6;
        i64 hi = base + 63;
        if ((u64)hi >= t.size[L]) hi = (i64)t.size[L] - 1;
        for (k = hi; k > base; k--)
            if (t.lev[L][k] < tau) break;
    }
    return k;
}

// ---- per-token views -----------------------------------------------------------------------------
struct Tokens {
    const u32 *pos;   // structural byte positions (stage 1 output)
    u32 n;
    const u8 *kind;
    const i32 *depth;     // depth AFTER the token
    const u32 *tape_off;  // tape index of the token's first word
    const u32 *str_off;   // Strings.B offset for copied strings
    const u32 *last_br;   // 1 + index of the last bracket token <= i (0: none)
    const u32 *match;     // brackets: index of the partner
    const u8 *ctxb;       // close brackets: context after the close
    // NDJSON shards: where this shard's tape / Strings.B / Message window start inside the merged
    // ParsedJson (all zero for an unsharded parse).  Every index the tape stores is rebased by these.
    u64 tape_base = 0, strings_base = 0, msg_base = 0;
};

// context of the gap in front of token i
SJ_HD u8 gap_ctx(const Tokens &t, u32 i) {
    if (i == 0) return CTX_ROOT;
    const u32 b = t.last_br[i - 1];
    if (b == 0) return CTX_ROOT;
    const u8 k = t.kind[b - 1];
    if (k == K_OPEN_OBJ) return CTX_OBJ;
    if (k == K_OPEN_ARR) return CTX_ARR;
    return t.ctxb[b - 1];
}

// bracket pass: for a close bracket find its partner and the context that resumes after it
SJ_HD void bracket_resolve(const MinTree &mt, const u8 *kind, const i32 *depth, u32 i, u32 *match, u8 *ctxb) {
    const i32 d = depth[i] + 1;  // depth before the close
    if (d <= 0) {                 // closes nothing: the grammar check rejects it (context is ROOT)
        match[i] = 0;
        ctxb[i] = CTX_ROOT;
        return;
    }
    const i64 j = psv(mt, (i64)i, d) + 1;  // the open bracket that raised the depth to d
    match[i] = (u32)j;
    match[j] = i;
    u8 ctx = CTX_ROOT;
    if (d - 1 > 0) {
        const i64 p = psv(mt, j, d - 1) + 1;  // the enclosing container's open bracket
        ctx = kind[p] == K_OPEN_OBJ ? CTX_OBJ : CTX_ARR;
    }
    ctxb[i] = ctx;
}

// The same on the compact bracket view (br_tok[c] = token index of the c-th bracket, the tree is built over
// br_depth[c] = depth after it): non-bracket tokens never change the depth, so the previous-smaller-value
// queries give the same brackets, over an array that is ~10x shorter.
SJ_HD void bracket_resolve_compact(const MinTree &mt, const u32 *br_tok, const u8 *kind, u32 c, u32 *match, u8 *ctxb) {
    const u32 i = br_tok[c];
    const i32 d = mt.lev[0][c] + 1;  // depth before the close
    if (d <= 0) {                    // closes nothing: the grammar check rejects it (context is ROOT)
        match[i] = 0;
        ctxb[i] = CTX_ROOT;
        return;
    }
    const i64 jc = psv(mt, (i64)c, d) + 1;  // the open bracket that raised the depth to d
    const u32 j = br_tok[jc];
    match[i] = j;
    match[j] = i;
    u8 ctx = CTX_ROOT;
    if (d - 1 > 0) {
        const i64 pc = psv(mt, jc, d - 1) + 1;  // the enclosing container's open bracket
        ctx = kind[br_tok[pc]] == K_OPEN_OBJ ? CTX_OBJ : CTX_ARR;
    }
    ctxb[i] = ctx;
}

// ---- grammar on values (the kernel preloads them; the pointer forms below feed the same functions) ----------
// a string token is an object key iff it sits in an object right after '{' or ','
SJ_HD bool string_is_key_v(u8 ctx, bool has_prev, u8 prev_kind) {
    return ctx == CTX_OBJ && has_prev && (prev_kind == K_OPEN_OBJ || prev_kind == K_COMMA);
}
// does a token of kind k (whose predecessor has kind prev_kind, if any) end a value of the container with context ctx?
SJ_HD bool ends_value_v(u8 k, u8 ctx, bool has_prev, u8 prev_kind) {
    if (k == K_CLOSE_OBJ || k == K_CLOSE_ARR || k == K_NUM || k == K_TRUE || k == K_FALSE || k == K_NULL) return true;
    if (k == K_STRING) return !string_is_key_v(ctx, has_prev, prev_kind);
    return false;
}
// Grammar check of token i (true = violation): k = its kind, pk / ppk = kinds of tokens i-1 / i-2 (if they
// exist), G = context of the gap in front of it.  Mirrors the transitions of unifiedMachine:
// continueRoot/startContinue (:176-221), object_begin/object_key_state/objectContinue (:225-325),
// arrayBegin/mainArraySwitch/arrayContinue (:346-426).
SJ_HD bool grammar_violation_v(u32 i, u8 k, u8 pk, u8 ppk, u8 G) {
    if (k == K_BAD) return true;
    if (i == 0) return !is_open(k);
    const bool has_pp = i > 1;
    switch (k) {
    case K_OPEN_OBJ:
    case K_OPEN_ARR:
        if (G == CTX_ROOT) return pk != K_NL;
        // fallthrough: a container is a value
    case K_NUM:
    case K_TRUE:
    case K_FALSE:
    case K_NULL:
        if (G == CTX_OBJ) return pk != K_COLON;
        if (G == CTX_ARR) return !(pk == K_OPEN_ARR || pk == K_COMMA);
        return true;
    case K_STRING:
        if (G == CTX_OBJ) return !(pk == K_OPEN_OBJ || pk == K_COMMA || pk == K_COLON);
        if (G == CTX_ARR) return !(pk == K_OPEN_ARR || pk == K_COMMA);
        return true;
    case K_COLON:
        return !(G == CTX_OBJ && pk == K_STRING && string_is_key_v(G, has_pp, ppk));
    case K_COMMA:
        return !(G != CTX_ROOT && ends_value_v(pk, G, has_pp, ppk));
    case K_CLOSE_OBJ:
        return !(G == CTX_OBJ && (pk == K_OPEN_OBJ || ends_value_v(pk, G, has_pp, ppk)));
    case K_CLOSE_ARR:
        return !(G == CTX_ARR && (pk == K_OPEN_ARR || ends_value_v(pk, G, has_pp, ppk)));
    case K_NL:
        return !(G == CTX_ROOT && (is_close(pk) || pk == K_NL));
    default:
        return true;
    }
}
// context from the last bracket in front of a token: b = its index + 1 (0: none), bk / bc = its kind / resume context
SJ_HD u8 gap_ctx_v(u32 b, u8 bk, u8 bc) {
    if (b == 0) return CTX_ROOT;
    if (bk == K_OPEN_OBJ) return CTX_OBJ;
    if (bk == K_OPEN_ARR) return CTX_ARR;
    return bc;
}

SJ_HD bool string_is_key(const Tokens &t, u32 j, u8 ctx) { return string_is_key_v(ctx, j > 0, j > 0 ? t.kind[j - 1] : (u8)K_BAD); }
SJ_HD bool grammar_violation(const Tokens &t, u32 i) {
    return grammar_violation_v(i, t.kind[i], i > 0 ? t.kind[i - 1] : (u8)K_BAD, i > 1 ? t.kind[i - 2] : (u8)K_BAD,
                               gap_ctx(t, i));
}

// ---- tape emission ---------------------------------------------------------------------------------
// brackets and atoms (write_tape call sites of unifiedMachine).  Returns true on a violation.
SJ_HD bool emit_simple(const Tokens &t, const MsgView &m, u32 i, u64 *tape) {
    const u8 k = t.kind[i];
    const u32 o = t.tape_off[i];
    switch (k) {
    case K_OPEN_OBJ:
    case K_OPEN_ARR: {  // payload: tape index just after the matching close (annotate_previousloc, :336)
        const u32 c = t.match[i];
        const u64 after = (c < t.n) ? t.tape_base + t.tape_off[c] + 1 : 0;
        tape[o] = ((u64)(k == K_OPEN_OBJ ? '{' : '[') << 56) | after;
        return false;
    }
    case K_CLOSE_OBJ:
    case K_CLOSE_ARR: {  // payload: tape index of the matching open (:335)
        const u32 op = t.match[i];
        tape[o] = ((u64)(k == K_CLOSE_OBJ ? '}' : ']') << 56) | (op < t.n ? t.tape_base + t.tape_off[op] : 0);
        return false;
    }
    case K_TRUE:
    case K_FALSE:
    case K_NULL: {
        tape[o] = (u64)(k == K_TRUE ? 't' : (k == K_FALSE ? 'f' : 'n')) << 56;
        return !atom_valid(m, t.pos[i], k);
    }
    default: return false;
    }
}

// strings (parseString, stage2_build_tape_amd64.go:72-113).  need_copy = copyStrings || src_len != dst_len
// (parse_string_amd64.go:40).
SJ_HD void emit_string(const Tokens &t, const MsgView &m, u32 i, bool need_copy, u32 dst_len, u64 *tape, u8 *strings) {
    const u32 o = t.tape_off[i];
    const u64 q = t.pos[i];
    if (!need_copy) {
        tape[o] = ((u64)'"' << 56) | (t.msg_base + q + 1);
    } else {
        if (strings) {  // nullptr: Strings.B is written by the byte-parallel path (sj_strings.h)
            u32 sl, dl;
            string_walk(m, q, strings + t.str_off[i], &sl, &dl);
        }
        tape[o] = ((u64)'"' << 56) | (STRINGBUFBIT + t.strings_base + t.str_off[i]);
    }
    tape[o + 1] = dst_len;
}

// root words: tape[0], tape[tape_len-1] and the close/open pair written by every record-separating
// newline run (startContinue, :196-221; succeed, :428-442).  nlb[r] = token index of the r-th such
// newline; R = number of them.
SJ_HD void emit_root(const u32 *nlb, u32 R, const u32 *tape_off, u32 tape_len, u32 r_plus1, u64 *tape, u64 tape_base = 0) {
    const u64 ROOT = (u64)'r' << 56;
    const u64 B = tape_base;
    if (r_plus1 == 0) {  // first and last word
        tape[0] = ROOT | (B + (R == 0 ? tape_len : tape_off[nlb[0]] + 1));
        tape[tape_len - 1] = ROOT | (B + (R == 0 ? 0u : tape_off[nlb[R - 1]] + 1));
        return;
    }
    const u32 r = r_plus1 - 1;
    const u32 o = tape_off[nlb[r]];
    tape[o] = ROOT | (B + (r == 0 ? 0u : tape_off[nlb[r - 1]] + 1));                 // close root of record r
    tape[o + 1] = ROOT | (B + (r + 1 == R ? tape_len : tape_off[nlb[r + 1]] + 1));  // open root of record r+1
}

}  // namespace sj
