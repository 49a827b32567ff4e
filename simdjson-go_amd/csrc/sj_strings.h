// sj_strings.h -- string validation / unescape as a byte-parallel pass over the message, host+device.
//
// Replaces the per-string walks of parseString (stage2_build_tape_amd64.go:72-113) and
// parse_string_amd64.s:72-479 when every string is copied (WithCopyStrings(true), the reference's
// default): Strings.B is then the concatenation of the unescaped contents of ALL strings in document
// order, so the destination of a string byte is simply the number of emitted bytes in front of it.
//
// Stage 1 leaves, per 64-byte chunk (bit j = byte j; chunks count from the 64-byte aligned base of the
// message), qm = in-string mask relative to the start of the 4 KiB unit, q = unescaped quotes,
// st = escape starters (backslashes that are not themselves escaped), and per unit the resolved state h.
//   SM  = (h ? ~qm : qm) & ~q        bytes strictly inside strings
//   esc = st << 1 (with carry)       escaped characters
//   EM  = emit mask: one bit per byte of Strings.B.  Plain content and simple escapes emit at their own
//         position (the starter emits nothing); a \uXXXX escape owns the five positions 'u',X,X,X,X and emits
//         its n = 1..3 UTF-8 bytes at the first n of them; a surrogate pair emits 4 bytes at the high
//         half's positions and nothing at the low half's.
//   UM  = positions of the 'u' of unicode escapes inside strings.
// E(a) = emitted bytes in front of aligned offset a = unit_base[a>>12] + chunk_pre[a>>6] + popc(EM & below),
// which gives every string token its Strings.B offset and unescaped length without walking the string.
// All quirks of the reference's string parser are kept (sj_stage2.h string_walk is the per-string
// statement of the same rules and is still used for WithCopyStrings(false)).
#pragma once
#include <stdint.h>

#include "sj_chunk.h"
#include "sj_stage2.h"

namespace sj {

struct StrView {
    const u8 *base;  // 64-byte aligned base of the message
    u64 lead, end;   // the message occupies [lead, end) of it
    const u64 *qm, *q, *st;
    const u8 *unit_h;
    SJ_HD u8 at(u64 a) const { return (a >= lead && a < end) ? base[a] : (u8)0; }  // zero padding like MsgView
    SJ_HD u64 sm(u64 c) const {
        const u64 m = qm[c];
        return (unit_h[c >> 6] ? ~m : m) & ~q[c];
    }
    // escaped characters of chunk c (characters that follow a starter)
    SJ_HD u64 esc(u64 c) const { return (st[c] << 1) | (c ? st[c - 1] >> 63 : 0); }
    SJ_HD bool is_starter(u64 a) const { return (st[a >> 6] >> (a & 63)) & 1u; }
    SJ_HD bool in_string(u64 a) const { return (sm(a >> 6) >> (a & 63)) & 1u; }
};

struct UEscape {
    u32 n;     // bytes emitted at the positions of 'u' and the hex digits (0: consumed as the low half of a pair)
    u8 b[4];
    bool ok;
};

SJ_HD u32 hex4_at(const StrView &m, u64 p) {
    const u32 d0 = (u32)hex_digit(m.at(p)), d1 = (u32)hex_digit(m.at(p + 1)), d2 = (u32)hex_digit(m.at(p + 2)),
              d3 = (u32)hex_digit(m.at(p + 3));
    return (d0 << 12) | (d1 << 8) | (d2 << 4) | d3;  // sign-extended -1 poisons the high bits
}

// is the unicode escape whose backslash is at `pos` a high surrogate (cp in D800..DBFF)?
SJ_HD bool is_high_surrogate_escape(const StrView &m, u64 pos) {
    return m.at(pos + 1) == 'u' && (hex4_at(m, pos + 2) & 0xfffffc00u) == 0xd800u;
}

// The escape whose 'u' sits at aligned offset au (its starter at au - 1), restating the \u branch of
// string_walk / parse_string_amd64.s: quote distance rule, the hex table quirk, surrogate pairs with an
// unchecked low half, code points above 0x10ffff rejected.
SJ_HD UEscape unicode_escape(const StrView &m, u64 au) {
    UEscape r;
    r.n = 0;
    r.ok = true;
    r.b[0] = r.b[1] = r.b[2] = r.b[3] = 0;
    const u64 pos = au - 1;
    // consumed as the low half of a pair?  count the unconsumed high-surrogate escapes directly in front
    // (6 bytes apart, each a real escape starter inside the string): odd => this one is a low half
    u32 highs = 0;
    for (u64 p = pos; p >= m.lead + 6; p -= 6) {
        const u64 pp = p - 6;
        if (!(m.is_starter(pp) && m.in_string(pp + 1) && is_high_surrogate_escape(m, pp))) break;
        highs++;
    }
    if (highs & 1u) return r;  // validated (as far as the reference validates it) by the high half
    u32 d = 12;
    for (u32 j = 1; j < 12; j++)
        if (m.at(pos + j) == '"') {
            d = j;
            break;
        }
    if (d < 6) {
        r.ok = false;
        return r;
    }
    u32 cp = hex4_at(m, pos + 2);
    if ((cp & 0xfffffc00u) == 0xd800u) {
        if (d < 12 || m.at(pos + 6) != '\\' || m.at(pos + 7) != 'u') {
            r.ok = false;
            return r;
        }
        const u32 cp2 = hex4_at(m, pos + 8);
        if ((cp | cp2) > 0xffffu) {
            r.ok = false;
            return r;
        }
        cp = (((cp << 10) + 0xfca00000u) | (cp2 + 0xffff2400u)) + 0x10000u;  // 32-bit wrap-around, low half unchecked
    }
    if (cp < 0x80u) {
        r.n = 1;
        r.b[0] = (u8)cp;
    } else if (cp < 0x800u) {
        r.n = 2;
        r.b[0] = (u8)((cp >> 6) + 192);
        r.b[1] = (u8)((cp & 63) | 128);
    } else if (cp < 0x10000u) {
        r.n = 3;
        r.b[0] = (u8)((cp >> 12) + 224);
        r.b[1] = (u8)(((cp >> 6) & 63) | 128);
        r.b[2] = (u8)((cp & 63) | 128);
    } else if (cp <= 0x10ffffu) {
        r.n = 4;
        r.b[0] = (u8)((cp >> 18) + 240);
        r.b[1] = (u8)(((cp >> 12) & 63) | 128);
        r.b[2] = (u8)(((cp >> 6) & 63) | 128);
        r.b[3] = (u8)((cp & 63) | 128);
    } else {
        r.ok = false;
    }
    return r;
}

// Pass 1, one chunk: the emit mask and the 'u' mask of chunk c; returns false on an invalid escape.
SJ_HD bool str_chunk_masks(const StrView &m, u64 c, u64 *em_out, u64 *um_out) {
    const u64 sm = m.sm(c);
    const u64 e = m.esc(c) & sm;  // escaped characters inside strings
    u64 em = sm & ~m.st[c];
    u64 um = 0;
    bool ok = true;
    // escapes whose 'u' lies in the last four bytes of the previous chunk reach into this one
    if (c > 0) {
        u64 pe = (m.esc(c - 1) & m.sm(c - 1)) >> 60;
        for (u32 k = 0; pe != 0; k++, pe >>= 1) {
            if (!(pe & 1u)) continue;
            const u64 au = (c - 1) * 64 + 60 + k;
            if (m.at(au) != 'u') continue;
            const UEscape u = unicode_escape(m, au);  // errors are reported by the owner of the 'u'
            for (u32 j = 1; j <= 4; j++) {
                const u64 a = au + j;
                if ((a >> 6) == c) {
                    const u64 bit = 1ull << (a & 63);
                    em = (j < u.n) ? (em | bit) : (em & ~bit);
                }
            }
        }
    }
    for (u64 r = e; r != 0; r &= r - 1) {
        const u32 p = (u32)ctz64(r);
        const u64 a = c * 64 + p;
        const u8 b = m.at(a);
        if (b != 'u') {
            if (escape_value(b) == 0) ok = false;
            continue;
        }
        um |= 1ull << p;
        const UEscape u = unicode_escape(m, a);
        if (!u.ok) ok = false;
        for (u32 j = 0; j <= 4; j++) {
            const u64 aj = a + j;
            if ((aj >> 6) == c) {
                const u64 bit = 1ull << (aj & 63);
                em = (j < u.n) ? (em | bit) : (em & ~bit);
            }
        }
    }
    *em_out = em;
    *um_out = um;
    return ok;
}

// Pass 2, one chunk: writes the popc(em) bytes of chunk c to dst (ascending positions).
// em / um are the masks of pass 1 (um of the previous chunk for escapes that reach into this one);
// byte_at(p) returns byte p of the chunk (the kernel serves it from LDS, the host replay from memory).
template <typename ByteAt>
SJ_HD void str_chunk_emit(const StrView &m, u64 c, u64 em, u64 um, u64 um_prev, u8 *dst, ByteAt byte_at) {
    const u64 e = m.esc(c);
    // positions 1..4 behind a 'u' (hex digit slots): handled by the unicode path
    const u64 uh = (um << 1) | (um << 2) | (um << 3) | (um << 4) | (um_prev >> 63) | (um_prev >> 62) | (um_prev >> 61) |
                   (um_prev >> 60);
    u32 o = 0;
    for (u64 r = em; r != 0; r &= r - 1) {
        const u32 p = (u32)ctz64(r);
        const u64 a = c * 64 + p;
        const u64 bit = 1ull << p;
        u8 v;
        if ((um | uh) & bit) {
            u64 au = a;  // find the 'u' this position belongs to (itself or 1..4 positions back)
            u32 j = 0;
            if (!(um & bit)) {
                for (j = 1; j <= 4; j++) {
                    const u64 cand = a - j;
                    const u64 cm = (cand >> 6) == c ? um : um_prev;
                    if ((cm >> (cand & 63)) & 1u) break;
                }
                au = a - j;
            }
            v = unicode_escape(m, au).b[j];
        } else if (e & bit) {
            v = escape_value(byte_at(p));
        } else {
            v = byte_at(p);
        }
        dst[o++] = v;
    }
}

// E(a): emitted bytes in front of aligned offset a
SJ_HD u64 emitted_before(const u32 *unit_base, const uint16_t *chunk_pre, const u64 *em, u64 a) {
    const u64 c = a >> 6;
    const u32 bit = (u32)(a & 63);
    const u64 below = bit ? (em[c] & (~0ull >> (64 - bit))) : 0ull;
    return (u64)unit_base[a >> 12] + chunk_pre[c] + (u64)popc64(below);
}

}  // namespace sj
