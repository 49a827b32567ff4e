// sj_strings.h -- string validation / unescape as a byte-parallel pass over the message, host+device.
//
// Replaces the per-string walks of parseString (stage2_build_tape_amd64.go:72-113) and
// parse_string_amd64.s:72-479 when every string is copied (WithCopyStrings(true), the reference's
// default): Strings.B is then the concatenation of the unescaped contents of ALL strings in document
// order, so the destination of a string byte is simply the number of emitted bytes in front of it.
//
// Stage 1 leaves, per 64-byte chunk (bit j = byte j; chunks count from the 64-byte aligned base of the
// message), qm = in-string mask relative to the start of the 4 KiB unit, st = escape starters (backslashes that are not
// themselves escaped), and per unit the resolved state h.  The unescaped quotes q are NOT stored (round 6: 8 of the 24 bytes
// per chunk that stage 1 wrote and the string kernels read back): qm is the running parity of the quotes inside the unit, so
// a quote sits exactly where qm changes -- q = qm ^ (qm << 1 | last bit of the chunk in front; 0 at the start of a unit).
//   SM  = (h ? ~qm : qm) & ~q        bytes strictly inside strings
//   esc = st << 1 (with carry)       escaped characters
//   EM  = emit mask: one bit per byte of Strings.B.  Plain content and simple escapes emit at their own
//         position (the starter emits nothing); a \uXXXX escape owns the five positions 'u',X,X,X,X and emits
//         its n = 1..3 UTF-8 bytes at the first n of them; a surrogate pair emits 4 bytes at the high
//         half's positions and nothing at the low half's.
//   UM  = positions of the 'u' of unicode escapes inside strings (kept for inspection; pass 2 re-derives them).
// E(a) = emitted bytes in front of aligned offset a = unit_base[a>>12] + chunk_pre[a>>6] + popc(EM & below),
// which gives every string its Strings.B offset and unescaped length without walking it: k_str_emit evaluates E at the
// opening quote of every string and leaves it under the string's number (round 5: soff[]; the tape kernels read it in order).
// All quirks of the reference's string parser are kept (sj_stage2.h string_walk is the per-string
// statement of the same rules and is still used for WithCopyStrings(false)).
#pragma once
#include <stdint.h>

#include "sj_chunk.h"
#include "sj_stage2.h"

namespace sj {

struct StrView {
    Arr<const u8> base;  // 64-byte aligned base of the message (Arr: sj_bounds.h, a plain pointer in the product build)
    u64 lead, end;   // the message occupies [lead, end) of it
    Arr<const u64> qm, st;
    Arr<const u8> unit_h;  // per unit: bit 0 = state at its start (1: inside a string), bit 1 = it holds an escape starter, bit 2 = an unescaped quote
    Arr<const u64> unit_slow;  // per unit: chunks that hold an escaped character other than " \\ / b f n r t (stage 1)
    SJ_HD u8 at(u64 a) const { return (a >= lead && a < end) ? base[a] : (u8)0; }  // zero padding like MsgView
    // the 16 bytes at a .. a+15 as two little-endian words (two unaligned 8-byte loads away from the message ends)
    SJ_HD void window16(u64 a, u64 &w0, u64 &w1) const {
        if (a >= lead && a + 16 <= end) {
            const u8 *w = arr_at(base, a, 16);
            w0 = load_u64(w);
            w1 = load_u64(w + 8);
        } else {
            w0 = w1 = 0;
            for (u32 k = 0; k < 8; k++) {
                w0 |= (u64)at(a + k) << (8 * k);
                w1 |= (u64)at(a + 8 + k) << (8 * k);
            }
        }
    }
    // unescaped quotes of chunk c: where the in-string mask changes (its value in front of a unit's first byte is 0)
    SJ_HD static u64 quotes_of(u64 m, u64 m_prev) { return m ^ ((m << 1) | (m_prev >> 63)); }
    SJ_HD u64 quotes(u64 c) const { return quotes_of(qm[c], (c & 63) ? qm[c - 1] : 0ull); }
    SJ_HD u64 sm(u64 c) const {
        const u64 m = qm[c];
        return ((unit_h[c >> 6] & 1u) ? ~m : m) & ~quotes_of(m, (c & 63) ? qm[c - 1] : 0ull);  // (bit 1 of unit_h: the unit holds an escape starter)
    }
    // escaped characters of chunk c (characters that follow a starter)
    SJ_HD u64 esc(u64 c) const { return (st[c] << 1) | (c ? st[c - 1] >> 63 : 0); }
    // does chunk c hold an escaped character that no simple escape names (hypothesis-free)?
    SJ_HD bool nonsimple(u64 c) const { return (unit_slow[c >> 6] >> (c & 63)) & 1u; }
    SJ_HD bool is_starter(u64 a) const { return (st[a >> 6] >> (a & 63)) & 1u; }
    SJ_HD bool in_string(u64 a) const { return (sm(a >> 6) >> (a & 63)) & 1u; }
};

struct UEscape {
    u32 n;     // bytes emitted at the positions of 'u' and the hex digits (0: consumed as the low half of a pair)
    u8 b[4];
    bool ok;
    bool overflow;  // the run of high-surrogate escapes in front is longer than SURROGATE_WALK_CAP: not decided here
};
// Bound of the backward walk over adjacent high-surrogate escapes (each escape of such a run walks to the start of
// the run: quadratic in the run length).  Real text stops after one or two steps; a document with a longer run is
// handed to the per-string path (string_walk: linear), see S2_ERR_SERIAL_STRINGS.
static constexpr u32 SURROGATE_WALK_CAP = 4096;

SJ_HD u32 hex4_of(u32 four) {  // four bytes, first digit in the low byte
    const u32 d0 = (u32)hex_digit((u8)four), d1 = (u32)hex_digit((u8)(four >> 8)), d2 = (u32)hex_digit((u8)(four >> 16)),
              d3 = (u32)hex_digit((u8)(four >> 24));
    return (d0 << 12) | (d1 << 8) | (d2 << 4) | d3;  // sign-extended -1 poisons the high bits
}

// is the unicode escape whose backslash is at `pos` a high surrogate (cp in D800..DBFF)?
SJ_HD bool is_high_surrogate_escape(const StrView &m, u64 pos) {
    u64 w0, w1;
    m.window16(pos, w0, w1);
    return (u8)(w0 >> 8) == 'u' && (hex4_of((u32)(w0 >> 16)) & 0xfffffc00u) == 0xd800u;
}

// The escape whose 'u' sits at aligned offset au (its starter at au - 1), restating the \u branch of
// string_walk / parse_string_amd64.s: quote distance rule, the hex table quirk, surrogate pairs with an
// unchecked low half, code points above 0x10ffff rejected.  Works on the 16 bytes from the backslash on.
SJ_HD UEscape unicode_escape(const StrView &m, u64 au) {
    UEscape r;
    r.n = 0;
    r.ok = true;
    r.overflow = false;
    r.b[0] = r.b[1] = r.b[2] = r.b[3] = 0;
    const u64 pos = au - 1;
    // consumed as the low half of a pair?  count the unconsumed high-surrogate escapes directly in front
    // (6 bytes apart, each a real escape starter inside the string): odd => this one is a low half
    u32 highs = 0;
    for (u64 p = pos; p >= m.lead + 6; p -= 6) {
        const u64 pp = p - 6;
        if (!(m.is_starter(pp) && m.in_string(pp + 1) && is_high_surrogate_escape(m, pp))) break;
        if (++highs > SURROGATE_WALK_CAP) {
            r.overflow = true;
            return r;
        }
    }
    if (highs & 1u) return r;  // validated (as far as the reference validates it) by the high half
    u64 w0, w1;
    m.window16(pos, w0, w1);
    // distance to the next raw quote among bytes 1..11 (12: none)
    const u64 Q8 = 0x2222222222222222ull;
    const u64 z0 = zero_bytes(w0 ^ Q8) & ~0xffull, z1 = zero_bytes(w1 ^ Q8) & 0x00000000ffffffffull;
    const u32 d = z0 ? (u32)ctz64(z0) >> 3 : (z1 ? 8u + ((u32)ctz64(z1) >> 3) : 12u);
    if (d < 6) {
        r.ok = false;
        return r;
    }
    u32 cp = hex4_of((u32)(w0 >> 16));  // bytes 2..5
    if ((cp & 0xfffffc00u) == 0xd800u) {
        if (d < 12 || (u8)(w0 >> 48) != '\\' || (u8)(w0 >> 56) != 'u') {
            r.ok = false;
            return r;
        }
        const u32 cp2 = hex4_of((u32)w1);  // bytes 8..11
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

// What pass 1 leaves per 64-byte chunk, one 16-byte record (one load for everything a consumer needs):
//   em   emit mask
//   pre  bits 0-14 the emitted bytes of the unit in front of the chunk, bit 15 "the chunk needs patching in
//        pass 2" (it holds an escape, or a unicode escape of the previous chunk reaches into it)
//   abs  the absolute Strings.B offset of the chunk's first emitted byte (unit prefix + pre), filled in by
//        k_str_emit once the unit scan has run
struct alignas(16) ChunkRec {
    u64 em;
    u32 pre;
    u32 abs;
};
// CHUNK_GENERAL: the chunk (or the one in front of it) holds an escaped character that no simple escape names: pass 2
// patches it with the general routine; without it the escapes of a CHUNK_SLOW chunk are simple ones
static constexpr u32 CHUNK_PRE_MASK = 0x7fffu, CHUNK_SLOW = 0x8000u, CHUNK_GENERAL = 0x10000u;
SJ_HD bool str_chunk_has_escapes(const StrView &m, u64 c) {
    if ((m.esc(c) & m.sm(c)) != 0) return true;
    return c > 0 && ((m.esc(c - 1) & m.sm(c - 1)) >> 60) != 0;
}

// Pass 1, one chunk: the emit mask and the 'u' mask of chunk c; returns false on an invalid escape.  *overflow_out
// is set when a surrogate run exceeds SURROGATE_WALK_CAP (the masks are then not valid).
SJ_HD bool str_chunk_masks(const StrView &m, u64 c, u64 *em_out, u64 *um_out, bool *escapes_out = nullptr,
                           bool *overflow_out = nullptr) {
    const u64 sm = m.sm(c);
    const u64 e = m.esc(c) & sm;  // escaped characters inside strings
    u64 em = sm & ~m.st[c];
    u64 um = 0;
    bool ok = true;
    bool escapes = e != 0;  // == str_chunk_has_escapes(m, c)
    bool overflow = false;
    // escapes whose 'u' lies in the last four bytes of the previous chunk reach into this one
    if (c > 0) {
        u64 pe = (m.esc(c - 1) & m.sm(c - 1)) >> 60;
        escapes |= pe != 0;
        for (u32 k = 0; pe != 0; k++, pe >>= 1) {
            if (!(pe & 1u)) continue;
            const u64 au = (c - 1) * 64 + 60 + k;
            if (m.at(au) != 'u') continue;
            const UEscape u = unicode_escape(m, au);  // errors are reported by the owner of the 'u'
            overflow |= u.overflow;
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
        overflow |= u.overflow;
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
    if (escapes_out) *escapes_out = escapes;
    if (overflow_out) *overflow_out = overflow;
    return ok;
}

// Pass 1 without touching the message: a chunk takes the general routine above only if it or the chunk in front of it
// holds an escaped character that no simple escape names (a \u, whose bytes may reach into this chunk, or an invalid
// escape).  Otherwise every escape of the chunk is a simple one -- valid by construction, the starter emits nothing, the
// escaped character emits one byte at its own position -- and the emit mask is SM & ~st.
SJ_HD bool str_chunk_needs_general(const StrView &m, u64 c) {
    return (m.nonsimple(c) && (m.esc(c) & m.sm(c)) != 0) || (c > 0 && m.nonsimple(c - 1));
}
SJ_HD bool str_chunk_masks_fast(const StrView &m, u64 c, u64 *em_out, u32 *flags_out, bool *overflow_out) {
    *overflow_out = false;
    if (!str_chunk_needs_general(m, c)) {
        const u64 sm = m.sm(c);
        *em_out = sm & ~m.st[c];
        *flags_out = (m.esc(c) & sm) != 0 ? CHUNK_SLOW : 0u;
        return true;
    }
    u64 um;
    bool escapes;
    const bool ok = str_chunk_masks(m, c, em_out, &um, &escapes, overflow_out);
    *flags_out = escapes ? (CHUNK_SLOW | CHUNK_GENERAL) : 0u;
    return ok;
}
// The same for one chunk from its three masks alone (every string copied, second half of round 5: k_str_emit derives this
// per chunk instead of reading a record): qm / q / st of the chunk, st of the chunk in front, h = state at the start of
// the unit.  esc = the escaped characters inside strings (they are all emitted: an escaped character is neither an
// unescaped quote nor a starter); oq = the OPENING quotes -- qm includes the opening and excludes the closing quote of
// every string.  String number k of the message is the k-th opening quote and the k-th string token.
struct ChunkFast {
    u64 em, esc, oq;
};
SJ_HD ChunkFast chunk_fast(u64 qm, u64 q, u64 st, u64 st_prev, u32 h) {
    const u64 in = h ? ~qm : qm;
    const u64 sm = in & ~q;
    ChunkFast r;
    r.em = sm & ~st;
    r.esc = ((st << 1) | (st_prev >> 63)) & sm;
    r.oq = q & in;
    return r;
}
// Pass 2 for a chunk whose escapes are all simple (CHUNK_SLOW without CHUNK_GENERAL): the escaped characters inside
// strings -- they are all emitted -- are translated in place.  put(p, v) as below; at(p) = the chunk's byte p.
template <typename At, typename Put>
SJ_HD void str_chunk_patch_simple(const StrView &m, u64 c, u64 em, At at, Put put) {
    for (u64 r = m.esc(c) & em; r != 0; r &= r - 1) {
        const u32 p = (u32)ctz64(r);
        put(p, escape_value(at(p)));
    }
}

// Pass 2, one chunk: Strings.B receives the bytes of chunk c selected by its emit mask, in order, after the
// bytes that differ from the message have been patched: put(p, v) is called for every position p (0..63) of the
// chunk whose emitted byte is v instead of the message byte -- the character behind a simple escape, and the
// first n positions of 'u',X,X,X,X of a unicode escape (also of one that starts in the last four bytes of the
// previous chunk).  Same traversal as pass 1.
template <typename Put>
SJ_HD void str_chunk_patch(const StrView &m, u64 c, Put put) {
    if (c > 0) {
        u64 pe = (m.esc(c - 1) & m.sm(c - 1)) >> 60;
        for (u32 k = 0; pe != 0; k++, pe >>= 1) {
            if (!(pe & 1u)) continue;
            const u64 au = (c - 1) * 64 + 60 + k;
            if (m.at(au) != 'u') continue;
            const UEscape u = unicode_escape(m, au);
            for (u32 j = 1; j <= 4; j++) {
                const u64 a = au + j;
                if ((a >> 6) == c && j < u.n) put((u32)(a & 63), u.b[j]);
            }
        }
    }
    for (u64 r = m.esc(c) & m.sm(c); r != 0; r &= r - 1) {
        const u32 p = (u32)ctz64(r);
        const u64 a = c * 64 + p;
        const u8 b = m.at(a);
        if (b != 'u') {
            put(p, escape_value(b));
            continue;
        }
        const UEscape u = unicode_escape(m, a);
        for (u32 j = 0; j <= 4; j++) {
            const u64 aj = a + j;
            if ((aj >> 6) == c && j < u.n) put((u32)(aj & 63), u.b[j]);
        }
    }
}

// ---- the general routine escape by escape (k_measure / k_str_emit: stage2.hip GenUnit; host replay: host_selftest.cpp) ---
// A unit's escaped in-string characters that need the general routine are listed and evaluated one by one, by any lane,
// in any order.  item = byte offset of the escaped character inside the unit that starts at aligned offset u0, or
// GEN_FOREIGN | k: escaped character k (0..3) of the last four bytes of the chunk in front of the unit (its bytes may
// reach into chunk 0; errors of a foreign item are reported by the unit that owns it).
// Mask pass: the fast formula has the bits of u,X,X,X,X set (plain in-string bytes); a valid escape that emits n bytes
// clears the bits j >= n -- clear(position inside the unit); overlapping escapes only exist in rejected documents.
static constexpr u32 GEN_FOREIGN = 0x8000u;
template <typename Clear>
SJ_HD void gen_item_masks(const StrView &m, u64 u0, u32 item, bool *bad, bool *over, Clear clear) {
    const bool foreign = (item & GEN_FOREIGN) != 0;
    const u64 a = foreign ? u0 - 4 + (item & 3u) : u0 + item;
    const u8 b = m.at(a);
    if (b != 'u') {
        if (!foreign && escape_value(b) == 0) *bad = true;
        return;
    }
    const UEscape ue = unicode_escape(m, a);
    if (!foreign) {
        if (!ue.ok) *bad = true;
        if (ue.overflow) *over = true;
    }
    if (!ue.ok || ue.overflow) return;  // (the parse fails or is repeated: the masks do not matter)
    for (u32 k = ue.n; k <= 4; k++) {
        const u64 ak = a + k;
        if (ak >= u0 && ak < u0 + 4096) clear((u32)(ak - u0));
    }
}
// Writing pass: the translated byte of a simple escape, the n bytes of a \u escape -- put(position inside the unit, byte)
template <typename Put>
SJ_HD void gen_item_patch(const StrView &m, u64 u0, u32 item, Put put) {
    const bool foreign = (item & GEN_FOREIGN) != 0;
    const u64 a = foreign ? u0 - 4 + (item & 3u) : u0 + item;
    const u8 b = m.at(a);
    if (b != 'u') {
        if (!foreign) put((u32)(a - u0), escape_value(b));
        return;
    }
    const UEscape ue = unicode_escape(m, a);
    for (u32 k = 0; k < ue.n; k++) {
        const u64 ak = a + k;
        if (ak >= u0 && ak < u0 + 4096) put((u32)(ak - u0), ue.b[k]);
    }
}

// E(a): emitted bytes in front of aligned offset a
SJ_HD u64 emitted_before(Arr<const u32> unit_base, Arr<const ChunkRec> rec, u64 a) {
    const ChunkRec r = rec[a >> 6];
    const u32 bit = (u32)(a & 63);
    const u64 below = bit ? (r.em & (~0ull >> (64 - bit))) : 0ull;
    return (u64)unit_base[a >> 12] + (r.pre & CHUNK_PRE_MASK) + (u64)popc64(below);
}

// ---- WithCopyStrings(false), byte-parallel (second half of round 5) ------------------------------------------------------
// parseString sends a string to Strings.B only if unescaping changed it (parseStringSimdValidateOnly + the src_len != dst_len
// rule, parse_string_amd64.go:33-42, stage2_build_tape_amd64.go:90-109), i.e. iff it holds an escape starter.  Strings.B is
// then the concatenation of the unescaped contents of THOSE strings -- still a property of the message: the emit mask
// restricted to the bytes of strings that hold a starter.  Which bytes those are is a segmented OR over the bytes of a
// string, evaluated 64 bytes at a time:
//   in    in-string mask of the chunk with the opening and without the closing quote of every string (qm under h)
//   mark  escape starters inside strings
//   a run of ones of `in` is the part of one string that lies in this chunk; a run with a mark is filled from the mark to
//   both ends by one addition each way (the carry of in + mark runs through the ones above the mark; the bytes below it
//   through the same addition on the bit-reversed words);
//   the run that touches byte 0 without starting there (head run: the string was open in front of the chunk) and the
//   run that touches byte 63 (tail run) also depend on the other chunks of their string: F = "the string open at the start
//   of the chunk holds a mark in front of the chunk", G = "the string open at the end of the chunk holds one behind it" --
//   two scans over the chunks (forward / backward) of the one-bit functions sel_fwd / sel_bwd, which compose like the
//   context functions of the token scan (x -> (x & a) | b).
// The k-th string of the message (k-th opening quote, k-th string token) is copied iff bytes are selected between its quotes;
// its Strings.B offset is the number of selected emit-mask bits in front of its opening quote, its unescaped length the
// selected bits up to the next opening quote, and the raw length of a string that is NOT copied is the distance to its
// closing quote -- the k-th closing quote of the message (strings do not nest).  k_str_emit leaves soff[k] and the position
// of the k-th closing quote in message order; k_s2_emit_planes reads both in token order.  No per-string walk, no measuring
// pass over the tokens, nothing is compacted that is not copied.
SJ_HD u64 brev64(u64 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brevll(x);
#else
    u64 r = 0;
    for (int i = 0; i < 64; i++) r |= ((x >> i) & 1ull) << (63 - i);
    return r;
#endif
}
// bits of the runs of ones of R at or above a bit of M inside the same run (M a subset of R)
SJ_HD u64 fill_up(u64 R, u64 M) { return (((R + M) ^ R) & R) | M; }
struct ChunkSel {
    u64 in, oq, cq;     // in-string mask (see above), opening quotes, closing quotes
    u64 hr, tr;         // head run (empty unless the string was open in front of the chunk), tail run (empty unless open behind it)
    u64 local;          // the bytes of runs that hold a mark inside this chunk
    u32 fwd, bwd;       // the scan elements: bit 0 = a, bit 1 = b of x -> (x & a) | b
};
SJ_HD ChunkSel chunk_sel(u64 qm, u64 q, u64 st, u32 h) {
    ChunkSel r;
    r.in = h ? ~qm : qm;
    r.oq = q & r.in;
    r.cq = q & ~r.in;
    const u64 R = r.in, M = st & R;
    const u64 Rr = brev64(R);
    r.local = fill_up(R, M) | brev64(fill_up(Rr, brev64(M)));
    const bool head_open = (R & ~r.oq & 1ull) != 0, tail_open = (R >> 63) != 0;
    r.hr = head_open ? R & ~(R + 1ull) : 0ull;                  // the trailing ones of R
    r.tr = tail_open ? brev64(Rr & ~(Rr + 1ull)) : 0ull;         // the leading ones of R
    const bool through = head_open && R == ~0ull;               // one string from the first to the last byte (then oq == 0)
    const u32 any = M != 0 ? 1u : 0u, hm = (r.hr & M) != 0 ? 1u : 0u, tm = (r.tr & M) != 0 ? 1u : 0u;
    // forward: state "the string open here holds a mark in front" at the start of the chunk -> at its end
    r.fwd = tail_open ? (through ? 1u | (any << 1) : tm << 1) : 0u;
    // backward: state "the string open here holds a mark behind" at the end of the chunk -> at its start
    r.bwd = head_open ? (through ? 1u | (any << 1) : hm << 1) : 0u;
    return r;
}
SJ_HD u32 sel_apply(u32 f, u32 x) { return (x & f & 1u) | (f >> 1); }
SJ_HD u32 sel_then(u32 first, u32 second) {  // first, then second
    const u32 a = first & second & 1u, b = ((first >> 1) & second & 1u) | (second >> 1);
    return a | (b << 1);
}
// the bytes of the chunk that belong to strings which are copied; F / G: the states at the start / at the end of the chunk
SJ_HD u64 chunk_sel_mask(const ChunkSel &c, u32 F, u32 G) { return c.local | (F ? c.hr : 0ull) | (G ? c.tr : 0ull); }

// the same from an absolute chunk offset in the record (rounds 3-4: what the emit pass gathered per string; the host replay
// still holds soff[] against both forms)
SJ_HD u64 emitted_before_abs(Arr<const ChunkRec> rec, u64 a) {
    const ChunkRec r = rec[a >> 6];
    const u32 bit = (u32)(a & 63);
    const u64 below = bit ? (r.em & (~0ull >> (64 - bit))) : 0ull;
    return (u64)r.abs + (u64)popc64(below);
}

}  // namespace sj
