/*
 * sjo_parse_string.c -- ORACLE (test infrastructure only, see sjo.h).
 * Restatement of parse_string_amd64.s (_parse_string_validate_only :72-258,
 * _parse_string :260-479) including its DATA tables (:4-70).
 *
 * `src` points at the first byte after the opening quote; `avail` is the number
 * of readable message bytes from there.  Bytes beyond `avail` read as 0, which
 * is what the Go caller provides by copying the tail of the message into a
 * zeroed buffer (stage2_build_tape_amd64.go:75-86).
 */
#include "sjo.h"
#include "sjo_internal.h"

#include <string.h>

/* digittoval as laid out by the DATA section (parse_string_amd64.s:12-37): the table
 * starts at LCDATA1+0x40 but no DATA is emitted for +0x40..+0x6f, so entries for
 * bytes 0x00..0x2f are ZERO (not -1 as in the C original).  Quirk Q3 -- parity unpinned. */
static int8_t digittoval(uint8_t b) {
    if (b < 0x30) return 0;
    if (b >= '0' && b <= '9') return (int8_t)(b - '0');
    if (b >= 'A' && b <= 'F') return (int8_t)(b - 'A' + 10);
    if (b >= 'a' && b <= 'f') return (int8_t)(b - 'a' + 10);
    return -1;
}

/* escape_map (parse_string_amd64.s:38-69) */
static uint8_t escape_map(uint8_t b) {
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

static inline uint8_t rd(const uint8_t *src, size_t avail, size_t i) { return i < avail ? src[i] : 0; }

/* movemask over a 32-byte window (VPCMPEQB + VPMOVMSKB) */
static uint32_t win_mask(const uint8_t *src, size_t avail, size_t pos, uint8_t c) {
    uint32_t m = 0;
    for (int j = 0; j < 32; j++)
        if (rd(src, avail, pos + (size_t)j) == c) m |= 1u << j;
    return m;
}

/* movsx-extended 4-hex-digit code point (LBB0_11 / LBB0_14) */
static uint32_t hex4(const uint8_t *src, size_t avail, size_t p) {
    int32_t d0 = digittoval(rd(src, avail, p + 0));
    int32_t d1 = digittoval(rd(src, avail, p + 1));
    int32_t d2 = digittoval(rd(src, avail, p + 2));
    int32_t d3 = digittoval(rd(src, avail, p + 3));
    return ((uint32_t)d0 << 12) | ((uint32_t)d1 << 8) | ((uint32_t)d2 << 4) | (uint32_t)d3;
}

/* Common walk.  If dst != NULL the unescaped bytes are written (the reference copies whole
 * 32-byte YMM words and patches; the visible result is the same byte sequence). */
int sjo_string_walk_from(const uint8_t *src, size_t avail, uint8_t *dst, size_t pos, size_t out, uint64_t *str_length,
                         uint64_t *dst_length) {
    /* pos: r13 - rdi; out: r14 (validate) / rsi - dst (parse) */
    for (;;) {
        if (pos > avail + 64) return 0; /* oracle guard: the reference would run off the buffer */
        uint32_t bs_bits = win_mask(src, avail, pos, '\\');
        uint32_t quote_bits = win_mask(src, avail, pos, '"');
        if (((bs_bits - 1) & quote_bits) != 0) { /* LBB0_3: quote before any backslash */
            unsigned q = (unsigned)__builtin_ctz(quote_bits);
            if (dst)
                for (unsigned j = 0; j < q; j++) dst[out + j] = rd(src, avail, pos + j);
            if (str_length) *str_length = pos + q;
            *dst_length = out + q;
            return 1;
        }
        if (((quote_bits - 1) & bs_bits) == 0) { /* LBB0_28: neither in this window */
            if (dst)
                for (unsigned j = 0; j < 32; j++) dst[out + j] = rd(src, avail, pos + j);
            pos += 32;
            out += 32;
            continue;
        }
        unsigned b = (unsigned)__builtin_ctz(bs_bits); /* r15 */
        uint8_t esc = rd(src, avail, pos + b + 1);
        if (dst)
            for (unsigned j = 0; j < b; j++) dst[out + j] = rd(src, avail, pos + j);
        if (esc != 'u') { /* LBB0_26 */
            uint8_t e = escape_map(esc);
            if (e == 0) return 0;
            if (dst) dst[out + b] = e;
            out += b + 1;
            pos += b + 2;
            continue;
        }
        /* distance from the backslash to the next raw quote (LBB0_8/LBB0_10) */
        uint32_t d;
        if (quote_bits != 0) {
            d = (uint32_t)__builtin_ctz(quote_bits) - b;
        } else {
            d = 32;
            if (b >= 21) {
                uint32_t q2 = win_mask(src, avail, pos + b - 20, '"');
                uint32_t t = q2 ? (uint32_t)__builtin_ctz(q2) : 32u;
                d = t + b - 20;
            }
            d -= b;
        }
        if (d < 6) return 0;
        size_t p = pos + b; /* LBB0_11: r13 += r15 -> the backslash */
        uint32_t cp = hex4(src, avail, p + 2);
        size_t next = p + 6;
        if ((cp & 0xfffffc00u) == 0xd800u) { /* LBB0_12 */
            if (d < 12) return 0;
            if (rd(src, avail, p + 6) != '\\') return 0;
            if (rd(src, avail, p + 7) != 'u') return 0;
            uint32_t cp2 = hex4(src, avail, p + 8);
            if ((cp2 | cp) > 0xffffu) return 0;
            cp = (cp << 10) + 0xfca00000u; /* add r12d, -56623104 */
            cp2 = cp2 + 0xffff2400u;       /* add esi, -56320 */
            cp = (cp2 | cp) + 0x10000u;
            next = p + 12;
        }
        unsigned n;
        uint8_t enc[4];
        if (cp < 0x80) {
            n = 1;
            enc[0] = (uint8_t)cp;
        } else if (cp < 0x800) {
            n = 2;
            enc[0] = (uint8_t)((cp >> 6) + 192);
            enc[1] = (uint8_t)((cp & 63) | 128);
        } else if (cp < 0x10000) {
            n = 3;
            enc[0] = (uint8_t)((cp >> 12) + 224);
            enc[1] = (uint8_t)(((cp >> 6) & 63) | 128);
            enc[2] = (uint8_t)((cp & 63) | 128);
        } else if (cp <= 0x10ffff) {
            n = 4;
            enc[0] = (uint8_t)((cp >> 18) + 240);
            enc[1] = (uint8_t)(((cp >> 12) & 63) | 128);
            enc[2] = (uint8_t)(((cp >> 6) & 63) | 128);
            enc[3] = (uint8_t)((cp & 63) | 128);
        } else {
            return 0;
        }
        if (dst) memcpy(dst + out + b, enc, n);
        out += b + n;
        pos = next;
    }
}

int sjo_parse_string_validate_only(const uint8_t *src, size_t avail, uint64_t *str_length, uint64_t *dst_length) {
    return sjo_string_walk_from(src, avail, NULL, 0, 0, str_length, dst_length);
}

int sjo_parse_string(const uint8_t *src, size_t avail, uint8_t *dst, uint64_t *dst_length) {
    return sjo_string_walk_from(src, avail, dst, 0, 0, NULL, dst_length);
}
