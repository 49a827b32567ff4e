// sj_number.h -- number parsing for stage 2 (one number per lane), host+device.
//
// Restates parseNumber (parse_number.go:65-135).  The reference hands the actual conversion to
// Go's strconv (ParseInt / ParseUint / ParseFloat, call sites parse_number.go:105,114,130);
// here that arithmetic is implemented for the GPU:
//   * integers: plain uint64 accumulation with range checks;
//   * floats:   Go's decimal float grammar (strconv/atof.go readFloat, restricted to the
//               characters isNumberRune admits), then
//                 1. Clinger's exact fast path (mantissa < 2^53, |exp10| <= 22),
//                 2. Eisel-Lemire with a 128-bit 5^q table (always decisive for <= 19 digits),
//                 3. for longer mantissas: Eisel-Lemire on the truncated mantissa w and on w+1;
//                    if they disagree the lane reports NUM_NEEDS_BIGNUM and the exact big-integer
//                    comparison in sj_bignum.h decides (still on the GPU).
//   The result must be the correctly rounded (round-half-even) binary64, which is what
//   strconv.ParseFloat returns; |x| >= 2^1024 after rounding is ErrRange => the parse fails.
#pragma once
#include <stdint.h>

#include "sj_chunk.h"
#include "sj_pow5_table.h"

namespace sj {

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__ static const u64 POW5_128[(POW5_MAX_Q - POW5_MIN_Q + 1) * 2] = SJ_POW5_TABLE_INIT;
#else
static const u64 POW5_128[(POW5_MAX_Q - POW5_MIN_Q + 1) * 2] = SJ_POW5_TABLE_INIT;
#endif

// parse_number.go:27-34
enum : u8 { NF_PART = 1, NF_FLOATONLY = 2, NF_MINUS = 4, NF_EOV = 8, NF_DIGIT = 16, NF_MUSTDIGIT = 32 };

SJ_HDC u8 number_rune_of(u8 c) {  // isNumberRune, parse_number.go:36-60
    if (c >= '0' && c <= '9') return NF_PART | NF_DIGIT;
    switch (c) {
    case '.': return NF_PART | NF_FLOATONLY | NF_MUSTDIGIT;
    case '+': return NF_PART;
    case '-': return NF_PART | NF_MINUS | NF_MUSTDIGIT;
    case 'e':
    case 'E': return NF_PART | NF_FLOATONLY;
    case ',':
    case '}':
    case ']':
    case ' ':
    case '\t':
    case '\r':
    case '\n':
    case ':': return NF_EOV;
    default: return 0;
    }
}
// the same as a table: one load instead of a compare chain under divergence (the reference uses a table too)
struct NumberRuneLut {
    u8 v[256];
};
constexpr NumberRuneLut make_number_rune_lut() {
    NumberRuneLut t{};
    for (u32 c = 0; c < 256; c++) t.v[c] = number_rune_of((u8)c);
    return t;
}
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__ static const NumberRuneLut NUMBER_RUNE_LUT = make_number_rune_lut();
#else
static constexpr NumberRuneLut NUMBER_RUNE_LUT = make_number_rune_lut();
#endif
SJ_HD u8 number_rune(u8 c) { return NUMBER_RUNE_LUT.v[c]; }

// ---- 64x64 -> 128 multiply ---------------------------------------------------------------------
struct U128 {
    u64 lo, hi;
};
SJ_HD U128 mul64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return U128{a * b, __umul64hi(a, b)};
#else
    unsigned __int128 p = (unsigned __int128)a * b;
    return U128{(u64)p, (u64)(p >> 64)};
#endif
}

// ---- Eisel-Lemire: w * 10^q -> binary64 bits (sign excluded) ---------------------------------------
// Returns the IEEE bits of the correctly rounded value for w != 0 (Mushtak & Lemire 2023: the
// 128-bit product is always sufficient for a 64-bit w).  Infinity is returned as 0x7ff0...0.
SJ_HD u64 eisel_lemire64(u64 w, int q) {
    if (q < POW5_MIN_Q) return 0;  // w * 10^q < 2^-1075: rounds to zero
    if (q > POW5_MAX_Q) return 0x7ff0000000000000ull;
    int lz = clz64(w);
    w <<= lz;
    const u64 t_hi = POW5_128[2 * (q - POW5_MIN_Q)], t_lo = POW5_128[2 * (q - POW5_MIN_Q) + 1];
    U128 first = mul64(w, t_hi);
    if ((first.hi & 0x1ffull) == 0x1ffull) {  // the low 9 bits could still change: refine
        const U128 second = mul64(w, t_lo);
        first.lo += second.hi;
        if (second.hi > first.lo) first.hi++;
    }
    const u64 lower = first.lo, upper = first.hi;
    const int upperbit = (int)(upper >> 63);
    u64 mantissa = upper >> (upperbit + 9);  // 64 - 52 - 3
    // floor(log2(10^q)) + 63 = ((152170 + 65536) * q >> 16) + 63
    int power2 = (int)(((long long)(152170 + 65536) * q) >> 16) + 63 + upperbit - lz + 1023;
    if (power2 <= 0) {  // subnormal (or zero)
        if (-power2 + 1 >= 64) return 0;
        mantissa >>= -power2 + 1;
        mantissa += mantissa & 1;
        mantissa >>= 1;
        power2 = mantissa < (1ull << 52) ? 0 : 1;
        return (mantissa & ~(1ull << 52)) | ((u64)power2 << 52);
    }
    // exactly halfway between two doubles? only possible for small |q| (5^q must fit the product)
    if (lower <= 1 && q >= -4 && q <= 23 && (mantissa & 3) == 1) {
        if ((mantissa << (upperbit + 9)) == upper) mantissa &= ~1ull;  // round to even
    }
    mantissa += mantissa & 1;
    mantissa >>= 1;
    if (mantissa >= (2ull << 52)) {
        mantissa = 1ull << 52;
        power2++;
    }
    mantissa &= ~(1ull << 52);
    if (power2 >= 0x7ff) return 0x7ff0000000000000ull;
    return mantissa | ((u64)power2 << 52);
}

// ---- decimal scanning (strconv readFloat, decimal branch) -------------------------------------------
struct Decimal {
    u64 mant;       // first <= 19 significant digits
    int exp10;      // value = mant * 10^exp10 (* (1 + tail) if trunc)
    bool neg;
    bool trunc;     // non-zero digits beyond the 19th were dropped
    bool ok;        // syntax ok (whole string consumed)
};

SJ_HD Decimal scan_decimal(const u8 *s, u32 n) {
    Decimal d{0, 0, false, false, false};
    u32 i = 0;
    if (i < n && (s[i] == '+' || s[i] == '-')) {
        d.neg = s[i] == '-';
        i++;
    }
    bool sawdot = false, sawdigits = false;
    int nd = 0, nd_mant = 0, dp = 0;
    for (; i < n; i++) {
        const u8 c = s[i];
        if (c == '.') {
            if (sawdot) break;
            sawdot = true;
            dp = nd;
            continue;
        }
        if (c >= '0' && c <= '9') {
            sawdigits = true;
            if (c == '0' && nd == 0) {  // ignore leading zeros
                dp--;
                continue;
            }
            nd++;
            if (nd_mant < 19) {
                d.mant = d.mant * 10 + (u64)(c - '0');
                nd_mant++;
            } else if (c != '0') {
                d.trunc = true;
            }
            continue;
        }
        break;
    }
    if (!sawdigits) return d;
    if (!sawdot) dp = nd;
    if (i < n && (s[i] == 'e' || s[i] == 'E')) {
        i++;
        if (i >= n) return d;
        int esign = 1;
        if (s[i] == '+') i++;
        else if (s[i] == '-') {
            i++;
            esign = -1;
        }
        if (i >= n || s[i] < '0' || s[i] > '9') return d;
        int e = 0;
        for (; i < n && s[i] >= '0' && s[i] <= '9'; i++)
            if (e < 10000) e = e * 10 + (s[i] - '0');
        dp += e * esign;
    }
    if (i != n) return d;
    if (d.mant != 0) d.exp10 = dp - nd_mant;
    d.ok = true;
    return d;
}

SJ_HD double bits_to_double(u64 b) {
    union {
        u64 u;
        double d;
    } x;
    x.u = b;
    return x.d;
}
SJ_HD u64 double_to_bits(double d) {
    union {
        u64 u;
        double d;
    } x;
    x.d = d;
    return x.u;
}

enum NumStatus : int { NUM_FAIL = 0, NUM_OK = 1, NUM_NEEDS_BIGNUM = 2 };

// ---- the common case in the token kernel: a plain integer of up to 18 digits ------------------------------------------
// parseNumber (parse_number.go:65-135) turns [-]digits without '.', 'e', 'E' into an int64 when strconv.ParseInt takes
// it; up to 18 digits always fit.  the emit pass has the token's position anyway: with the first 24 bytes of the number in
// three words (first byte = lowest byte of w0, zero behind the end of the message) this decides in registers whether
// the number is of that shape -- optional '-', 1..18 digits, no leading zero in front of another digit, and then one of
// the bytes that end a value for parseNumber (',' '}' ']' ' ' '\t' '\r' '\n' ':', :75-84) -- and returns its value.  Every
// other number (floats, 19 and 20 digit integers, anything malformed) is left to the general routine (k_numbers), so
// nothing is decided here that parse_number would decide differently; the host replay compares the two on every number.
SJ_HD u64 digits8(u64 d) {  // eight digit VALUES (0..9) per byte, first digit in the lowest byte -> their decimal value
    d = (d * 2561u) >> 8;                                       // pairs: 10 * a + b
    d = ((d & 0x00ff00ff00ff00ffull) * 6553601u) >> 16;         // quads
    return ((d & 0x0000ffff0000ffffull) * 42949672960001ull) >> 32;
}
SJ_HD bool parse_int_fast(u64 w0, u64 w1, u64 w2, u64 *val) {
    const bool neg = (u8)w0 == '-';
    if (neg) {  // drop the sign byte: the window moves one byte down
        w0 = (w0 >> 8) | (w1 << 56);
        w1 = (w1 >> 8) | (w2 << 56);
        w2 >>= 8;
    }
    const u64 Z = 0x3030303030303030ull, K7 = 0x7f7f7f7f7f7f7f7full, K76 = 0x7676767676767676ull, H = 0x8080808080808080ull;
    const u64 d0 = w0 ^ Z, d1 = w1 ^ Z, d2 = w2 ^ Z;
    // 0x80 in every byte that is not a digit (exact: no carry crosses a byte)
    const u64 n0 = (((d0 & K7) + K76) | d0) & H, n1 = (((d1 & K7) + K76) | d1) & H, n2 = (((d2 & K7) + K76) | d2) & H;
    u32 n;  // digits in front of the first other byte
    if (n0) n = (u32)ctz64(n0) >> 3;
    else if (n1) n = 8u + ((u32)ctz64(n1) >> 3);
    else if (n2) n = 16u + ((u32)ctz64(n2) >> 3);
    else return false;
    if (n == 0 || n > 18) return false;
    if (n > 1 && (u8)w0 == '0') return false;  // "0123": the general routine rejects it
    const u8 t = n < 8 ? (u8)(w0 >> (8 * n)) : (n < 16 ? (u8)(w1 >> (8 * (n - 8))) : (u8)(w2 >> (8 * (n - 16))));
    if (!(t == ',' || t == '}' || t == ']' || t == ' ' || t == '\t' || t == '\r' || t == '\n' || t == ':')) return false;
    // value: the digits of a partly filled word are moved to its high bytes (the low bytes become leading zeros)
    u64 v;
    if (n <= 8) {
        v = digits8(d0 << (8 * (8 - n)));
    } else if (n <= 16) {
        const u32 k = n - 8;  // 1..8 digits in the second word
        u64 p10 = 10;
        for (u32 i = 1; i < k; i++) p10 *= 10;
        v = digits8(d0) * p10 + digits8(d1 << (8 * (8 - k)));
    } else {
        const u32 k = n - 16;  // 1..2
        v = (digits8(d0) * 100000000ull + digits8(d1)) * (k == 1 ? 10ull : 100ull) + digits8(d2 << (8 * (8 - k)));
    }
    *val = neg ? (u64)0 - v : v;
    return true;
}

// ParseFloat(s, 64) on a syntactically scanned decimal.  *bits excludes nothing (sign applied).
// NUM_NEEDS_BIGNUM: *bits holds the candidate for the truncated mantissa (the lower neighbour).
SJ_HD int decimal_to_double(const Decimal &d, u64 *bits) {
    const u64 sign = d.neg ? 0x8000000000000000ull : 0;
    if (d.mant == 0) {
        *bits = sign;
        return NUM_OK;
    }
    if (!d.trunc && d.mant < (1ull << 53) && d.exp10 >= -22 && d.exp10 <= 22) {
        // Clinger: both operands exact, one correctly rounded IEEE operation
        const double p10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
        double f = (double)d.mant;
        f = d.exp10 < 0 ? f / p10[-d.exp10] : f * p10[d.exp10];
        *bits = double_to_bits(f) | sign;
        return NUM_OK;
    }
    const u64 b0 = eisel_lemire64(d.mant, d.exp10);
    if (d.trunc) {
        const u64 b1 = eisel_lemire64(d.mant + 1, d.exp10);
        if (b0 != b1) {
            *bits = b0 | sign;
            return NUM_NEEDS_BIGNUM;
        }
    }
    if (b0 == 0x7ff0000000000000ull) return NUM_FAIL;  // strconv.ErrRange
    *bits = b0 | sign;
    return NUM_OK;
}

// ---- parseNumber (parse_number.go:65-135) -------------------------------------------------------------
// buf[0..avail) is the rest of the message starting at the number.  On NUM_OK / NUM_NEEDS_BIGNUM
// *tag is the first tape word ('l' / 'u' / 'd' << 56 | flags) and *val the second; *numlen the
// length of the number text.
SJ_HD int parse_number(const u8 *buf, u32 avail, u64 *tag, u64 *val, u32 *numlen) {
    u32 pos = 0;
    u8 found = 0;
    for (u32 i = 0; i < avail; i++) {
        const u8 t = number_rune(buf[i]);
        if (t == 0) return NUM_FAIL;
        if (t == NF_EOV) break;
        if (t & NF_MUSTDIGIT) {
            if (i + 1 >= avail || (number_rune(buf[i + 1]) & NF_DIGIT) == 0) return NUM_FAIL;
        }
        found |= t;
        pos = i + 1;
    }
    if (pos == 0) return NUM_FAIL;
    *numlen = pos;
    u64 float_tag = (u64)'d' << 56;

    if ((found & NF_FLOATONLY) == 0 && pos <= 20) {
        if ((found & NF_MINUS) == 0) {
            if (pos > 1 && buf[0] == '0') return NUM_FAIL;
        } else {
            if (pos > 2 && buf[1] == '0') return NUM_FAIL;
        }
        // strconv.ParseInt(s, 10, 64)
        u32 i = 0;
        bool neg = false;
        if (buf[0] == '+' || buf[0] == '-') {
            neg = buf[0] == '-';
            i = 1;
        }
        bool syntax = i >= pos, range = false;
        u64 v = 0;
        u32 nd = 0;  // nineteen digits always fit 64 bits: only the twentieth needs the overflow test (a 64-bit division)
        for (; i < pos; i++) {
            const u8 c = buf[i];
            if (c < '0' || c > '9') {
                syntax = true;
                break;
            }
            const u64 dgt = (u64)(c - '0');
            if (nd < 19) v = v * 10 + dgt;
            else if (range || v > (0xffffffffffffffffull - dgt) / 10) range = true;
            else v = v * 10 + dgt;
            nd++;
        }
        if (!syntax) {
            if (!range && ((!neg && v <= 0x7fffffffffffffffull) || (neg && v <= 0x8000000000000000ull))) {
                *tag = (u64)'l' << 56;
                *val = neg ? (0 - v) : v;
                return NUM_OK;
            }
            float_tag |= 1;  // ErrRange -> FloatOverflowedInteger
            if ((found & NF_MINUS) == 0) {
                // strconv.ParseUint: no sign allowed
                if (buf[0] != '+') {
                    if (!range) {
                        *tag = (u64)'u' << 56;
                        *val = v;
                        return NUM_OK;
                    }
                    float_tag |= 1;
                }
            }
        }
    } else if ((found & NF_FLOATONLY) == 0) {
        float_tag |= 1;
    }

    if (pos > 1 && buf[0] == '0' && (number_rune(buf[1]) & NF_FLOATONLY) == 0) return NUM_FAIL;

    const Decimal d = scan_decimal(buf, pos);
    if (!d.ok) return NUM_FAIL;
    u64 bits;
    const int st = decimal_to_double(d, &bits);
    if (st == NUM_FAIL) return NUM_FAIL;
    *tag = float_tag;
    *val = bits;
    return st;
}

// The kernels parse from a 32-byte copy of the head of the number (LDS: no dependent global byte loads); `full`
// points at the number in the message, `rest` = bytes from there to the end of the message.  The copy may cut a
// longer number anywhere -- behind a sign or a '.', where the pre-validation fails -- so both "used all 32 bytes"
// and "failed" send a longer text to the message itself.
// The common case ahead of the general routine: [-] 1..18 digits and an end-of-value rune behind them, all inside the
// window, no leading zero.  Two 32-bit accumulators of nine digits each (no 64-bit multiply per digit, no rune
// pre-scan); such a number cannot overflow int64, so the result is the reference's ParseInt branch
// (parse_number.go:95-105).  Returns -1 for everything else: the general routine decides.
SJ_HD int parse_int_fast(const u8 *head, u32 avail, u64 *tag, u64 *val, u32 *numlen) {
    const bool neg = avail > 0 && head[0] == '-';
    u32 i = neg ? 1u : 0u, nd = 0, g0 = 0, g1 = 0;
    for (; nd < 18 && i < avail; i++, nd++) {
        const u32 d = (u32)head[i] - (u32)'0';
        if (d > 9u) break;
        if (nd < 9) g0 = g0 * 10u + d;
        else g1 = g1 * 10u + d;
    }
    if (nd == 0 || i >= avail) return -1;              // no digit, or the text runs out of the window
    if (number_rune(head[i]) != NF_EOV) return -1;     // a 19th digit, '.', 'e', '+', or not a number at all
    if (nd > 1 && head[neg ? 1 : 0] == '0') return -1;  // leading zero
    u64 v = g0;
    if (nd > 9) {
        u32 p10 = 10;
        for (u32 k = 10; k < nd; k++) p10 *= 10u;  // 10^(nd - 9) <= 10^9 < 2^32
        v = (u64)g0 * p10 + g1;
    }
    *tag = (u64)'l' << 56;
    *val = neg ? (0 - v) : v;
    *numlen = i;
    return NUM_OK;
}

SJ_HD int parse_number_head32(const u8 *head, const u8 *full, u64 rest, u64 *tag, u64 *val, u32 *numlen) {
    const u32 avail = rest < 32 ? (u32)rest : 32u;
    *numlen = 0;
    int st = parse_int_fast(head, avail, tag, val, numlen);
    if (st >= 0) return st;
    st = parse_number(head, avail, tag, val, numlen);
    if (rest > 32 && (st == NUM_FAIL || *numlen == 32)) st = parse_number(full, (u32)rest, tag, val, numlen);
    return st;
}

}  // namespace sj
