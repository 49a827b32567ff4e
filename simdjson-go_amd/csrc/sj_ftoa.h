// sj_ftoa.h -- number formatting for tape -> JSON text (Iter.MarshalJSONBuffer, parsed_json.go:401-556), host+device.
//
// What has to match the reference byte for byte is the TEXT:
//   appendFloat       parsed_json.go:1250-1272   ES6-style choice between %f and %e, "e-09" -> "e-9"
//   appendFloatF/fmtF appendfloat_f.go:11-84     %f with the shortest precision
//   strconv %e        (Go standard library, fmtE): d.ddddde+XX with at least two exponent digits
//   strconv.AppendInt / AppendUint
// over the shortest round-trip digits of the float64.  The reference gets those digits from its copy of Go's Ryu
// (ftoaryu.go); any correct shortest-digits routine yields the same ones, and the one here is built for lanes that run
// in lockstep: three 64 x 128-bit multiplications against one table entry and a fixed-shape digit emission
// (shortest_decimal below; tests/test_host_ftoa.py checks > 250 000 doubles incl. the ends of every binade).  The
// 128-bit powers of ten are tools/gen_pow10_table.py's output.
#pragma once
#include <stdint.h>

#include "sj_chunk.h"
#include "sj_number.h"  // mul64 / U128
#include "sj_pow10_table.h"

namespace sj {

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__ static const u64 POW10_128[(POW10_MAX_Q - POW10_MIN_Q + 1) * 2] = SJ_POW10_TABLE_INIT;
#else
static const u64 POW10_128[(POW10_MAX_Q - POW10_MIN_Q + 1) * 2] = SJ_POW10_TABLE_INIT;
#endif

// ---- shortest round-trip digits ----------------------------------------------------------------------------------
// Any correct shortest-digits routine prints the same digits as the reference's copy of Go's strconv (the shortest
// decimal that reads back as the same float64, the closest such one, ties to even).  This one works on the rounding
// interval directly (Giulietti's "Schubfach" formulation): with v = c * 2^q, k = floor(log10(2^q)) and g = the 128-bit
// power of ten 10^-k rounded up, the three products of 4c - 2 (or 4c - 1 below a power of two), 4c and 4c + 2 with g --
// each one 64 x 128-bit multiplication whose discarded bits only survive as a sticky "odd" bit -- are the interval and v
// itself scaled to integers; the answer is the multiple of 10 (if any) or of 1 inside the interval, closest to v.  One
// multiplication per bound, no digit-by-digit loop: the lanes of a wave stay together.
SJ_HD int floor_log10_pow2(int e) { return (e * 1262611) >> 22; }                       // |e| <= 1500
SJ_HD int floor_log10_three_quarters_pow2(int e) { return (e * 1262611 - 524031) >> 22; }
SJ_HD int floor_log2_pow10(int e) { return (e * 1741647) >> 19; }                       // |e| <= 1233

// floor(cp * g / 2^128) with the dropped bits folded into bit 0 ("round to odd"); g = {hi, lo}
SJ_HD u64 mul_round_to_odd(u64 ghi, u64 glo, u64 cp) {
    const U128 x = mul64(cp, glo), y = mul64(cp, ghi);
    const u64 z = y.lo + x.hi;
    const u64 y1 = y.hi + (z < y.lo ? 1u : 0u);
    return y1 | (z > 1 ? 1u : 0u);
}

// decimal significand and exponent of the shortest representation: value = *dec * 10^(*k10); c = binary significand
// (hidden bit included), q = binary exponent of its unit, `narrow` = the interval below v is half as wide (c is a power of two)
SJ_HD void shortest_decimal(u64 c, int q, bool narrow, u64 *dec, int *k10) {
    const bool even = (c & 1u) == 0;  // the interval includes its ends iff the significand is even (round-half-even reading)
    const u64 cbl = 4 * c - 2 + (narrow ? 1u : 0u), cb = 4 * c, cbr = 4 * c + 2;
    const int k = narrow ? floor_log10_three_quarters_pow2(q) : floor_log10_pow2(q);
    const int h = q + floor_log2_pow10(-k) + 1;  // 1 <= h <= 4
    // 10^-k as a 128-bit mantissa, rounded UP (the table holds the rounded-down mantissas; 10^0 .. 10^55 are exact)
    u64 glo = POW10_128[2 * (-k - POW10_MIN_Q)], ghi = POW10_128[2 * (-k - POW10_MIN_Q) + 1];
    if (-k < 0 || -k > 55) {
        glo += 1;
        ghi += glo == 0 ? 1u : 0u;
    }
    const u64 vbl = mul_round_to_odd(ghi, glo, cbl << h);
    const u64 vb = mul_round_to_odd(ghi, glo, cb << h);
    const u64 vbr = mul_round_to_odd(ghi, glo, cbr << h);
    const u64 lower = vbl + (even ? 0u : 1u), upper = vbr - (even ? 0u : 1u);
    const u64 s = vb >> 2;  // floor(v * 10^-k)
    if (s >= 10) {  // a multiple of 10 inside the interval is one digit shorter
        const u64 sp = s / 10;
        const bool up_inside = lower <= 40 * sp, wp_inside = 40 * sp + 40 <= upper;
        if (up_inside != wp_inside) {
            *dec = sp + (wp_inside ? 1u : 0u);
            *k10 = k + 1;
            return;
        }
    }
    const bool u_inside = lower <= 4 * s, w_inside = 4 * s + 4 <= upper;
    *k10 = k;
    if (u_inside != w_inside) {
        *dec = s + (w_inside ? 1u : 0u);
        return;
    }
    const u64 mid = 4 * s + 2;  // both (or neither) inside: the one closer to v, ties to even
    const bool round_up = vb > mid || (vb == mid && (s & 1u) != 0);
    *dec = s + (round_up ? 1u : 0u);
}

// The decimal digits live in REGISTERS (round 5: the byte buffers this file used to fill -- 17 digits, the trimmed window, the
// caller's 32-byte text -- were 80 bytes of scratch per lane in k_ms_tile): eight digits of x < 10^8 as the bytes of one
// 64-bit word, most significant digit in byte 0, values 0..9 (two four-digit halves, two two-digit quarters, no loop)
SJ_HD u64 digits8(u32 x) {
    const u32 hi = x / 10000u, lo = x - hi * 10000u;
    const u32 a = hi / 100u, b = hi - a * 100u, c = lo / 100u, d = lo - c * 100u;
    const u32 w0 = (a / 10u) | (a % 10u) << 8 | (b / 10u) << 16 | (b % 10u) << 24;
    const u32 w1 = (c / 10u) | (c % 10u) << 8 | (d / 10u) << 16 | (d % 10u) << 24;
    return (u64)w0 | ((u64)w1 << 32);
}
SJ_HD int low_byte_index(u64 v) {  // index of the lowest non-zero byte (v != 0)
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)(__ffsll((unsigned long long)v) - 1) >> 3;
#else
    return __builtin_ctzll(v) >> 3;
#endif
}
SJ_HD int high_byte_index(u64 v) {  // index of the highest non-zero byte (v != 0)
#if defined(__HIP_DEVICE_COMPILE__)
    return (63 - __clzll((long long)v)) >> 3;
#else
    return (63 - __builtin_clzll(v)) >> 3;
#endif
}

// value = 0.d(0)d(1)...d(nd-1) * 10^dp, no trailing zeros (nd == 0: zero); the 17 digits of dec < 10^17 are `top`, the
// bytes of `a` (digits 1..8) and the bytes of `b` (digits 9..16), d(0) is digit `first`
struct Digits {
    u64 a, b;
    u32 top;
    int first, nd, dp;
    SJ_HD u8 digit(int i) const {  // ASCII of d(i), 0 <= i < nd
        const int k = first + i;
        const u32 v = k == 0 ? top : (u32)((k < 9 ? a >> (8 * (k - 1)) : b >> (8 * (k - 9))) & 0xffu);
        return (u8)('0' + v);
    }
};

// the digits of mant * 2^exp2 (mant != 0 unless the value is zero; denormals come with their own exponent)
SJ_HD Digits shortest_digits(u64 mant, int exp2, bool narrow) {
    Digits d{0, 0, 0, 0, 0, 0};
    if (mant == 0) return d;
    u64 dec;
    int k10;
    shortest_decimal(mant, exp2, narrow, &dec, &k10);
    // 0 < dec < 10^17: 1 + 8 + 8 digits, then the window between the first and the last non-zero digit
    const u64 top = dec / 100000000ull;           // < 10^9
    const u32 low8 = (u32)(dec - top * 100000000ull);
    d.top = (u32)(top / 100000000ull);            // the 17th digit
    d.a = digits8((u32)(top - (u64)d.top * 100000000ull));
    d.b = digits8(low8);
    d.first = d.top ? 0 : (d.a ? 1 + low_byte_index(d.a) : 9 + low_byte_index(d.b));
    const int last = d.b ? 9 + high_byte_index(d.b) : (d.a ? 1 + high_byte_index(d.a) : 0);
    d.nd = last - d.first + 1;
    d.dp = k10 + (17 - d.first);  // dec has 17 - first digits: value = 0.d... * 10^(k10 + digits)
    return d;
}

// appendFloat (parsed_json.go:1250-1272): exactly the bytes of the text are written (at most 25: sign, 17 digits, '.',
// "e-308"; %f prints at most 21 digits in front of the point or "0." and 5 zeros and 17 digits behind it); returns the
// length, 0 for Inf / NaN (an error there).  WRITE = false: the length alone (the counting pass of k_ms_tile).
template <bool WRITE>
SJ_HD u32 format_float_t(u64 bits, u8 *out) {
    const bool neg = (bits >> 63) != 0;
    int exp = (int)((bits >> 52) & 0x7ff);
    u64 mant = bits & ((1ull << 52) - 1);
    if (exp == 0x7ff) return 0;  // "INF or NaN number found"
    if (exp == 0) exp++;         // denormal
    else mant |= 1ull << 52;
    exp += -1023;
    // below a power of two the interval's lower half is narrower (not for the smallest normal exponent and denormals)
    const Digits d = shortest_digits(mant, exp - 52, (bits & ((1ull << 52) - 1)) == 0 && ((bits >> 52) & 0x7ff) > 1);
    u32 n = 0;
    auto put = [&](u8 c) {
        if (WRITE) out[n] = c;
        n++;
    };
    if (neg) put('-');
    // abs >= 1e-6 && abs < 1e21, or zero  <=>  %f  (the comparisons are exact on the decimal exponent of the
    // shortest digits: abs < 1e21 <=> dp <= 21, abs >= 1e-6 <=> dp >= -5, because 1e21 and 1e-6 round-trip as "1")
    const u64 absbits = bits & 0x7fffffffffffffffull;
    const bool use_f = absbits == 0 || (absbits >= 0x3eb0c6f7a0b5ed8dull /* 1e-6 */ && absbits < 0x444b1ae4d6e2ef50ull /* 1e21 */);
    if (use_f) {  // fmtF with prec = max(nd - dp, 0) (appendfloat_f.go:43-84)
        const int prec = d.nd - d.dp > 0 ? d.nd - d.dp : 0;
        if (!WRITE) return n + (u32)(d.dp > 0 ? d.dp : 1) + (prec > 0 ? 1u + (u32)prec : 0u);
        if (d.dp > 0) {
            const int m = d.nd < d.dp ? d.nd : d.dp;
            for (int k = 0; k < m; k++) put(d.digit(k));
            for (int k = m; k < d.dp; k++) put('0');
        } else {
            put('0');
        }
        if (prec > 0) {
            put('.');
            for (int i = 0; i < prec; i++) {
                const int j = d.dp + i;
                put((0 <= j && j < d.nd) ? d.digit(j) : (u8)'0');
            }
        }
        return n;
    }
    // strconv 'e' with the shortest precision (fmtE): first digit, '.', the rest, 'e', sign, >= 2 exponent digits --
    // but "e-09" is cleaned up to "e-9" (parsed_json.go:1265-1270: the text ends with 'e', '-', '0', digit exactly for the
    // negative one-digit exponents; positive exponents start at 21 here)
    int e = d.nd == 0 ? 0 : d.dp - 1;
    const bool eneg = e < 0;
    if (eneg) e = -e;
    const u32 elen = e >= 100 ? 3u : ((e >= 10 || !eneg) ? 2u : 1u);
    if (!WRITE) return n + 1u + (d.nd > 1 ? (u32)d.nd : 0u) + 2u + elen;
    put(d.nd > 0 ? d.digit(0) : (u8)'0');
    if (d.nd > 1) {
        put('.');
        for (int k = 1; k < d.nd; k++) put(d.digit(k));
    }
    put('e');
    put(eneg ? '-' : '+');
    if (elen == 3) put((u8)('0' + e / 100));
    if (elen >= 2) put((u8)('0' + (e / 10) % 10));
    put((u8)('0' + e % 10));
    return n;
}
SJ_HD u32 format_float(u64 bits, u8 *out) { return format_float_t<true>(bits, out); }
SJ_HD u32 float_text_len(u64 bits) { return format_float_t<false>(bits, nullptr); }

// strconv.AppendUint / AppendInt, base 10: exactly the digits are written (`out` needs up to 20 bytes)
SJ_HD u32 digit_count(u64 v) {
    u32 n = 1;
    for (u64 p = 10; n < 20 && v >= p; p *= 10) n++;  // (10^19 still fits; n = 20 ends the loop before p overflows)
    return n;
}
SJ_HD u32 format_uint(u64 v, u8 *out) {
    const u32 n = digit_count(v);
    u8 *e = out + n;
    while (v >= 1000000000ull) {  // nine digits at a time in 32-bit arithmetic
        const u64 q = v / 1000000000ull;
        u32 r = (u32)(v - q * 1000000000ull);
        for (int k = 0; k < 9; k++) {
            *--e = (u8)('0' + r % 10u);
            r /= 10u;
        }
        v = q;
    }
    u32 r = (u32)v;
    do {
        *--e = (u8)('0' + r % 10u);
        r /= 10u;
    } while (r != 0);
    return n;
}
SJ_HD u32 format_int(u64 raw, u8 *out) {  // raw: the two's complement tape word
    if ((raw >> 63) == 0) return format_uint(raw, out);
    out[0] = '-';
    return 1 + format_uint(0 - raw, out + 1);
}
SJ_HD u32 int_text_len(u64 raw) { return (raw >> 63) ? 1u + digit_count(0 - raw) : digit_count(raw); }

}  // namespace sj
