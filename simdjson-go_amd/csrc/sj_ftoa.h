// sj_ftoa.h -- number formatting for tape -> JSON text (Iter.MarshalJSONBuffer, parsed_json.go:401-556), host+device.
//
// Restates, for float64 only,
//   appendFloat       parsed_json.go:1250-1272   ES6-style choice between %f and %e, "e-09" -> "e-9"
//   appendFloatF/fmtF appendfloat_f.go:11-84     %f with the shortest precision
//   ryuFtoaShortest   ftoaryu.go:22-118          shortest round-trip digits (Ryu; a copy of Go's strconv)
//   computeBounds, ryuDigits, ryuDigits32, mult128bitPow10, divisibleByPower5   ftoaryu.go:139-367
//   strconv %e        (Go standard library, fmtE): d.ddddde+XX with at least two exponent digits
// and strconv.AppendInt / AppendUint.  The 128-bit powers of ten are tools/gen_pow10_table.py's output, verified
// entry by entry against the table in ftoaryu.go:392-1089.
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

struct Digits {  // decimalSlice: value = 0.d[0]d[1]...d[nd-1] * 10^dp
    u8 d[24];
    int nd, dp;
};

SJ_HD int mul_log2_log10(int x) { return (x * 78913) >> 18; }   // floor(x * log10(2)), |x| <= 1600 (ftoaryu.go:121)
SJ_HD int mul_log10_log2(int x) { return (x * 108853) >> 15; }  // floor(x * log2(10)), |x| <= 500  (ftoaryu.go:131)

// mult128bitPow10 (ftoaryu.go:330-352): m * 10^q as a 64-bit mantissa; *exact = no bit was dropped
SJ_HD u64 mult128_pow10(u64 m, int *e2, int q, bool *exact) {
    if (q == 0) {
        *e2 -= 8;
        *exact = true;
        return m << 8;
    }
    u64 p0 = POW10_128[2 * (q - POW10_MIN_Q)], p1 = POW10_128[2 * (q - POW10_MIN_Q) + 1];
    if (q < 0) p0 += 1;  // inverse powers of ten must be rounded up
    *e2 += mul_log10_log2(q) - 127 + 119;
    const U128 l = mul64(m, p0), h = mul64(m, p1);
    const u64 mid = l.hi + h.lo;
    const u64 h1 = h.hi + (mid < l.hi ? 1u : 0u);
    *exact = (mid << 9) == 0 && l.lo == 0;
    return (h1 << 9) | (mid >> 55);
}

SJ_HD bool divisible_by_pow5(u64 m, int k) {  // ftoaryu.go:354-366
    if (m == 0) return true;
    for (int i = 0; i < k; i++) {
        if (m % 5 != 0) return false;
        m /= 5;
    }
    return true;
}

// ryuDigits32 (ftoaryu.go:213-287); d->d[0 .. d->nd) already holds the high part
SJ_HD void ryu_digits32(Digits *d, u32 lower, u32 central, u32 upper, bool c0, bool cup, int endindex) {
    if (upper == 0) {
        d->dp = endindex + 1;
        return;
    }
    int trimmed = 0, c_next = 0;
    while (upper > 0) {
        const u32 l = (lower + 9) / 10;
        u32 c = central / 10, cdigit = central % 10;
        const u32 u = upper / 10;
        if (l > u) break;
        if (l == c + 1 && c < u) {
            c++;
            cdigit = 0;
            cup = false;
        }
        trimmed++;
        c0 = c0 && c_next == 0;
        c_next = (int)cdigit;
        lower = l;
        central = c;
        upper = u;
    }
    if (trimmed > 0) cup = c_next > 5 || (c_next == 5 && !c0) || (c_next == 5 && c0 && (central & 1u) == 1u);
    if (central < upper && cup) central++;
    endindex -= trimmed;
    u32 v = central;
    int n = endindex;
    while (n > d->nd) {
        const u32 v1 = v / 100, v2 = v % 100;
        d->d[n] = (u8)('0' + v2 % 10);
        d->d[n - 1] = (u8)('0' + v2 / 10);
        n -= 2;
        v = v1;
    }
    if (n == d->nd) d->d[n] = (u8)(v + '0');
    d->nd = endindex + 1;
    d->dp = d->nd + trimmed;
}

// ryuDigits (ftoaryu.go:156-199).  `first` tracks the reference's re-slicing of d.d (d.d = d.d[n:]).
SJ_HD void ryu_digits(Digits *d, u64 lower, u64 central, u64 upper, bool c0, bool cup) {
    u32 lhi = (u32)(lower / 1000000000ull), llo = (u32)(lower % 1000000000ull);
    const u32 chi = (u32)(central / 1000000000ull), clo = (u32)(central % 1000000000ull);
    const u32 uhi = (u32)(upper / 1000000000ull), ulo = (u32)(upper % 1000000000ull);
    d->nd = 0;
    if (uhi == 0) {
        ryu_digits32(d, llo, clo, ulo, c0, cup, 8);
    } else if (lhi < uhi) {
        if (llo != 0) lhi++;
        c0 = c0 && clo == 0;
        cup = (clo > 500000000u) || (clo == 500000000u && cup);
        ryu_digits32(d, lhi, chi, uhi, c0, cup, 8);
        d->dp += 9;
    } else {
        // emit the high part left-aligned, then the low nine digits behind it
        u8 tmp[9];
        int n = 9;
        for (u32 v = chi; v > 0; v /= 10) tmp[--n] = (u8)(v % 10 + '0');
        d->nd = 9 - n;
        for (int k = 0; k < d->nd; k++) d->d[k] = tmp[n + k];
        ryu_digits32(d, llo, clo, ulo, c0, cup, d->nd + 8);
    }
    while (d->nd > 0 && d->d[d->nd - 1] == '0') d->nd--;  // trailing zeros
    int lead = 0;                                            // initial zeros
    while (lead < d->nd && d->d[lead] == '0') lead++;
    if (lead) {
        for (int k = lead; k < d->nd; k++) d->d[k - lead] = d->d[k];
        d->nd -= lead;
        d->dp -= lead;
    }
}

// ryuFtoaShortest (ftoaryu.go:22-118): shortest digits of mant * 2^exp
SJ_HD void ryu_shortest(Digits *d, u64 mant, int exp) {
    if (mant == 0) {
        d->nd = d->dp = 0;
        return;
    }
    if (exp <= 0) {  // an exact integer with fewer bits than the mantissa
        int tz = 0;
        while (tz < 64 && ((mant >> tz) & 1u) == 0) tz++;
        if (tz >= -exp) {
            mant >>= (u32)(-exp);
            ryu_digits(d, mant, mant, mant, true, false);
            return;
        }
    }
    // computeBounds (ftoaryu.go:139-154)
    u64 ml, mc, mu;
    int e2;
    if (mant != (1ull << 52) || exp == -1023 + 1 - 52) {
        ml = 2 * mant - 1;
        mc = 2 * mant;
        mu = 2 * mant + 1;
        e2 = exp - 1;
    } else {
        ml = 4 * mant - 1;
        mc = 4 * mant;
        mu = 4 * mant + 2;
        e2 = exp - 2;
    }
    if (e2 == 0) {
        ryu_digits(d, ml, mc, mu, true, false);
        return;
    }
    const int q = mul_log2_log10(-e2) + 1;  // 10^q larger than 2^-e2
    bool dl0, dc0, du0;
    int el = e2, ec = e2, eu = e2;
    u64 dl = mult128_pow10(ml, &el, q, &dl0);
    u64 dc = mult128_pow10(mc, &ec, q, &dc0);
    u64 du = mult128_pow10(mu, &eu, q, &du0);
    e2 = eu;
    if (q > 55) dl0 = dc0 = du0 = false;  // large positive powers of ten are not exact
    if (q < 0 && q >= -24) {               // division by a power of ten may be exact
        if (divisible_by_pow5(ml, -q)) dl0 = true;
        if (divisible_by_pow5(mc, -q)) dc0 = true;
        if (divisible_by_pow5(mu, -q)) du0 = true;
    }
    const u32 extra = (u32)(-e2);
    const u64 extra_mask = (1ull << extra) - 1;
    const u64 fracl = dl & extra_mask, fracc = dc & extra_mask, fracu = du & extra_mask;
    dl >>= extra;
    dc >>= extra;
    du >>= extra;
    bool uok = !du0 || fracu > 0;
    if (du0 && fracu == 0) uok = (mant & 1u) == 0;
    if (!uok) du--;
    bool cup;
    if (dc0) cup = fracc > (1ull << (extra - 1)) || (fracc == (1ull << (extra - 1)) && (dc & 1u) == 1u);
    else cup = (fracc >> (extra - 1)) == 1;
    const bool lok = dl0 && fracl == 0 && (mant & 1u) == 0;
    if (!lok) dl++;
    const bool c0 = dc0 && fracc == 0;
    ryu_digits(d, dl, dc, du, c0, cup);
    d->dp -= q;
}

// appendFloat (parsed_json.go:1250-1272): `out` needs 32 bytes; returns the length, 0 for Inf / NaN (an error there)
SJ_HD u32 format_float(u64 bits, u8 *out) {
    const bool neg = (bits >> 63) != 0;
    int exp = (int)((bits >> 52) & 0x7ff);
    u64 mant = bits & ((1ull << 52) - 1);
    if (exp == 0x7ff) return 0;  // "INF or NaN number found"
    if (exp == 0) exp++;         // denormal
    else mant |= 1ull << 52;
    exp += -1023;
    Digits d;
    ryu_shortest(&d, mant, exp - 52);
    u32 n = 0;
    if (neg) out[n++] = '-';
    // abs >= 1e-6 && abs < 1e21, or zero  <=>  %f  (the comparisons are exact on the decimal exponent of the
    // shortest digits: abs < 1e21 <=> dp <= 21, abs >= 1e-6 <=> dp >= -5, because 1e21 and 1e-6 round-trip as "1")
    const u64 absbits = bits & 0x7fffffffffffffffull;
    const bool use_f = absbits == 0 || (absbits >= 0x3eb0c6f7a0b5ed8dull /* 1e-6 */ && absbits < 0x444b1ae4d6e2ef50ull /* 1e21 */);
    if (use_f) {  // fmtF with prec = max(nd - dp, 0) (appendfloat_f.go:43-84)
        if (d.dp > 0) {
            const int m = d.nd < d.dp ? d.nd : d.dp;
            for (int k = 0; k < m; k++) out[n++] = d.d[k];
            for (int k = m; k < d.dp; k++) out[n++] = '0';
        } else {
            out[n++] = '0';
        }
        const int prec = d.nd - d.dp > 0 ? d.nd - d.dp : 0;
        if (prec > 0) {
            out[n++] = '.';
            for (int i = 0; i < prec; i++) {
                const int j = d.dp + i;
                out[n++] = (0 <= j && j < d.nd) ? d.d[j] : (u8)'0';
            }
        }
        return n;
    }
    // strconv 'e' with the shortest precision (fmtE): first digit, '.', the rest, 'e', sign, >= 2 exponent digits
    out[n++] = d.nd > 0 ? d.d[0] : (u8)'0';
    if (d.nd > 1) {
        out[n++] = '.';
        for (int k = 1; k < d.nd; k++) out[n++] = d.d[k];
    }
    out[n++] = 'e';
    int e = d.nd == 0 ? 0 : d.dp - 1;
    if (e < 0) {
        out[n++] = '-';
        e = -e;
    } else {
        out[n++] = '+';
    }
    if (e < 10) {
        out[n++] = '0';
        out[n++] = (u8)('0' + e);
    } else if (e < 100) {
        out[n++] = (u8)('0' + e / 10);
        out[n++] = (u8)('0' + e % 10);
    } else {
        out[n++] = (u8)('0' + e / 100);
        out[n++] = (u8)('0' + (e / 10) % 10);
        out[n++] = (u8)('0' + e % 10);
    }
    // clean up e-09 to e-9 (parsed_json.go:1265-1270)
    if (n >= 4 && out[n - 4] == 'e' && out[n - 3] == '-' && out[n - 2] == '0') {
        out[n - 2] = out[n - 1];
        n--;
    }
    return n;
}

// strconv.AppendUint / AppendInt, base 10: `out` needs 20 bytes
SJ_HD u32 format_uint(u64 v, u8 *out) {
    u8 tmp[20];
    int n = 0;
    do {
        tmp[n++] = (u8)('0' + v % 10);
        v /= 10;
    } while (v != 0);
    for (int k = 0; k < n; k++) out[k] = tmp[n - 1 - k];
    return (u32)n;
}
SJ_HD u32 format_int(u64 raw, u8 *out) {  // raw: the two's complement tape word
    if ((raw >> 63) == 0) return format_uint(raw, out);
    out[0] = '-';
    return 1 + format_uint(0 - raw, out + 1);
}

}  // namespace sj
