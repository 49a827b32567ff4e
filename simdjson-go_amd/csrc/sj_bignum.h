// sj_bignum.h -- exact tie-breaker for decimal->binary64 conversion (host+device, one number per lane).
//
// Used only when the mantissa has more than 19 significant digits AND Eisel-Lemire on the
// truncated mantissa w and on w+1 give two different (adjacent) doubles b0 < b1.  The exact
// decimal value X = D * 10^e (D = up to 800 significant digits + sticky, as in Go's
// strconv `decimal`) is compared with the midpoint (2M+1) * 2^(E-1) of b0 = M * 2^E using
// big-integer multiplication by powers of five and shifts; no division is needed.
#pragma once
#include <stdint.h>

#include "sj_chunk.h"

namespace sj {

struct Big {
    static constexpr int LIMBS = 112;  // 3584 bits: 800 decimal digits (2658 bits) + headroom
    u32 n;
    u32 w[LIMBS];
};

SJ_HD void big_set(Big &b, u64 v) {
    b.n = 0;
    if (v) b.w[b.n++] = (u32)v;
    if (v >> 32) b.w[b.n++] = (u32)(v >> 32);
}
SJ_HD bool big_mul_add(Big &b, u32 m, u32 add) {  // b = b*m + add; false on overflow
    u64 carry = add;
    for (u32 i = 0; i < b.n; i++) {
        const u64 t = (u64)b.w[i] * m + carry;
        b.w[i] = (u32)t;
        carry = t >> 32;
    }
    if (carry) {
        if (b.n >= (u32)Big::LIMBS) return false;
        b.w[b.n++] = (u32)carry;
    }
    return true;
}
SJ_HD bool big_mul_pow5(Big &b, u32 e) {
    while (e >= 13) {
        if (!big_mul_add(b, 1220703125u, 0)) return false;  // 5^13
        e -= 13;
    }
    u32 m = 1;
    for (u32 i = 0; i < e; i++) m *= 5;
    return e == 0 || big_mul_add(b, m, 0);
}
SJ_HD bool big_shl(Big &b, u32 bits) {
    if (b.n == 0 || bits == 0) return true;
    const u32 limbs = bits >> 5, r = bits & 31;
    if (b.n + limbs + 1 > (u32)Big::LIMBS) return false;
    if (r) {
        u32 carry = 0;
        for (u32 i = 0; i < b.n; i++) {
            const u32 v = b.w[i];
            b.w[i] = (v << r) | carry;
            carry = v >> (32 - r);
        }
        if (carry) b.w[b.n++] = carry;
    }
    if (limbs) {
        for (int i = (int)b.n - 1; i >= 0; i--) b.w[i + limbs] = b.w[i];
        for (u32 i = 0; i < limbs; i++) b.w[i] = 0;
        b.n += limbs;
    }
    return true;
}
SJ_HD int big_cmp(const Big &a, const Big &b) {
    if (a.n != b.n) return a.n > b.n ? 1 : -1;
    for (int i = (int)a.n - 1; i >= 0; i--)
        if (a.w[i] != b.w[i]) return a.w[i] > b.w[i] ? 1 : -1;
    return 0;
}

// s[0..n) is a syntactically valid Go decimal float (scan_decimal said ok); b0 = candidate bits of the
// lower neighbour (sign stripped).  Returns the correctly rounded bits (sign stripped); 0x7ff0..0 = overflow.
SJ_HD u64 bignum_round(const u8 *s, u32 n, u64 b0, Big &X, Big &Y) {
    // ---- digits -> D, e, sticky (Go keeps 800 digits) ----
    u32 i = 0;
    if (i < n && (s[i] == '+' || s[i] == '-')) i++;
    big_set(X, 0);
    int nd = 0, kept = 0, dp = 0;
    bool sawdot = false, sticky = false;
    u32 chunk = 0, chunk_digits = 0;
    const u32 P10[10] = {1u, 10u, 100u, 1000u, 10000u, 100000u, 1000000u, 10000000u, 100000000u, 1000000000u};
    for (; i < n; i++) {
        const u8 c = s[i];
        if (c == '.') {
            sawdot = true;
            dp = nd;
            continue;
        }
        if (c < '0' || c > '9') break;
        if (c == '0' && nd == 0) {
            dp--;
            continue;
        }
        nd++;
        if (kept < 800) {
            chunk = chunk * 10 + (u32)(c - '0');
            chunk_digits++;
            kept++;
            if (chunk_digits == 9) {
                big_mul_add(X, 1000000000u, chunk);
                chunk = 0;
                chunk_digits = 0;
            }
        } else if (c != '0') {
            sticky = true;
        }
    }
    if (chunk_digits) big_mul_add(X, P10[chunk_digits], chunk);
    if (!sawdot) dp = nd;
    int e10 = 0;
    if (i < n && (s[i] == 'e' || s[i] == 'E')) {
        i++;
        int esign = 1;
        if (s[i] == '+') i++;
        else if (s[i] == '-') {
            i++;
            esign = -1;
        }
        int e = 0;
        for (; i < n && s[i] >= '0' && s[i] <= '9'; i++)
            if (e < 10000) e = e * 10 + (s[i] - '0');
        e10 = e * esign;
    }
    const int e = dp + e10 - kept;  // X_value = D * 10^e (+ dropped tail)
    // ---- midpoint above b0 ----
    const u64 frac = b0 & 0x000fffffffffffffull;
    const int ef = (int)(b0 >> 52) & 0x7ff;
    const u64 M = ef ? (frac | (1ull << 52)) : frac;
    const int E = ef ? ef - 1075 : -1074;
    big_set(Y, 2 * M + 1);
    int x_pow2 = 0, y_pow2 = E - 1;
    bool ok = true;
    if (e >= 0) {
        ok &= big_mul_pow5(X, (u32)e);
        x_pow2 += e;
    } else {
        ok &= big_mul_pow5(Y, (u32)(-e));
        y_pow2 += -e;
    }
    const int sh = x_pow2 - y_pow2;
    if (sh > 0) ok &= big_shl(X, (u32)sh);
    else if (sh < 0) ok &= big_shl(Y, (u32)(-sh));
    int c = big_cmp(X, Y);
    if (!ok) c = 1;  // cannot happen for inputs that reach this path (see header); keep monotone
    if (c == 0 && sticky) c = 1;
    if (c > 0) return b0 + 1;
    if (c < 0) return b0;
    return (M & 1) ? b0 + 1 : b0;  // exact tie: round half to even
}

}  // namespace sj
