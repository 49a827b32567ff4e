/*
 * sjo_parse_number.c -- ORACLE (test infrastructure only, see sjo.h).
 * Restatement of parse_number.go:36-135.  The reference delegates the actual
 * conversions to the Go standard library (strconv.ParseInt / ParseUint /
 * ParseFloat; call sites parse_number.go:105,114,130).  They are restated here:
 *   - ParseInt/ParseUint(base 10, 64 bit): sign + decimal digits, range check;
 *   - ParseFloat(…, 64): Go's decimal float grammar (strconv/atof.go readFloat,
 *     restricted to the characters isNumberRune lets through) + glibc strtod,
 *     which is correctly rounded (round-half-even) exactly like ParseFloat;
 *     |x| overflowing to Inf is ErrRange (=> failure), underflow is silent.
 */
#include "sjo.h"

#include <errno.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { /* parse_number.go:27-34 */
    isPartOfNumberFlag = 1,
    isFloatOnlyFlag = 2,
    isMinusFlag = 4,
    isEOVFlag = 8,
    isDigitFlag = 16,
    isMustHaveDigitNext = 32,
};

static uint8_t is_number_rune(uint8_t c) { /* parse_number.go:36-60 */
    if (c >= '0' && c <= '9') return isPartOfNumberFlag | isDigitFlag;
    switch (c) {
    case '.': return isPartOfNumberFlag | isFloatOnlyFlag | isMustHaveDigitNext;
    case '+': return isPartOfNumberFlag;
    case '-': return isPartOfNumberFlag | isMinusFlag | isMustHaveDigitNext;
    case 'e':
    case 'E': return isPartOfNumberFlag | isFloatOnlyFlag;
    case ',':
    case '}':
    case ']':
    case ' ':
    case '\t':
    case '\r':
    case '\n':
    case ':': return isEOVFlag;
    default: return 0;
    }
}

/* strconv.ParseInt(s, 10, 64): 0 ok, 1 syntax error, 2 range error */
static int go_parse_int(const uint8_t *s, size_t n, int64_t *out) {
    if (n == 0) return 1;
    size_t i = 0;
    int neg = 0;
    if (s[0] == '+' || s[0] == '-') {
        neg = s[0] == '-';
        i = 1;
        if (n == 1) return 1;
    }
    uint64_t v = 0;
    int range = 0;
    for (; i < n; i++) {
        if (s[i] < '0' || s[i] > '9') return 1; /* syntax errors win: ParseUint scans the whole string */
        unsigned d = (unsigned)(s[i] - '0');
        if (!range) {
            if (v > (UINT64_MAX - d) / 10) range = 1;
            else v = v * 10 + d;
        }
    }
    if (range) return 2;
    if (!neg && v > (uint64_t)INT64_MAX) return 2;
    if (neg && v > (uint64_t)INT64_MAX + 1) return 2;
    *out = neg ? (int64_t)(0 - v) : (int64_t)v;
    return 0;
}

/* strconv.ParseUint(s, 10, 64) */
static int go_parse_uint(const uint8_t *s, size_t n, uint64_t *out) {
    if (n == 0) return 1;
    uint64_t v = 0;
    int range = 0;
    for (size_t i = 0; i < n; i++) {
        if (s[i] < '0' || s[i] > '9') return 1;
        unsigned d = (unsigned)(s[i] - '0');
        if (!range) {
            if (v > (UINT64_MAX - d) / 10) range = 1;
            else v = v * 10 + d;
        }
    }
    if (range) return 2;
    *out = v;
    return 0;
}

/* Go decimal float syntax restricted to [0-9.+-eE]: [+-]? digits* [. digits*]? ([eE][+-]?digits+)?
 * with at least one mantissa digit, whole string consumed (strconv/atof.go readFloat + ParseFloat). */
static int go_float_syntax_ok(const uint8_t *s, size_t n) {
    size_t i = 0;
    if (i < n && (s[i] == '+' || s[i] == '-')) i++;
    int sawdigits = 0, sawdot = 0;
    for (; i < n; i++) {
        if (s[i] == '.') {
            if (sawdot) break;
            sawdot = 1;
            continue;
        }
        if (s[i] >= '0' && s[i] <= '9') {
            sawdigits = 1;
            continue;
        }
        break;
    }
    if (!sawdigits) return 0;
    if (i < n && (s[i] == 'e' || s[i] == 'E')) {
        i++;
        if (i >= n) return 0;
        if (s[i] == '+' || s[i] == '-') i++;
        if (i >= n || s[i] < '0' || s[i] > '9') return 0;
        while (i < n && s[i] >= '0' && s[i] <= '9') i++;
    }
    return i == n;
}

/* parseNumber: parse_number.go:65-135.  Returns the tag word (tag<<56 | flags) or 0. */
uint64_t sjo_parse_number(const uint8_t *buf, size_t len, uint64_t *val) {
    size_t pos = 0;
    uint8_t found = 0;
    *val = 0;
    for (size_t i = 0; i < len; i++) {
        uint8_t t = is_number_rune(buf[i]);
        if (t == 0) return 0;
        if (t == isEOVFlag) break;
        if (t & isMustHaveDigitNext) {
            /* A period and minus must be followed by a digit */
            if (len < i + 2 || (is_number_rune(buf[i + 1]) & isDigitFlag) == 0) return 0;
        }
        found |= t;
        pos = i + 1;
    }
    if (pos == 0) return 0;
    const size_t maxIntLen = 20;
    uint64_t float_tag = (uint64_t)'d' << SJO_JSONTAGOFFSET;

    if ((found & isFloatOnlyFlag) == 0 && pos <= maxIntLen) {
        if ((found & isMinusFlag) == 0) {
            if (pos > 1 && buf[0] == '0') return 0;
        } else {
            if (pos > 2 && buf[1] == '0') return 0;
        }
        int64_t i64;
        int e = go_parse_int(buf, pos, &i64);
        if (e == 0) {
            *val = (uint64_t)i64;
            return (uint64_t)'l' << SJO_JSONTAGOFFSET;
        }
        if (e == 2) float_tag |= 1; /* FloatOverflowedInteger */
        if ((found & isMinusFlag) == 0) {
            uint64_t u64;
            e = go_parse_uint(buf, pos, &u64);
            if (e == 0) {
                *val = u64;
                return (uint64_t)'u' << SJO_JSONTAGOFFSET;
            }
            if (e == 2) float_tag |= 1;
        }
    } else if ((found & isFloatOnlyFlag) == 0) {
        float_tag |= 1;
    }

    if (pos > 1 && buf[0] == '0' && (is_number_rune(buf[1]) & isFloatOnlyFlag) == 0) return 0;

    if (!go_float_syntax_ok(buf, pos)) return 0;
    char stackbuf[512];
    char *tmp = pos + 1 <= sizeof stackbuf ? stackbuf : (char *)malloc(pos + 1);
    memcpy(tmp, buf, pos);
    tmp[pos] = 0;
    errno = 0;
    double f = strtod(tmp, NULL);
    if (tmp != stackbuf) free(tmp);
    if (isinf(f)) return 0; /* strconv.ErrRange */
    memcpy(val, &f, 8);
    return float_tag;
}
