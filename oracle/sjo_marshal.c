/*
 * sjo_marshal.c -- ORACLE (test infrastructure only, see sjo.h).
 * Restatement of Iter.MarshalJSONBuffer on a whole ParsedJson (parsed_json.go:401-556): tape -> JSON text, records
 * separated by '\n', with escapeBytes (:1171-1238), appendFloat (:1250-1272), strconv.AppendInt / AppendUint.
 *
 * Floats: the reference formats with a copy of Go's Ryu (ftoaryu.go).  The oracle deliberately does NOT restate Ryu
 * a second time (the product's csrc/sj_ftoa.h does): it finds the shortest round-trip digits with glibc -- the
 * smallest precision p whose correctly rounded "%.{p}e" reads back to the same double -- and lays them out by the
 * reference's rules.  tests/test_oracle_marshal.py pins this against the reference's own expected texts
 * (simdjson_amd64_test.go:701-955, :33-86) and against Python's repr (Gay's shortest digits) on every binade.
 */
#define _GNU_SOURCE
#include "sjo.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint8_t *p;
    size_t len, cap;
} out_t;

static void put(out_t *o, const void *src, size_t n) {
    if (o->len + n > o->cap) {
        o->cap = (o->len + n) * 2 + 256;
        o->p = (uint8_t *)realloc(o->p, o->cap);
    }
    memcpy(o->p + o->len, src, n);
    o->len += n;
}
static void putc_(out_t *o, char c) { put(o, &c, 1); }

/* escapeBytes, parsed_json.go:1190-1238 */
static void escape_bytes(out_t *o, const uint8_t *s, size_t n) {
    static const char hex[] = "0123456789abcdef";
    for (size_t i = 0; i < n; i++) {
        const uint8_t c = s[i];
        switch (c) {
        case '\b': put(o, "\\b", 2); break;
        case '\f': put(o, "\\f", 2); break;
        case '\n': put(o, "\\n", 2); break;
        case '\r': put(o, "\\r", 2); break;
        case '"': put(o, "\\\"", 2); break;
        case '\t': put(o, "\\t", 2); break;
        case '\\': put(o, "\\\\", 2); break;
        default:
            if (c < 0x20) {
                char u[6] = {'\\', 'u', '0', '0', hex[c >> 4], hex[c & 15]};
                put(o, u, 6);
            } else {
                putc_(o, (char)c);
            }
        }
    }
}

/* appendFloat, parsed_json.go:1250-1272; 0 = "INF or NaN number found" */
int sjo_format_float(uint64_t bits, char *out) {
    double f;
    memcpy(&f, &bits, 8);
    if (isinf(f) || isnan(f)) return 0;
    char digits[32];
    int nd = 0, dp = 0;
    if (f != 0) {
        /* Shortest digits that read back to f.  For each precision p the candidates are the correctly rounded p-digit
         * decimal M and its neighbours M +- 1: next to a power of two the rounding interval is lopsided, and a
         * neighbour can lie inside it when M itself does not (e.g. 2^-1017 = 7.120236347223045e-307, whose correctly
         * rounded 16-digit decimal ends in ...44).  The closest candidate that reads back wins. */
        const double a = fabs(f);
        char buf[64];
        unsigned long long best_m = 0;
        int best_e = 0, found = 0;
        for (int p = 1; p <= 17 && !found; p++) {
            snprintf(buf, sizeof buf, "%.*e", p - 1, a);
            const char *e = strchr(buf, 'e');
            unsigned long long m = 0;
            for (const char *c = buf; c < e; c++)
                if (*c != '.') m = m * 10 + (unsigned long long)(*c - '0');
            const int e10 = atoi(e + 1) - (p - 1); /* value = m * 10^e10 */
            long double best_d = 0;
            for (int k = 0; k < 3; k++) {
                const unsigned long long cand = k == 0 ? m : (k == 1 ? m + 1 : m - 1);
                if (cand == 0) continue;
                char cb[64];
                snprintf(cb, sizeof cb, "%llue%d", cand, e10);
                if (strtod(cb, NULL) != a) continue;
                const long double d = fabsl(strtold(cb, NULL) - (long double)a);
                if (!found || d < best_d) {
                    found = 1;
                    best_d = d;
                    best_m = cand;
                    best_e = e10;
                }
            }
        }
        char mb[32];
        nd = snprintf(mb, sizeof mb, "%llu", best_m);
        memcpy(digits, mb, (size_t)nd);
        dp = nd + best_e;
        while (nd > 1 && digits[nd - 1] == '0') nd--;
    }
    int n = 0;
    if (bits >> 63) out[n++] = '-';
    const double a = fabs(f);
    if ((a >= 1e-6 && a < 1e21) || a == 0) { /* appendFloatF / fmtF, appendfloat_f.go:43-84 */
        if (dp > 0) {
            const int m = nd < dp ? nd : dp;
            memcpy(out + n, digits, (size_t)m);
            n += m;
            for (int k = m; k < dp; k++) out[n++] = '0';
        } else {
            out[n++] = '0';
        }
        const int prec = nd - dp > 0 ? nd - dp : 0;
        if (prec > 0) {
            out[n++] = '.';
            for (int i = 0; i < prec; i++) {
                const int j = dp + i;
                out[n++] = (0 <= j && j < nd) ? digits[j] : '0';
            }
        }
        return n;
    }
    /* strconv.AppendFloat(dst, f, 'e', -1, 64), then e-09 -> e-9 */
    out[n++] = digits[0];
    if (nd > 1) {
        out[n++] = '.';
        memcpy(out + n, digits + 1, (size_t)nd - 1);
        n += nd - 1;
    }
    int ex = dp - 1;
    n += sprintf(out + n, "e%c%02d", ex < 0 ? '-' : '+', abs(ex));
    if (n >= 4 && out[n - 4] == 'e' && out[n - 3] == '-' && out[n - 2] == '0') {
        out[n - 2] = out[n - 1];
        n--;
    }
    return n;
}

enum { ST_NONE, ST_ARRAY, ST_OBJECT, ST_ROOT };

/* MarshalJSONBuffer over the whole tape.  rc: 0 ok, negative = one of the reference's error returns. */
int sjo_marshal_json(const uint64_t *tape, size_t n, const uint8_t *strings, const uint8_t *msg, uint8_t **out,
                     size_t *out_len) {
    out_t o = {0};
    uint8_t *stack = (uint8_t *)malloc(n + 2);
    size_t sp = 0;
    stack[sp++] = ST_NONE;
    size_t i = 0;
    int rc = 0;
    char num[40];
#define STR(idx, ptr, len)                                                                                             \
    do {                                                                                                               \
        const uint64_t w_ = tape[idx] & SJO_JSONVALUEMASK;                                                             \
        ptr = (w_ & SJO_STRINGBUFBIT) ? strings + (w_ & (SJO_STRINGBUFBIT - 1)) : msg + w_;                            \
        len = tape[(idx) + 1];                                                                                         \
    } while (0)
    while (i < n) {
        uint8_t t = (uint8_t)(tape[i] >> 56);
        if (stack[sp - 1] == ST_OBJECT && t != '}') { /* key names (:422-434) */
            if (t != '"') {
                rc = -1;
                break;
            }
            const uint8_t *s;
            uint64_t l;
            STR(i, s, l);
            putc_(&o, '"');
            escape_bytes(&o, s, l);
            put(&o, "\":", 2);
            i += 2;
            if (i >= n) {
                rc = -2;
                break;
            }
            t = (uint8_t)(tape[i] >> 56);
        }
        size_t next = i + 1;
        int value_done = 1;
        switch (t) {
        case 'r': {
            const int is_open = (tape[i] & SJO_JSONVALUEMASK) > i;
            if (sp > 1) {
                if (is_open || stack[sp - 1] != ST_ROOT) {
                    rc = -3;
                    goto done;
                }
                if (i + 1 < n) putc_(&o, '\n'); /* PeekNextTag() != TagEnd (:451-453) */
                sp--;
                value_done = 0;
                break;
            }
            stack[sp++] = ST_ROOT;
            value_done = 0;
            break;
        }
        case '"': {
            const uint8_t *s;
            uint64_t l;
            STR(i, s, l);
            putc_(&o, '"');
            escape_bytes(&o, s, l);
            putc_(&o, '"');
            next = i + 2;
            break;
        }
        case 'l':
            put(&o, num, (size_t)sprintf(num, "%lld", (long long)tape[i + 1]));
            next = i + 2;
            break;
        case 'u':
            put(&o, num, (size_t)sprintf(num, "%llu", (unsigned long long)tape[i + 1]));
            next = i + 2;
            break;
        case 'd': {
            const int k = sjo_format_float(tape[i + 1], num);
            if (k == 0) {
                rc = -4;
                goto done;
            }
            put(&o, num, (size_t)k);
            next = i + 2;
            break;
        }
        case 'n': put(&o, "null", 4); break;
        case 't': put(&o, "true", 4); break;
        case 'f': put(&o, "false", 5); break;
        case '{':
            putc_(&o, '{');
            stack[sp++] = ST_OBJECT;
            value_done = 0;
            break;
        case '[':
            putc_(&o, '[');
            stack[sp++] = ST_ARRAY;
            value_done = 0;
            break;
        case '}':
            putc_(&o, '}');
            if (stack[sp - 1] != ST_OBJECT) {
                rc = -5;
                goto done;
            }
            sp--;
            break;
        case ']':
            putc_(&o, ']');
            if (stack[sp - 1] != ST_ARRAY) {
                rc = -6;
                goto done;
            }
            sp--;
            break;
        default:
            rc = -7;
            goto done;
        }
        i = next;
        if (value_done && i < n) { /* separators (:534-549) */
            const uint8_t nt = (uint8_t)(tape[i] >> 56);
            if (stack[sp - 1] == ST_ARRAY && nt != ']') putc_(&o, ',');
            if (stack[sp - 1] == ST_OBJECT && nt != '}') putc_(&o, ',');
        }
    }
    if (rc == 0 && sp > 1) rc = -8; /* objects or arrays not closed */
done:
    free(stack);
    if (rc) {
        free(o.p);
        *out = NULL;
        *out_len = 0;
        return rc;
    }
    *out = o.p ? o.p : (uint8_t *)malloc(1);
    *out_len = o.len;
    return 0;
}
