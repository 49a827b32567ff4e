// sj_host.h -- host-side helpers of the parse driver (no device code).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace sj {

// bytes.TrimSpace as called by parseMessage (parse_json_amd64.go:55): ASCII fast path, and the
// Unicode White_Space set (unicode.IsSpace) once a byte >= 0x80 is met at either end.
namespace trim_detail {
inline bool ascii_ws(uint8_t c) { return c == ' ' || (c >= '\t' && c <= '\r'); }
inline bool uni_ws(uint32_t r) {
    if (r < 0x100) return r == ' ' || (r >= '\t' && r <= '\r') || r == 0x85 || r == 0xa0;
    return r == 0x1680 || (r >= 0x2000 && r <= 0x200a) || r == 0x2028 || r == 0x2029 || r == 0x202f || r == 0x205f ||
           r == 0x3000;
}
// strict UTF-8 decoder (utf8.DecodeRune): malformed => U+FFFD, width 1
inline uint32_t decode(const uint8_t *p, size_t n, size_t *w) {
    *w = 1;
    if (n == 0) return 0xfffd;
    const uint8_t a = p[0];
    if (a < 0x80) return a;
    int need;
    uint32_t r, lo = 0x80, hi = 0xbf;
    if (a >= 0xc2 && a <= 0xdf) { need = 1; r = a & 0x1f; }
    else if (a >= 0xe0 && a <= 0xef) { need = 2; r = a & 0x0f; if (a == 0xe0) lo = 0xa0; if (a == 0xed) hi = 0x9f; }
    else if (a >= 0xf0 && a <= 0xf4) { need = 3; r = a & 0x07; if (a == 0xf0) lo = 0x90; if (a == 0xf4) hi = 0x8f; }
    else return 0xfffd;
    if (n < (size_t)need + 1) return 0xfffd;
    for (int k = 1; k <= need; k++) {
        const uint8_t c = p[k];
        const uint32_t l = k == 1 ? lo : 0x80, h = k == 1 ? hi : 0xbf;
        if (c < l || c > h) return 0xfffd;
        r = (r << 6) | (c & 0x3f);
    }
    *w = (size_t)need + 1;
    return r;
}
inline uint32_t decode_last(const uint8_t *p, size_t n, size_t *w) {  // utf8.DecodeLastRune
    *w = 1;
    if (n == 0) return 0xfffd;
    if (p[n - 1] < 0x80) return p[n - 1];
    const size_t lim = n >= 4 ? n - 4 : 0;
    size_t start = n - 1;
    while (start > lim && (p[start] & 0xc0) == 0x80) start--;
    size_t ww;
    const uint32_t r = decode(p + start, n - start, &ww);
    if (start + ww != n) return 0xfffd;
    *w = ww;
    return r;
}
}  // namespace trim_detail

inline void trim_space(const uint8_t *s, size_t n, size_t *off, size_t *len) {
    using namespace trim_detail;
    size_t a = 0, b = n;
    bool unicode = false;
    for (; a < b; a++) {
        if (s[a] >= 0x80) { unicode = true; break; }
        if (!ascii_ws(s[a])) break;
    }
    if (!unicode)
        for (; b > a; b--) {
            if (s[b - 1] >= 0x80) { unicode = true; break; }
            if (!ascii_ws(s[b - 1])) break;
        }
    if (unicode) {  // TrimFunc(s[a:b], unicode.IsSpace)
        while (a < b) {
            size_t w;
            if (!uni_ws(decode(s + a, b - a, &w))) break;
            a += w;
        }
        while (b > a) {
            size_t w;
            if (!uni_ws(decode_last(s + a, b - a, &w))) break;
            b -= w;
        }
    }
    *off = a;
    *len = b - a;
}

}  // namespace sj
