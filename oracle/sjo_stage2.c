/*
 * sjo_stage2.c -- ORACLE (test infrastructure only, see sjo.h).
 * Restatement of stage 2 (stage2_build_tape_amd64.go) and of the parse driver
 * (parse_json_amd64.go:28-127, simdjson_amd64.go:66-94).
 */
#include "sjo.h"
#include "sjo_internal.h"

#include <stdlib.h>
#include <string.h>

/* ---------------- bytes.TrimSpace (Go std bytes package) ---------------- */
static int ascii_space(uint8_t c) { return c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r' || c == ' '; }

static int unicode_is_space(uint32_t r) { /* unicode.IsSpace */
    if (r <= 0xff) return r == '\t' || r == '\n' || r == '\v' || r == '\f' || r == '\r' || r == ' ' || r == 0x85 || r == 0xa0;
    return r == 0x1680 || (r >= 0x2000 && r <= 0x200a) || r == 0x2028 || r == 0x2029 || r == 0x202f ||
           r == 0x205f || r == 0x3000;
}

/* utf8.DecodeRune: returns rune and width; invalid encodings give (0xFFFD, 1) */
static uint32_t decode_rune(const uint8_t *p, size_t n, size_t *w) {
    *w = 1;
    if (n == 0) return 0xfffd;
    uint8_t b0 = p[0];
    if (b0 < 0x80) return b0;
    if (b0 < 0xc2) return 0xfffd;
    if (b0 < 0xe0) {
        if (n < 2 || (p[1] & 0xc0) != 0x80) return 0xfffd;
        *w = 2;
        return ((uint32_t)(b0 & 0x1f) << 6) | (p[1] & 0x3f);
    }
    if (b0 < 0xf0) {
        if (n < 3) return 0xfffd;
        uint8_t lo = 0x80, hi = 0xbf;
        if (b0 == 0xe0) lo = 0xa0;
        if (b0 == 0xed) hi = 0x9f;
        if (p[1] < lo || p[1] > hi || (p[2] & 0xc0) != 0x80) return 0xfffd;
        *w = 3;
        return ((uint32_t)(b0 & 0x0f) << 12) | ((uint32_t)(p[1] & 0x3f) << 6) | (p[2] & 0x3f);
    }
    if (b0 < 0xf5) {
        if (n < 4) return 0xfffd;
        uint8_t lo = 0x80, hi = 0xbf;
        if (b0 == 0xf0) lo = 0x90;
        if (b0 == 0xf4) hi = 0x8f;
        if (p[1] < lo || p[1] > hi || (p[2] & 0xc0) != 0x80 || (p[3] & 0xc0) != 0x80) return 0xfffd;
        *w = 4;
        return ((uint32_t)(b0 & 0x07) << 18) | ((uint32_t)(p[1] & 0x3f) << 12) |
               ((uint32_t)(p[2] & 0x3f) << 6) | (p[3] & 0x3f);
    }
    return 0xfffd;
}

/* utf8.DecodeLastRune */
static uint32_t decode_last_rune(const uint8_t *p, size_t n, size_t *w) {
    *w = 1;
    if (n == 0) return 0xfffd;
    if (p[n - 1] < 0x80) return p[n - 1];
    size_t lim = n >= 4 ? n - 4 : 0;
    size_t start = n - 1;
    while (start > lim && (p[start] & 0xc0) == 0x80) start--; /* back up to a rune start */
    size_t ww;
    uint32_t r = decode_rune(p + start, n - start, &ww);
    if (start + ww != n) return 0xfffd; /* width 1 */
    *w = ww;
    return r;
}

static void trim_func_unicode(const uint8_t *s, size_t n, size_t *off, size_t *len) {
    size_t a = 0;
    while (a < n) {
        size_t w;
        uint32_t r = decode_rune(s + a, n - a, &w);
        if (!unicode_is_space(r)) break;
        a += w;
    }
    size_t b = n;
    while (b > a) {
        size_t w;
        uint32_t r = decode_last_rune(s + a, b - a, &w);
        if (!unicode_is_space(r)) break;
        b -= w;
    }
    *off = a;
    *len = b - a;
}

void sjo_trim_space(const uint8_t *s, size_t n, size_t *off, size_t *out_len) {
    size_t start = 0;
    for (; start < n; start++) {
        uint8_t c = s[start];
        if (c >= 0x80) {
            size_t o, l;
            trim_func_unicode(s + start, n - start, &o, &l);
            *off = start + o;
            *out_len = l;
            return;
        }
        if (!ascii_space(c)) break;
    }
    size_t stop = n;
    for (; stop > start; stop--) {
        uint8_t c = s[stop - 1];
        if (c >= 0x80) {
            size_t o, l;
            trim_func_unicode(s + start, stop - start, &o, &l);
            *off = start + o;
            *out_len = l;
            return;
        }
        if (!ascii_space(c)) break;
    }
    *off = start;
    *out_len = stop - start;
}

/* ---------------- atoms: stage2_build_tape_amd64.go:124-158, 455-476 ---------------- */
static int is_not_structural_or_whitespace(uint8_t c) {
    /* structuralOrWhitespaceNegated: zero for NUL \t \n \r space , : [ ] { } */
    switch (c) {
    case 0: case '\t': case '\n': case '\r': case ' ': case ',': case ':': case '[': case ']': case '{': case '}':
        return 0;
    default:
        return 1;
    }
}

int sjo_is_valid_true_atom(const uint8_t *buf, size_t len) {
    if (len >= 5) return memcmp(buf, "true", 4) == 0 && !is_not_structural_or_whitespace(buf[4]);
    return 0;
}

int sjo_is_valid_false_atom(const uint8_t *buf, size_t len) {
    if (len >= 6) return memcmp(buf, "false", 5) == 0 && !is_not_structural_or_whitespace(buf[5]);
    return 0;
}

int sjo_is_valid_null_atom(const uint8_t *buf, size_t len) {
    if (len >= 5) return memcmp(buf, "null", 4) == 0 && !is_not_structural_or_whitespace(buf[4]);
    return 0;
}

/* ---------------- growable outputs: pj_t lives in sjo_internal.h ---------------- */
static void tape_push(pj_t *pj, uint64_t w) {
    if (pj->tape_len == pj->tape_cap) {
        pj->tape_cap = pj->tape_cap ? pj->tape_cap * 2 : 1024;
        pj->tape = (uint64_t *)realloc(pj->tape, pj->tape_cap * 8);
    }
    pj->tape[pj->tape_len++] = w;
}
static void write_tape(pj_t *pj, uint64_t val, uint8_t c) { tape_push(pj, val | ((uint64_t)c << 56)); }
static void scope_push(pj_t *pj, uint64_t v) {
    if (pj->scope_len == pj->scope_cap) {
        pj->scope_cap = pj->scope_cap ? pj->scope_cap * 2 : 128;
        pj->scope = (uint64_t *)realloc(pj->scope, pj->scope_cap * 8);
    }
    pj->scope[pj->scope_len++] = v;
}

/* parseString: stage2_build_tape_amd64.go:72-113 */
static int parse_string(pj_t *pj, uint64_t idx) {
    const uint8_t *src = pj->msg + idx + 1;
    size_t avail = pj->len - (size_t)idx - 1;
    uint64_t src_len = 0, size = 0;
    if (!(pj->validate_string ? pj->validate_string : sjo_parse_string_validate_only)(src, avail, &src_len, &size)) return 0;
    int need_copy = pj->copy_strings || src_len != size; /* parse_string_amd64.go:40 */
    if (!need_copy) {
        write_tape(pj, idx + 1, '"');
    } else {
        size_t need = pj->strs_len + (size_t)size + 64;
        if (need > pj->strs_cap) {
            pj->strs_cap = need * 2;
            pj->strs = (uint8_t *)realloc(pj->strs, pj->strs_cap);
        }
        uint64_t written = 0;
        (pj->copy_string ? pj->copy_string : sjo_parse_string)(src, avail, pj->strs + pj->strs_len, &written);
        write_tape(pj, SJO_STRINGBUFBIT + pj->strs_len, '"');
        pj->strs_len += (size_t)written;
        size = written;
    }
    tape_push(pj, size);
    return 1;
}

static int add_number(pj_t *pj, uint64_t idx) { /* stage2_build_tape_amd64.go:115-122 */
    uint64_t val;
    uint64_t tag = sjo_parse_number(pj->msg + idx, pj->len - (size_t)idx, &val);
    if (tag == 0) return 0;
    tape_push(pj, tag);
    tape_push(pj, val);
    return 1;
}

enum { RET_START = 1, RET_OBJECT = 2, RET_ARRAY = 3 }; /* :27-32 */

/* updateChar (:34-46): the delta stream is pre-summed into absolute positions */
#define UPDATE_CHAR()                                                   \
    do {                                                                \
        if (pj->ipos >= pj->npos && !more_indexes(pj)) goto succeed;    \
        idx = pj->pos[pj->ipos++];                                      \
    } while (0)

/* Two-thread shape (parse_json_amd64.go:75-95): stage 1 keeps appending to pos[] on another thread and publishes
 * its count; the consumer refreshes its view when it runs dry (the role of `<-pj.indexChans`, :48-61). */
static int more_indexes(pj_t *pj) {
    if (!pj->live_npos) return 0;
    for (;;) {
        const int done = __atomic_load_n(pj->live_done, __ATOMIC_ACQUIRE);
        pj->npos = __atomic_load_n(pj->live_npos, __ATOMIC_ACQUIRE);
        if (pj->ipos < pj->npos) return 1;
        if (done) return 0;
        __builtin_ia32_pause();
    }
}

/* unifiedMachine: stage2_build_tape_amd64.go:160-446 */
int sjo_unified_machine(pj_t *pj) {
    const uint8_t *buf = pj->msg;
    uint64_t idx = 0;
    uint64_t offset;
    const size_t len = pj->len;

    scope_push(pj, ((uint64_t)pj->tape_len << 2) | RET_START);
    write_tape(pj, 0, 'r');
    UPDATE_CHAR();

continue_root:
    switch (buf[idx]) {
    case '{':
        scope_push(pj, ((uint64_t)pj->tape_len << 2) | RET_START);
        write_tape(pj, 0, '{');
        goto object_begin;
    case '[':
        scope_push(pj, ((uint64_t)pj->tape_len << 2) | RET_START);
        write_tape(pj, 0, '[');
        goto array_begin;
    default:
        goto fail;
    }

start_continue:
    UPDATE_CHAR();
    if (buf[idx] != '\n') goto fail;
    while (buf[idx] == '\n') UPDATE_CHAR();
    offset = pj->scope[--pj->scope_len];
    pj->tape[offset >> 2] |= (uint64_t)pj->tape_len + 1;
    write_tape(pj, offset >> 2, 'r');
    scope_push(pj, ((uint64_t)pj->tape_len << 2) | RET_START);
    write_tape(pj, 0, 'r');
    goto continue_root;

object_begin:
    UPDATE_CHAR();
    switch (buf[idx]) {
    case '"':
        if (!parse_string(pj, idx)) goto fail;
        goto object_key_state;
    case '}':
        goto scope_end;
    default:
        goto fail;
    }

object_key_state:
    UPDATE_CHAR();
    if (buf[idx] != ':') goto fail;
    UPDATE_CHAR();
    switch (buf[idx]) {
    case '"':
        if (!parse_string(pj, idx)) goto fail;
        break;
    case 't':
        if (!sjo_is_valid_true_atom(buf + idx, len - idx)) goto fail;
        write_tape(pj, 0, 't');
        break;
    case 'f':
        if (!sjo_is_valid_false_atom(buf + idx, len - idx)) goto fail;
        write_tape(pj, 0, 'f');
        break;
    case 'n':
        if (!sjo_is_valid_null_atom(buf + idx, len - idx)) goto fail;
        write_tape(pj, 0, 'n');
        break;
    case '-':
        if (!add_number(pj, idx)) goto fail;
        break;
    case '{':
        scope_push(pj, ((uint64_t)pj->tape_len << 2) | RET_OBJECT);
        write_tape(pj, 0, '{');
        goto object_begin;
    case '[':
        scope_push(pj, ((uint64_t)pj->tape_len << 2) | RET_OBJECT);
        write_tape(pj, 0, '[');
        goto array_begin;
    default:
        if (buf[idx] >= '0' && buf[idx] <= '9') {
            if (!add_number(pj, idx)) goto fail;
            break;
        }
        goto fail;
    }

object_continue:
    UPDATE_CHAR();
    switch (buf[idx]) {
    case ',':
        UPDATE_CHAR();
        if (buf[idx] != '"') goto fail;
        if (!parse_string(pj, idx)) goto fail;
        goto object_key_state;
    case '}':
        goto scope_end;
    default:
        goto fail;
    }

scope_end:
    offset = pj->scope[--pj->scope_len];
    write_tape(pj, offset >> 2, buf[idx]);
    pj->tape[offset >> 2] |= (uint64_t)pj->tape_len;
    switch (offset & 3) {
    case RET_ARRAY: goto array_continue;
    case RET_OBJECT: goto object_continue;
    default: goto start_continue;
    }

array_begin:
    UPDATE_CHAR();
    if (buf[idx] == ']') goto scope_end;

main_array_switch:
    switch (buf[idx]) {
    case '"':
        if (!parse_string(pj, idx)) goto fail;
        break;
    case 't':
        if (!sjo_is_valid_true_atom(buf + idx, len - idx)) goto fail;
        write_tape(pj, 0, 't');
        break;
    case 'f':
        if (!sjo_is_valid_false_atom(buf + idx, len - idx)) goto fail;
        write_tape(pj, 0, 'f');
        break;
    case 'n':
        if (!sjo_is_valid_null_atom(buf + idx, len - idx)) goto fail;
        write_tape(pj, 0, 'n');
        break;
    case '-':
        if (!add_number(pj, idx)) goto fail;
        break;
    case '{':
        scope_push(pj, ((uint64_t)pj->tape_len << 2) | RET_ARRAY);
        write_tape(pj, 0, '{');
        goto object_begin;
    case '[':
        scope_push(pj, ((uint64_t)pj->tape_len << 2) | RET_ARRAY);
        write_tape(pj, 0, '[');
        goto array_begin;
    default:
        if (buf[idx] >= '0' && buf[idx] <= '9') {
            if (!add_number(pj, idx)) goto fail;
            break;
        }
        goto fail;
    }

array_continue:
    UPDATE_CHAR();
    switch (buf[idx]) {
    case ',':
        UPDATE_CHAR();
        goto main_array_switch;
    case ']':
        goto scope_end;
    default:
        goto fail;
    }

succeed:
    offset = pj->scope[--pj->scope_len];
    if (pj->scope_len != 0) return 0;
    pj->tape[offset >> 2] |= (uint64_t)pj->tape_len + 1;
    write_tape(pj, offset >> 2, 'r');
    return 1;

fail:
    return 0;
}

/* parseMessage: parse_json_amd64.go:52-127 (stage-1 error wins, :97-105 / :123-126) */
int sjo_parse(const uint8_t *msg, size_t len, uint32_t flags, uint64_t **tape, size_t *tape_len,
              uint8_t **strings, size_t *strings_len, size_t *msg_off, size_t *msg_len) {
    size_t off, mlen;
    sjo_trim_space(msg, len, &off, &mlen);
    if (msg_off) *msg_off = off;
    if (msg_len) *msg_len = mlen;
    *tape = NULL;
    *tape_len = 0;
    *strings = NULL;
    *strings_len = 0;

    size_t cap = mlen + 64;
    uint32_t *pos = (uint32_t *)malloc(sizeof(uint32_t) * cap);
    size_t npos = 0;
    int ok1 = sjo_find_structural_indices(msg + off, mlen, (flags & SJO_FLAG_NDJSON) != 0, pos, cap, &npos);
    if (!ok1) {
        free(pos);
        return SJO_ERR_STAGE1;
    }
    pj_t pj;
    memset(&pj, 0, sizeof pj);
    pj.msg = msg + off;
    pj.len = mlen;
    pj.copy_strings = (flags & SJO_FLAG_COPY_STRINGS) != 0;
    pj.pos = pos;
    pj.npos = npos;
    pj.strs_cap = 128;
    pj.strs = (uint8_t *)malloc(pj.strs_cap);
    int ok2 = sjo_unified_machine(&pj);
    free(pos);
    free(pj.scope);
    if (!ok2) {
        free(pj.tape);
        free(pj.strs);
        return SJO_ERR_STAGE2;
    }
    *tape = pj.tape;
    *tape_len = pj.tape_len;
    *strings = pj.strs;
    *strings_len = pj.strs_len;
    return SJO_OK;
}

void sjo_free(void *p) { free(p); }
