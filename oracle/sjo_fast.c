/*
 * sjo_fast.c -- ORACLE, CPU-baseline variant (test infrastructure only, see sjo.h).
 *
 * The scalar restatement in sjo_stage1.c / sjo_parse_string.c follows the reference instruction by instruction but
 * one byte at a time; timed next to a GPU it is a strawman.  This file restates the same routines with the
 * instructions the reference's assembly uses (AVX2 + PCLMULQDQ + BMI), so that the CPU baseline of bench.py runs in
 * the reference's own shapes:
 *   B1  one thread, stage 1 only                        find_structural_bits_amd64.s:49-155
 *   B2  two threads, stage 1 feeding stage 2            parse_json_amd64.go:75-95 (indexChans)
 *   B3  N threads over 10 MiB NDJSON blocks             simdjson_amd64.go:116-216 (ParseNDStream)
 * Stage 2 is the oracle's unifiedMachine (scalar Go in the reference too); strings use the reference's 32-byte
 * windows (parse_string_amd64.s) with the scalar walk for \u escapes and string tails.
 * Results are bit-identical to the scalar oracle (tests/test_oracle_fast.py); kind = "port" in bench.py.
 */
#define _GNU_SOURCE
#include "sjo.h"
#include "sjo_internal.h"

#include <immintrin.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define TGT __attribute__((target("avx2,pclmul,bmi,bmi2,lzcnt,popcnt")))

int sjo_avx2_available(void) {
    return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("bmi");
}

/* ------------------------------------------------------------------------------------------------------------
 * stage 1, one 64-byte chunk in two YMM registers
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t ends_odd_backslash, inside_quote, pseudo_pred, error_mask;
} s1_carry;

TGT static inline uint64_t eq64(__m256i lo, __m256i hi, char c) { /* VPCMPEQB + VPMOVMSKB pairs */
    const __m256i k = _mm256_set1_epi8(c);
    return (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(lo, k)) |
           ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(hi, k)) << 32);
}

TGT static inline uint64_t chunk_structurals(__m256i lo, __m256i hi, s1_carry *c, int ndjson) {
    const uint64_t even_bits = 0x5555555555555555ULL, odd_bits = ~even_bits;
    /* find_odd_backslash_sequences_amd64.s:34-58 */
    const uint64_t bs = eq64(lo, hi, '\\');
    const uint64_t start_edges = bs & ~(bs << 1);
    const uint64_t prev = c->ends_odd_backslash;
    const uint64_t even_starts = start_edges & (even_bits ^ prev);
    const uint64_t odd_starts = start_edges & (odd_bits ^ prev);
    const uint64_t even_carries = bs + even_starts;
    uint64_t odd_carries;
    c->ends_odd_backslash = __builtin_add_overflow(bs, odd_starts, &odd_carries) ? 1 : 0;
    odd_carries |= prev;
    const uint64_t odd_ends = ((even_carries & ~bs) & odd_bits) | ((odd_carries & ~bs) & even_bits);
    /* find_quote_mask_and_bits_amd64.s:52-83: PCLMULQDQ by all-ones = prefix XOR */
    const uint64_t quote_bits = eq64(lo, hi, '"') & ~odd_ends;
    uint64_t quote_mask =
        (uint64_t)_mm_cvtsi128_si64(_mm_clmulepi64_si128(_mm_set_epi64x(0, (long long)quote_bits), _mm_set1_epi8((char)0xff), 0));
    quote_mask ^= c->inside_quote;
    const __m256i ctl = _mm256_set1_epi8(0x1f);
    const uint64_t unescaped = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_max_epu8(lo, ctl), ctl)) |
                               ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_max_epu8(hi, ctl), ctl)) << 32);
    c->error_mask |= unescaped & quote_mask;
    c->inside_quote = (uint64_t)((int64_t)quote_mask >> 63);
    /* find_whitespace_and_structurals_amd64.s:62-103: two VPSHUFB nibble look-ups, AND, test against 0x07 / 0x18 */
    const __m256i low_nibble_mask = _mm256_setr_epi8(16, 0, 0, 0, 0, 0, 0, 0, 0, 8, 12, 1, 2, 9, 0, 0, 16, 0, 0, 0, 0, 0, 0, 0, 0, 8, 12, 1, 2, 9, 0, 0);
    const __m256i high_nibble_mask = _mm256_setr_epi8(8, 0, 18, 4, 0, 1, 0, 1, 0, 0, 0, 3, 2, 1, 0, 0, 8, 0, 18, 4, 0, 1, 0, 1, 0, 0, 0, 3, 2, 1, 0, 0);
    const __m256i m7f = _mm256_set1_epi8(0x7f), zero = _mm256_setzero_si256();
    const __m256i vlo = _mm256_and_si256(_mm256_shuffle_epi8(low_nibble_mask, lo),
                                         _mm256_shuffle_epi8(high_nibble_mask, _mm256_and_si256(_mm256_srli_epi32(lo, 4), m7f)));
    const __m256i vhi = _mm256_and_si256(_mm256_shuffle_epi8(low_nibble_mask, hi),
                                         _mm256_shuffle_epi8(high_nibble_mask, _mm256_and_si256(_mm256_srli_epi32(hi, 4), m7f)));
    const __m256i s7 = _mm256_set1_epi8(0x07), w18 = _mm256_set1_epi8(0x18);
    const uint64_t structurals0 = ~((uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_and_si256(vlo, s7), zero)) |
                                    ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_and_si256(vhi, s7), zero)) << 32));
    const uint64_t whitespace = ~((uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_and_si256(vlo, w18), zero)) |
                                  ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_and_si256(vhi, w18), zero)) << 32));
    /* finalize_structurals_amd64.s:19-36 */
    uint64_t structurals = (structurals0 & ~quote_mask) | quote_bits;
    const uint64_t pseudo_pred = structurals | whitespace;
    const uint64_t shifted = (pseudo_pred << 1) | c->pseudo_pred;
    c->pseudo_pred = pseudo_pred >> 63;
    structurals |= shifted & ~whitespace & ~quote_mask;
    structurals &= ~(quote_bits & ~quote_mask);
    if (ndjson) structurals |= eq64(lo, hi, '\n') & ~quote_mask; /* find_newline_delimiters_amd64.s:16-28 */
    return structurals;
}

/* flatten_bits_amd64.s:26-60 (TZCNT / BLSR loop), absolute positions instead of deltas */
TGT static inline uint32_t *flatten(uint32_t *out, uint64_t mask, uint32_t base) {
    while (mask) {
        *out++ = base + (uint32_t)_tzcnt_u64(mask);
        mask = _blsr_u64(mask);
    }
    return out;
}

/* findStructuralIndices + _find_structural_bits_in_slice with the index ring replaced by one array; `live` (may be
 * NULL) receives the number of positions written so far every `publish_every` positions (2-thread shape). */
TGT static int stage1_avx2(const uint8_t *msg, size_t len, int ndjson, uint32_t *pos, size_t pos_cap, size_t *n_out,
                           size_t *live) {
    s1_carry c = {0, 0, 1, 0};
    uint32_t *out = pos, *published = pos;
    size_t off = 0;
    const size_t full = len & ~(size_t)63;
    if (pos_cap < len + 64) return 0; /* caller sizes for the worst case: no per-store check */
    for (; off < full; off += 64) {
        const __m256i lo = _mm256_loadu_si256((const __m256i *)(msg + off));
        const __m256i hi = _mm256_loadu_si256((const __m256i *)(msg + off + 32));
        out = flatten(out, chunk_structurals(lo, hi, &c, ndjson), (uint32_t)off);
        if (live && (size_t)(out - published) >= 1408) { /* indexSizeWithSafetyBuffer: one channel send per buffer */
            __atomic_store_n(live, (size_t)(out - pos), __ATOMIC_RELEASE);
            published = out;
        }
    }
    if (off < len) { /* space-masked tail, find_structural_bits_amd64.s:134-155 */
        uint8_t tail[64];
        memset(tail, 0x20, sizeof tail);
        memcpy(tail, msg + off, len - off);
        const __m256i lo = _mm256_loadu_si256((const __m256i *)tail);
        const __m256i hi = _mm256_loadu_si256((const __m256i *)(tail + 32));
        out = flatten(out, chunk_structurals(lo, hi, &c, ndjson), (uint32_t)off);
    }
    const size_t n = (size_t)(out - pos);
    if (live) __atomic_store_n(live, n, __ATOMIC_RELEASE);
    *n_out = n;
    /* stage1_find_marks_amd64.go:115-129,147 */
    if (c.error_mask || n == 0 || c.inside_quote) return 0;
    const uint8_t last = msg[pos[n - 1]];
    return last == '}' || last == ']';
}

int sjo_find_structural_indices_avx2(const uint8_t *msg, size_t len, int ndjson, uint32_t *pos_out, size_t pos_cap,
                                     size_t *n_out) {
    *n_out = 0;
    if (!sjo_avx2_available() || len == 0) return sjo_find_structural_indices(msg, len, ndjson, pos_out, pos_cap, n_out);
    return stage1_avx2(msg, len, ndjson, pos_out, pos_cap, n_out, NULL);
}

/* ------------------------------------------------------------------------------------------------------------
 * strings: parse_string_amd64.s with its 32-byte windows; \u escapes and the last < 32 + 12 bytes of the message
 * take the scalar walk
 * ---------------------------------------------------------------------------------------------------------- */
static const uint8_t ESCAPE_MAP[256] = {['"'] = 0x22, ['/'] = 0x2f, ['\\'] = 0x5c, ['b'] = 0x08, ['f'] = 0x0c,
                                        ['n'] = 0x0a, ['r'] = 0x0d, ['t'] = 0x09};

/* digittoval as the DATA section lays it out (parse_string_amd64.s:12-37, quirk Q3: bytes below 0x30 read as 0) */
static int8_t DIGIT[256];
__attribute__((constructor)) static void init_digit(void) {
    for (int b = 0; b < 256; b++) {
        int8_t v = -1;
        if (b < 0x30) v = 0;
        else if (b >= '0' && b <= '9') v = (int8_t)(b - '0');
        else if (b >= 'A' && b <= 'F') v = (int8_t)(b - 'A' + 10);
        else if (b >= 'a' && b <= 'f') v = (int8_t)(b - 'a' + 10);
        DIGIT[b] = v;
    }
}

TGT static int string_avx2(const uint8_t *src, size_t avail, uint8_t *dst, uint64_t *str_length, uint64_t *dst_length) {
    size_t pos = 0, out = 0;
    const __m256i kb = _mm256_set1_epi8('\\'), kq = _mm256_set1_epi8('"');
    while (pos + 34 <= avail) {
        const __m256i v = _mm256_loadu_si256((const __m256i *)(src + pos));
        const uint32_t bs_bits = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, kb));
        const uint32_t quote_bits = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, kq));
        if (dst) _mm256_storeu_si256((__m256i *)(dst + out), v); /* the caller keeps 64 bytes of slack (stage2:93-104) */
        if (((bs_bits - 1) & quote_bits) != 0) {
            const unsigned q = (unsigned)_tzcnt_u32(quote_bits);
            if (str_length) *str_length = pos + q;
            *dst_length = out + q;
            return 1;
        }
        if (((quote_bits - 1) & bs_bits) == 0) {
            pos += 32;
            out += 32;
            continue;
        }
        const unsigned b = (unsigned)_tzcnt_u32(bs_bits);
        const uint8_t esc = src[pos + b + 1];
        if (esc != 'u') { /* LBB0_26 */
            const uint8_t e = ESCAPE_MAP[esc];
            if (e == 0) return 0;
            if (dst) dst[out + b] = e;
            out += b + 1;
            pos += b + 2;
            continue;
        }
        /* \\uXXXX (LBB0_8 .. LBB0_14): needs the distance from the backslash to the next raw quote; when the window
         * shows no quote and the backslash sits in its last 11 bytes the reference looks at a second window -- that
         * case, and the last bytes of the message, go to the scalar walk */
        if (quote_bits == 0 && b >= 21) break;
        if (pos + b + 12 > avail) break;
        const uint32_t d = quote_bits ? (uint32_t)_tzcnt_u32(quote_bits) - b : 32u - b;
        if (d < 6) return 0;
        const uint8_t *p = src + pos + b;
        uint32_t cp = ((uint32_t)(int32_t)DIGIT[p[2]] << 12) | ((uint32_t)(int32_t)DIGIT[p[3]] << 8) |
                      ((uint32_t)(int32_t)DIGIT[p[4]] << 4) | (uint32_t)(int32_t)DIGIT[p[5]];
        unsigned adv = 6;
        if ((cp & 0xfffffc00u) == 0xd800u) { /* LBB0_12: surrogate pair, low half unchecked, 32-bit wrap-around */
            if (d < 12 || p[6] != '\\' || p[7] != 'u') return 0;
            const uint32_t cp2 = ((uint32_t)(int32_t)DIGIT[p[8]] << 12) | ((uint32_t)(int32_t)DIGIT[p[9]] << 8) |
                                 ((uint32_t)(int32_t)DIGIT[p[10]] << 4) | (uint32_t)(int32_t)DIGIT[p[11]];
            if ((cp2 | cp) > 0xffffu) return 0;
            cp = (((cp << 10) + 0xfca00000u) | (cp2 + 0xffff2400u)) + 0x10000u;
            adv = 12;
        }
        uint8_t enc[4];
        unsigned n;
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
        pos += b + adv;
    }
    return sjo_string_walk_from(src, avail, dst, pos, out, str_length, dst_length);
}
static int validate_avx2(const uint8_t *src, size_t avail, uint64_t *str_length, uint64_t *dst_length) {
    return string_avx2(src, avail, NULL, str_length, dst_length);
}
static int copy_avx2(const uint8_t *src, size_t avail, uint8_t *dst, uint64_t *dst_length) {
    return string_avx2(src, avail, dst, NULL, dst_length);
}

/* ------------------------------------------------------------------------------------------------------------
 * whole parse with recycled buffers (`reuse *ParsedJson`), 1 or 2 threads
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    const uint8_t *msg;
    size_t len;
    int ndjson;
    uint32_t *pos;
    size_t pos_cap, n;
    size_t live;
    int done, ok;
} s1_job;

TGT static int stage1_avx2(const uint8_t *msg, size_t len, int ndjson, uint32_t *pos, size_t pos_cap, size_t *n_out,
                           size_t *live);

typedef struct sjo_fast {
    uint32_t *pos;
    size_t pos_cap;
    pj_t pj; /* tape / strs / scope capacities are kept across calls */
    /* the stage-1 goroutine of the 2-thread shape: one helper thread per workspace, parked on a condition variable */
    pthread_t helper;
    int helper_started, helper_quit, job_pending;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    s1_job job;
} sjo_fast;

static void *helper_main(void *arg) {
    sjo_fast *w = (sjo_fast *)arg;
    pthread_mutex_lock(&w->mu);
    for (;;) {
        while (!w->job_pending && !w->helper_quit) pthread_cond_wait(&w->cv, &w->mu);
        if (w->helper_quit) break;
        w->job_pending = 0;
        pthread_mutex_unlock(&w->mu);
        s1_job *j = &w->job;
        j->ok = stage1_avx2(j->msg, j->len, j->ndjson, j->pos, j->pos_cap, &j->n, &j->live);
        __atomic_store_n(&j->done, 1, __ATOMIC_RELEASE);
        pthread_mutex_lock(&w->mu);
    }
    pthread_mutex_unlock(&w->mu);
    return NULL;
}

sjo_fast *sjo_fast_create(void) {
    sjo_fast *w = (sjo_fast *)calloc(1, sizeof(sjo_fast));
    pthread_mutex_init(&w->mu, NULL);
    pthread_cond_init(&w->cv, NULL);
    return w;
}
void sjo_fast_destroy(sjo_fast *w) {
    if (!w) return;
    if (w->helper_started) {
        pthread_mutex_lock(&w->mu);
        w->helper_quit = 1;
        pthread_cond_signal(&w->cv);
        pthread_mutex_unlock(&w->mu);
        pthread_join(w->helper, NULL);
    }
    free(w->pos);
    free(w->pj.tape);
    free(w->pj.strs);
    free(w->pj.scope);
    free(w);
}

/* parseMessage (parse_json_amd64.go:52-127).  threads = 2: stage 1 on its own thread, stage 2 consumes the
 * positions as they are published (the reference does this for messages above 8 KiB).  The outputs stay owned by
 * the workspace (valid until the next call). */
int sjo_fast_parse(sjo_fast *w, const uint8_t *msg, size_t len, uint32_t flags, int threads, const uint64_t **tape,
                   size_t *tape_len, const uint8_t **strings, size_t *strings_len) {
    size_t off, mlen;
    sjo_trim_space(msg, len, &off, &mlen);
    *tape_len = *strings_len = 0;
    if (mlen == 0) return SJO_ERR_STAGE1;
    if (!sjo_avx2_available()) return -1;
    if (w->pos_cap < mlen + 64) {
        free(w->pos);
        w->pos_cap = mlen + 64 + mlen / 8;
        w->pos = (uint32_t *)malloc(w->pos_cap * sizeof(uint32_t));
    }
    pj_t *pj = &w->pj;
    pj->msg = msg + off;
    pj->len = mlen;
    pj->copy_strings = (flags & SJO_FLAG_COPY_STRINGS) != 0;
    pj->tape_len = pj->strs_len = pj->scope_len = 0;
    pj->pos = w->pos;
    pj->ipos = 0;
    pj->validate_string = validate_avx2;
    pj->copy_string = copy_avx2;
    if (!pj->strs) {
        pj->strs_cap = 128;
        pj->strs = (uint8_t *)malloc(pj->strs_cap);
    }
    int ok1, ok2;
    if (threads >= 2) {
        const s1_job job = {msg + off, mlen, (flags & SJO_FLAG_NDJSON) != 0, w->pos, w->pos_cap, 0, 0, 0, 0};
        w->job = job;
        pj->npos = 0;
        pj->live_npos = &w->job.live;
        pj->live_done = &w->job.done;
        pthread_mutex_lock(&w->mu);
        if (!w->helper_started) {
            pthread_create(&w->helper, NULL, helper_main, w);
            w->helper_started = 1;
        }
        w->job_pending = 1;
        pthread_cond_signal(&w->cv);
        pthread_mutex_unlock(&w->mu);
        ok2 = sjo_unified_machine(pj);
        while (!__atomic_load_n(&w->job.done, __ATOMIC_ACQUIRE)) __builtin_ia32_pause(); /* wg.Wait() */
        pj->live_npos = NULL;
        pj->live_done = NULL;
        ok1 = w->job.ok;
    } else {
        size_t n = 0;
        pj->live_npos = NULL;
        pj->live_done = NULL;
        ok1 = stage1_avx2(msg + off, mlen, (flags & SJO_FLAG_NDJSON) != 0, w->pos, w->pos_cap, &n, NULL);
        pj->npos = n;
        ok2 = ok1 ? sjo_unified_machine(pj) : 0;
    }
    if (!ok1) return SJO_ERR_STAGE1; /* the stage-1 error wins (:97-105, :123-126) */
    if (!ok2) return SJO_ERR_STAGE2;
    *tape = pj->tape;
    *tape_len = pj->tape_len;
    *strings = pj->strs;
    *strings_len = pj->strs_len;
    return SJO_OK;
}

/* ------------------------------------------------------------------------------------------------------------
 * timed loops for bench.py (seconds per pass, best of `iters`)
 * ---------------------------------------------------------------------------------------------------------- */
static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* B1: stage 1 only, one thread.  avx2 = 0 times the scalar restatement. */
double sjo_bench_stage1(const uint8_t *msg, size_t len, int ndjson, int iters, int avx2, size_t *n_out) {
    uint32_t *pos = (uint32_t *)malloc((len + 64) * sizeof(uint32_t));
    double best = 1e30;
    size_t n = 0;
    for (int i = 0; i < iters; i++) {
        const double t0 = now_s();
        if (avx2) sjo_find_structural_indices_avx2(msg, len, ndjson, pos, len + 64, &n);
        else sjo_find_structural_indices(msg, len, ndjson, pos, len + 64, &n);
        const double dt = now_s() - t0;
        if (dt < best) best = dt;
    }
    if (n_out) *n_out = n;
    free(pos);
    return best;
}

/* B1 full parse (threads = 1) / B2 (threads = 2): the reference's benchmark loop (reuse, bytes / time) */
double sjo_bench_parse(const uint8_t *msg, size_t len, uint32_t flags, int threads, int iters, int *rc_out,
                       size_t *tape_len_out) {
    sjo_fast *w = sjo_fast_create();
    const uint64_t *tape;
    const uint8_t *strs;
    size_t tl = 0, sl = 0;
    double best = 1e30;
    int rc = 0;
    for (int i = 0; i < iters + 1; i++) { /* pass 0 sizes the buffers (like the reference's first b.N iteration) */
        const double t0 = now_s();
        rc = sjo_fast_parse(w, msg, len, flags, threads, &tape, &tl, &strs, &sl);
        const double dt = now_s() - t0;
        if (i > 0 && dt < best) best = dt;
    }
    if (rc_out) *rc_out = rc;
    if (tape_len_out) *tape_len_out = tl;
    sjo_fast_destroy(w);
    return best;
}

/* B3: ParseNDStream's shape -- the input is cut into blocks of about `block_bytes` at newlines
 * (simdjson_amd64.go:127-155), `threads` workers parse blocks independently with recycled buffers. */
typedef struct {
    const uint8_t *msg;
    const size_t *cut; /* nblocks + 1 offsets */
    size_t nblocks;
    size_t next;
    int failed;
} nd_pool;

static void *nd_worker(void *arg) {
    nd_pool *p = (nd_pool *)arg;
    sjo_fast *w = sjo_fast_create();
    for (;;) {
        const size_t b = __atomic_fetch_add(&p->next, 1, __ATOMIC_RELAXED);
        if (b >= p->nblocks) break;
        const uint64_t *tape;
        const uint8_t *strs;
        size_t tl, sl;
        const int rc = sjo_fast_parse(w, p->msg + p->cut[b], p->cut[b + 1] - p->cut[b], SJO_FLAG_NDJSON | SJO_FLAG_COPY_STRINGS,
                                      1, &tape, &tl, &strs, &sl);
        if (rc != SJO_OK) __atomic_store_n(&p->failed, 1, __ATOMIC_RELAXED);
    }
    sjo_fast_destroy(w);
    return NULL;
}

double sjo_bench_nd_blocks(const uint8_t *msg, size_t len, int threads, size_t block_bytes, int iters, int *failed_out) {
    size_t cap = len / block_bytes + 4, nb = 0;
    size_t *cut = (size_t *)malloc(cap * sizeof(size_t));
    cut[0] = 0;
    while (cut[nb] < len) {
        size_t e = cut[nb] + block_bytes;
        if (e >= len) e = len;
        else {
            const uint8_t *nl = (const uint8_t *)memchr(msg + e, '\n', len - e);
            e = nl ? (size_t)(nl - msg) + 1 : len;
        }
        cut[++nb] = e;
    }
    double best = 1e30;
    int failed = 0;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    for (int i = 0; i < iters; i++) {
        nd_pool pool = {msg, cut, nb, 0, 0};
        const double t0 = now_s();
        for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, nd_worker, &pool);
        for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
        const double dt = now_s() - t0;
        if (dt < best) best = dt;
        failed |= pool.failed;
    }
    if (failed_out) *failed_out = failed;
    free(th);
    free(cut);
    return best;
}
