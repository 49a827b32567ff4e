// query.hip -- queries over the DEVICE-RESIDENT result of the last parse of a context.
//
// The tape is up to 1.7x the input and PCIe moves ~55 GB/s, so a parse whose result has to cross to the host is
// D2H-bound by more than 10x; what callers of ParseND usually want is an aggregate or a subset of the records:
//     countWhere("Make", "HOND", pj)                      ndjson_test.go:421-471 (README example :226-269)
//     Object.FindKey(key) per record + string compare     parsed_object.go:97-138
// sjhip_count_where evaluates exactly that on the device and returns 8 bytes; sjhip_filter_where compacts the
// matching records into a new, self-contained (Tape, Strings.B) on the device -- bit-identical to what ParseND
// produces for the document made of the matching lines -- so that only the subset crosses PCIe.
//
// Semantics of a match (the reference's countWhere): the record's root value is an object; the FIRST member of that
// object whose key equals `key` (top level only, FindKey does not descend) has a string value equal to `value`.
// Strings are compared after unescaping (they are read from Strings.B / the message exactly as the Iter API does).
#include <hip/hip_runtime.h>
#include <string.h>

#include "../../include/sjhip.h"
#include "sj_ctx.h"
#include "sj_device.h"
#include "sj_stage2.h"

using namespace sj;

#define HIPCHK(call, what)                                        \
    do {                                                          \
        hipError_t e_ = (call);                                   \
        if (e_ != hipSuccess) return ctx_hip_fail(ctx, e_, what); \
    } while (0)

namespace {

static constexpr u64 PAYLOAD = 0x00ffffffffffffffull;  // JSONVALUEMASK, parsed_json.go:27
static constexpr u32 NONE32 = 0xffffffffu;
static constexpr int QMAX = 1024;  // longest key / value a query may name (they travel as kernel arguments)

struct QView {
    const u64 *tape;
    u64 tape_len;
    const u8 *strings;
    u64 strings_len;
    const u8 *msg;     // device copy of the message (strings that were not copied point into it)
    const u32 *nl_off; // tape offset of the close root of record r (r < R)
    u32 R;             // record-separating newline runs: R + 1 records
    u8 key[QMAX], val[QMAX];
    u32 klen, vlen;
};

__device__ __forceinline__ u32 rec_open(const QView &q, u32 r) { return r == 0 ? 0u : q.nl_off[r - 1] + 1u; }
__device__ __forceinline__ u32 rec_close(const QView &q, u32 r) { return r == q.R ? (u32)q.tape_len - 1u : q.nl_off[r]; }

__device__ __forceinline__ const u8 *str_bytes(const QView &q, u64 word) {
    const u64 p = word & PAYLOAD;
    return (p & STRINGBUFBIT) ? q.strings + (p & (STRINGBUFBIT - 1)) : q.msg + p;
}
__device__ __forceinline__ bool str_equals(const QView &q, u64 word, u64 len, const u8 *want, u32 wlen) {
    if (len != wlen) return false;
    const u8 *s = str_bytes(q, word);
    for (u32 k = 0; k < wlen; k++)
        if (s[k] != want[k]) return false;
    return true;
}

// FindKey(key) on the root object of record r + the string compare of countWhere
__device__ bool record_matches(const QView &q, u32 r) {
    const u32 o = rec_open(q, r);
    const u64 w = q.tape[o + 1];
    if ((w >> 56) != '{') return false;
    const u64 end = (w & PAYLOAD) - 1;  // index of the closing '}'
    u64 i = (u64)o + 2;
    while (i < end) {
        const u64 kw = q.tape[i], kl = q.tape[i + 1];  // member key
        const u64 v = i + 2, vw = q.tape[v];
        const u32 vt = (u32)(vw >> 56);
        if (str_equals(q, kw, kl, q.key, q.klen))
            return vt == '"' && str_equals(q, vw, q.tape[v + 1], q.val, q.vlen);  // FindKey returns the first match
        if (vt == '{' || vt == '[') i = vw & PAYLOAD;  // behind the matching close
        else if (vt == '"' || vt == 'l' || vt == 'u' || vt == 'd') i = v + 2;
        else i = v + 1;  // t f n
    }
    return false;
}

__global__ __launch_bounds__(256) void k_q_count(QView q, unsigned long long *count) {
    const u32 r = blockIdx.x * 256 + threadIdx.x;
    const bool m = r <= q.R && record_matches(q, r);
    const u64 b = __ballot(m);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(count, (unsigned long long)__popcll(b));
}

// ---- filter: pass 1, one lane per record ------------------------------------------------------------------------
struct QRec {
    u32 *flag;      // [R+1] 1 if the record matches
    u32 *words;     // [R+1] its tape words if it matches, else 0      -> exclusive prefix = new index of its open root
    u32 *first_str; // [R+1] Strings.B offset of its first string, NONE32 if it has none
    u32 *s_len;     // [R+1] bytes of Strings.B a matching record owns (pass 2)
    u32 *s_pre;     // [R+1] their exclusive prefix = new Strings.B offset of its first string (pass 2)
    unsigned long long *totals;  // matching records, tape words, Strings.B bytes
};

__global__ __launch_bounds__(256) void k_q_mark(QView q, QRec o) {
    const u32 r = blockIdx.x * 256 + threadIdx.x;
    if (r > q.R) return;
    const bool m = record_matches(q, r);
    const u32 a = rec_open(q, r), c = rec_close(q, r);
    o.flag[r] = m ? 1u : 0u;
    o.words[r] = m ? c - a + 1u : 0u;
    // first string of the record: walk its items (a number's second word is raw data and must be stepped over)
    u32 fs = NONE32;
    for (u64 i = (u64)a + 1; i < c;) {
        const u64 w = q.tape[i];
        const u32 t = (u32)(w >> 56);
        if (t == '"') {
            fs = (u32)(w & (STRINGBUFBIT - 1));  // every string is copied (checked by the host): a Strings.B offset
            break;
        }
        i += (t == 'l' || t == 'u' || t == 'd') ? 2 : 1;
    }
    o.first_str[r] = fs;
}

// ---- pass 2: one block.  Exclusive prefixes of words / string bytes over the records, and the Strings.B range of
// every record: [its first string, the first string of any later record) -- strings are laid out in document order.
__global__ __launch_bounds__(1024) void k_q_scan(QView q, QRec o) {
    __shared__ unsigned long long s_w[1024], s_c[1024], s_b[1024];
    __shared__ u32 s_first[1024];
    const u32 n = q.R + 1, tid = threadIdx.x;
    const u32 per = (n + 1023u) / 1024u;
    const u32 lo = tid * per < n ? tid * per : n, hi = lo + per < n ? lo + per : n;
    unsigned long long w = 0, c = 0;
    u32 first = NONE32;
    for (u32 r = lo; r < hi; r++) {
        w += o.words[r];
        c += o.flag[r];
        if (first == NONE32) first = o.first_str[r];
    }
    s_w[tid] = w;
    s_c[tid] = c;
    s_first[tid] = first;
    __syncthreads();
    if (tid == 0) {  // 1024 partial sums: serial is fine next to the record walks
        unsigned long long aw = 0, ac = 0;
        for (int k = 0; k < 1024; k++) {
            const unsigned long long tw = s_w[k], tc = s_c[k];
            s_w[k] = aw;
            aw += tw;
            ac += tc;
        }
        o.totals[0] = ac;
        o.totals[1] = aw;
        u32 nxt = (u32)q.strings_len;  // s_first[k] := the first string behind chunk k
        for (int k = 1023; k >= 0; k--) {
            const u32 f = s_first[k];
            s_first[k] = nxt;
            if (f != NONE32) nxt = f;
        }
    }
    __syncthreads();
    u32 nxt = s_first[tid];  // Strings.B ranges, walking the chunk backwards
    unsigned long long b = 0;
    for (u32 r = hi; r-- > lo;) {
        const u32 f = o.first_str[r];
        u32 len = 0;
        if (f != NONE32) {
            len = nxt - f;
            nxt = f;
        }
        len = o.flag[r] ? len : 0u;
        o.s_len[r] = len;
        b += len;
    }
    s_b[tid] = b;
    __syncthreads();
    if (tid == 0) {
        unsigned long long ab = 0;
        for (int k = 0; k < 1024; k++) {
            const unsigned long long tb = s_b[k];
            s_b[k] = ab;
            ab += tb;
        }
        o.totals[2] = ab;
    }
    __syncthreads();
    unsigned long long pw = s_w[tid], pb = s_b[tid];
    for (u32 r = lo; r < hi; r++) {
        const u32 tw = o.words[r];
        o.words[r] = (u32)pw;
        o.s_pre[r] = (u32)pb;
        pw += tw;
        pb += o.s_len[r];
    }
}

// ---- pass 3: one wave per record: its tape words with every stored index rebased, then its strings -------------------
// Which words are tags?  A number's value and a string's length are raw 64-bit data whose top byte can look like any
// tag.  Raw words only ever follow a two-word tag (" l u d) that is itself not raw, so with c(i) = "the top byte of
// word i is one of \" l u d" and p = the last index below i with c(p) = 0 (the opening root word always is one):
//     word i is raw  <=>  i - p - 1 is odd
// which a wave evaluates for 64 words at a time from one ballot.
__global__ __launch_bounds__(256) void k_q_copy(QView q, QRec o, u64 *out_tape, u8 *out_strings, u32 total_words) {
    const u32 r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r > q.R || !o.flag[r]) return;  // wave-uniform
    const u32 a = rec_open(q, r), c = rec_close(q, r);
    const u32 na = o.words[r];  // new index of the opening root
    const u32 nwords = c - a + 1u;
    const long long dw = (long long)na - (long long)a;
    const u32 sb = o.first_str[r], slen = o.s_len[r], ns = o.s_pre[r];
    const u64 ds = (u64)((long long)ns - (long long)sb);  // only used when the record has a string
    long long p_prev = -1;  // last index below the group with c = 0, relative to the record (none: word 0 is the first)
    for (u32 g = 0; g < nwords; g += 64) {
        const u32 i = g + (u32)lane;
        const bool in = i < nwords;
        const u64 w = in ? q.tape[a + i] : 0;
        const u32 t = (u32)(w >> 56);
        const bool two = t == '"' || t == 'l' || t == 'u' || t == 'd';
        const u64 inmask = __ballot(in);
        const u64 zm = ~__ballot(in && two) & inmask;  // words of the group with c = 0
        const u64 zeros_below = zm & (lane ? (~0ull >> (64 - lane)) : 0ull);
        const long long p = zeros_below ? (long long)g + (63 - __builtin_clzll(zeros_below)) : p_prev;
        const bool raw = p >= 0 && ((((long long)i - p - 1) & 1) != 0);
        if (in) {
            u64 v = w;
            if (!raw) {
                if (i == 0) v = ((u64)'r' << 56) | (u64)(na + nwords);  // the next record's open root, or the tape length
                else if (i == nwords - 1) v = ((u64)'r' << 56) | (u64)na;  // its own open root
                else if (t == '{' || t == '[' || t == '}' || t == ']') v = (w & ~PAYLOAD) | (u64)((long long)(w & PAYLOAD) + dw);
                else if (t == '"') v = w + ds;
            }
            out_tape[na + i] = v;
        }
        if (zm) p_prev = (long long)g + (63 - __builtin_clzll(zm));
    }
    (void)total_words;
    for (u32 k = (u32)lane; k < slen; k += 64) out_strings[ns + k] = q.strings[sb + k];
}

}  // namespace

namespace sj {
// nl_off and the record count of the last stage-2 run in this workspace (stage2.hip)
void stage2_records_view(void *ws, size_t n_tokens, const uint32_t **nl_off, const S2State **st);
}

static int make_view(sjhip_ctx *ctx, const uint8_t *key, size_t klen, const uint8_t *val, size_t vlen, QView *q,
                     uint32_t *records) {
    if (!ctx || !key || !val) return SJHIP_ERR_ARG;
    if (klen > QMAX || vlen > QMAX) {
        ctx_set_error(ctx, "query key / value longer than %d bytes", QMAX);
        return SJHIP_ERR_ARG;
    }
    if (!ctx->q_valid || ctx->tape_len == 0) {
        ctx_set_error(ctx, "no parse result on the device (queries follow a successful sjhip_parse / sjhip_parse_device)");
        return SJHIP_ERR_ARG;
    }
    const uint32_t *nl = nullptr;
    const S2State *st = nullptr;
    stage2_records_view(ctx->d_s2.p, ctx->p_n, &nl, &st);
    q->tape = (const u64 *)ctx->d_tape.p;
    q->tape_len = ctx->tape_len;
    q->strings = (const u8 *)ctx->d_strings.p;
    q->strings_len = ctx->strings_len;
    q->msg = (const u8 *)ctx->p_msg;
    q->nl_off = nl;
    q->R = ctx->q_records;
    memset(q->key, 0, QMAX);
    memset(q->val, 0, QMAX);
    memcpy(q->key, key, klen);
    memcpy(q->val, val, vlen);
    q->klen = (u32)klen;
    q->vlen = (u32)vlen;
    *records = ctx->q_records + 1u;
    return SJHIP_OK;
}

int sjhip_count_where(sjhip_ctx *ctx, const uint8_t *key, size_t klen, const uint8_t *value, size_t vlen, uint64_t *count) {
    if (!count) return SJHIP_ERR_ARG;
    QView q;
    uint32_t n = 0;
    int rc = make_view(ctx, key, klen, value, vlen, &q, &n);
    if (rc) return rc;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    rc = arena_reserve(ctx, ctx->d_kat, 64);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(ctx->d_kat.p, 0, 8, ctx->stream), "count memset");
    hipLaunchKernelGGL(k_q_count, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, q, (unsigned long long *)ctx->d_kat.p);
    HIPCHK(hipGetLastError(), "count launch");
    unsigned long long *h = (unsigned long long *)(ctx->h_scratch + 512);
    HIPCHK(hipMemcpyAsync(h, ctx->d_kat.p, 8, hipMemcpyDeviceToHost, ctx->stream), "D2H count");
    HIPCHK(hipStreamSynchronize(ctx->stream), "count sync");
    *count = *h;
    return SJHIP_OK;
}

int sjhip_filter_where(sjhip_ctx *ctx, const uint8_t *key, size_t klen, const uint8_t *value, size_t vlen,
                       uint64_t *n_records, size_t *tape_len, size_t *strings_len) {
    QView q;
    uint32_t n = 0;
    int rc = make_view(ctx, key, klen, value, vlen, &q, &n);
    if (rc) return rc;
    if (!(ctx->p_flags & SJHIP_FLAG_COPY_STRINGS)) {
        ctx_set_error(ctx, "sjhip_filter_where needs a parse with SJHIP_FLAG_COPY_STRINGS (the filtered Strings.B is self-contained)");
        return SJHIP_ERR_ARG;
    }
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    ctx->ser_valid = 0;  // the serializer's columns live in the same arenas
    ctx->ms_valid = 0;
    const size_t per = ((size_t)n * 4 + 255) / 256 * 256;
    rc = arena_reserve(ctx, ctx->d_q, per * 5 + 256);
    if (rc) return rc;
    rc = arena_reserve(ctx, ctx->d_qtape, ctx->tape_len * 8 + 64);
    if (rc) return rc;
    rc = arena_reserve(ctx, ctx->d_qstrings, ctx->strings_len + 64);
    if (rc) return rc;
    char *w = (char *)ctx->d_q.p;
    QRec o;
    o.totals = (unsigned long long *)w;
    w += 256;
    o.flag = (u32 *)w;
    w += per;
    o.words = (u32 *)w;
    w += per;
    o.first_str = (u32 *)w;
    w += per;
    o.s_len = (u32 *)w;
    w += per;
    o.s_pre = (u32 *)w;
    hipLaunchKernelGGL(k_q_mark, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, q, o);
    hipLaunchKernelGGL(k_q_scan, dim3(1), dim3(1024), 0, ctx->stream, q, o);
    HIPCHK(hipGetLastError(), "filter launch");
    unsigned long long *h = (unsigned long long *)(ctx->h_scratch + 512);
    HIPCHK(hipMemcpyAsync(h, o.totals, 24, hipMemcpyDeviceToHost, ctx->stream), "D2H totals");
    HIPCHK(hipStreamSynchronize(ctx->stream), "filter sync");
    ctx->q_tape_len = (size_t)h[1];
    ctx->q_strings_len = (size_t)h[2];
    if (n_records) *n_records = h[0];
    if (tape_len) *tape_len = ctx->q_tape_len;
    if (strings_len) *strings_len = ctx->q_strings_len;
    if (h[0] == 0) return SJHIP_OK;
    hipLaunchKernelGGL(k_q_copy, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, q, o, (u64 *)ctx->d_qtape.p, (u8 *)ctx->d_qstrings.p,
                       (u32)h[1]);
    HIPCHK(hipGetLastError(), "filter copy launch");
    return SJHIP_OK;
}

int sjhip_fetch_filtered(sjhip_ctx *ctx, uint64_t *tape_dst, uint8_t *strings_dst) {
    if (!ctx) return SJHIP_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    if (ctx->q_tape_len && tape_dst)
        HIPCHK(hipMemcpyAsync(tape_dst, ctx->d_qtape.p, ctx->q_tape_len * 8, hipMemcpyDeviceToHost, ctx->stream), "D2H filtered tape");
    if (ctx->q_strings_len && strings_dst)
        HIPCHK(hipMemcpyAsync(strings_dst, ctx->d_qstrings.p, ctx->q_strings_len, hipMemcpyDeviceToHost, ctx->stream),
               "D2H filtered strings");
    HIPCHK(hipStreamSynchronize(ctx->stream), "fetch sync");
    return SJHIP_OK;
}
