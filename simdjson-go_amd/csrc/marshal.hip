// marshal.hip -- Iter.MarshalJSONBuffer on the device-resident tape (parsed_json.go:401-556):
// tape + Strings.B -> compact JSON text, records separated by '\n'.
//
// The reference walks the tape with a stack and appends to a byte slice.  Here every tape entry computes the length of
// its own text, a prefix sum gives every entry its position, and a second pass writes:
//   { [ } ]                 the character
//   "..."                   '"' + escapeBytes (:1190-1238) + '"'
//   l / u / d               strconv.AppendInt / AppendUint / appendFloat (sj_ftoa.h: the reference's Ryu copy)
//   t f n                   true false null
//   closing root            '\n' unless it is the last word of the tape (:451-453)
// plus the separator behind an entry that completes a value (a scalar, a string that is not a key, a closing
// bracket): ',' unless the next entry is a closing bracket or a closing root (:534-549).  A key is followed by ':'
// instead -- and because a key is always followed by a value (never by a closing bracket), ':' and the ',' a value
// would get have the same length: the length pass does not need to know which strings are keys, only the writing pass.
// Keys are identified from the parser's own token array, which is still on the device: string token k of the message is
// string entry k of the tape, and it is a key iff the token behind it is ':' (k_ms_keys).
// Tag words are told from raw words (the second word of a string / number entry) with the parity rule of sj_tapewalk.h.
#include <hip/hip_runtime.h>
#include <string.h>

#include "../../include/sjhip.h"
#include "sj_ctx.h"
#include "sj_device.h"
#include "sj_ftoa.h"
#include "sj_stage2.h"
#include "sj_tapewalk.h"

using namespace sj;

#define HIPCHK(call, what)                                        \
    do {                                                          \
        hipError_t e_ = (call);                                   \
        if (e_ != hipSuccess) return ctx_hip_fail(ctx, e_, what); \
    } while (0)

namespace {

struct MsView {
    const u64 *tape;
    u64 n;
    u32 tiles;
    const u8 *strings;
    const u8 *msg;
    long long *tile_last;        // [tiles] sj_tapewalk.h
    unsigned long long *cnt_b;   // [tiles] text bytes of the tile           -> exclusive prefix
    unsigned long long *cnt_s;   // [tiles] string entries of the tile       -> exclusive prefix
    unsigned long long *totals;  // text bytes, string entries, error flag
    const u8 *keyflag;           // [string entries] 1: the string is an object key
    u32 *slen;                   // [n] escaped length of the string whose tag word is tape[i] (counting pass -> writing pass)
    u8 *text;
};

struct KeyView {
    const u8 *kind;  // [n] token kinds (stage 1)
    u32 n;
    u32 tiles;       // 4096 tokens each
    unsigned long long *cnt;  // [tiles] string tokens -> exclusive prefix
    u8 *keyflag;
};

__device__ __forceinline__ const u8 *entry_string(const MsView &p, u64 word) {
    const u64 v = word & TW_PAYLOAD;
    return (v & STRINGBUFBIT) ? p.strings + (v & (STRINGBUFBIT - 1)) : p.msg + v;
}

// escapeBytes: bytes below 0x20, '"' and '\\' are escaped (shouldEscape, parsed_json.go:1171-1186)
__device__ __forceinline__ u32 escaped_size(u8 c) {
    if (c == '"' || c == '\\') return 2;
    if (c >= 0x20) return 1;
    return (c == '\b' || c == '\f' || c == '\n' || c == '\r' || c == '\t') ? 2u : 6u;
}
__device__ u64 escaped_length(const u8 *s, u64 len) {
    u64 n = 0, k = 0;
    for (; k + 8 <= len; k += 8) {  // eight bytes at a time while nothing needs an escape
        u64 w;
        memcpy(&w, s + k, 8);
        const u64 lo = (w & 0x7f7f7f7f7f7f7f7full);
        const u64 ctl = ~((lo + 0x6060606060606060ull) | w) & 0x8080808080808080ull;              // byte < 0x20
        const u64 q = zero_bytes(w ^ 0x2222222222222222ull) | zero_bytes(w ^ 0x5c5c5c5c5c5c5c5cull);  // '"' '\\'
        if ((ctl | q) == 0) {
            n += 8;
            continue;
        }
        for (int j = 0; j < 8; j++) n += escaped_size(s[k + j]);
    }
    for (; k < len; k++) n += escaped_size(s[k]);
    return n;
}
__device__ __forceinline__ u8 *write_escaped_byte(u8 *o, u8 c) {
    const char *hex = "0123456789abcdef";
    const u32 sz = escaped_size(c);
    if (sz == 1) {
        *o++ = c;
    } else if (sz == 2) {
        *o++ = '\\';
        *o++ = c == '\b' ? 'b' : c == '\f' ? 'f' : c == '\n' ? 'n' : c == '\r' ? 'r' : c == '\t' ? 't' : c;
    } else {
        o[0] = '\\';
        o[1] = 'u';
        o[2] = '0';
        o[3] = '0';
        o[4] = (u8)hex[c >> 4];
        o[5] = (u8)hex[c & 15];
        o += 6;
    }
    return o;
}
__device__ u8 *write_escaped(u8 *o, const u8 *s, u64 len) {
    u64 k = 0;
    for (; k + 8 <= len; k += 8) {  // eight bytes at a time while nothing needs an escape (the test of escaped_length)
        u64 w;
        memcpy(&w, s + k, 8);
        const u64 lo = (w & 0x7f7f7f7f7f7f7f7full);
        const u64 ctl = ~((lo + 0x6060606060606060ull) | w) & 0x8080808080808080ull;
        const u64 q = zero_bytes(w ^ 0x2222222222222222ull) | zero_bytes(w ^ 0x5c5c5c5c5c5c5c5cull);
        if ((ctl | q) == 0) {
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = (u8)(w >> (8 * j));  // (the destination has no alignment: byte stores)
            o += 8;
        } else {
            for (int j = 0; j < 8; j++) o = write_escaped_byte(o, (u8)(w >> (8 * j)));
        }
    }
    for (; k < len; k++) o = write_escaped_byte(o, s[k]);
    return o;
}

// ---- keys: string token k is a key iff the next token is ':' ------------------------------------------------------
template <bool EMIT>
__global__ __launch_bounds__(256) void k_ms_keys(KeyView p) {
    __shared__ unsigned long long s_s[4];
    const int tid = threadIdx.x;
    const u32 base = blockIdx.x * 4096u + (u32)tid * 16u;
    u32 cnt = 0;
    u8 kd[17];
#pragma unroll
    for (int k = 0; k <= 16; k++) kd[k] = base + k < p.n ? p.kind[base + k] : (u8)K_BAD;
#pragma unroll
    for (int k = 0; k < 16; k++) cnt += kd[k] == K_STRING ? 1u : 0u;
    unsigned long long tot = 0;
    const unsigned long long ex = block_excl_sum(cnt, s_s, tid, &tot);
    if (!EMIT) {
        if (tid == 0) p.cnt[blockIdx.x] = tot;
        return;
    }
    u64 o = p.cnt[blockIdx.x] + ex;
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (kd[k] == K_STRING) p.keyflag[o++] = kd[k + 1] == K_COLON ? 1 : 0;
}

// one block: exclusive prefix sums of up to two per-tile counts; totals[0..1]
__global__ __launch_bounds__(1024) void k_ms_scan(unsigned long long *a, unsigned long long *b, u32 tiles,
                                                  unsigned long long *totals) {
    __shared__ unsigned long long s_a[1024], s_b[1024];
    const u32 tid = threadIdx.x, per = (tiles + 1023u) / 1024u;
    const u32 lo = tid * per < tiles ? tid * per : tiles, hi = lo + per < tiles ? lo + per : tiles;
    unsigned long long x = 0, y = 0;
    for (u32 t = lo; t < hi; t++) {
        x += a[t];
        if (b) y += b[t];
    }
    s_a[tid] = x;
    s_b[tid] = y;
    __syncthreads();
    if (tid == 0) {
        unsigned long long ra = 0, rb = 0;
        for (int k = 0; k < 1024; k++) {
            const unsigned long long va = s_a[k], vb = s_b[k];
            s_a[k] = ra;
            s_b[k] = rb;
            ra += va;
            rb += vb;
        }
        if (totals) {
            totals[0] = ra;
            totals[1] = rb;
        }
    }
    __syncthreads();
    unsigned long long ra = s_a[tid], rb = s_b[tid];
    for (u32 t = lo; t < hi; t++) {
        const unsigned long long va = a[t];
        a[t] = ra;
        ra += va;
        if (b) {
            const unsigned long long vb = b[t];
            b[t] = rb;
            rb += vb;
        }
    }
}

// ---- the tape pass: EMIT = false lengths, EMIT = true text ---------------------------------------------------------
static constexpr u32 MS_WINDOW = 40960;  // bytes of text a tile stages in LDS (3 blocks per CU)
template <bool EMIT>
__global__ __launch_bounds__(TW_THREADS) void k_ms_tile(MsView p) {
    __shared__ long long s_l[TW_THREADS / 64];
    __shared__ unsigned long long s_s[TW_THREADS / 64];
    const int tid = threadIdx.x;
    const u64 base = (u64)blockIdx.x * TW_TILE + (u64)tid * TW_ITEMS;
    u64 w[TW_ITEMS + 2];  // the thread's words and the two behind them (an entry's second word, the next entry's tag)
#pragma unroll
    for (int k = 0; k < TW_ITEMS + 2; k++) w[k] = base + k < p.n ? p.tape[base + k] : 0;
    long long last = -1;
#pragma unroll
    for (int k = 0; k < TW_ITEMS; k++)
        if (base + k < p.n && !two_word_tag(w[k])) last = (long long)(base + k);
    long long anchor = block_excl_max(last, s_l, tid);
    const long long carry = p.tile_last[blockIdx.x];
    anchor = anchor > carry ? anchor : carry;

    u32 len[TW_ITEMS];
    u8 isent[TW_ITEMS];
    u64 bytes = 0;
    u32 nstr = 0;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < TW_ITEMS; k++) {
        const u64 i = base + k;
        len[k] = 0;
        isent[k] = 0;
        if (i >= p.n) continue;
        const bool raw = anchor >= 0 && ((((long long)i - anchor - 1) & 1) != 0);
        if (!two_word_tag(w[k])) anchor = (long long)i;
        if (raw) continue;
        isent[k] = 1;
        const u32 t = (u32)(w[k] >> 56);
        const bool two = two_word_tag(w[k]);
        // separator behind a completed value: ',' unless the next entry closes something (for a key: ':', same length)
        const u64 nw = two ? w[k + 2] : w[k + 1];
        const u32 nt = (u32)(nw >> 56);
        const bool last_entry = i + (two ? 2 : 1) >= p.n;
        const u32 sep = (!last_entry && nt != '}' && nt != ']' && nt != 'r') ? 1u : 0u;
        u32 l = 0;
        if (t == '"') {
            u32 el;
            if (EMIT) {
                el = p.slen[i];  // measured by the counting pass: no second walk over the string
            } else {
                el = (u32)escaped_length(entry_string(p, w[k]), w[k + 1]);
                p.slen[i] = el;
            }
            l = 2 + el + sep;
            nstr++;
        } else if (t == 'l' || t == 'u' || t == 'd') {
            u8 tmp[32];
            u32 nl = t == 'd' ? format_float(w[k + 1], tmp) : (t == 'l' ? format_int(w[k + 1], tmp) : format_uint(w[k + 1], tmp));
            if (nl == 0) bad = true;  // Inf / NaN: "INF or NaN number found"
            l = nl + sep;
        } else if (t == 't' || t == 'n') {
            l = 4 + sep;
        } else if (t == 'f') {
            l = 5 + sep;
        } else if (t == '{' || t == '[') {
            l = 1;
        } else if (t == '}' || t == ']') {
            l = 1 + sep;
        } else if (t == 'r') {
            const bool is_open = (w[k] & TW_PAYLOAD) > i;  // isOpenRoot (:441)
            l = (!is_open && i + 1 < p.n) ? 1u : 0u;       // '\n' between records
        } else {
            bad = true;
        }
        len[k] = l;
        bytes += l;
    }
    unsigned long long tot = 0;
    // text bytes of a tile: < 2^44 even for degenerate strings; string entries: <= 2048
    const unsigned long long packed = (bytes << 16) | nstr;
    const unsigned long long ex = block_excl_sum(packed, s_s, tid, &tot);
    if (!EMIT) {
        if (tid == 0) {
            p.cnt_b[blockIdx.x] = tot >> 16;
            p.cnt_s[blockIdx.x] = tot & 0xffffu;
        }
        if (bad) atomicOr(&p.totals[2], 1ull);
        return;
    }
    // The text of a tile is one contiguous range.  When it fits the window the threads write it into LDS (byte stores
    // that cost a fraction of scattered global ones) and the block copies the window out with coalesced 4-byte
    // stores; a tile with more text (long strings) writes straight to memory.
    __shared__ __attribute__((aligned(16))) u8 s_text[EMIT ? MS_WINDOW : 16];
    const u64 tile_bytes = tot >> 16;
    const bool staged = tile_bytes <= MS_WINDOW;  // block-uniform
    u8 *const gdst = p.text + p.cnt_b[blockIdx.x];
    u8 *o = (staged ? s_text : gdst) + (ex >> 16);
    u64 si = p.cnt_s[blockIdx.x] + (ex & 0xffffu);  // ordinal of the thread's first string entry
#pragma unroll
    for (int k = 0; k < TW_ITEMS; k++) {
        if (!isent[k]) continue;
        const u32 t = (u32)(w[k] >> 56);
        u8 *const end = o + len[k];
        bool sep_is_colon = false;
        if (t == '"') {
            *o++ = '"';
            o = write_escaped(o, entry_string(p, w[k]), w[k + 1]);
            *o++ = '"';
            sep_is_colon = p.keyflag[si++] != 0;
        } else if (t == 'l') {
            o += format_int(w[k + 1], o);
        } else if (t == 'u') {
            o += format_uint(w[k + 1], o);
        } else if (t == 'd') {
            u8 tmp[32];  // format_float writes up to 32 bytes: not straight into the neighbours' text
            const u32 nl = format_float(w[k + 1], tmp);
            for (u32 j = 0; j < nl; j++) o[j] = tmp[j];
            o += nl;
        } else if (t == 't') {
            o[0] = 't'; o[1] = 'r'; o[2] = 'u'; o[3] = 'e';
            o += 4;
        } else if (t == 'n') {
            o[0] = 'n'; o[1] = 'u'; o[2] = 'l'; o[3] = 'l';
            o += 4;
        } else if (t == 'f') {
            o[0] = 'f'; o[1] = 'a'; o[2] = 'l'; o[3] = 's'; o[4] = 'e';
            o += 5;
        } else if (t == 'r') {
            if (len[k]) *o++ = '\n';
        } else {
            *o++ = (u8)t;  // { [ } ]
        }
        if (o < end) *o++ = sep_is_colon ? ':' : ',';
    }
    if (staged) {
        __syncthreads();
        const u32 nb = (u32)tile_bytes, nw = nb >> 2;
        for (u32 i = (u32)tid; i < nw; i += TW_THREADS)  // unaligned 4-byte global stores are fine on gfx950
            *reinterpret_cast<u32 *>(gdst + 4 * i) = *reinterpret_cast<const u32 *>(s_text + 4 * i);
        const u32 tail = nw * 4 + (u32)tid;
        if (tail < nb) gdst[tail] = s_text[tail];
    }
}

}  // namespace

int sjhip_marshal_json(sjhip_ctx *ctx, size_t *text_len) {
    if (!ctx) return SJHIP_ERR_ARG;
    ctx->ms_len = 0;
    ctx->ms_valid = 0;
    if (!ctx->q_valid || ctx->tape_len == 0) {
        ctx_set_error(ctx, "no parse result on the device (sjhip_marshal_json follows a successful sjhip_parse / sjhip_parse_device)");
        return SJHIP_ERR_ARG;
    }
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    ctx->ser_valid = 0;  // shares d_q with the serializer and the filter
    ctx->q_tape_len = ctx->q_strings_len = 0;
    ctx->f_valid = 0;
    MsView p;
    p.tape = (const u64 *)ctx->d_tape.p;
    p.n = ctx->tape_len;
    p.tiles = (u32)((p.n + TW_TILE - 1) / TW_TILE);
    p.strings = (const u8 *)ctx->d_strings.p;
    p.msg = (const u8 *)ctx->p_msg;
    KeyView kv;
    kv.kind = ctx->p_kind;
    kv.n = (u32)ctx->p_n;
    kv.tiles = (kv.n + 4095u) / 4096u;
    const size_t per = ((size_t)p.tiles * 8 + 255) / 256 * 256, perk = ((size_t)kv.tiles * 8 + 255) / 256 * 256;
    const size_t flags = ((size_t)kv.n + 255) / 256 * 256;
    int rc = arena_reserve(ctx, ctx->d_q, 256 + per * 3 + perk + flags + (size_t)p.n * 4 + 256);
    if (rc) return rc;
    char *w = (char *)ctx->d_q.p;
    p.totals = (unsigned long long *)w;
    w += 256;
    p.tile_last = (long long *)w;
    w += per;
    p.cnt_b = (unsigned long long *)w;
    w += per;
    p.cnt_s = (unsigned long long *)w;
    w += per;
    kv.cnt = (unsigned long long *)w;
    w += perk;
    kv.keyflag = (u8 *)w;
    w += flags;
    p.keyflag = kv.keyflag;
    p.slen = (u32 *)w;
    p.text = nullptr;
    HIPCHK(hipMemsetAsync(p.totals, 0, 256, ctx->stream), "marshal memset");
    // keys from the token array of the parse
    hipLaunchKernelGGL(k_ms_keys<false>, dim3(kv.tiles), dim3(256), 0, ctx->stream, kv);
    hipLaunchKernelGGL(k_ms_scan, dim3(1), dim3(1024), 0, ctx->stream, kv.cnt, (unsigned long long *)nullptr, kv.tiles,
                       (unsigned long long *)nullptr);
    hipLaunchKernelGGL(k_ms_keys<true>, dim3(kv.tiles), dim3(256), 0, ctx->stream, kv);
    // tag / raw classification, lengths, positions
    hipLaunchKernelGGL(k_tw_last, dim3(p.tiles), dim3(TW_THREADS), 0, ctx->stream, p.tape, p.n, p.tile_last);
    hipLaunchKernelGGL(k_tw_scan_last, dim3(1), dim3(1024), 0, ctx->stream, p.tile_last, p.tiles);
    hipLaunchKernelGGL(k_ms_tile<false>, dim3(p.tiles), dim3(TW_THREADS), 0, ctx->stream, p);
    hipLaunchKernelGGL(k_ms_scan, dim3(1), dim3(1024), 0, ctx->stream, p.cnt_b, p.cnt_s, p.tiles, p.totals);
    HIPCHK(hipGetLastError(), "marshal launch");
    unsigned long long *h = (unsigned long long *)(ctx->h_scratch + 512);
    HIPCHK(hipMemcpyAsync(h, p.totals, 24, hipMemcpyDeviceToHost, ctx->stream), "D2H totals");
    HIPCHK(hipStreamSynchronize(ctx->stream), "marshal sync");
    if (h[2]) {
        ctx_set_error(ctx, "INF or NaN number found");  // the reference's error (parsed_json.go:1252)
        return SJHIP_ERR_ARG;
    }
    rc = arena_reserve(ctx, ctx->d_qtape, (size_t)h[0] + 64);
    if (rc) return rc;
    p.text = (u8 *)ctx->d_qtape.p;
    hipLaunchKernelGGL(k_ms_tile<true>, dim3(p.tiles), dim3(TW_THREADS), 0, ctx->stream, p);
    HIPCHK(hipGetLastError(), "marshal emit launch");
    ctx->ms_len = (size_t)h[0];
    ctx->ms_valid = 1;
    if (text_len) *text_len = ctx->ms_len;
    return SJHIP_OK;
}

int sjhip_fetch_marshaled(sjhip_ctx *ctx, uint8_t *dst) {
    if (!ctx || !ctx->ms_valid) return SJHIP_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    if (ctx->ms_len && dst)
        HIPCHK(hipMemcpyAsync(dst, ctx->d_qtape.p, ctx->ms_len, hipMemcpyDeviceToHost, ctx->stream), "D2H JSON text");
    HIPCHK(hipStreamSynchronize(ctx->stream), "fetch sync");
    return SJHIP_OK;
}
