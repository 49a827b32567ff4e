// serialize.hip -- Serializer.Serialize (parsed_serialize.go:200-431, format version 3) on the device-resident tape
// (SURVEY.md section 8f, N3).
//
// The reference walks the tape once and splits it into three columns -- one tag byte per tape ENTRY, the values of the
// entries that have one (8 or 16 bytes), and the bytes of all strings -- then compresses each column (S2 / zstd) and
// frames them.  The split is a scatter driven by two prefix sums over the tape, which is what a GPU does well; the
// compressors are byte-serial CPU libraries and stay on the host: this file produces the CompressNone stream (block
// type 0), i.e. exactly the input the host compressors would be handed.
//   tags    one byte per entry: the tag; 'e' for a float whose tag word carries a flag (tagFloatWithFlag, :313-320)
//   values  "  : (offset into the string column, length)      l u d : the 64-bit value      e : tag word, value
//           { [ r : payload - own index (closing tags are rebuilt from their opening tag, :324-331)
//   strings the reference appends every string it has not seen at the same hash slot (indexString, :836-857 -- keyed by
//           Go's per-process random memhash, so its own output is not reproducible); here the column IS Strings.B: the
//           parser already laid all strings out in tape order, offsets are the tape's offsets, nothing is copied.
//           (= the reference's algorithm with the de-duplication table never hitting; Deserialize cannot tell.)
// Which tape words are entries and which are the second word of an entry is decided with the parity rule of query.hip
// (a value can look like any tag), here document-wide: the last "anchor" index in front of every 2048-word tile comes
// from a max-scan over the tiles.
#include <hip/hip_runtime.h>
#include <string.h>

#include "../../include/sjhip.h"
#include "sj_ctx.h"
#include "sj_device.h"
#include "sj_stage2.h"
#include "sj_tapewalk.h"

using namespace sj;

#define HIPCHK(call, what)                                        \
    do {                                                          \
        hipError_t e_ = (call);                                   \
        if (e_ != hipSuccess) return ctx_hip_fail(ctx, e_, what); \
    } while (0)

namespace {

static constexpr u64 PAYLOAD = TW_PAYLOAD;
static constexpr int ST_THREADS = TW_THREADS, ST_ITEMS = TW_ITEMS, ST_TILE = TW_TILE;

struct SerView {
    const u64 *tape;
    u64 n;
    u32 tiles;
    long long *tile_last;        // [tiles] last anchor (word whose top byte is not one of " l u d) of the tile, -1: none
    unsigned long long *cnt_t;   // [tiles] entries of the tile            -> exclusive prefix
    unsigned long long *cnt_v;   // [tiles] value bytes of the tile        -> exclusive prefix
    unsigned long long *totals;  // entries, value bytes
    u8 *tags;
    u8 *vals;
};

// one block: exclusive prefix sums of the two per-tile counts + totals
__global__ __launch_bounds__(1024) void k_ser_scan_cnt(SerView p) {
    __shared__ unsigned long long s_a[1024], s_b[1024];
    const u32 tid = threadIdx.x, per = (p.tiles + 1023u) / 1024u;
    const u32 lo = tid * per < p.tiles ? tid * per : p.tiles, hi = lo + per < p.tiles ? lo + per : p.tiles;
    unsigned long long a = 0, b = 0;
    for (u32 t = lo; t < hi; t++) {
        a += p.cnt_t[t];
        b += p.cnt_v[t];
    }
    s_a[tid] = a;
    s_b[tid] = b;
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
        p.totals[0] = ra;
        p.totals[1] = rb;
    }
    __syncthreads();
    unsigned long long ra = s_a[tid], rb = s_b[tid];
    for (u32 t = lo; t < hi; t++) {
        const unsigned long long va = p.cnt_t[t], vb = p.cnt_v[t];
        p.cnt_t[t] = ra;
        p.cnt_v[t] = rb;
        ra += va;
        rb += vb;
    }
}

// EMIT = false: per-tile counts; EMIT = true: the tag and value columns
template <bool EMIT>
__global__ __launch_bounds__(ST_THREADS) void k_ser_tile(SerView p) {
    __shared__ long long s_l[ST_THREADS / 64];
    __shared__ unsigned long long s_s[ST_THREADS / 64];
    const int tid = threadIdx.x;
    const u64 base = (u64)blockIdx.x * ST_TILE + (u64)tid * ST_ITEMS;
    u64 w[ST_ITEMS + 1];
#pragma unroll
    for (int k = 0; k <= ST_ITEMS; k++) w[k] = base + k < p.n ? p.tape[base + k] : 0;
    long long last = -1;
#pragma unroll
    for (int k = 0; k < ST_ITEMS; k++)
        if (base + k < p.n && !two_word_tag(w[k])) last = (long long)(base + k);
    long long anchor = block_excl_max(last, s_l, tid);  // last anchor in front of this thread's words, inside the tile
    const long long carry = p.tile_last[blockIdx.x];
    anchor = anchor > carry ? anchor : carry;
    // entries of this thread: tag byte and value bytes
    u8 tg[ST_ITEMS];
    u8 nb[ST_ITEMS];
    u32 ntag = 0, nval = 0;
#pragma unroll
    for (int k = 0; k < ST_ITEMS; k++) {
        const u64 i = base + k;
        tg[k] = 0xff;  // not an entry
        nb[k] = 0;
        if (i >= p.n) continue;
        const bool raw = anchor >= 0 && ((((long long)i - anchor - 1) & 1) != 0);
        if (!two_word_tag(w[k])) anchor = (long long)i;
        if (raw) continue;
        const u32 t = (u32)(w[k] >> 56);
        u8 out = (u8)t, bytes = 0;
        if (t == '"') bytes = 16;
        else if (t == 'l' || t == 'u') bytes = 8;
        else if (t == 'd') {
            if ((w[k] & PAYLOAD) == 0) bytes = 8;
            else {
                out = 'e';
                bytes = 16;
            }
        } else if (t == '{' || t == '[' || t == 'r') bytes = 8;
        tg[k] = out;
        nb[k] = bytes;
        ntag++;
        nval += bytes;
    }
    unsigned long long tot = 0;
    const unsigned long long packed = ((unsigned long long)nval << 20) | ntag;  // <= 8 entries, <= 128 bytes per thread
    const unsigned long long ex = block_excl_sum(packed, s_s, tid, &tot);
    if (!EMIT) {
        if (tid == 0) {
            p.cnt_t[blockIdx.x] = tot & 0xfffffu;
            p.cnt_v[blockIdx.x] = tot >> 20;
        }
        return;
    }
    u64 to = p.cnt_t[blockIdx.x] + (ex & 0xfffffu);
    u64 vo = p.cnt_v[blockIdx.x] + (ex >> 20);
#pragma unroll
    for (int k = 0; k < ST_ITEMS; k++) {
        if (tg[k] == 0xff) continue;
        const u64 i = base + k;
        p.tags[to++] = tg[k];
        if (nb[k] == 0) continue;
        u64 *v = reinterpret_cast<u64 *>(p.vals + vo);
        const u32 t = (u32)(w[k] >> 56);
        if (t == '"') {
            v[0] = w[k] & PAYLOAD & ~STRINGBUFBIT;  // offset into the string column (= Strings.B)
            v[1] = w[k + 1];
        } else if (tg[k] == 'e') {
            v[0] = w[k];
            v[1] = w[k + 1];
        } else if (t == 'l' || t == 'u' || t == 'd') {
            v[0] = w[k + 1];
        } else {
            v[0] = (w[k] & PAYLOAD) - i;  // { [ r: distance to the partner (roots: may wrap, :324-328)
        }
        vo += nb[k];
    }
}

// binary.PutUvarint
size_t put_uvarint(uint8_t *dst, uint64_t v) {
    size_t n = 0;
    while (v >= 0x80) {
        dst[n++] = (uint8_t)v | 0x80;
        v >>= 7;
    }
    dst[n++] = (uint8_t)v;
    return n;
}

}  // namespace

int sjhip_serialize(sjhip_ctx *ctx, size_t *tags_len, size_t *values_len, size_t *strings_len, size_t *stream_len) {
    if (!ctx) return SJHIP_ERR_ARG;
    ctx->ser_valid = 0;
    ctx->ms_valid = 0;
    if (!ctx->q_valid || ctx->tape_len == 0) {
        ctx_set_error(ctx, "no parse result on the device (sjhip_serialize follows a successful sjhip_parse / sjhip_parse_device)");
        return SJHIP_ERR_ARG;
    }
    if (!(ctx->p_flags & SJHIP_FLAG_COPY_STRINGS)) {
        ctx_set_error(ctx, "sjhip_serialize needs a parse with SJHIP_FLAG_COPY_STRINGS (Strings.B is the string column)");
        return SJHIP_ERR_ARG;
    }
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    SerView p;
    p.tape = (const u64 *)ctx->d_tape.p;
    p.n = ctx->tape_len;
    p.tiles = (u32)((p.n + ST_TILE - 1) / ST_TILE);
    const size_t per = ((size_t)p.tiles * 8 + 255) / 256 * 256;
    int rc = arena_reserve(ctx, ctx->d_q, per * 3 + 256);
    if (rc) return rc;
    rc = arena_reserve(ctx, ctx->d_qtape, p.n * 8 + 64);  // values: at most 8 bytes per tape word
    if (rc) return rc;
    rc = arena_reserve(ctx, ctx->d_qstrings, p.n + 64);   // tags: at most one per tape word
    if (rc) return rc;
    char *w = (char *)ctx->d_q.p;
    p.totals = (unsigned long long *)w;
    w += 256;
    p.tile_last = (long long *)w;
    w += per;
    p.cnt_t = (unsigned long long *)w;
    w += per;
    p.cnt_v = (unsigned long long *)w;
    p.vals = (u8 *)ctx->d_qtape.p;
    p.tags = (u8 *)ctx->d_qstrings.p;
    hipLaunchKernelGGL(k_tw_last, dim3(p.tiles), dim3(ST_THREADS), 0, ctx->stream, p.tape, p.n, p.tile_last);
    hipLaunchKernelGGL(k_tw_scan_last, dim3(1), dim3(1024), 0, ctx->stream, p.tile_last, p.tiles);
    hipLaunchKernelGGL(k_ser_tile<false>, dim3(p.tiles), dim3(ST_THREADS), 0, ctx->stream, p);
    hipLaunchKernelGGL(k_ser_scan_cnt, dim3(1), dim3(1024), 0, ctx->stream, p);
    hipLaunchKernelGGL(k_ser_tile<true>, dim3(p.tiles), dim3(ST_THREADS), 0, ctx->stream, p);
    HIPCHK(hipGetLastError(), "serialize launch");
    unsigned long long *h = (unsigned long long *)(ctx->h_scratch + 512);
    HIPCHK(hipMemcpyAsync(h, p.totals, 16, hipMemcpyDeviceToHost, ctx->stream), "D2H totals");
    HIPCHK(hipStreamSynchronize(ctx->stream), "serialize sync");
    ctx->ser_tags = (size_t)h[0];
    ctx->ser_vals = (size_t)h[1];
    ctx->ser_valid = 1;
    ctx->q_tape_len = ctx->q_strings_len = 0;  // the filter result shared these arenas
    ctx->f_valid = 0;
    // size of the framed stream (parsed_serialize.go:381-426)
    uint8_t tmp[16];
    const size_t sl = ctx->strings_len;
    size_t rest = put_uvarint(tmp, ctx->tape_len) + 2 + put_uvarint(tmp, sl) + put_uvarint(tmp, sl + 1) + 1 + sl +
                  put_uvarint(tmp, ctx->ser_tags) + put_uvarint(tmp, ctx->ser_tags + 1) + 1 + ctx->ser_tags +
                  put_uvarint(tmp, ctx->ser_vals) + put_uvarint(tmp, ctx->ser_vals + 1) + 1 + ctx->ser_vals;
    ctx->ser_rest = rest;
    ctx->ser_stream = 1 + put_uvarint(tmp, rest) + rest;
    if (tags_len) *tags_len = ctx->ser_tags;
    if (values_len) *values_len = ctx->ser_vals;
    if (strings_len) *strings_len = sl;
    if (stream_len) *stream_len = ctx->ser_stream;
    return SJHIP_OK;
}

int sjhip_fetch_serialized(sjhip_ctx *ctx, uint8_t *dst, size_t cap, size_t *len) {
    if (!ctx || !dst) return SJHIP_ERR_ARG;
    if (!ctx->ser_valid || cap < ctx->ser_stream) {
        ctx_set_error(ctx, "sjhip_fetch_serialized: no serialized result, or destination smaller than %zu bytes", ctx->ser_stream);
        return SJHIP_ERR_ARG;
    }
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    const size_t sl = ctx->strings_len, tl = ctx->ser_tags, vl = ctx->ser_vals;
    size_t o = 0;
    dst[o++] = 3;  // serializedVersion
    const size_t rest = ctx->ser_rest;  // the size field covers everything behind it
    o += put_uvarint(dst + o, rest);
    o += put_uvarint(dst + o, ctx->tape_len);
    dst[o++] = 0;  // Strings: uncompressed size 0
    dst[o++] = 0;  // Strings: empty block
    o += put_uvarint(dst + o, sl);      // Message (the string column): uncompressed size
    o += put_uvarint(dst + o, sl + 1);  // block size = type byte + data
    dst[o++] = 0;                       // blockTypeUncompressed
    if (sl) HIPCHK(hipMemcpyAsync(dst + o, ctx->d_strings.p, sl, hipMemcpyDeviceToHost, ctx->stream), "D2H string column");
    o += sl;
    o += put_uvarint(dst + o, tl);
    o += put_uvarint(dst + o, tl + 1);
    dst[o++] = 0;
    if (tl) HIPCHK(hipMemcpyAsync(dst + o, ctx->d_qstrings.p, tl, hipMemcpyDeviceToHost, ctx->stream), "D2H tag column");
    o += tl;
    o += put_uvarint(dst + o, vl);
    o += put_uvarint(dst + o, vl + 1);
    dst[o++] = 0;
    if (vl) HIPCHK(hipMemcpyAsync(dst + o, ctx->d_qtape.p, vl, hipMemcpyDeviceToHost, ctx->stream), "D2H value column");
    o += vl;
    HIPCHK(hipStreamSynchronize(ctx->stream), "fetch sync");
    if (len) *len = o;
    if (o != ctx->ser_stream) {
        ctx_set_error(ctx, "serialized stream: %zu bytes written, %zu announced", o, ctx->ser_stream);
        return SJHIP_ERR_HIP;
    }
    return SJHIP_OK;
}
