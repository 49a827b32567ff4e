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
//           Go's per-process random memhash, so its own output is not reproducible).  Two forms here:
//           * plain (flags 0): the column IS Strings.B -- the parser already laid all strings out in tape order, offsets
//             are the tape's offsets, nothing is copied (= the reference's algorithm with the table never hitting;
//             Deserialize cannot tell; byte-identical to the oracle's stream);
//           * SJHIP_SER_DEDUP: a string is dropped when it equals the FIRST string of the document that hashes to its
//             slot (a table of 2^20 slots filled with an atomic minimum over the tape index: deterministic, unlike the
//             reference's replace-on-miss order); the kept strings form the column in tape order, every entry stores
//             the column offset of its representative.  Like the reference's, the stream is valid for any table policy.
// Which tape words are entries and which are the second word of an entry is decided with the parity rule of query.hip
// (a value can look like any tag), here document-wide: the last "anchor" index in front of every 2048-word tile comes
// from a max-scan over the tiles.
#include <hip/hip_runtime.h>
#include <string.h>

#include "../../include/sjhip.h"
#include "sj_bounds.h"
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
    // de-duplication of the string column (null table: the column is Strings.B itself)
    u32 *table;                  // [SER_SLOTS] smallest tape index of a string entry that hashes to the slot
    u32 *slot_off;               // [SER_SLOTS] column offset of that string
    unsigned long long *cnt_s;   // [tiles] kept string bytes of the tile -> exclusive prefix
    const u8 *strings;           // Strings.B
    u64 strings_len;             // (the debug build checks the offset and length a tape word names against it, ser_string)
    u8 *scol;                    // the string column
};
static constexpr u32 SER_SLOT_BITS = 20, SER_SLOTS = 1u << SER_SLOT_BITS;

// the bytes of a string entry of Strings.B: offset and length come out of tape words -- checked in the debug build
__device__ __forceinline__ const u8 *ser_string(const SerView &p, u64 off, u64 len) {
#if defined(SJ_DEBUG_BOUNDS)
    if (off > p.strings_len || len > p.strings_len - off) {
        bounds_report(A_STRINGS, off + len, p.strings_len);
        off = 0;
    }
#else
    (void)len;
#endif
    return p.strings + off;
}
// hash of a string for the de-duplication table (any deterministic function: equal bytes are compared afterwards)
__device__ __forceinline__ u32 ser_hash(const u8 *p, u64 len) {
    u64 h = 0x9e3779b97f4a7c15ull ^ len;
    u64 i = 0;
    for (; i + 8 <= len; i += 8) h = (h ^ *reinterpret_cast<const u64 *>(p + i)) * 0xff51afd7ed558ccdull + (h >> 29);
    u64 t = 0;
    for (u64 k = 0; i + k < len; k++) t |= (u64)p[i + k] << (8 * k);
    h = (h ^ t) * 0xc4ceb9fe1a85ec53ull;
    return (u32)((h ^ (h >> 31)) >> (64 - SER_SLOT_BITS));
}
__device__ __forceinline__ bool ser_equal(const u8 *a, const u8 *b, u64 len) {
    u64 i = 0;
    for (; i + 8 <= len; i += 8)
        if (*reinterpret_cast<const u64 *>(a + i) != *reinterpret_cast<const u64 *>(b + i)) return false;
    for (; i < len; i++)
        if (a[i] != b[i]) return false;
    return true;
}

// one block: exclusive prefix sums of the per-tile counts + totals (a: entries, b: value bytes, c: kept string bytes)
__global__ __launch_bounds__(1024) void k_ser_scan_cnt(unsigned long long *cnt_a, unsigned long long *cnt_b,
                                                       unsigned long long *cnt_c, u32 tiles, unsigned long long *totals) {
    __shared__ long long s_w[16];
#pragma unroll 1
    for (int k = 0; k < 3; k++) {  // (one copy of the scan: three inlined ones spilled 84 bytes per lane under the 1024-thread bound)
        unsigned long long *const cnt = k == 0 ? cnt_a : (k == 1 ? cnt_b : cnt_c);
        const long long t = cnt ? block1024_scan_array<false>((long long *)cnt, tiles, s_w, (int)threadIdx.x) : 0;
        if (threadIdx.x == 0) totals[k] = (unsigned long long)t;
        __syncthreads();
    }
}

// The passes over the tape.  MODE 0: (de-duplication) every string entry enters its tape index into its table slot with
// an atomic minimum; 1: per-tile counts (entries, value bytes, kept string bytes); 2: (de-duplication) the kept strings
// are copied to the column and the first string of every slot publishes its column offset; 3: the tag and value columns.
// A string entry is kept unless it equals the first string of its slot (empty strings are never kept: offset 0).
template <int MODE>
__global__ __launch_bounds__(ST_THREADS) void k_ser_tile(SerView p) {
    constexpr bool EMIT = MODE == 3;
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
    __shared__ long long s_carry;
    const long long carry = p.tile_last ? p.tile_last[blockIdx.x] : tw_local_anchor(p.tape, (u64)blockIdx.x * ST_TILE, tid, &s_carry);
    if (carry == -2) {  // (block-uniform, sj_tapewalk.h) nothing of this tile can be classified: report and leave
        if (tid == 0) {
            atomicOr(&p.totals[3], 1ull);
            if (MODE == 1) {
                p.cnt_t[blockIdx.x] = p.cnt_v[blockIdx.x] = 0;
                if (p.table) p.cnt_s[blockIdx.x] = 0;
            }
        }
        return;
    }
    anchor = anchor > carry ? anchor : carry;
    // entries of this thread: tag byte and value bytes
    u8 tg[ST_ITEMS];
    u8 nb[ST_ITEMS];
    u32 ntag = 0, nval = 0;
    const bool dedup = p.table != nullptr;
    bool keep[ST_ITEMS];           // string entries that go to the column
    u32 slot[ST_ITEMS];
    unsigned long long nstr = 0;   // kept string bytes of this thread
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
        keep[k] = false;
        slot[k] = 0;
        if (dedup && t == '"') {
            const u64 so = w[k] & PAYLOAD & ~STRINGBUFBIT, sl = w[k + 1];
            if (sl != 0) {
                slot[k] = ser_hash(ser_string(p, so, sl), sl);
                if (MODE == 0) {
                    atomicMin(&p.table[slot[k]], (u32)i);
                } else {
                    const u32 first = p.table[slot[k]];
                    if (first == (u32)i) {
                        keep[k] = true;
                    } else {  // the first string of the slot lies in front of this one
                        const u64 fo = p.tape[first] & PAYLOAD & ~STRINGBUFBIT, fl = p.tape[first + 1];
                        keep[k] = !(fl == sl && ser_equal(ser_string(p, fo, fl), ser_string(p, so, sl), sl));
                    }
                    if (keep[k]) nstr += sl;
                }
            }
        }
    }
    if (MODE == 0) return;
    unsigned long long tot = 0;
    const unsigned long long packed = ((unsigned long long)nval << 20) | ntag;  // <= 8 entries, <= 128 bytes per thread
    const unsigned long long ex = block_excl_sum(packed, s_s, tid, &tot);
    unsigned long long stot = 0, sex = 0;
    if (dedup) sex = block_excl_sum(nstr, s_s, tid, &stot);
    if (MODE == 1) {
        if (tid == 0) {
            p.cnt_t[blockIdx.x] = tot & 0xfffffu;
            p.cnt_v[blockIdx.x] = tot >> 20;
            if (dedup) p.cnt_s[blockIdx.x] = stot;
        }
        return;
    }
    u64 co = dedup ? p.cnt_s[blockIdx.x] + sex : 0;  // column offset of this thread's first kept string
    if (MODE == 2) {
#pragma unroll
        for (int k = 0; k < ST_ITEMS; k++) {
            if (!keep[k]) continue;
            const u64 so = w[k] & PAYLOAD & ~STRINGBUFBIT, sl = w[k + 1];
            {
                const u8 *src = ser_string(p, so, sl);
                for (u64 b = 0; b < sl; b++) p.scol[co + b] = src[b];
            }
            if (p.table[slot[k]] == (u32)(base + k)) p.slot_off[slot[k]] = (u32)co;
            co += sl;
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
            u64 so = w[k] & PAYLOAD & ~STRINGBUFBIT;  // offset into the string column (= Strings.B without de-duplication)
            if (dedup) {
                so = keep[k] ? co : (w[k + 1] ? (u64)p.slot_off[slot[k]] : 0ull);
                if (keep[k]) co += w[k + 1];
            }
            v[0] = so;
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

int sjhip_serialize_ex(sjhip_ctx *ctx, uint32_t flags, size_t *tags_len, size_t *values_len, size_t *strings_len, size_t *stream_len) {
    if (!ctx) return SJHIP_ERR_ARG;
    ctx->ser_valid = 0;
    ctx->ms_valid = 0;
    if (!ctx->q_valid || ctx->tape_len == 0) {
        if (ctx->big_valid) ctx_set_error(ctx, "sjhip_serialize works on the result of one context; this ND result was parsed shard by shard");
        else ctx_set_error(ctx, "no parse result on the device (sjhip_serialize follows a successful sjhip_parse / sjhip_parse_device)");
        return SJHIP_ERR_ARG;
    }
    if (!(ctx->p_flags & SJHIP_FLAG_COPY_STRINGS)) {
        ctx_set_error(ctx, "sjhip_serialize needs a parse with SJHIP_FLAG_COPY_STRINGS (Strings.B is the string column)");
        return SJHIP_ERR_ARG;
    }
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    const bool dedup = (flags & SJHIP_SER_DEDUP) != 0;
    SerView p;
    p.tape = (const u64 *)ctx->d_tape.p;
    p.n = ctx->tape_len;
    p.tiles = (u32)((p.n + ST_TILE - 1) / ST_TILE);
    const size_t per = ((size_t)p.tiles * 8 + 255) / 256 * 256;
    int rc = arena_reserve(ctx, ctx->d_q, per * 4 + 256);
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
    w += per;
    p.cnt_s = (unsigned long long *)w;
    p.vals = (u8 *)ctx->d_qtape.p;
    p.tags = (u8 *)ctx->d_qstrings.p;
    p.table = p.slot_off = nullptr;
    p.strings = (const u8 *)ctx->d_strings.p;
    p.strings_len = ctx->strings_len;
    p.scol = nullptr;
    if (dedup) {
        rc = arena_reserve(ctx, ctx->d_stab, (size_t)SER_SLOTS * 8);
        if (rc) return rc;
        rc = arena_reserve(ctx, ctx->d_scol, ctx->strings_len + 64);
        if (rc) return rc;
        p.table = (u32 *)ctx->d_stab.p;
        p.slot_off = p.table + SER_SLOTS;
        p.scol = (u8 *)ctx->d_scol.p;
        HIPCHK(hipMemsetAsync(p.table, 0xff, (size_t)SER_SLOTS * 4, ctx->stream), "table reset");
    }
    // The tiles find the anchor of the tag / raw classification among the 64 words in front of them; a tile that cannot
    // (sj_tapewalk.h) raises totals[3] and the walk is repeated with the anchors of the global pass.
    long long *const tile_last = p.tile_last;
    unsigned long long *h = (unsigned long long *)(ctx->h_scratch + 512);
    for (int attempt = 0; attempt < 2; attempt++) {
        p.tile_last = attempt == 0 ? nullptr : tile_last;
        HIPCHK(hipMemsetAsync(p.totals, 0, 256, ctx->stream), "serialize memset");
        if (attempt == 1) {
            hipLaunchKernelGGL(k_tw_last, dim3(p.tiles), dim3(ST_THREADS), 0, ctx->stream, p.tape, p.n, p.tile_last);
            hipLaunchKernelGGL(k_tw_scan_last, dim3(1), dim3(1024), 0, ctx->stream, p.tile_last, p.tiles);
            if (dedup) HIPCHK(hipMemsetAsync(p.table, 0xff, (size_t)SER_SLOTS * 4, ctx->stream), "table reset");
        }
        if (dedup) hipLaunchKernelGGL(k_ser_tile<0>, dim3(p.tiles), dim3(ST_THREADS), 0, ctx->stream, p);
        hipLaunchKernelGGL(k_ser_tile<1>, dim3(p.tiles), dim3(ST_THREADS), 0, ctx->stream, p);
        hipLaunchKernelGGL(k_ser_scan_cnt, dim3(1), dim3(1024), 0, ctx->stream, p.cnt_t, p.cnt_v, dedup ? p.cnt_s : nullptr, p.tiles, p.totals);
        if (dedup) hipLaunchKernelGGL(k_ser_tile<2>, dim3(p.tiles), dim3(ST_THREADS), 0, ctx->stream, p);
        hipLaunchKernelGGL(k_ser_tile<3>, dim3(p.tiles), dim3(ST_THREADS), 0, ctx->stream, p);
        HIPCHK(hipGetLastError(), "serialize launch");
        HIPCHK(hipMemcpyAsync(h, p.totals, 32, hipMemcpyDeviceToHost, ctx->stream), "D2H totals");
        HIPCHK(hipStreamSynchronize(ctx->stream), "serialize sync");
        if (h[3] == 0) break;
    }
#if defined(SJ_DEBUG_BOUNDS)
    {   // debug build: a string entry outside Strings.B fails the call (this translation unit's record, sj_bounds.h)
        BoundsHit hit = {};
        if (hipMemcpyFromSymbol(&hit, HIP_SYMBOL(g_bounds_hit), sizeof hit) == hipSuccess && hit.hits) {
            const BoundsHit zero = {};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bounds_hit), &zero, sizeof zero);
            ctx_set_error(ctx, "bounds check (Serialize): %u out-of-bounds strings, the first at byte %llu of %llu", hit.hits, hit.index, hit.size);
            return SJHIP_ERR_HIP;
        }
    }
#endif
    ctx->ser_tags = (size_t)h[0];
    ctx->ser_vals = (size_t)h[1];
    ctx->ser_dedup = dedup;
    ctx->ser_slen = dedup ? (size_t)h[2] : ctx->strings_len;
    ctx->ser_valid = 1;
    ctx->q_tape_len = ctx->q_strings_len = 0;  // the filter result shared these arenas
    ctx->f_valid = 0;
    // size of the framed stream (parsed_serialize.go:381-426)
    uint8_t tmp[16];
    const size_t sl = ctx->ser_slen;
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

int sjhip_serialize(sjhip_ctx *ctx, size_t *tags_len, size_t *values_len, size_t *strings_len, size_t *stream_len) {
    return sjhip_serialize_ex(ctx, 0, tags_len, values_len, strings_len, stream_len);
}

int sjhip_fetch_serialized(sjhip_ctx *ctx, uint8_t *dst, size_t cap, size_t *len) {
    if (!ctx || !dst) return SJHIP_ERR_ARG;
    if (!ctx->ser_valid || cap < ctx->ser_stream) {
        ctx_set_error(ctx, "sjhip_fetch_serialized: no serialized result, or destination smaller than %zu bytes", ctx->ser_stream);
        return SJHIP_ERR_ARG;
    }
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    const size_t sl = ctx->ser_slen, tl = ctx->ser_tags, vl = ctx->ser_vals;
    const void *scol = ctx->ser_dedup ? ctx->d_scol.p : ctx->d_strings.p;
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
    if (sl) HIPCHK(hipMemcpyAsync(dst + o, scol, sl, hipMemcpyDeviceToHost, ctx->stream), "D2H string column");
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

// ---- Serializer.Deserialize (parsed_serialize.go:466-695) on the device ------------------------------------------------
// The reference rebuilds the tape with one sequential walk over the tag column.  Here every tag knows how many tape
// words (1 or 2) and value bytes (0 / 8 / 16) it stands for, two prefix sums give it its tape offset and its place in
// the value column, and one scatter pass writes the words: a string entry points into the string column (which becomes
// pj.Message, as in the reference: no STRINGBUFBIT), an opening bracket stores index + distance and also writes the
// word of its closing bracket (the stream does not carry values for those).  The columns must be uncompressed blocks
// (type 0, what this library writes; S2 / zstd blocks are decompressed on the host first).  TagNop entries (left by
// the reference's in-place deletions) do not occur in a freshly parsed document and are rejected.
namespace {
static constexpr int DS_THREADS = 256, DS_ITEMS = 16, DS_TILE = DS_THREADS * DS_ITEMS;
struct DesView {
    const u8 *tags;
    const u8 *vals;
    u64 n_tags, n_vals, tape_len;
    u32 tiles;
    unsigned long long *cnt_w, *cnt_v, *totals;  // per tile: tape words, value bytes -> exclusive prefixes; totals[3] = error
    u64 *tape;
};
__device__ __forceinline__ void des_shape(u8 t, u32 &words, u32 &bytes, bool &ok) {
    ok = true;
    switch (t) {
    case '"': words = 2; bytes = 16; break;
    case 'e': words = 2; bytes = 16; break;  // tagFloatWithFlag
    case 'l': case 'u': case 'd': words = 2; bytes = 8; break;
    case '{': case '[': case 'r': words = 1; bytes = 8; break;
    case '}': case ']': case 'n': case 't': case 'f': case 0: words = 1; bytes = 0; break;
    default: words = 0; bytes = 0; ok = false; break;  // incl. TagNop
    }
}
// MODE 0: per-tile counts; 1: the scatter pass; 2: verification of what the scatter pass left (closing brackets)
template <int MODE>
__global__ __launch_bounds__(DS_THREADS) void k_des_tile(DesView p) {
    __shared__ unsigned long long s_s[DS_THREADS / 64];
    const int tid = threadIdx.x;
    const u64 base = (u64)blockIdx.x * DS_TILE + (u64)tid * DS_ITEMS;
    u8 tg[DS_ITEMS];
    u32 nw = 0, nv = 0;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < DS_ITEMS; k++) {
        tg[k] = 0;
        if (base + k >= p.n_tags) continue;
        tg[k] = p.tags[base + k];
        u32 words, bytes;
        bool ok;
        des_shape(tg[k], words, bytes, ok);
        bad |= !ok;
        nw += words;
        nv += bytes;
    }
    unsigned long long tot = 0;
    const unsigned long long ex = block_excl_sum(((unsigned long long)nv << 24) | nw, s_s, tid, &tot);
    if (MODE == 0) {
        if (tid == 0) {
            p.cnt_w[blockIdx.x] = tot & 0xffffffu;
            p.cnt_v[blockIdx.x] = tot >> 24;
        }
        if (bad) atomicOr(&p.totals[3], 1ull);
        return;
    }
    u64 off = p.cnt_w[blockIdx.x] + (ex & 0xffffffu);
    u64 vo = p.cnt_v[blockIdx.x] + (ex >> 24);
#pragma unroll
    for (int k = 0; k < DS_ITEMS; k++) {
        if (base + k >= p.n_tags) continue;
        u32 words, bytes;
        bool ok;
        des_shape(tg[k], words, bytes, ok);
        if (!ok || off + words > p.tape_len || vo + bytes > p.n_vals) {
            bad = true;
            continue;
        }
        const u64 tag = (u64)tg[k] << 56;
        const u64 *v = reinterpret_cast<const u64 *>(p.vals + vo);
        if (MODE == 2) {
            // The reference walks the tags in order and checks, when it reaches a closing bracket, that the word its
            // opener left there carries the same tag (parsed_serialize.go:666-671).  Here the scatter pass ran in
            // parallel, so both ends are checked: a closing tag must sit on a word of its kind whose payload is an
            // opener in front of it that points right behind it -- a slot no opener of THIS stream wrote would still hold
            // a word of the previous parse -- and an opener must still own its closing slot (two openers claiming one
            // slot, or a distance that does not reach past the opener itself, are corrupt streams).
            if (tg[k] == '}' || tg[k] == ']') {
                const u64 w = p.tape[off], j = w & PAYLOAD;
                const u8 open = tg[k] == '}' ? (u8)'{' : (u8)'[';
                if ((w >> 56) != tg[k] || j >= off || p.tape[j] != (((u64)open << 56) | (off + 1))) bad = true;
            } else if (tg[k] == '{' || tg[k] == '[') {
                const u64 val = v[0] + off;
                if (val < off + 2 || val > p.tape_len ||
                    p.tape[val - 1] != (((u64)(tg[k] == '{' ? '}' : ']') << 56) | off))
                    bad = true;
            }
            off += words;
            vo += bytes;
            continue;
        }
        switch (tg[k]) {
        case '"':
            p.tape[off] = tag | v[0];
            p.tape[off + 1] = v[1];
            break;
        case 'e':
            p.tape[off] = v[0];
            p.tape[off + 1] = v[1];
            break;
        case 'l': case 'u': case 'd':
            p.tape[off] = tag;
            p.tape[off + 1] = v[0];
            break;
        case '{': case '[': {
            const u64 val = v[0] + off;  // always forward, and past the opener itself
            if (val > p.tape_len || val < off + 2) {
                bad = true;
                break;
            }
            p.tape[off] = tag | val;
            p.tape[val - 1] = ((u64)(tg[k] == '{' ? '}' : ']') << 56) | off;  // the closing bracket is rebuilt from its opener
            break;
        }
        case 'r':
            if (v[0] + off > p.tape_len) bad = true;
            p.tape[off] = tag | ((v[0] + off) & PAYLOAD);
            break;
        case '}': case ']':
            break;  // written by its opening bracket
        default:
            p.tape[off] = tag;
            break;
        }
        off += words;
        vo += bytes;
    }
    if (bad) atomicOr(&p.totals[3], 1ull);
}

// binary.ReadUvarint
bool get_uvarint(const uint8_t *src, size_t len, size_t *o, uint64_t *v) {
    uint64_t x = 0;
    for (unsigned s = 0; *o < len && s < 64; s += 7) {
        const uint8_t b = src[(*o)++];
        x |= (uint64_t)(b & 0x7f) << s;
        if (!(b & 0x80)) {
            *v = x;
            return true;
        }
    }
    return false;
}
// one column: uncompressed size, block size, block type 0, data
bool get_block(const uint8_t *src, size_t len, size_t *o, uint64_t *size, size_t *data_off) {
    uint64_t usize = 0, bsize = 0;
    if (!get_uvarint(src, len, o, &usize) || !get_uvarint(src, len, o, &bsize)) return false;
    if (bsize == 0) {  // empty block (the reference writes one for an empty column)
        if (usize != 0) return false;
        *size = 0;
        *data_off = *o;
        return true;
    }
    if (*o + bsize > len || src[*o] != 0 || bsize != usize + 1) return false;  // blockTypeUncompressed only
    *size = usize;
    *data_off = *o + 1;
    *o += bsize;
    return true;
}
}  // namespace

int sjhip_deserialize(sjhip_ctx *ctx, const uint8_t *stream, size_t len, size_t *tape_len, size_t *strings_len, size_t *message_len) {
    if (!ctx || !stream) return SJHIP_ERR_ARG;
    ctx->q_valid = ctx->r_valid = ctx->ser_valid = ctx->ms_valid = ctx->f_valid = ctx->kf_valid = 0;
    ctx->pending = 0;
    ctx->pack_valid = 0;
    ctx->tape_len = ctx->strings_len = 0;
    ctx->des_msg_len = 0;
    auto corrupt = [&](const char *what) {
        ctx_set_error(ctx, "sjhip_deserialize: %s", what);
        return SJHIP_ERR_ARG;
    };
    size_t o = 0;
    if (len < 2 || stream[o++] > 3) return corrupt("unknown version");
    uint64_t comp = 0, tl = 0, ss = 0, ms = 0, nt = 0, nv = 0;
    size_t off_s = 0, off_m = 0, off_t = 0, off_v = 0;
    if (!get_uvarint(stream, len, &o, &comp) || comp > len - o) return corrupt("stream too short");
    if (!get_uvarint(stream, len, &o, &tl)) return corrupt("tape size");
    if (!get_block(stream, len, &o, &ss, &off_s)) return corrupt("Strings block (only uncompressed blocks are read on the device)");
    if (!get_block(stream, len, &o, &ms, &off_m)) return corrupt("Message block (only uncompressed blocks are read on the device)");
    if (!get_block(stream, len, &o, &nt, &off_t)) return corrupt("tag block (only uncompressed blocks are read on the device)");
    if (!get_block(stream, len, &o, &nv, &off_v)) return corrupt("value block (only uncompressed blocks are read on the device)");
    if (tl >= 0xfffffff0ull || (nv & 7) != 0) return corrupt("sizes");
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    DesView p;
    p.n_tags = nt;
    p.n_vals = nv;
    p.tape_len = tl;
    p.tiles = (u32)((nt + DS_TILE - 1) / DS_TILE);
    const size_t per = ((size_t)p.tiles * 8 + 255) / 256 * 256;
    int rc = arena_reserve(ctx, ctx->d_q, per * 2 + 256 + nt + 64 + 256);
    if (rc) return rc;
    rc = arena_reserve(ctx, ctx->d_qtape, nv + 64);
    if (rc) return rc;
    rc = arena_reserve(ctx, ctx->d_tape, (tl + 2) * 8);
    if (rc) return rc;
    rc = arena_reserve(ctx, ctx->d_strings, ss + 64);
    if (rc) return rc;
    rc = arena_reserve(ctx, ctx->d_msg, ms + 128);
    if (rc) return rc;
    char *w = (char *)ctx->d_q.p;
    p.totals = (unsigned long long *)w;
    w += 256;
    p.cnt_w = (unsigned long long *)w;
    w += per;
    p.cnt_v = (unsigned long long *)w;
    w += per;
    p.tags = (const u8 *)w;
    p.vals = (const u8 *)ctx->d_qtape.p;
    p.tape = (u64 *)ctx->d_tape.p;
    HIPCHK(hipMemsetAsync(p.totals, 0, 32, ctx->stream), "state reset");
    if (nt) HIPCHK(hipMemcpyAsync((void *)p.tags, stream + off_t, nt, hipMemcpyHostToDevice, ctx->stream), "H2D tags");
    if (nv) HIPCHK(hipMemcpyAsync((void *)p.vals, stream + off_v, nv, hipMemcpyHostToDevice, ctx->stream), "H2D values");
    if (ss) HIPCHK(hipMemcpyAsync(ctx->d_strings.p, stream + off_s, ss, hipMemcpyHostToDevice, ctx->stream), "H2D strings");
    if (ms) HIPCHK(hipMemcpyAsync(ctx->d_msg.p, stream + off_m, ms, hipMemcpyHostToDevice, ctx->stream), "H2D message");
    if (p.tiles) {
        hipLaunchKernelGGL(k_des_tile<0>, dim3(p.tiles), dim3(DS_THREADS), 0, ctx->stream, p);
        hipLaunchKernelGGL(k_ser_scan_cnt, dim3(1), dim3(1024), 0, ctx->stream, p.cnt_w, p.cnt_v, (unsigned long long *)nullptr, p.tiles, p.totals);
        hipLaunchKernelGGL(k_des_tile<1>, dim3(p.tiles), dim3(DS_THREADS), 0, ctx->stream, p);
        hipLaunchKernelGGL(k_des_tile<2>, dim3(p.tiles), dim3(DS_THREADS), 0, ctx->stream, p);
        HIPCHK(hipGetLastError(), "deserialize launch");
    }
    unsigned long long *h = (unsigned long long *)(ctx->h_scratch + 512);
    HIPCHK(hipMemcpyAsync(h, p.totals, 32, hipMemcpyDeviceToHost, ctx->stream), "D2H totals");
    HIPCHK(hipStreamSynchronize(ctx->stream), "deserialize sync");
    if (h[3] != 0) return corrupt("unknown tag, a value beyond the tape, or a closing bracket without its opener (TagNop entries are not read on the device)");
    if (h[0] != tl) return corrupt("tags did not fill tape");
    if (h[1] != nv) return corrupt("values did not fill tape");
    ctx->tape_len = (size_t)tl;
    ctx->strings_len = (size_t)ss;
    ctx->des_msg_len = (size_t)ms;
    if (tape_len) *tape_len = (size_t)tl;
    if (strings_len) *strings_len = (size_t)ss;
    if (message_len) *message_len = (size_t)ms;
    return SJHIP_OK;
}

// pj.Message of the last sjhip_deserialize (the string column of the stream); Tape / Strings.B come through sjhip_fetch
int sjhip_fetch_message(sjhip_ctx *ctx, uint8_t *dst) {
    if (!ctx) return SJHIP_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    if (ctx->des_msg_len && dst)
        HIPCHK(hipMemcpyAsync(dst, ctx->d_msg.p, ctx->des_msg_len, hipMemcpyDeviceToHost, ctx->stream), "D2H message");
    HIPCHK(hipStreamSynchronize(ctx->stream), "fetch sync");
    return SJHIP_OK;
}
