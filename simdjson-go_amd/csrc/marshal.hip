// marshal.hip -- Iter.MarshalJSONBuffer on the device-resident tape (parsed_json.go:401-556):
// tape + Strings.B -> compact JSON text, records separated by '\n'.
//
// The reference walks the tape with a stack and appends to a byte slice.  Here every tape entry computes the length of
// its own text, a prefix sum gives every entry its position, and the text is written -- in one pass over the tape when
// the parser left the key flags (per-tile sizes travel through descriptors, k_ms_tile<2>), else in two (counting pass,
// scan, writing pass):
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
#include <stdlib.h>
#include <string.h>

#include "../../include/sjhip.h"
#include "sj_bounds.h"
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
    u64 strings_len, msg_len;    // (the debug build checks every string a tape word names against them, ms_string)
    long long *tile_last;        // [tiles] sj_tapewalk.h, or null: the tiles look for their anchor themselves
    unsigned long long *cnt_b;   // [tiles] text bytes of the tile           -> exclusive prefix
    unsigned long long *cnt_s;   // [tiles] string entries of the tile       -> exclusive prefix
    unsigned long long *totals;  // text bytes, string entries, error flag
    const u8 *keyflag;           // [string entries] 1: the string is an object key
    const u8 *kf_tape;           // null, or the flags the parser left (SJHIP_FLAG_KEY_FLAGS): [tape index of the entry >> 1]
    u32 *slen;                   // [tiles][1024] escaped length of the tile's k-th string (counting pass -> writing pass)
    const u8 *strings_end, *msg_end;  // ends of the buffers the strings live in (8-byte loads stop there)
    // a shard of a sharded ParseND (an ND message beyond one context's reach) stores its indices in the merged index space: what
    // a root word points at, where a string lies in Strings.B / the message (0 for an unsharded result)
    u64 tape_base, strings_base, msg_base;
    u8 *text;
    // the single-pass form (k_ms_tile<2>): one descriptor per tile (0: nothing yet, MS_DESC_AGG | size, MS_DESC_PREFIX | size of
    // everything up to and including the tile), the ticket counter that numbers the tiles, the capacity of `text`
    unsigned long long *desc;
    u32 *ticket;
    u64 text_cap;
    u32 exp;  // SJ_EXP builds only (SJHIP_MS_EXP): parts of k_ms_tile to leave out (A/B timing; the text is wrong)
};
#if defined(SJ_EXP)
#define MS_EXPBIT(p, b) ((((p).exp >> (b)) & 1u) != 0)
// bit 31: thread 0 of every block adds the time since its previous stamp to totals[8 + k] (phase profile of k_ms_tile)
#define MS_STAMP(k)                                                                              \
    do {                                                                                         \
        if (MS_EXPBIT(p, 31) && threadIdx.x == 0 && (blockIdx.x & 63u) == 0) {                   \
            const unsigned long long t_now = __builtin_readcyclecounter();                       \
            atomicAdd(&p.totals[8 + (k)], t_now - t_prev);                                       \
            t_prev = t_now;                                                                      \
        }                                                                                        \
    } while (0)
#else
#define MS_STAMP(k) do { } while (0)
#define MS_EXPBIT(p, b) false
#endif

struct KeyView {
    const u8 *kind;  // [n] token kinds (stage 1)
    u32 n;
    u32 tiles;       // 4096 tokens each
    unsigned long long *cnt;  // [tiles] string tokens -> exclusive prefix
    u8 *keyflag;
};

// the bytes of a string entry: offset and length come out of tape words -- in the debug build (-DSJ_DEBUG_BOUNDS) they are
// checked against the buffer they point into, a violation is recorded (sj_bounds.h) and the call fails
__device__ __forceinline__ const u8 *ms_string(const MsView &p, bool inbuf, u64 off, u64 len) {
#if defined(SJ_DEBUG_BOUNDS)
    const u64 size = inbuf ? p.strings_len : p.msg_len;
    if (off > size || len > size - off) {
        bounds_report(inbuf ? A_STRINGS : A_MSG, off + len, size);
        off = 0;
    }
#else
    (void)len;
#endif
    return (inbuf ? p.strings : p.msg) + off;
}

// e: queue entry of a string (its ordinal inside the tile in bits 11..20), at: its tape index
__device__ __forceinline__ bool entry_is_key(const MsView &p, u64 key_base, u32 e, u64 at) {
    (void)at;
    return p.kf_tape ? ((e >> 29) & 1u) != 0 : p.keyflag[key_base + ((e >> 11) & 0x3ffu)] != 0;
}

// escapeBytes: bytes below 0x20, '"' and '\\' are escaped (shouldEscape, parsed_json.go:1171-1186)
__device__ __forceinline__ u32 escaped_size(u8 c) {
    if (c == '"' || c == '\\') return 2;
    if (c >= 0x20) return 1;
    return (c == '\b' || c == '\f' || c == '\n' || c == '\r' || c == '\t') ? 2u : 6u;
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
    __shared__ long long s_w[16];
    const long long ta = block1024_scan_array<false>((long long *)a, tiles, s_w, (int)threadIdx.x);
    const long long tb = b ? block1024_scan_array<false>((long long *)b, tiles, s_w, (int)threadIdx.x) : 0;
    if (threadIdx.x == 0 && totals) {
        totals[0] = (unsigned long long)ta;
        totals[1] = (unsigned long long)tb;
    }
}

// ---- descriptors of the single-pass form: status in the two top bits, a byte count below -----------------------------
static constexpr u64 MS_DESC_AGG = 1ull << 62, MS_DESC_PREFIX = 2ull << 62, MS_DESC_VALUE = (1ull << 62) - 1;
__device__ __forceinline__ u64 ms_desc_load(const unsigned long long *d) {
    return __hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ms_desc_store(unsigned long long *d, u64 v) {
    __hip_atomic_store(d, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// One wave: publishes the tile's own size, sums the sizes of the tiles in front of it (64 descriptors per step, nearest
// first, up to the nearest one that already holds a prefix), publishes the tile's prefix and returns the sum.  Tiles are
// numbered by a ticket, so every tile in front is running or finished and publishes its size without waiting for anyone.
// (the tile's own size has been published before: ms_desc_store(&desc[tile], MS_DESC_AGG | own), as early as it is known)
__device__ __forceinline__ u64 ms_lookback(unsigned long long *desc, u32 tile, u64 own, int lane, unsigned long long *flags) {
    u64 sum = 0;
    long long j = (long long)tile - 1;
    u32 spins = 0;
    while (j >= 0) {  // (wave-uniform)
        const long long idx = j - lane;
        const u64 d = idx >= 0 ? ms_desc_load(&desc[idx]) : MS_DESC_PREFIX;  // in front of tile 0: nothing
        const u32 st = (u32)(d >> 62);
        const u64 m_pre = __ballot(st == 2), m_inv = __ballot(st == 0);
        const int first_pre = m_pre ? (int)__builtin_ctzll(m_pre) : 64;
        const int first_inv = m_inv ? (int)__builtin_ctzll(m_inv) : 64;
        if (first_inv < first_pre) {  // a nearer tile has not published yet
            if (++spins > (1u << 22)) {  // (never: a bounded loop instead of a hang; the host reports an internal error)
                if (lane == 0) atomicOr(flags, 16ull);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        u64 v = lane <= first_pre ? (d & MS_DESC_VALUE) : 0ull;
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) v += (u64)__shfl_xor((long long)v, sft, 64);
        sum += v;
        if (first_pre < 64) break;
        j -= 64;
    }
    if (lane == 0) ms_desc_store(&desc[tile], MS_DESC_PREFIX | (sum + own));
    return sum;
}

// ---- the tape pass: EMIT = false lengths, EMIT = true text ---------------------------------------------------------
// A tile is 2048 tape words.  The entries of a tile are SORTED BY KIND into LDS queues and each queue is worked on with
// the lanes packed densely (the first version let every thread walk its own eight words: under divergence a wave ran the
// float formatter, the integer formatter and the string walk for every one of the eight steps, and waited for its
// longest string):
//   1. classify: tag / raw (sj_tapewalk.h), separator, the fixed part of every entry's length -> s_len[word];
//      strings, integers and floats enter their queues (strings with their ordinal inside the tile)
//   2. measure, queue by queue: digits of the integers, shortest digits of the floats, escaped length of the strings
//      (counting pass; the writing pass reads them back: slen[tile][ordinal]) -- strings of MS_LONG bytes and more are
//      taken by a whole wave, 8 bytes per lane and step
//   3. one block scan over the per-thread sums: counting pass -> per-tile totals; writing pass -> the offset of every
//      entry inside the tile's text (s_len turns into offsets), literals and brackets are written right there
//   4. write, queue by queue, into the tile's LDS window (or straight to memory when the tile's text is larger), then
//      the block copies the window out with coalesced stores.
static constexpr u32 MS_WINDOW = 32768;  // bytes of text a tile stages in LDS at most (the launcher picks 21 KiB: 39.4 KB per block
                                         // with the queues, 4 blocks per CU)
static constexpr u32 MS_LONG = 64;       // strings from this length on are measured / written by a whole wave
static constexpr u32 MS_QCAP = TW_TILE / 2;  // a string or a number takes two words

// eight bytes of a string at offset k (k + 8 may run past its end: bytes behind the end read as 'a'), without reading
// past `lim` (the end of the buffer the string lives in)
__device__ __forceinline__ u64 str_load8(const u8 *s, u64 k, u64 len, const u8 *lim) {
    u64 w;
    if (s + k + 8 <= lim) {
        memcpy(&w, s + k, 8);
    } else {
        w = 0;
#pragma unroll 1
        for (u32 j = 0; j < 8 && s + k + j < lim; j++) w |= (u64)s[k + j] << (8 * j);
    }
    const u64 rem = len - k;
    if (rem < 8) {
        const u64 keep = (1ull << (8 * rem)) - 1;
        w = (w & keep) | (0x6161616161616161ull & ~keep);
    }
    return w;
}
// (the same padding for a word that comes from elsewhere: `rem` bytes of it belong to the string)
__device__ __forceinline__ u64 pad_a(u64 w, u64 rem) {
    if (rem >= 8) return w;
    const u64 keep = (1ull << (8 * rem)) - 1;
    return (w & keep) | (0x6161616161616161ull & ~keep);
}
// escaped size of the (up to) eight valid bytes of such a word
__device__ __forceinline__ u32 esc_size8(u64 w, u32 valid) {
    const u64 lo = (w & 0x7f7f7f7f7f7f7f7full);
    const u64 ctl = ~((lo + 0x6060606060606060ull) | w) & 0x8080808080808080ull;              // byte < 0x20
    const u64 q = zero_bytes(w ^ 0x2222222222222222ull) | zero_bytes(w ^ 0x5c5c5c5c5c5c5c5cull);  // '"' '\\'
    if ((ctl | q) == 0) return valid;
    u32 n = 0;
#pragma unroll 1  // (the rare path: one copy of the loop body per call site -- the kernel is larger than the instruction cache as it is)
    for (u32 j = 0; j < valid; j++) n += escaped_size((u8)(w >> (8 * j)));
    return n;
}
__device__ __forceinline__ u8 *write_esc8(u8 *o, u64 w, u32 valid) {
    const u64 lo = (w & 0x7f7f7f7f7f7f7f7full);
    const u64 ctl = ~((lo + 0x6060606060606060ull) | w) & 0x8080808080808080ull;
    const u64 q = zero_bytes(w ^ 0x2222222222222222ull) | zero_bytes(w ^ 0x5c5c5c5c5c5c5c5cull);
    if ((ctl | q) == 0) {
        for (u32 j = 0; j < valid; j++) o[j] = (u8)(w >> (8 * j));  // (the destination has no alignment: byte stores)
        return o + valid;
    }
#pragma unroll 1
    for (u32 j = 0; j < valid; j++) o = write_escaped_byte(o, (u8)(w >> (8 * j)));
    return o;
}

// ---- entry classes (round 6) ----------------------------------------------------------------------------------------------
// One byte per value of a tape word's top byte, in a 256-byte LDS table every thread fills one entry of when the block starts: the
// classification asks the table instead of walking a chain of tag comparisons for every word (eight words per thread, twice, and
// two_word_tag four more comparisons on ten words: the kernel is bound by instruction issue at four waves per SIMD, and ~35 % of a
// tile's time went into those chains, profiles/r06_marshal_parts_ab.txt).  bits 0-2: bytes of text the entry itself brings (without
// separator, string bytes and digits), bit 3: the entry closes something (no separator in front of it), bit 4: a separator may
// follow it, bits 5-7: kind.
static constexpr u32 CLS_CLOSES = 8u, CLS_SEP = 16u, CLS_KIND = 0xe0u, CLS_PLAIN = 0x00u, CLS_STR = 0x20u, CLS_INT = 0x40u, CLS_FLT = 0x60u,
                     CLS_ROOT = 0x80u, CLS_BAD = 0xe0u;
__device__ __forceinline__ u32 tag_class(u32 t) {
    if (t == '"') return CLS_STR | CLS_SEP | 2u;
    if (t == 'l' || t == 'u') return CLS_INT | CLS_SEP;
    if (t == 'd') return CLS_FLT | CLS_SEP;
    if (t == 't' || t == 'n') return CLS_PLAIN | CLS_SEP | 4u;
    if (t == 'f') return CLS_PLAIN | CLS_SEP | 5u;
    if (t == '{' || t == '[') return CLS_PLAIN | 1u;
    if (t == '}' || t == ']') return CLS_PLAIN | CLS_SEP | CLS_CLOSES | 1u;
    if (t == 'r') return CLS_ROOT | CLS_CLOSES;
    return CLS_BAD;
}
__device__ __forceinline__ bool cls_two_word(u32 c) { return ((c >> 5) - 1u) < 3u; }  // string, integer, float

// ---- text into the LDS window eight bytes at a time (round 6) ------------------------------------------------------------
// The window is zeroed when the block starts and a string's text -- quote, bytes, quote, separator -- is shifted together in
// a 64-bit accumulator that is ORed into the window slot by slot with ALIGNED 8-byte LDS atomics (ds_or_b64; the neighbours'
// bytes of a shared slot are zeros in this lane's word): one LDS operation per eight bytes of text instead of one byte store
// -- and a shift, an address and a trip of a data-dependent loop -- per byte (the byte loop was 0.31 of configs[4]'s 1.30 ms,
// profiles/r06_marshal_parts_ab.txt).  Plain byte stores of other entries into the same slots mix freely with the atomics: the LDS
// executes both one operation at a time.  A word with a byte to escape leaves through the byte path (acc_flush, write_esc8,
// acc_init behind it).
struct TextAcc {
    u64 v;                     // bytes not yet in the window, at their place inside the slot
    u32 n;                     // of them (plus the bytes in front of the entry in the first slot): 0..7
    unsigned long long *slot;  // the aligned slot they belong to
};
__device__ __forceinline__ void acc_init(TextAcc &a, u8 *win, u32 off) {
    a.slot = reinterpret_cast<unsigned long long *>(win + (off & ~7u));
    a.n = off & 7u;
    a.v = 0;
}
// the low m bytes of x (1 <= m <= 8; the bytes above them are zero)
__device__ __forceinline__ void acc_push(TextAcc &a, u64 x, u32 m) {
    a.v |= x << (8u * a.n);
    const u32 t = a.n + m;
    if (t >= 8u) {
        atomicOr(a.slot, (unsigned long long)a.v);
        a.slot++;
        a.v = a.n ? x >> (8u * (8u - a.n)) : 0ull;
        a.n = t - 8u;
    } else {
        a.n = t;
    }
}
__device__ __forceinline__ void acc_flush(TextAcc &a) {
    if (a.v) atomicOr(a.slot, (unsigned long long)a.v);
    a.v = 0;
}
__device__ __forceinline__ u32 acc_offset(const TextAcc &a, const u8 *win) {  // window offset of the next byte
    return (u32)(reinterpret_cast<const u8 *>(a.slot) - win) + a.n;
}
__device__ __forceinline__ bool word_is_clean(u64 w) {  // no byte of w is escaped in the text (bytes behind a string's end read as 'a')
    const u64 lo = (w & 0x7f7f7f7f7f7f7f7full);
    const u64 ctl = ~((lo + 0x6060606060606060ull) | w) & 0x8080808080808080ull;
    const u64 q = zero_bytes(w ^ 0x2222222222222222ull) | zero_bytes(w ^ 0x5c5c5c5c5c5c5c5cull);
    return (ctl | q) == 0;
}
__device__ __forceinline__ u64 low_bytes(u64 w, u32 valid) { return valid >= 8u ? w : w & ((1ull << (8u * valid)) - 1ull); }

// WPE: waves per SIMD the register allocation aims at (launch bound), WINDOW: bytes of text a tile stages in LDS
// MODE 0: the counting pass (per-tile sizes; escaped lengths of the strings kept in slen), MODE 1: the writing pass of that
// pair (sizes scanned by k_ms_scan in between), MODE 2: both in ONE pass -- a tile measures, publishes its size in a
// descriptor, takes the sum of the tiles in front of it from their descriptors (decoupled look-back, tiles numbered by a
// ticket so that every predecessor is running or done) and writes; the text buffer is sized by a bound (ms_text_bound).
// STAGE (round 6): bytes of Strings.B a tile reads into LDS with coalesced 16-byte loads before it looks at its short strings --
// the strings of a tile lie side by side there (every string copied, in tape order), ~6.6 KB per tile on configs[4].  A thread
// fetching the first sixteen bytes of ITS strings from memory is a gather of 64 different cache lines per load instruction: the
// texture addresser works through them one line at a time, and those gathers alone were 0.24 of configs[4]'s 1.21 ms
// (profiles/r06_marshal_parts_ab.txt: 1.045 -> 0.806 without them, everything else in place).  From LDS the same sixteen bytes are three
// aligned 8-byte reads and two funnel shifts.  A string outside the staged range (a long stretch, a string in the message) is
// read from memory as before.
template <int MODE, int WPE, u32 WINDOW, u32 STAGE = 0>
__global__ __launch_bounds__(TW_THREADS, WPE) void k_ms_tile(MsView p) {
    constexpr bool EMIT = MODE != 0, ONEPASS = MODE == 2;
    __shared__ long long s_l[TW_THREADS / 64];
    __shared__ unsigned long long s_s[TW_THREADS / 64];
    __shared__ u32 s_len[TW_TILE];   // per word: text bytes of the entry that starts there (0: none); writing pass: then its offset
    // strings: short ones from the front, long ones from the back; idx | ordinal << 11 | sep << 21 | in Strings.B << 22 |
    // length << 23 (short ones) | offset << 32: a short string needs no second look at the tape
    __shared__ u64 s_qs[MS_QCAP];
    __shared__ uint16_t s_qn[MS_QCAP];  // numbers: integers from the front, floats from the back; idx | sep << 11 (12 bits)
    __shared__ __attribute__((aligned(16))) u8 s_text[EMIT ? WINDOW : 16];
    __shared__ __attribute__((aligned(16))) u8 s_str[STAGE ? STAGE + 32 : 16];
    __shared__ u32 s_lo;  // STAGE: the lowest Strings.B offset among the tile's short strings
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#if defined(SJ_EXP)
    unsigned long long t_prev = __builtin_readcyclecounter();
#endif
    __shared__ u8 s_cls[256];
    static_assert(TW_THREADS == 256, "one table entry per thread");
    s_cls[tid] = (u8)tag_class((u32)tid);
    __shared__ u32 s_tile;
    if (ONEPASS && tid == 0) s_tile = atomicAdd(p.ticket, 1u);
    __syncthreads();  // (the ticket and the table)
    MS_STAMP(0);  // ticket
    const u32 tile = ONEPASS ? s_tile : blockIdx.x;
    const u64 tb = (u64)tile * TW_TILE;
    const u64 base = tb + (u64)tid * TW_ITEMS;
    // requested with the tile's words, used later: the word the local anchor search of this thread looks at (64 words in front of the
    // tile, one per thread of wave 0) and the key flags of the thread's entries -- one aligned 4-byte load (tb and base are multiples
    // of eight words, a flag per two words) instead of a byte load per string in the middle of the classification
    u64 aw = 0;
    const bool a_have = tid < 64 && !p.tile_last && tb >= 1 + (u64)tid;
    if (a_have) aw = p.tape[tb - 1 - (u64)tid];
    u32 kf4 = 0;
    if (p.kf_tape && base < p.n) kf4 = *reinterpret_cast<const u32 *>(p.kf_tape + (base >> 1));
    u64 w[TW_ITEMS + 2];  // the thread's words and the two behind them (an entry's second word, the next entry's tag)
    if (base + TW_ITEMS + 2 <= p.n) {  // five 16-byte loads (the tape arena is 256-byte aligned, base a multiple of 8 words)
        const uint4 *q4 = reinterpret_cast<const uint4 *>(p.tape + base);
#pragma unroll
        for (int k = 0; k < (TW_ITEMS + 2) / 2; k++) {
            const uint4 v = q4[k];
            w[2 * k] = (u64)v.x | ((u64)v.y << 32);
            w[2 * k + 1] = (u64)v.z | ((u64)v.w << 32);
        }
    } else {
#pragma unroll
        for (int k = 0; k < TW_ITEMS + 2; k++) w[k] = base + k < p.n ? p.tape[base + k] : 0;
    }
    u32 clp[3] = {0u, 0u, 0u};  // the classes of the ten words, a byte each (ten LDS reads in flight, one wait)
#pragma unroll
    for (int k = 0; k < TW_ITEMS + 2; k++) clp[k >> 2] |= (u32)s_cls[(u32)(w[k] >> 56)] << (8 * (k & 3));
    u32 tgp[2] = {0u, 0u};  // ... and the tags of the thread's own eight
#pragma unroll
    for (int k = 0; k < TW_ITEMS; k++) tgp[k >> 2] |= (u32)(w[k] >> 56) << (8 * (k & 3));
    auto CL = [&](int k) -> u32 { return (clp[k >> 2] >> (8 * (k & 3))) & 0xffu; };  // (k is a constant wherever this is called)
    long long last = -1;
#pragma unroll
    for (int k = 0; k < TW_ITEMS; k++)
        if (base + k < p.n && !cls_two_word(CL(k))) last = (long long)(base + k);
    // the anchor in front of the tile: from the global scan (tile_last), or -- the common case, no extra pass over the
    // tape -- the closest of the 64 words in front of the tile that is not a two-word tag; a tile that finds none
    // (64 raw words that all look like string / number tags) reports it and the host repeats the walk with tile_last
    __shared__ long long s_carry;
    long long carry;
    if (p.tile_last) {
        carry = p.tile_last[tile];
    } else {  // (sj_tapewalk.h tw_local_anchor, with the word already in a register)
        if (tid < 64) {
            const u64 b = __ballot(a_have && !two_word_tag(aw));
            if (tid == 0) s_carry = b ? (long long)(tb - 1 - (u64)ctz64(b)) : (tb > 64 ? -2ll : -1ll);
        }
        __syncthreads();
        carry = s_carry;
    }
    // queue slots are drawn with LDS atomics, in whatever order the lanes arrive.  (Slots in document order from one packed
    // block scan -- no atomics, adjacent lanes on adjacent strings -- were measured in round 4: configs[4] 2.01 instead of
    // 1.75 ms, tools/gpu_ab_marshal.sh; and again in round 6, with the text leaving in aligned 8-byte pieces: 1.30 instead of
    // 1.14 ms, 1.33 with the lanes of a wave spread over distant slots of the ordered queue -- profiles/r06_marshal_parts_ab.txt.)
    __shared__ u32 s_cnt[4];         // short strings, long strings, integers, floats
    if (tid < 4) s_cnt[tid] = 0;
    if (STAGE && tid == 4) s_lo = 0xffffffffu;
    if (EMIT) {  // the window starts as zeros (TextAcc); the barriers of the scans below lie between this and the first byte of text
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        for (u32 i = (u32)tid; i < WINDOW / 16; i += TW_THREADS) reinterpret_cast<uint4 *>(s_text)[i] = z;
    }
    MS_STAMP(1);  // tape words, local anchor
    long long anchor = block_excl_max(last, s_l, tid);  // (its barriers also publish s_cnt = 0)
    MS_STAMP(2);  // block max-scan
    if (carry == -2) {  // (block-uniform) nothing of this tile can be classified: report and leave -- with a wrong anchor
        if (tid == 0) {  // raw words would be read as tags, their neighbours as string lengths
            atomicOr(&p.totals[2], 4ull);
            if (!EMIT) p.cnt_b[tile] = p.cnt_s[tile] = 0;
            if (ONEPASS) ms_desc_store(&p.desc[tile], MS_DESC_AGG);  // (no text: the tiles behind it must not wait for it)
        }
        return;
    }
    anchor = anchor > carry ? anchor : carry;

    // ---- 1. classify
    u8 isent[TW_ITEMS], sepf[TW_ITEMS];
    u32 nstr = 0;
    bool bad = false, toobig = false;
#pragma unroll
    for (int k = 0; k < TW_ITEMS; k++) {
        const u64 i = base + k;
        isent[k] = 0;
        sepf[k] = 0;
        if (i >= p.n) continue;
        const bool raw = anchor >= 0 && ((((long long)i - anchor - 1) & 1) != 0);
        const bool two = cls_two_word(CL(k));
        if (!two) anchor = (long long)i;
        if (raw) continue;
        isent[k] = 1;
        // separator behind a completed value: ',' unless the next entry closes something (for a key: ':', same length)
        const u32 nc = two ? CL(k + 2) : CL(k + 1);
        const bool last_entry = i + (two ? 2 : 1) >= p.n;
        sepf[k] = (!last_entry && !(nc & CLS_CLOSES)) ? 1 : 0;
        if ((CL(k) & CLS_KIND) == CLS_STR) nstr++;
    }
    MS_STAMP(3);  // classify (flags)
    unsigned long long tot_s = 0;
    u32 ord = (u32)block_excl_sum(nstr, s_s, tid, &tot_s);
    MS_STAMP(4);  // string ordinals (block scan)  // ordinal of the thread's first string inside the tile
    const u64 slen_base = (u64)tile * MS_QCAP;
    u32 my_lo = 0xffffffffu;  // STAGE: the lowest Strings.B offset among this thread's short strings
#pragma unroll
    for (int k = 0; k < TW_ITEMS; k++) {
        const u32 idx = (u32)tid * TW_ITEMS + (u32)k;
        u32 l = 0;
        if (isent[k]) {
            const u32 c = CL(k), kind = c & CLS_KIND, sep = sepf[k];
            l = (c & 7u) + ((c & CLS_SEP) ? sep : 0u);  // literals and brackets are done with this
            if (kind == CLS_STR) {
                const bool lng = w[k + 1] >= MS_LONG;
                const u32 slot = lng ? MS_QCAP - 1 - atomicAdd(&s_cnt[1], 1u) : atomicAdd(&s_cnt[0], 1u);
                const u64 vr = w[k] & TW_PAYLOAD;
                const u64 inbuf = (vr & STRINGBUFBIT) ? 1u : 0u;
                const u64 v = inbuf ? (vr & ~STRINGBUFBIT) - p.strings_base : vr - p.msg_base;  // (inside this context's buffers)
                // (the key flag of the parser rides in the entry: read here, neighbouring threads read neighbouring bytes; read
                // when the string is written it was one more gather per string)
                const u64 key = ((kf4 >> (8 * (k >> 1))) & 0xffu) != 0 ? 1ull : 0ull;  // (flag of word pair k / 2; 0 without parser flags)
                if (STAGE && inbuf && !lng && (u32)v < my_lo) my_lo = (u32)v;
                s_qs[slot] = (u64)(idx | (ord << 11) | (sep << 21)) | (inbuf << 22) | ((lng ? 0ull : w[k + 1]) << 23) | (key << 29) |
                             (v << 32);
                if (MODE == 1) l += p.slen[slen_base + ord];
                ord++;
            } else if (kind == CLS_INT) {
                s_qn[atomicAdd(&s_cnt[2], 1u)] = (uint16_t)(idx | (sep << 11));
            } else if (kind == CLS_FLT) {
                s_qn[MS_QCAP - 1 - atomicAdd(&s_cnt[3], 1u)] = (uint16_t)(idx | (sep << 11));
            } else if (kind == CLS_ROOT) {
                const bool is_open = (w[k] & TW_PAYLOAD) > p.tape_base + base + k;  // isOpenRoot (:441)
                l = (!is_open && base + k + 1 < p.n) ? 1u : 0u;       // '\n' between records
            } else if (kind == CLS_BAD) {
                bad = true;
            }
        }
        s_len[idx] = l;
    }
    if (STAGE) {  // one LDS atomic per wave (256 atomics on one address are executed one after the other -- by the LDS all four
        // blocks of the CU share: the first version of the staging was 0.29 ms SLOWER for it)
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) {
            const u32 o = (u32)__shfl_xor((int)my_lo, sft, 64);
            my_lo = o < my_lo ? o : my_lo;
        }
        if (lane == 0 && my_lo != 0xffffffffu) atomicMin(&s_lo, my_lo);
    }
    __syncthreads();
    MS_STAMP(5);  // queues
    const u32 n_short = s_cnt[0], n_long = s_cnt[1], n_int = s_cnt[2], n_flt = s_cnt[3];
    u64 st_base = 0;   // STAGE: bytes [st_base, st_base + st_avail) of Strings.B are in s_str
    u32 st_avail = 0;
    if (STAGE) {
        const u32 lo = s_lo;
        if (lo != 0xffffffffu) {
            st_base = (u64)(lo & ~15u);
            const u64 have = (u64)(p.strings_end - p.strings);
            const u64 left = have > st_base ? have - st_base : 0;
            st_avail = (u32)(left < STAGE ? left & ~15ull : (u64)STAGE);
            for (u32 i = (u32)tid * 16u; i < st_avail; i += TW_THREADS * 16u)
                *reinterpret_cast<uint4 *>(s_str + i) = *reinterpret_cast<const uint4 *>(p.strings + st_base + i);
        }
    }

    // ---- 2. measure (numbers in both passes: their digits are not kept; strings in the counting pass only)
    for (u32 j = (u32)tid; j < (MS_EXPBIT(p, 2) ? 0u : n_int); j += TW_THREADS) {
        const u32 idx = s_qn[j] & 0x7ffu;
        const u64 tw = p.tape[tb + idx], v = p.tape[tb + idx + 1];
        s_len[idx] += (u32)(tw >> 56) == 'l' ? int_text_len(v) : digit_count(v);
    }
    for (u32 j = (u32)tid; j < n_flt; j += TW_THREADS) {
        const u32 idx = s_qn[MS_QCAP - 1 - j] & 0x7ffu;
        const u32 nl = float_text_len(p.tape[tb + idx + 1]);
        if (nl == 0) bad = true;  // Inf / NaN: "INF or NaN number found"
        s_len[idx] += nl;
    }
    // The first sixteen bytes of the thread's short strings (thread t owns queue slots t, t + 256, ...: at most four) are
    // requested TOGETHER and stay in registers from the measuring to the writing: a string of up to 16 bytes -- nearly all of
    // configs[4]'s -- is read once, and the round trips of a thread's strings overlap instead of following one another inside
    // two loops with data-dependent inner loops (what the byte-store experiment really measured: without the loads of the writing
    // loop configs[4] took 0.99 instead of 1.30 ms, without those of the measuring loop 1.09; with the byte stores replaced by
    // 8-byte LDS operations and the loads left alone 1.25).
    constexpr int SPT = MS_QCAP / TW_THREADS;
    u64 sw0[SPT], sw1[SPT];
    if (STAGE) __syncthreads();  // s_str is loaded (the number loops above ran beside the loads)
    auto heads = [&]() {
#pragma unroll
        for (int k = 0; k < SPT; k++) {
            sw0[k] = sw1[k] = 0;
            const u32 j = (u32)tid + (u32)k * TW_THREADS;
            if (j < n_short && !MS_EXPBIT(p, 1)) {
                const u64 e64 = s_qs[j];
                const u32 e = (u32)e64;
                const u64 len = (e >> 23) & 0x3fu;
                const bool inbuf = (e >> 22) & 1u;
                const u8 *sp = ms_string(p, inbuf, e64 >> 32, len);
                const u8 *lim = inbuf ? p.strings_end : p.msg_end;
                const u64 off = e64 >> 32;
                if (STAGE && inbuf && off >= st_base && off - st_base + 24 <= (u64)st_avail) {
                    const u32 a = (u32)(off - st_base), sh = 8u * (a & 7u);
                    const u64 *q = reinterpret_cast<const u64 *>(s_str + (a & ~7u));
                    const u64 q0 = q[0], q1 = q[1], q2 = q[2];
                    sw0[k] = pad_a(sh ? (q0 >> sh) | (q1 << (64u - sh)) : q0, len);
                    sw1[k] = len > 8 ? pad_a(sh ? (q1 >> sh) | (q2 << (64u - sh)) : q1, len - 8) : 0ull;
                    continue;
                }
                if (len > 0) sw0[k] = str_load8(sp, 0, len, lim);
                if (len > 8) sw1[k] = str_load8(sp, 8, len, lim);
            }
        }
    };
    heads();
    static_assert(SPT == 4, "head() selects among four");
    auto head = [&](int k, bool second) -> u64 {  // (k is a loop counter of a loop that is NOT unrolled: selects, not indexing)
        const u64 a = second ? sw1[0] : sw0[0], b = second ? sw1[1] : sw0[1], c = second ? sw1[2] : sw0[2], d = second ? sw1[3] : sw0[3];
        return k == 0 ? a : k == 1 ? b : k == 2 ? c : d;
    };
    if (MODE != 1) {
#pragma unroll 1
        for (int k = 0; k < SPT; k++) {
            const u32 j = (u32)tid + (u32)k * TW_THREADS;
            if (j >= n_short) continue;
            const u64 e64 = s_qs[j];
            const u32 e = (u32)e64, idx = e & 0x7ffu;
            const u64 len = (e >> 23) & 0x3fu;
            const bool inbuf = (e >> 22) & 1u;
            const u8 *sp = ms_string(p, inbuf, e64 >> 32, len);
            const u8 *lim = inbuf ? p.strings_end : p.msg_end;
            u32 el = 0;
            if (MS_EXPBIT(p, 1)) el = (u32)len;
            else
            for (u64 q = 0; q < len; q += 8)
                el += esc_size8(q < 16 ? head(k, q == 8) : str_load8(sp, q, len, lim), (u32)(len - q < 8 ? len - q : 8));
            s_len[idx] += el;
            if (MODE == 0) p.slen[slen_base + ((e >> 11) & 0x3ffu)] = el;
        }
        for (u32 j = (u32)wave; j < n_long; j += TW_THREADS / 64) {  // one wave per long string
            const u64 e64 = s_qs[MS_QCAP - 1 - j];
            const u32 e = (u32)e64, idx = e & 0x7ffu;
            const u64 len = p.tape[tb + idx + 1];
            const bool inbuf = (e >> 22) & 1u;
            const u8 *sp = ms_string(p, inbuf, e64 >> 32, len);
            const u8 *lim = inbuf ? p.strings_end : p.msg_end;
            u64 el = 0;
            for (u64 q = (u64)lane * 8; q < len; q += 512) el += esc_size8(str_load8(sp, q, len, lim), (u32)(len - q < 8 ? len - q : 8));
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) el += (u64)__shfl_xor((long long)el, sft, 64);
            if (el > 0xfffffff0ull) toobig = true;  // (a single string of more than 4 GiB of text)
            if (lane == 0) {
                s_len[idx] += (u32)el;
                if (MODE == 0) p.slen[slen_base + ((e >> 11) & 0x3ffu)] = (u32)el;
            }
        }
    }
    __syncthreads();

    MS_STAMP(6);  // measure
    // ---- 3. positions
    u64 bytes = 0;
#pragma unroll
    for (int k = 0; k < TW_ITEMS; k++) bytes += s_len[tid * TW_ITEMS + k];
    unsigned long long tot = 0;
    const unsigned long long ex = block_excl_sum(bytes, s_s, tid, &tot);
    if (tot > 0xfffffff0ull) toobig = true;  // entry offsets inside a tile are 32-bit
    if (!EMIT) {
        if (tid == 0) {
            p.cnt_b[tile] = tot;
            p.cnt_s[tile] = tot_s;
        }
        if (bad) atomicOr(&p.totals[2], 1ull);
        if (toobig) atomicOr(&p.totals[2], 2ull);
        return;
    }
    // One pass: where the tile's text starts = the sizes of all tiles in front of it, from their descriptors.  The tile's own
    // size is published at once; the sum is only needed when bytes leave the block -- behind all the writing for a tile
    // that assembles its text in the LDS window (by then the tiles in front have published theirs: nothing to wait for),
    // in front of it for a tile that writes straight to memory.
    __shared__ unsigned long long s_off;
    const u64 tile_bytes = tot;
    const bool staged = tile_bytes <= WINDOW;  // block-uniform
    auto resolve = [&]() -> bool {  // block-uniform; false: nothing may be written
        if (wave == 0) {
            const unsigned long long off = MS_EXPBIT(p, 6) ? (unsigned long long)tile * 9600ull : ms_lookback(p.desc, tile, tot, lane, &p.totals[2]);
            if (lane == 0) {
                s_off = off;
                if ((u64)tile + 1 == p.tiles) p.totals[0] = off + tot;           // the length of the whole text
                if (off + tot > p.text_cap) atomicOr(&p.totals[2], 8ull);         // the bound did not hold: two passes
            }
        }
        __syncthreads();
        return s_off + tot <= p.text_cap;
    };
    if (ONEPASS) {
        if (bad) atomicOr(&p.totals[2], 1ull);
        const bool big = __syncthreads_or(toobig ? 1 : 0) != 0;  // (also: every wave is behind its last use of s_s)
        if (big) {  // reported; the tiles behind this one must not wait for it
            if (tid == 0) {
                atomicOr(&p.totals[2], 2ull);
                ms_desc_store(&p.desc[tile], MS_DESC_AGG);
            }
            return;
        }
        if (tid == 0) ms_desc_store(&p.desc[tile], (tile ? MS_DESC_AGG : MS_DESC_PREFIX) | tot);  // (tile 0 knows its prefix)
        if (!staged && !resolve()) return;
    }
    // The text of a tile is one contiguous range.  When it fits the window the block writes it into LDS (byte stores
    // that cost a fraction of scattered global ones) and copies the window out with coalesced 4-byte stores; a tile
    // with more text (long strings) writes straight to memory.
    MS_STAMP(7);  // positions scan, descriptor
    u8 *gdst = ONEPASS ? (staged ? nullptr : p.text + s_off) : p.text + p.cnt_b[tile];
    u8 *const tbase = staged ? s_text : gdst;
    {
        u32 run = (u32)ex;
#pragma unroll
        for (int k = 0; k < TW_ITEMS; k++) {
            const u32 idx = (u32)tid * TW_ITEMS + (u32)k;
            const u32 l = s_len[idx];
            s_len[idx] = run;
            const u32 c = CL(k), kind = c & CLS_KIND;
            if (isent[k] && l != 0 && (kind == CLS_PLAIN || kind == CLS_ROOT) && !MS_EXPBIT(p, 5)) {
                // literals, brackets and record separators are written here: the text is a constant selected by the tag (a bracket
                // stands for itself), the separator behind it, and the whole leaves as one piece
                const u32 t = (tgp[k >> 2] >> (8 * (k & 3))) & 0xffu;  // (the words themselves are dead by now: 20 registers less)
                u64 piece = t == 't' ? 0x65757274ull : t == 'n' ? 0x6c6c756eull : t == 'f' ? 0x65736c6166ull : t == 'r' ? 0x0aull : (u64)t;
                const u32 nb = c & 7u;  // (a root word brings no byte of its own: its '\n' is the whole text, l == 1)
                if (kind == CLS_PLAIN && sepf[k] && (c & CLS_SEP)) piece |= (u64)',' << (8u * nb);
                if (staged) {
                    const u32 sh = 8u * (run & 7u);
                    unsigned long long *slot = reinterpret_cast<unsigned long long *>(s_text + (run & ~7u));
                    atomicOr(slot, (unsigned long long)(piece << sh));
                    if (sh && (piece >> (64u - sh))) atomicOr(slot + 1, (unsigned long long)(piece >> (64u - sh)));
                } else {
                    u8 *o = tbase + run;
                    for (u32 j = 0; j < l; j++) o[j] = (u8)(piece >> (8u * j));
                }
            }
            run += l;
        }
    }
    __syncthreads();

    MS_STAMP(8);  // offsets, literals
    // ---- 4. write, queue by queue
    for (u32 j = (u32)tid; j < (MS_EXPBIT(p, 2) ? 0u : n_int); j += TW_THREADS) {
        const u32 e = s_qn[j], idx = e & 0x7ffu;
        const u64 tw = p.tape[tb + idx], v = p.tape[tb + idx + 1];
        u8 *o = tbase + s_len[idx];
        o += (u32)(tw >> 56) == 'l' ? format_int(v, o) : format_uint(v, o);
        if ((e >> 11) & 1u) *o = ',';
    }
    for (u32 j = (u32)tid; j < n_flt; j += TW_THREADS) {
        const u32 e = s_qn[MS_QCAP - 1 - j], idx = e & 0x7ffu;
        u8 *o = tbase + s_len[idx];
        const u32 nl = format_float(p.tape[tb + idx + 1], o);  // (exactly the bytes of the text: straight into the tile's window)
        if ((e >> 11) & 1u) o[nl] = ',';
    }
    const u64 key_base = p.kf_tape ? 0 : p.cnt_s[tile];
#pragma unroll 1
    for (int k = 0; k < SPT; k++) {
        const u32 j = (u32)tid + (u32)k * TW_THREADS;
        if (j >= (MS_EXPBIT(p, 4) ? 0u : n_short)) continue;
        const u64 e64 = s_qs[j];
        const u32 e = (u32)e64, idx = e & 0x7ffu;
        const u64 len = (e >> 23) & 0x3fu;
        const bool inbuf = (e >> 22) & 1u;
        const u8 *sp = ms_string(p, inbuf, e64 >> 32, len);
        const u8 *lim = inbuf ? p.strings_end : p.msg_end;
        if (staged) {  // (block-uniform) the text goes into the window eight bytes at a time
            TextAcc a;
            acc_init(a, s_text, s_len[idx]);
            acc_push(a, (u64)'"', 1);
            for (u64 q = 0; q < len && !MS_EXPBIT(p, 0); q += 8) {
                const u32 valid = (u32)(len - q < 8 ? len - q : 8);
                const u64 x = q < 16 ? head(k, q == 8) : str_load8(sp, q, len, lim);
                if (word_is_clean(x)) {
                    acc_push(a, low_bytes(x, valid), valid);
                } else {
                    acc_flush(a);
                    const u32 at = acc_offset(a, s_text);
                    acc_init(a, s_text, (u32)(write_esc8(s_text + at, x, valid) - s_text));
                }
            }
            u64 tail = (u64)'"';
            u32 nt = 1;
            if ((e >> 21) & 1u) {
                tail |= (u64)(entry_is_key(p, key_base, e, tb + idx) ? ':' : ',') << 8;
                nt = 2;
            }
            acc_push(a, tail, nt);
            acc_flush(a);
            continue;
        }
        u8 *o = tbase + s_len[idx];
        *o++ = '"';
        if (MS_EXPBIT(p, 0)) o += len;
        else
        for (u64 q = 0; q < len; q += 8)
            o = write_esc8(o, q < 16 ? head(k, q == 8) : str_load8(sp, q, len, lim), (u32)(len - q < 8 ? len - q : 8));
        *o++ = '"';
        if ((e >> 21) & 1u) *o = entry_is_key(p, key_base, e, tb + idx) ? ':' : ',';
    }
    for (u32 j = (u32)wave; j < n_long; j += TW_THREADS / 64) {  // one wave per long string: 512 bytes per step
        const u64 e64 = s_qs[MS_QCAP - 1 - j];
        const u32 e = (u32)e64, idx = e & 0x7ffu;
        const u64 len = p.tape[tb + idx + 1];
        const bool inbuf = (e >> 22) & 1u;
        const u8 *sp = ms_string(p, inbuf, e64 >> 32, len);
        const u8 *lim = inbuf ? p.strings_end : p.msg_end;
        u8 *o = tbase + s_len[idx];
        if (lane == 0) *o = '"';
        o++;
        for (u64 c0 = 0; c0 < len; c0 += 512) {
            const u64 q = c0 + (u64)lane * 8;
            u64 x = 0;
            u32 valid = 0, sz = 0;
            if (q < len) {
                valid = (u32)(len - q < 8 ? len - q : 8);
                x = str_load8(sp, q, len, lim);
                sz = esc_size8(x, valid);
            }
            u32 incl = sz;
#pragma unroll
            for (int sft = 1; sft < 64; sft <<= 1) {
                const u32 up = (u32)__shfl_up((int)incl, sft, 64);
                if (lane >= sft) incl += up;
            }
            if (valid) {
                if (staged && word_is_clean(x)) {  // eight bytes, two aligned slots
                    const u32 at = (u32)(o - s_text) + (incl - sz), sh = 8u * (at & 7u);
                    unsigned long long *slot = reinterpret_cast<unsigned long long *>(s_text + (at & ~7u));
                    const u64 piece = low_bytes(x, valid);
                    atomicOr(slot, (unsigned long long)(piece << sh));
                    if (sh && (piece >> (64u - sh))) atomicOr(slot + 1, (unsigned long long)(piece >> (64u - sh)));
                } else {
                    write_esc8(o + (incl - sz), x, valid);
                }
            }
            o += (u32)__shfl((int)incl, 63, 64);
        }
        if (lane == 0) {
            *o++ = '"';
            if ((e >> 21) & 1u) *o = entry_is_key(p, key_base, e, tb + idx) ? ':' : ',';
        }
    }
    if (staged) {
        __syncthreads();
        MS_STAMP(9);  // numbers and strings written
        if (ONEPASS) {
            if (!resolve()) return;
            gdst = p.text + s_off;
        }
        MS_STAMP(10);  // look-back
        const u32 nb = MS_EXPBIT(p, 3) ? 0u : (u32)tile_bytes, nw = nb >> 2;
        for (u32 i = (u32)tid; i < nw; i += TW_THREADS)  // unaligned 4-byte global stores are fine on gfx950
            *reinterpret_cast<u32 *>(gdst + 4 * i) = *reinterpret_cast<const u32 *>(s_text + 4 * i);
        const u32 tail = nw * 4 + (u32)tid;
        if (tail < nb) gdst[tail] = s_text[tail];
        MS_STAMP(11);  // copy-out
    }
}

// SJHIP_MS_VARIANT (experiments): register budget / LDS window of the tile kernel
static int ms_variant() {
    static const int v = [] {
        const char *e = getenv("SJHIP_MS_VARIANT");
        return e ? atoi(e) : 8;
    }();
    return v;
}
// Measured on configs[4] / configs[1] (tools/gpu_marshal_variants.sh, key flags from the parser): (4 waves, 32 KiB) 1.94 /
// 1.54 ms, (6, 32 KiB) 1.93 / 1.47, (6, 16 KiB) 1.74 / 1.32, (6 -- 8 is not reachable with this much LDS --, 8 KiB) 1.80 / 1.38:
// a tile of parking-citations is 9.5 KB of text, one of twitter.json 19 KB; four blocks per CU instead of three in the
// writing pass pay for the tiles that no longer fit the window.  With the single pass (same workloads, 1.30 / 1.28 ms at
// 16 KiB): 19 KiB 1.30 / 1.17, 21 KiB -- the largest window that leaves four blocks per CU, the number queue holding 16-bit
// entries -- 1.31 / 1.13: every tile that assembles its text in LDS resolves its offset late and leaves with coalesced stores.
static int ms_onepass() {  // SJHIP_MS_ONEPASS=0: always the two-pass form
    static const int v = [] {
        const char *e = getenv("SJHIP_MS_ONEPASS");
        return e ? atoi(e) : 1;
    }();
    return v;
}
template <int MODE>
static void launch_ms_tile(const MsView &p, hipStream_t st) {
    const dim3 g(p.tiles), b(TW_THREADS);
    switch (ms_variant()) {
        // (the second template argument is the occupancy the LDS of that window leaves: asking for more only earns hipcc's
        // "failed to meet occupancy target" -- the registers were never the limit, 80-91 VGPRs without scratch)
        case 0:
        case 1: hipLaunchKernelGGL((k_ms_tile<MODE, 3, MS_WINDOW>), g, b, 0, st, p); break;
        case 3: hipLaunchKernelGGL((k_ms_tile<MODE, 4, MS_WINDOW / 2>), g, b, 0, st, p); break;
        case 5: hipLaunchKernelGGL((k_ms_tile<MODE, 4, 19456u>), g, b, 0, st, p); break;
        case 7: hipLaunchKernelGGL((k_ms_tile<MODE, 4, 13312u, 8192u>), g, b, 0, st, p); break;
        case 6: hipLaunchKernelGGL((k_ms_tile<MODE, 4, 21504u>), g, b, 0, st, p); break;  // the largest window with four blocks per CU
        default:
            // Window + stage share what four blocks per CU leave (21.5 KB).  A tape whose tiles average at most ~6.5 KB of
            // Strings.B and ~11 KB of text (configs[4]: 6.6 / 9.5) takes the staged form; twitter.json's (15 / 19 KB) the large window.
            if (p.tiles && (u64)(p.strings_end - p.strings) / p.tiles <= 7168u && p.msg_len / p.tiles <= 11264u)
                hipLaunchKernelGGL((k_ms_tile<MODE, 4, 13312u, 8192u>), g, b, 0, st, p);
            else
                hipLaunchKernelGGL((k_ms_tile<MODE, 4, 21504u>), g, b, 0, st, p);
            break;
    }
}
}  // namespace

// debug build (-DSJ_DEBUG_BOUNDS): an out-of-bounds string of a MarshalJSON kernel fails the call (this translation unit's record)
static int marshal_bounds_check(sjhip_ctx *ctx) {
#if defined(SJ_DEBUG_BOUNDS)
    BoundsHit hit = {};
    if (hipMemcpyFromSymbol(&hit, HIP_SYMBOL(g_bounds_hit), sizeof hit) != hipSuccess) return SJHIP_OK;
    if (hit.hits) {
        const BoundsHit zero = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bounds_hit), &zero, sizeof zero);
        ctx_set_error(ctx, "bounds check (MarshalJSON): %u out-of-bounds strings, the first in array %u (sj_bounds.h ArrId) at byte %llu of %llu",
                      hit.hits, hit.id, hit.index, hit.size);
        return SJHIP_ERR_HIP;
    }
#else
    (void)ctx;
#endif
    return SJHIP_OK;
}

// Iter.MarshalJSON of the result `part` holds (ctx itself, or one shard of ctx's sharded ND result); errors are left in ctx
static int marshal_part(sjhip_ctx *ctx, sjhip_ctx *part, size_t *text_len) {
    part->ms_len = 0;
    part->ms_valid = 0;
    if (!part->r_valid || part->tape_len == 0) {
        ctx_set_error(ctx, "no parse result on the device (sjhip_marshal_json follows a successful sjhip_parse / sjhip_parse_device)");
        return SJHIP_ERR_ARG;
    }
    if (part->p_len >= (1ull << 32) && !(part->p_flags & SJHIP_FLAG_COPY_STRINGS)) {
        // k_ms_tile keeps the offset of a string in 32 bits of its queue entry: with WithCopyStrings(false) the strings that are not
        // copied lie at MESSAGE offsets, which pass 2^32 in a document of 4 GiB or more (Strings.B itself is checked to stay below)
        ctx_set_error(ctx, "sjhip_marshal_json: a document of 4 GiB or more parsed without SJHIP_FLAG_COPY_STRINGS (message offsets beyond 32 bits)");
        return SJHIP_ERR_TOOBIG;
    }
    HIPCHK(hipSetDevice(part->device), "hipSetDevice");
    part->ser_valid = 0;  // shares d_q with the serializer and the filter
    part->q_tape_len = part->q_strings_len = 0;
    part->f_valid = 0;
    MsView p;
    p.tape = (const u64 *)part->d_tape.p;
    p.n = part->tape_len;
    p.tiles = (u32)((p.n + TW_TILE - 1) / TW_TILE);
    p.tape_base = part->r_tape_base;
    p.strings_base = part->r_strings_base;
    p.msg_base = part->r_msg_base;
    p.strings = (const u8 *)part->d_strings.p;
    p.msg = (const u8 *)part->p_msg;
    p.strings_len = part->strings_len;
    p.msg_len = part->p_msg ? part->p_len : 0;
    p.strings_end = p.strings + part->strings_len;
    p.msg_end = p.msg ? p.msg + part->p_len : nullptr;
    KeyView kv;
    kv.kind = part->p_kind;
    kv.n = (u32)part->p_n;
    kv.tiles = (kv.n + 4095u) / 4096u;
    const size_t per = ((size_t)p.tiles * 8 + 255) / 256 * 256, perk = ((size_t)kv.tiles * 8 + 255) / 256 * 256;
    const size_t flags = ((size_t)kv.n + 255) / 256 * 256;
    int rc = arena_reserve(part, part->d_q, 256 + per * 4 + 256 + perk + flags + (size_t)p.tiles * MS_QCAP * 4 + 256);
    if (rc) return rc;
    char *w = (char *)part->d_q.p;
    p.totals = (unsigned long long *)w;
    w += 256;
    p.tile_last = (long long *)w;
    w += per;
    p.cnt_b = (unsigned long long *)w;
    w += per;
    p.cnt_s = (unsigned long long *)w;
    w += per;
    p.desc = (unsigned long long *)w;
    w += per;
    p.ticket = (u32 *)w;
    w += 256;
    kv.cnt = (unsigned long long *)w;
    w += perk;
    kv.keyflag = (u8 *)w;
    w += flags;
    p.keyflag = kv.keyflag;
    p.slen = (u32 *)w;
    p.text = nullptr;
    p.text_cap = 0;
    p.exp = 0;
#if defined(SJ_EXP)
    if (const char *e = getenv("SJHIP_MS_EXP")) p.exp = (u32)strtoul(e, nullptr, 0);
#endif
    HIPCHK(hipMemsetAsync(p.totals, 0, 256, part->stream), "marshal memset");
    // keys: the flags the parser left (SJHIP_FLAG_KEY_FLAGS), or from the token array of the parse (three launches)
    p.kf_tape = (part->kf_valid) ? (const u8 *)part->d_keyflag.p : nullptr;
    if (!p.kf_tape) {
        hipLaunchKernelGGL(k_ms_keys<false>, dim3(kv.tiles), dim3(256), 0, part->stream, kv);
        hipLaunchKernelGGL(k_ms_scan, dim3(1), dim3(1024), 0, part->stream, kv.cnt, (unsigned long long *)nullptr, kv.tiles,
                           (unsigned long long *)nullptr);
        hipLaunchKernelGGL(k_ms_keys<true>, dim3(kv.tiles), dim3(256), 0, part->stream, kv);
    }
    long long *const tile_last = p.tile_last;
    unsigned long long *h = (unsigned long long *)(part->h_scratch + 512);
    // With the key flags at hand, ONE pass over the tape (k_ms_tile<2>): the text buffer is sized by a bound instead of by a
    // counting pass.  The text of an entry is never longer than its source, except a number's: "1e20" prints as 21
    // digits (appendFloat leaves exponent form only from 1e21 on), 17 bytes more -- strings come out as they went in or
    // shorter (a raw control character is a stage-1 error, every escape the writer produces is at most as long as the
    // one the parser consumed), literals and brackets as they are, separators are a subset of the source's.  A tile that
    // would write past the bound (it cannot) or that finds no anchor raises a flag and the two-pass form below runs.
    bool no_local_anchor = false;
    bool one_pass = p.kf_tape && ms_onepass();
    size_t bound = part->p_len + 20 * (p.n / 2 + 1) + 64;
    if (one_pass) {
        // The bound is ~10 bytes per tape word above the real text: for a tape of several hundred million words that is
        // gigabytes of a grow-only arena.  Beyond SJHIP_MS_BOUND_LIMIT (default 8 GiB), or when the device cannot give
        // the block, the exact two-pass form below runs instead of failing.
        static const size_t limit = getenv("SJHIP_MS_BOUND_LIMIT") ? (size_t)strtoull(getenv("SJHIP_MS_BOUND_LIMIT"), nullptr, 0) : (size_t)8 << 30;
        if (const char *e = getenv("SJHIP_MS_TEST_BOUND")) bound = (size_t)strtoull(e, nullptr, 0);  // tests: make the bound fail
        if (bound > limit) one_pass = false;
        else if (arena_reserve(part, part->d_qtape, bound + 64) != SJHIP_OK) {
            (void)hipGetLastError();
            one_pass = false;
        }
    }
    if (one_pass) {
        p.tile_last = nullptr;
        p.text = (u8 *)part->d_qtape.p;
        p.text_cap = bound;
        HIPCHK(hipMemsetAsync(p.desc, 0, per + 256, part->stream), "marshal memset (descriptors)");  // (and the ticket behind them)
        launch_ms_tile<2>(p, part->stream);
        HIPCHK(hipGetLastError(), "marshal launch (one pass)");
        HIPCHK(hipMemcpyAsync(h, p.totals, 24, hipMemcpyDeviceToHost, part->stream), "D2H totals");
        HIPCHK(hipStreamSynchronize(part->stream), "marshal sync");
#if defined(SJ_EXP)
        if (p.exp >> 31) {
            unsigned long long t[32];
            if (hipMemcpy(t, p.totals, 256, hipMemcpyDeviceToHost) == hipSuccess) {
                unsigned long long sum = 0;
                for (int k = 0; k < 12; k++) sum += t[8 + k];
                fprintf(stderr, "k_ms_tile phases (share of thread 0's time, %u tiles):", p.tiles);
                for (int k = 0; k < 12; k++) fprintf(stderr, " %d:%.1f%%", k, sum ? 100.0 * (double)t[8 + k] / (double)sum : 0.0);
                fprintf(stderr, "  avg per sampled tile %.0f ticks\n", p.tiles ? (double)sum / ((p.tiles + 63) / 64) : 0.0);
            }
        }
#endif
        if (h[2] & 16ull) {
            ctx_set_error(ctx, "MarshalJSON: look-back aborted (internal synchronisation timeout)");
            return SJHIP_ERR_HIP;
        }
        if (!(h[2] & (4ull | 8ull))) {
            if (h[2] & 1ull) {
                ctx_set_error(ctx, "INF or NaN number found");  // the reference's error (parsed_json.go:1252)
                return SJHIP_ERR_ARG;
            }
            if (h[2] & 2ull) {
                ctx_set_error(ctx, "MarshalJSON: 2048 consecutive tape words produce more than 4 GiB of text");
                return SJHIP_ERR_TOOBIG;
            }
            part->ms_len = (size_t)h[0];
            part->ms_valid = 1;
            if (text_len) *text_len = part->ms_len;
            return marshal_bounds_check(ctx);
        }
        no_local_anchor = (h[2] & 4ull) != 0;  // (the counting pass below starts with the global anchors right away)
        HIPCHK(hipMemsetAsync(p.totals, 0, 256, part->stream), "marshal memset");
        p.text = nullptr;
    }
    // lengths and positions; the tag / raw anchors of the tiles are found locally unless a tile reports that it cannot
    for (int attempt = no_local_anchor ? 1 : 0; attempt < 2; attempt++) {
        if (attempt == 0) {
            p.tile_last = nullptr;
        } else {
            p.tile_last = tile_last;
            HIPCHK(hipMemsetAsync(p.totals, 0, 256, part->stream), "marshal memset");
            hipLaunchKernelGGL(k_tw_last, dim3(p.tiles), dim3(TW_THREADS), 0, part->stream, p.tape, p.n, p.tile_last);
            hipLaunchKernelGGL(k_tw_scan_last, dim3(1), dim3(1024), 0, part->stream, p.tile_last, p.tiles);
        }
        launch_ms_tile<0>(p, part->stream);
        // (the prefix of the string counts is only needed to index the recovered key flags)
        hipLaunchKernelGGL(k_ms_scan, dim3(1), dim3(1024), 0, part->stream, p.cnt_b, p.kf_tape ? (unsigned long long *)nullptr : p.cnt_s,
                           p.tiles, p.totals);
        HIPCHK(hipGetLastError(), "marshal launch");
        HIPCHK(hipMemcpyAsync(h, p.totals, 24, hipMemcpyDeviceToHost, part->stream), "D2H totals");
        HIPCHK(hipStreamSynchronize(part->stream), "marshal sync");
        if (!(h[2] & 4ull)) break;
    }
    if (h[2] & 1ull) {
        ctx_set_error(ctx, "INF or NaN number found");  // the reference's error (parsed_json.go:1252)
        return SJHIP_ERR_ARG;
    }
    if (h[2] & 2ull) {
        ctx_set_error(ctx, "MarshalJSON: 2048 consecutive tape words produce more than 4 GiB of text");
        return SJHIP_ERR_TOOBIG;
    }
    rc = arena_reserve(part, part->d_qtape, (size_t)h[0] + 64);
    if (rc) return rc;
    p.text = (u8 *)part->d_qtape.p;
    launch_ms_tile<1>(p, part->stream);
    HIPCHK(hipGetLastError(), "marshal emit launch");
    part->ms_len = (size_t)h[0];
    part->ms_valid = 1;
    if (text_len) *text_len = part->ms_len;
    return SJHIP_OK;
}

// The device-resident result of the last parse as text.  A sharded ND result (parse_nd_big) is marshaled shard by shard -- a shard
// ends at a record boundary, so the text of the whole result is the shards' texts joined with the '\n' that separates two records
// (parsed_json.go:401-556 writes one between records and none behind the last).
int sjhip_marshal_json(sjhip_ctx *ctx, size_t *text_len) {
    if (!ctx) return SJHIP_ERR_ARG;
    ctx->ms_len = 0;
    ctx->ms_valid = 0;
    if (!ctx->big_valid) return marshal_part(ctx, ctx, text_len);
    size_t total = 0;
    int np = 0;
    for (int k = 0; k < nd_big_shards(ctx); k++) {
        sjhip_ctx *part = nd_big_shard(ctx, k);
        if (!part) continue;
        size_t len = 0;
        const int rc = marshal_part(ctx, part, &len);
        if (rc) {
            if (part->err[0] && !ctx->err[0]) ctx_set_error(ctx, "%s", part->err);
            (void)hipSetDevice(ctx->device);
            return rc;
        }
        total += len + (np ? 1 : 0);
        np++;
    }
    (void)hipSetDevice(ctx->device);
    if (np == 0) {
        ctx_set_error(ctx, "no parse result on the device (sjhip_marshal_json follows a successful sjhip_parse / sjhip_parse_device)");
        return SJHIP_ERR_ARG;
    }
    ctx->ms_len = total;
    ctx->ms_valid = 1;
    if (text_len) *text_len = total;
    return SJHIP_OK;
}

int sjhip_fetch_marshaled(sjhip_ctx *ctx, uint8_t *dst) {
    if (!ctx || !ctx->ms_valid) return SJHIP_ERR_ARG;
    if (ctx->big_valid) {  // the shards' texts, joined with the newline between two records
        size_t at = 0;
        int np = 0;
        for (int k = 0; k < nd_big_shards(ctx); k++) {
            sjhip_ctx *part = nd_big_shard(ctx, k);
            if (!part || !part->ms_valid) continue;
            if (np++ && dst) dst[at++] = '\n';
            HIPCHK(hipSetDevice(part->device), "hipSetDevice");
            if (part->ms_len && dst)
                HIPCHK(hipMemcpyAsync(dst + at, part->d_qtape.p, part->ms_len, hipMemcpyDeviceToHost, part->stream), "D2H JSON text");
            HIPCHK(hipStreamSynchronize(part->stream), "fetch sync");
            at += part->ms_len;
        }
        HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
        return at == ctx->ms_len ? marshal_bounds_check(ctx) : SJHIP_ERR_ARG;
    }
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    if (ctx->ms_len && dst)
        HIPCHK(hipMemcpyAsync(dst, ctx->d_qtape.p, ctx->ms_len, hipMemcpyDeviceToHost, ctx->stream), "D2H JSON text");
    HIPCHK(hipStreamSynchronize(ctx->stream), "fetch sync");
    return marshal_bounds_check(ctx);  // (debug build: the writing pass has finished here)
}
