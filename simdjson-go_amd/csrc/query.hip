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
#include "sj_bounds.h"
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

// (Arr: sj_bounds.h -- a plain pointer in the product build; in the debug build (-DSJ_DEBUG_BOUNDS) every index that comes out
// of a tape word -- the end of a container, the offset and length of a string -- is checked against the array it is used on,
// and a violation fails the query instead of reading outside the arenas)
struct QView {
    Arr<const u64> tape;
    u64 tape_len;
    Arr<const u8> strings;
    u64 strings_len;
    Arr<const u8> msg; // device copy of the message (strings that were not copied point into it)
    u64 msg_len;
    Arr<const u32> nl_off; // tape offset (inside this context's tape) of the close root of record r (r < R)
    u32 R;             // record-separating newline runs: R + 1 records
    // A shard of a sharded ParseND (parse_nd_big: an ND message beyond one context's reach) stores its indices in the MERGED index
    // space: tape indices + tape_base, Strings.B offsets + the shard's Strings.B base, message offsets + its message base.  The view
    // of such a shard holds `tape`, `strings` and `msg` as pointers moved DOWN by those bases, so that every index a tape word holds
    // -- and every index the path queries hand out -- is used as it stands; only the record bounds (nl_off: local) add tape_base.
    u64 tape_base;     // 0 for an unsharded result; tape_len: END of this context's stretch in the merged index space
    u8 key[QMAX], val[QMAX];
    u32 klen, vlen;
};

__device__ __forceinline__ u64 rec_open(const QView &q, u32 r) { return q.tape_base + (r == 0 ? 0u : q.nl_off[r - 1] + 1u); }
__device__ __forceinline__ u64 rec_close(const QView &q, u32 r) { return r == q.R ? q.tape_len - 1u : q.tape_base + q.nl_off[r]; }

__device__ __forceinline__ const u8 *str_bytes(const QView &q, u64 word, u64 len) {
    const u64 p = word & PAYLOAD;
    return (p & STRINGBUFBIT) ? arr_at(q.strings, p & (STRINGBUFBIT - 1), len) : arr_at(q.msg, p, len);
}
__device__ __forceinline__ bool str_equals(const QView &q, u64 word, u64 len, const u8 *want, u32 wlen) {
    if (len != wlen) return false;
    const u8 *s = str_bytes(q, word, len);
    for (u32 k = 0; k < wlen; k++)
        if (s[k] != want[k]) return false;
    return true;
}

// FindKey(key) on the root object of record r + the string compare of countWhere
__device__ bool record_matches(const QView &q, u32 r) {
    const u64 o = rec_open(q, r);
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

// ---- pass 2: exclusive prefixes of words / string bytes over the records, and the Strings.B range of every record:
// [its first string, the first string of any later record) -- strings are laid out in document order, so "the first
// string of any later record" is a minimum over the records behind it.  Tiles of 1024 records (256 threads x 4
// consecutive records), tile sums scanned by one block, then applied: sums -> scan -> apply (word prefixes, string
// lengths and their tile sums) -> scan -> apply (string prefixes).
static constexpr int QT = 256, QI = 4, QTILE = QT * QI;
struct QTiles {
    unsigned long long *tw;  // [tiles] tape words of the tile's matching records -> their exclusive prefix
    unsigned long long *tb;  // [tiles] Strings.B bytes of the tile's matching records -> their exclusive prefix
    u32 *tc;                 // [tiles] matching records of the tile
    u32 *tf;                 // [tiles] first string of the tile -> first string of any later tile
};
__device__ __forceinline__ unsigned long long q_block_excl_sum(unsigned long long v, unsigned long long *s_w, int tid,
                                                                unsigned long long *total) {
    const int lane = tid & 63, wave = tid >> 6;
    unsigned long long incl = v;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const unsigned long long o = (unsigned long long)__shfl_up((long long)incl, s, 64);
        if (lane >= s) incl += o;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    unsigned long long before = 0, tot = 0;
    for (int w = 0; w < QT / 64; w++) {
        if (w < wave) before += s_w[w];
        tot += s_w[w];
    }
    if (total) *total = tot;
    __syncthreads();
    return before + incl - v;
}
// minimum over the threads behind this one (NONE32 if there is none)
__device__ __forceinline__ u32 q_block_excl_suffix_min(u32 v, u32 *s_w, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    u32 incl = v;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const u32 o = (u32)__shfl_down((int)incl, s, 64);
        if (lane + s < 64) incl = o < incl ? o : incl;
    }
    if (lane == 0) s_w[wave] = incl;
    __syncthreads();
    u32 after = NONE32;
    for (int w = 0; w < QT / 64; w++)
        if (w > wave) after = s_w[w] < after ? s_w[w] : after;
    u32 ex = (u32)__shfl_down((int)incl, 1, 64);
    if (lane == 63) ex = NONE32;
    __syncthreads();
    return ex < after ? ex : after;
}
template <typename T>
__device__ __forceinline__ void q_load4(const T *p, u32 base, u32 n, T fill, T (&v)[QI]) {
#pragma unroll
    for (int k = 0; k < QI; k++) v[k] = base + k < n ? p[base + k] : fill;  // consecutive: 16 bytes per thread
}

__global__ __launch_bounds__(QT) void k_q_tile_sums(QRec o, u32 n, QTiles T) {
    __shared__ unsigned long long s_w[QT / 64];
    __shared__ u32 s_m[QT / 64];
    const int tid = threadIdx.x;
    const u32 base = blockIdx.x * QTILE + (u32)tid * QI;
    u32 w[QI], fl[QI], f[QI];
    q_load4(o.words, base, n, 0u, w);
    q_load4(o.flag, base, n, 0u, fl);
    q_load4(o.first_str, base, n, NONE32, f);
    unsigned long long tw = 0, tc = 0;
    u32 mn = NONE32;
#pragma unroll
    for (int k = 0; k < QI; k++) {
        tw += w[k];
        tc += fl[k];
        mn = f[k] < mn ? f[k] : mn;
    }
    unsigned long long tot_w = 0, tot_c = 0;
    (void)q_block_excl_sum(tw, s_w, tid, &tot_w);
    (void)q_block_excl_sum(tc, s_w, tid, &tot_c);
    const u32 later = q_block_excl_suffix_min(mn, s_m, tid);
    if (tid == 0) {
        T.tw[blockIdx.x] = tot_w;
        T.tc[blockIdx.x] = (u32)tot_c;
        T.tf[blockIdx.x] = mn < later ? mn : later;
    }
}

// one block over the tiles.  FIRST: tw -> exclusive prefix, tf -> first string of any later tile, totals[0], [1];
// otherwise tb -> exclusive prefix, totals[2]
template <bool FIRST>
__global__ __launch_bounds__(1024) void k_q_tile_scan(QTiles T, u32 tiles, unsigned long long *totals, u32 strings_len) {
    __shared__ unsigned long long s_a[1024], s_c[1024];
    __shared__ u32 s_f[1024];
    const u32 tid = threadIdx.x, per = (tiles + 1023u) / 1024u;
    const u32 lo = tid * per < tiles ? tid * per : tiles, hi = lo + per < tiles ? lo + per : tiles;
    unsigned long long *col = FIRST ? T.tw : T.tb;
    unsigned long long a = 0, c = 0;
    u32 first = NONE32;
    for (u32 t = lo; t < hi; t++) {
        a += col[t];
        if (FIRST) {
            c += T.tc[t];
            first = T.tf[t] < first ? T.tf[t] : first;
        }
    }
    s_a[tid] = a;
    s_c[tid] = c;
    s_f[tid] = first;
    __syncthreads();
    if (tid == 0) {  // 1024 partials: serial is fine
        unsigned long long ra = 0, rc = 0;
        for (int k = 0; k < 1024; k++) {
            const unsigned long long va = s_a[k];
            s_a[k] = ra;
            ra += va;
            rc += s_c[k];
        }
        if (FIRST) {
            totals[0] = rc;
            totals[1] = ra;
            u32 nxt = strings_len;
            for (int k = 1023; k >= 0; k--) {
                const u32 v = s_f[k];
                s_f[k] = nxt;
                nxt = v < nxt ? v : nxt;
            }
        } else {
            totals[2] = ra;
        }
    }
    __syncthreads();
    unsigned long long run = s_a[tid];
    for (u32 t = lo; t < hi; t++) {
        const unsigned long long v = col[t];
        col[t] = run;
        run += v;
    }
    if (FIRST) {
        u32 nxt = s_f[tid];
        for (u32 t = hi; t > lo; t--) {
            const u32 v = T.tf[t - 1];
            T.tf[t - 1] = nxt;
            nxt = v < nxt ? v : nxt;
        }
    }
}

// words[r] := new index of the record's open root; s_len[r]; tile sums of s_len
__global__ __launch_bounds__(QT) void k_q_tile_apply1(QRec o, u32 n, QTiles T) {
    __shared__ unsigned long long s_w[QT / 64];
    __shared__ u32 s_m[QT / 64];
    const int tid = threadIdx.x;
    const u32 base = blockIdx.x * QTILE + (u32)tid * QI;
    u32 w[QI], fl[QI], f[QI];
    q_load4(o.words, base, n, 0u, w);
    q_load4(o.flag, base, n, 0u, fl);
    q_load4(o.first_str, base, n, NONE32, f);
    unsigned long long tw = 0;
    u32 mn = NONE32;
#pragma unroll
    for (int k = 0; k < QI; k++) {
        tw += w[k];
        mn = f[k] < mn ? f[k] : mn;
    }
    unsigned long long pw = T.tw[blockIdx.x] + q_block_excl_sum(tw, s_w, tid, nullptr);
    const u32 later = q_block_excl_suffix_min(mn, s_m, tid), behind_tile = T.tf[blockIdx.x];
    u32 nxt = later < behind_tile ? later : behind_tile;  // first string of any record behind this thread's four
    u32 len[QI];
    unsigned long long tb = 0;
#pragma unroll
    for (int k = QI - 1; k >= 0; k--) {
        u32 l = 0;
        if (f[k] != NONE32) {
            l = nxt - f[k];
            nxt = f[k];
        }
        len[k] = fl[k] ? l : 0u;
        tb += len[k];
    }
#pragma unroll
    for (int k = 0; k < QI; k++) {
        if (base + k < n) {
            o.words[base + k] = (u32)pw;
            o.s_len[base + k] = len[k];
        }
        pw += w[k];
    }
    unsigned long long tot_b = 0;
    (void)q_block_excl_sum(tb, s_w, tid, &tot_b);
    if (tid == 0) T.tb[blockIdx.x] = tot_b;
}

// s_pre[r] := new Strings.B offset of the record's first string
__global__ __launch_bounds__(QT) void k_q_tile_apply2(QRec o, u32 n, QTiles T) {
    __shared__ unsigned long long s_w[QT / 64];
    const int tid = threadIdx.x;
    const u32 base = blockIdx.x * QTILE + (u32)tid * QI;
    u32 len[QI];
    q_load4(o.s_len, base, n, 0u, len);
    unsigned long long tb = 0;
#pragma unroll
    for (int k = 0; k < QI; k++) tb += len[k];
    unsigned long long pb = T.tb[blockIdx.x] + q_block_excl_sum(tb, s_w, tid, nullptr);
#pragma unroll
    for (int k = 0; k < QI; k++) {
        if (base + k < n) o.s_pre[base + k] = (u32)pb;
        pb += len[k];
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

// ---- paths, typed values, key sets (round 5: Iter.FindElement parsed_json.go:833-865, Object.FindPath parsed_object.go:256-313,
// Object.ForEach with onlyKeys parsed_object.go:142-196) ----------------------------------------------------------------------
// The keys of a path / of a key set travel concatenated in QView::key; QPath holds where each one ends.
static constexpr int QPATH_MAX = 16;
struct QPath {
    u32 end[QPATH_MAX];  // key j = key[end[j - 1] .. end[j])
    u32 n;
};
__device__ __forceinline__ bool key_is(const QView &q, const QPath &pth, u32 j, u64 word, u64 len) {
    const u32 b = j ? pth.end[j - 1] : 0u;
    return str_equals(q, word, len, q.key + b, pth.end[j] - b);
}
__device__ __forceinline__ u64 skip_value(const QView &q, u64 v) {  // index behind the value whose first word is tape[v]
    const u64 vw = q.tape[v];
    const u32 vt = (u32)(vw >> 56);
    if (vt == '{' || vt == '[') return vw & PAYLOAD;  // behind the matching close
    return (vt == '"' || vt == 'l' || vt == 'u' || vt == 'd') ? v + 2 : v + 1;
}
// FindElement on record r: into the root, into objects, not into arrays; the first member with the key wins at every
// level.  Returns the tape index of the element's value, SJHIP_PATH_NOT_FOUND (ErrPathNotFound) or SJHIP_PATH_NOT_OBJECT
// ("type ... found before object was found" / "value of key ... is not an object").
__device__ u64 record_find_path(const QView &q, const QPath &pth, u32 r) {
    const u64 o = rec_open(q, r);
    const u64 w = q.tape[o + 1];
    if ((w >> 56) != '{') return SJHIP_PATH_NOT_OBJECT;
    u64 end = (w & PAYLOAD) - 1;  // index of the closing '}'
    u64 i = (u64)o + 2;
    u32 seg = 0;
    while (i < end) {
        const u64 v = i + 2;
        if (key_is(q, pth, seg, q.tape[i], q.tape[i + 1])) {
            if (seg + 1 == pth.n) return v;
            const u64 vw = q.tape[v];
            if ((vw >> 56) != '{') return SJHIP_PATH_NOT_OBJECT;
            end = (vw & PAYLOAD) - 1;
            i = v + 1;
            seg++;
            continue;
        }
        i = skip_value(q, v);
    }
    return SJHIP_PATH_NOT_FOUND;
}
// the typed comparisons: what Iter.String / Int / Uint / Float / Bool return for the element (parsed_json.go:560-749:
// integers, unsigned integers and floats convert into each other where the value fits), compared with the wanted value
__device__ bool element_is(const QView &q, u64 v, int op, u64 want) {
    const u64 w = q.tape[v];
    const u32 t = (u32)(w >> 56);
    switch (op) {
    case SJHIP_OP_EXISTS: return true;
    case SJHIP_OP_EQ_STRING: return t == '"' && str_equals(q, w, q.tape[v + 1], q.val, q.vlen);
    case SJHIP_OP_EQ_BOOL: return (t == 't' && want != 0) || (t == 'f' && want == 0);
    case SJHIP_OP_IS_NULL: return t == 'n';
    case SJHIP_OP_EQ_INT: {
        const u64 raw = (t == 'l' || t == 'u' || t == 'd') ? q.tape[v + 1] : 0;
        if (t == 'l') return (long long)raw == (long long)want;
        if (t == 'u') return raw <= 0x7fffffffffffffffull && (long long)raw == (long long)want;
        if (t == 'd') {
            // Iter.Int: an error above math.MaxInt64 / below math.MinInt64 (as float64 constants: 2^63 and -2^63), else int64(v) --
            // which for v == 2^63 is the amd64 conversion's "integer indefinite", MinInt64
            const double d = __longlong_as_double((long long)raw);
            if (d > 9223372036854775808.0 || d < -9223372036854775808.0) return false;
            const long long iv = d >= 9223372036854775808.0 || d != d ? (long long)0x8000000000000000ull : (long long)d;
            return iv == (long long)want;
        }
        return false;
    }
    case SJHIP_OP_EQ_UINT: {
        const u64 raw = (t == 'l' || t == 'u' || t == 'd') ? q.tape[v + 1] : 0;
        if (t == 'u') return raw == want;
        if (t == 'l') return (long long)raw >= 0 && raw == want;
        if (t == 'd') {
            // Iter.Uint (parsed_json.go:679-692): an error only for v > math.MaxUint64 -- which as a float64 constant is 2^64 --
            // and for v < 0; uint64(v) of exactly 2^64 is the amd64 conversion's result: (v - 2^63) converts to the integer
            // indefinite 0x8000000000000000, XORed with the sign bit = 0 (the mirror image of EQ_INT's 2^63 edge above)
            const double d = __longlong_as_double((long long)raw);
            if (d != d) return want == 0x8000000000000000ull;  // NaN never reaches the tape (parse_number rejects it); amd64: indefinite
            if (d < 0.0 || d > 18446744073709551616.0) return false;
            const u64 uv = d >= 18446744073709551616.0 ? 0ull : (u64)d;
            return uv == want;
        }
        return false;
    }
    case SJHIP_OP_EQ_FLOAT: {
        const double wd = __longlong_as_double((long long)want);
        if (t == 'd') return __longlong_as_double((long long)q.tape[v + 1]) == wd;
        if (t == 'l') return (double)(long long)q.tape[v + 1] == wd;
        if (t == 'u') return (double)q.tape[v + 1] == wd;
        return false;
    }
    }
    return false;
}
__global__ __launch_bounds__(256) void k_q_find_path(QView q, QPath pth, u64 *out) {
    const u32 r = blockIdx.x * 256 + threadIdx.x;
    if (r <= q.R) out[r] = record_find_path(q, pth, r);
}
__global__ __launch_bounds__(256) void k_q_count_path(QView q, QPath pth, int op, u64 want, unsigned long long *count) {
    const u32 r = blockIdx.x * 256 + threadIdx.x;
    bool m = false;
    if (r <= q.R) {
        const u64 v = record_find_path(q, pth, r);
        m = v < SJHIP_PATH_NOT_OBJECT && element_is(q, v, op, want);
    }
    const u64 b = __ballot(m);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(count, (unsigned long long)__popcll(b));
}
// ForEach(fn, onlyKeys) on the root object of record r: the members whose key is in the set, in document order, until as
// many members as the set has keys have been delivered (parsed_object.go:190-194: a key that occurs twice counts twice).
// out[r * n + j] = key number << 56 | tape index of the value of the j-th delivered member; ~0: no further member
__global__ __launch_bounds__(256) void k_q_project(QView q, QPath set, u64 *out) {
    const u32 r = blockIdx.x * 256 + threadIdx.x;
    if (r > q.R) return;
    u64 *dst = out + (u64)r * set.n;
    u32 n = 0;
    const u64 o = rec_open(q, r);
    const u64 w = q.tape[o + 1];
    if ((w >> 56) == '{') {
        const u64 end = (w & PAYLOAD) - 1;
        for (u64 i = (u64)o + 2; i < end && n < set.n;) {
            const u64 v = i + 2;
            for (u32 j = 0; j < set.n; j++)
                if (key_is(q, set, j, q.tape[i], q.tape[i + 1])) {
                    dst[n++] = ((u64)j << 56) | v;
                    break;
                }
            i = skip_value(q, v);
        }
    }
    for (; n < set.n; n++) dst[n] = ~0ull;
}

}  // namespace

namespace sj {
// nl_off and the record count of the last stage-2 run in this workspace (stage2.hip)
void stage2_records_view(void *ws, size_t n_tokens, const uint32_t **nl_off);
}

// debug build (-DSJ_DEBUG_BOUNDS): an out-of-bounds access of a query kernel fails the call (this translation unit's record)
static int query_bounds_check(sjhip_ctx *ctx) {
#if defined(SJ_DEBUG_BOUNDS)
    BoundsHit h = {};
    if (hipMemcpyFromSymbol(&h, HIP_SYMBOL(g_bounds_hit), sizeof h) != hipSuccess) return SJHIP_OK;
    if (h.hits) {
        const BoundsHit zero = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bounds_hit), &zero, sizeof zero);
        ctx_set_error(ctx, "bounds check (query): %u out-of-bounds accesses, the first to array %u (sj_bounds.h ArrId) at element %llu of %llu",
                      h.hits, h.id, h.index, h.size);
        return SJHIP_ERR_HIP;
    }
#else
    (void)ctx;
#endif
    return SJHIP_OK;
}

// The contexts whose device-resident results make up the last parse of `ctx`, in document order: the context itself, or --
// after an ND message beyond one context's reach (parse_nd_big) -- the contexts of its shards.  Iter / ForEach / FindElement of
// the reference work on any ParsedJson (parsed_json.go:96,125,833); here the count and path queries run shard by shard in the
// merged index space (QView) and the host adds the counts up / lays the per-record answers end to end.
static int result_parts(sjhip_ctx *ctx, sjhip_ctx **parts, int cap) {
    if (!ctx->big_valid) {
        parts[0] = ctx;
        return 1;
    }
    int n = 0;
    for (int k = 0; k < nd_big_shards(ctx) && n < cap; k++)
        if (sjhip_ctx *c = nd_big_shard(ctx, k)) parts[n++] = c;
    return n;
}
static constexpr int MAX_PARTS = 4096;  // (parse_nd_big's own limit)

// view of the result held by `part` (ctx itself, or one shard context of ctx's sharded result); errors are left in ctx
static int make_view(sjhip_ctx *ctx, sjhip_ctx *part, const uint8_t *key, size_t klen, const uint8_t *val, size_t vlen, QView *q,
                     uint32_t *records) {
    if (!ctx || !key || !val) return SJHIP_ERR_ARG;
    if (klen > QMAX || vlen > QMAX) {
        ctx_set_error(ctx, "query key / value longer than %d bytes", QMAX);
        return SJHIP_ERR_ARG;
    }
    if (!part || !part->r_valid || part->tape_len == 0) {
        ctx_set_error(ctx, "no parse result on the device (queries follow a successful sjhip_parse / sjhip_parse_device)");
        return SJHIP_ERR_ARG;
    }
    const uint32_t *nl = nullptr;
    stage2_records_view(part->d_s2.p, part->p_nlay, &nl);
    // (pointers moved down by the shard's bases: see QView; all zero for an unsharded result)
    q->tape_base = part->r_tape_base;
    q->tape = SJ_ARR((const u64 *)part->d_tape.p - part->r_tape_base, part->r_tape_base + part->tape_len, A_TAPE);
    q->tape_len = part->r_tape_base + part->tape_len;
    q->strings = SJ_ARR((const u8 *)part->d_strings.p - part->r_strings_base, part->r_strings_base + part->strings_len, A_STRINGS);
    q->strings_len = part->strings_len;
    q->msg = SJ_ARR((const u8 *)part->p_msg - part->r_msg_base, part->r_msg_base + part->p_len, A_MSG);
    q->msg_len = part->p_len;
    q->nl_off = SJ_ARR(nl, part->q_records, A_NL_OFF);
    q->R = part->q_records;
    memset(q->key, 0, QMAX);
    memset(q->val, 0, QMAX);
    memcpy(q->key, key, klen);
    memcpy(q->val, val, vlen);
    q->klen = (u32)klen;
    q->vlen = (u32)vlen;
    *records = part->q_records + 1u;
    return SJHIP_OK;
}
// the unsharded result of ctx itself (filter / what needs q_valid)
static int make_view(sjhip_ctx *ctx, const uint8_t *key, size_t klen, const uint8_t *val, size_t vlen, QView *q, uint32_t *records) {
    if (ctx && !ctx->q_valid) {
        if (ctx->big_valid) ctx_set_error(ctx, "sjhip_filter_where works on the result of one context; this ND result was parsed shard by shard");
        else ctx_set_error(ctx, "no parse result on the device (queries follow a successful sjhip_parse / sjhip_parse_device)");
        return SJHIP_ERR_ARG;
    }
    return make_view(ctx, ctx, key, klen, val, vlen, q, records);
}
// Runs `launch(part, q, n, d_count)` on every part of ctx's result and adds the 8-byte counts up.
template <typename F>
static int count_over_parts(sjhip_ctx *ctx, const uint8_t *key, size_t klen, const uint8_t *val, size_t vlen, uint64_t *count, F launch) {
    if (!ctx) return SJHIP_ERR_ARG;
    static thread_local sjhip_ctx *parts[MAX_PARTS];
    const int np = result_parts(ctx, parts, MAX_PARTS);
    if (np == 0) return make_view(ctx, nullptr, key, klen, val, vlen, nullptr, nullptr);  // (the error text)
    uint64_t total = 0;
    for (int k = 0; k < np; k++) {  // every part is queued on its own stream (its own device) before any is waited for
        sjhip_ctx *part = parts[k];
        QView q;
        uint32_t n = 0;
        int rc = make_view(ctx, part, key, klen, val, vlen, &q, &n);
        if (rc) return rc;
        HIPCHK(hipSetDevice(part->device), "hipSetDevice");
        rc = arena_reserve(part, part->d_kat, 64);
        if (rc) return rc;
        HIPCHK(hipMemsetAsync(part->d_kat.p, 0, 8, part->stream), "count memset");
        launch(part, q, n, (unsigned long long *)part->d_kat.p);
        HIPCHK(hipGetLastError(), "count launch");
        HIPCHK(hipMemcpyAsync(part->h_scratch + 512, part->d_kat.p, 8, hipMemcpyDeviceToHost, part->stream), "D2H count");
    }
    for (int k = 0; k < np; k++) {
        HIPCHK(hipSetDevice(parts[k]->device), "hipSetDevice");
        HIPCHK(hipStreamSynchronize(parts[k]->stream), "count sync");
        total += *(const unsigned long long *)(parts[k]->h_scratch + 512);
    }
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    *count = total;
    return query_bounds_check(ctx);
}

int sjhip_count_where(sjhip_ctx *ctx, const uint8_t *key, size_t klen, const uint8_t *value, size_t vlen, uint64_t *count) {
    if (!count) return SJHIP_ERR_ARG;
    return count_over_parts(ctx, key, klen, value, vlen, count, [](sjhip_ctx *part, const QView &q, uint32_t n, unsigned long long *d) {
        hipLaunchKernelGGL(k_q_count, dim3((n + 255) / 256), dim3(256), 0, part->stream, q, d);
    });
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
    ctx->f_valid = 0;
    const size_t per = ((size_t)n * 4 + 255) / 256 * 256;
    const u32 tiles = (n + QTILE - 1) / QTILE;
    const size_t per_t = ((size_t)tiles * 8 + 255) / 256 * 256;
    rc = arena_reserve(ctx, ctx->d_q, per * 5 + per_t * 3 + 256);
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
    w += per;
    QTiles T;
    T.tw = (unsigned long long *)w;
    w += per_t;
    T.tb = (unsigned long long *)w;
    w += per_t;
    T.tc = (u32 *)w;
    T.tf = (u32 *)(w + per_t / 2);
    hipLaunchKernelGGL(k_q_mark, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, q, o);
    hipLaunchKernelGGL(k_q_tile_sums, dim3(tiles), dim3(QT), 0, ctx->stream, o, n, T);
    hipLaunchKernelGGL(k_q_tile_scan<true>, dim3(1), dim3(1024), 0, ctx->stream, T, tiles, o.totals, (u32)q.strings_len);
    hipLaunchKernelGGL(k_q_tile_apply1, dim3(tiles), dim3(QT), 0, ctx->stream, o, n, T);
    hipLaunchKernelGGL(k_q_tile_scan<false>, dim3(1), dim3(1024), 0, ctx->stream, T, tiles, o.totals, (u32)q.strings_len);
    hipLaunchKernelGGL(k_q_tile_apply2, dim3(tiles), dim3(QT), 0, ctx->stream, o, n, T);
    HIPCHK(hipGetLastError(), "filter launch");
    unsigned long long *h = (unsigned long long *)(ctx->h_scratch + 512);
    HIPCHK(hipMemcpyAsync(h, o.totals, 24, hipMemcpyDeviceToHost, ctx->stream), "D2H totals");
    HIPCHK(hipStreamSynchronize(ctx->stream), "filter sync");
    ctx->q_tape_len = (size_t)h[1];
    ctx->q_strings_len = (size_t)h[2];
    ctx->f_valid = 1;
    if (n_records) *n_records = h[0];
    if (tape_len) *tape_len = ctx->q_tape_len;
    if (strings_len) *strings_len = ctx->q_strings_len;
    if (h[0] == 0) return query_bounds_check(ctx);
    hipLaunchKernelGGL(k_q_copy, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, q, o, (u64 *)ctx->d_qtape.p, (u8 *)ctx->d_qstrings.p,
                       (u32)h[1]);
    HIPCHK(hipGetLastError(), "filter copy launch");
    return SJHIP_OK;
}

int sjhip_fetch_filtered(sjhip_ctx *ctx, uint64_t *tape_dst, uint8_t *strings_dst) {
    if (!ctx) return SJHIP_ERR_ARG;
    if (!ctx->f_valid) {  // no filter ran, or a later parse / serialize / marshal call re-used its arenas
        ctx_set_error(ctx, "no filtered result on the device (sjhip_fetch_filtered follows sjhip_filter_where)");
        return SJHIP_ERR_ARG;
    }
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    if (ctx->q_tape_len && tape_dst)
        HIPCHK(hipMemcpyAsync(tape_dst, ctx->d_qtape.p, ctx->q_tape_len * 8, hipMemcpyDeviceToHost, ctx->stream), "D2H filtered tape");
    if (ctx->q_strings_len && strings_dst)
        HIPCHK(hipMemcpyAsync(strings_dst, ctx->d_qstrings.p, ctx->q_strings_len, hipMemcpyDeviceToHost, ctx->stream),
               "D2H filtered strings");
    HIPCHK(hipStreamSynchronize(ctx->stream), "fetch sync");
    return query_bounds_check(ctx);  // (debug build: the copy kernel of sjhip_filter_where has finished here)
}

// ---- paths, typed values, key sets ------------------------------------------------------------------------------------------
// the keys of a path / key set: where each one ends in the concatenation (QView::key holds the bytes)
static int make_path(sjhip_ctx *ctx, const uint8_t *keys, const uint32_t *key_lens, uint32_t n_keys, QPath *pth, size_t *total_out) {
    if (!ctx || !keys || !key_lens || n_keys == 0) return SJHIP_ERR_ARG;
    if (n_keys > (uint32_t)QPATH_MAX) {
        ctx_set_error(ctx, "a path / key set holds at most %d keys", QPATH_MAX);
        return SJHIP_ERR_ARG;
    }
    size_t total = 0;
    for (uint32_t j = 0; j < n_keys; j++) {
        total += key_lens[j];
        if (total > (size_t)QMAX) {
            ctx_set_error(ctx, "the keys of a path / key set are longer than %d bytes together", QMAX);
            return SJHIP_ERR_ARG;
        }
        pth->end[j] = (u32)total;
    }
    for (uint32_t j = n_keys; j < (uint32_t)QPATH_MAX; j++) pth->end[j] = (u32)total;
    pth->n = n_keys;
    *total_out = total;
    return SJHIP_OK;
}

// Per-record answers (`per` 8-byte words for every record) of every part of ctx's result, laid end to end in `out` in document
// order: launch(part, q, n, d_out) fills n * per words on the part's device.  cap_records: room in `out`; *records: records of
// the whole result.
template <typename F>
static int records_over_parts(sjhip_ctx *ctx, const uint8_t *keys, size_t klen, uint32_t per, uint64_t *out, size_t cap_records,
                              size_t *records, const char *who, F launch) {
    static const uint8_t none = 0;
    static thread_local sjhip_ctx *parts[MAX_PARTS];
    const int np = result_parts(ctx, parts, MAX_PARTS);
    if (np == 0) return make_view(ctx, nullptr, keys, klen, &none, 0, nullptr, nullptr);
    size_t total = 0;
    for (int k = 0; k < np; k++) total += (size_t)parts[k]->q_records + 1u;
    *records = total;
    if (cap_records < total) {
        ctx_set_error(ctx, "%s: room for %zu records, the result holds %zu", who, cap_records, total);
        return SJHIP_ERR_ARG;
    }
    size_t at = 0;
    for (int k = 0; k < np; k++) {
        sjhip_ctx *part = parts[k];
        QView q;
        uint32_t n = 0;
        int rc = make_view(ctx, part, keys, klen, &none, 0, &q, &n);
        if (rc) return rc;
        HIPCHK(hipSetDevice(part->device), "hipSetDevice");
        const size_t bytes = (size_t)n * per * 8;
        rc = arena_reserve(part, part->d_kat, bytes + 64);
        if (rc) return rc;
        launch(part, q, n, (u64 *)part->d_kat.p);
        HIPCHK(hipGetLastError(), "query launch");
        HIPCHK(hipMemcpyAsync(out + at * per, part->d_kat.p, bytes, hipMemcpyDeviceToHost, part->stream), "D2H per-record answers");
        at += n;
    }
    for (int k = 0; k < np; k++) {
        HIPCHK(hipSetDevice(parts[k]->device), "hipSetDevice");
        HIPCHK(hipStreamSynchronize(parts[k]->stream), "query sync");
    }
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    return query_bounds_check(ctx);
}

int sjhip_find_path(sjhip_ctx *ctx, const uint8_t *keys, const uint32_t *key_lens, uint32_t n_keys, uint64_t *index_out,
                    size_t cap, size_t *records) {
    if (!index_out || !records) return SJHIP_ERR_ARG;
    QPath pth;
    size_t klen = 0;
    const int rc = make_path(ctx, keys, key_lens, n_keys, &pth, &klen);
    if (rc) return rc;
    return records_over_parts(ctx, keys, klen, 1, index_out, cap, records, "sjhip_find_path",
                              [&](sjhip_ctx *part, const QView &q, uint32_t n, u64 *d) {
                                  hipLaunchKernelGGL(k_q_find_path, dim3((n + 255) / 256), dim3(256), 0, part->stream, q, pth, d);
                              });
}

int sjhip_count_where_path(sjhip_ctx *ctx, const uint8_t *keys, const uint32_t *key_lens, uint32_t n_keys, int op,
                           const void *value, size_t vlen, uint64_t *count) {
    if (!count || op < SJHIP_OP_EXISTS || op > SJHIP_OP_IS_NULL) return SJHIP_ERR_ARG;
    const bool is_str = op == SJHIP_OP_EQ_STRING;
    u64 want = 0;
    if (op == SJHIP_OP_EQ_INT || op == SJHIP_OP_EQ_UINT || op == SJHIP_OP_EQ_FLOAT) {
        if (!value || vlen != 8) return SJHIP_ERR_ARG;  // int64_t / uint64_t / double
        memcpy(&want, value, 8);
    } else if (op == SJHIP_OP_EQ_BOOL) {
        if (!value || vlen != 1) return SJHIP_ERR_ARG;
        want = *(const uint8_t *)value != 0;
    } else if (is_str && !value && vlen) {
        return SJHIP_ERR_ARG;
    }
    QPath pth;
    size_t klen = 0;
    const int rc = make_path(ctx, keys, key_lens, n_keys, &pth, &klen);
    if (rc) return rc;
    static const uint8_t none = 0;
    return count_over_parts(ctx, keys, klen, is_str && value ? (const uint8_t *)value : &none, is_str ? vlen : 0, count,
                            [&](sjhip_ctx *part, const QView &q, uint32_t n, unsigned long long *d) {
                                hipLaunchKernelGGL(k_q_count_path, dim3((n + 255) / 256), dim3(256), 0, part->stream, q, pth, op, want, d);
                            });
}

int sjhip_project_keys(sjhip_ctx *ctx, const uint8_t *keys, const uint32_t *key_lens, uint32_t n_keys, uint64_t *out,
                       size_t cap_records, size_t *records) {
    if (!out || !records) return SJHIP_ERR_ARG;
    QPath set;
    size_t klen = 0;
    const int rc = make_path(ctx, keys, key_lens, n_keys, &set, &klen);
    if (rc) return rc;
    for (uint32_t a = 0; a < n_keys; a++)  // a set: the reference's onlyKeys is a map
        for (uint32_t b = a + 1; b < n_keys; b++) {
            const u32 ab = a ? set.end[a - 1] : 0, bb = set.end[b - 1];
            if (set.end[a] - ab == set.end[b] - bb && memcmp(keys + ab, keys + bb, set.end[a] - ab) == 0) {
                ctx_set_error(ctx, "sjhip_project_keys: key %u and key %u are equal", a, b);
                return SJHIP_ERR_ARG;
            }
        }
    return records_over_parts(ctx, keys, klen, n_keys, out, cap_records, records, "sjhip_project_keys",
                              [&](sjhip_ctx *part, const QView &q, uint32_t n, u64 *d) {
                                  hipLaunchKernelGGL(k_q_project, dim3((n + 255) / 256), dim3(256), 0, part->stream, q, set, d);
                              });
}
