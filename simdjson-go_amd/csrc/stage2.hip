// stage2.hip -- stage 2 (tape build, string unescape, number parse) as follow-on kernels over the
// structural-index array produced by stage 1.  Replaces unifiedMachine
// (stage2_build_tape_amd64.go:160-446), parseString (:72-113), parse_string_amd64.s and
// parseNumber (parse_number.go:65-135).  The per-token logic lives in sj_stage2.h / sj_number.h /
// sj_bignum.h (host+device, replayed on the CPU by the test-suite); this file holds the kernels,
// the device-wide scans and the launcher.  No host synchronisation happens between the kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sj_bignum.h"
#include "sj_chunk.h"
#include "sj_device.h"
#include "sj_number.h"
#include "sj_stage2.h"
#include "sj_strings.h"

namespace sj {

static constexpr int S2_BLOCK = 256;
static constexpr int S2_ITEMS = 4;
static constexpr int S2_TILE = S2_BLOCK * S2_ITEMS;
static constexpr u32 DLEN_INVALID = 0xffffffffu;
static constexpr u32 DLEN_COPY = 0x80000000u;

// device view of all stage-2 arrays (carved out of one workspace by the launcher)
struct S2Dev {
    const u8 *msg;
    u64 len;
    const u32 *pos;
    u32 n;
    u32 ndjson, copy_strings;
    u8 *kind;      // [n]
    u32 *dlen;     // [n] strings: unescaped length | DLEN_COPY, or DLEN_INVALID
    i32 *depth;    // [n]
    u32 *tape_off; // [n]
    u32 *str_off;  // [n]
    u32 *last_br;  // [n]
    u32 *match;    // [n] brackets: partner; record-separating newline: its ordinal
    u8 *ctxb;      // [n]
    u32 *nlb;      // [n] token indexes of record-separating newlines
    u32 *bigq;     // [n] queue of number tokens that need the big-integer tie-break
    u32 *br_tok;   // [n] compact bracket view: token index of the c-th bracket
    i32 *br_depth; // [n]                       depth after it (level 0 of the min tree)
    // tile aggregates / exclusive prefixes
    i32 *agg_d;
    u32 *agg_w, *agg_s, *agg_lb, *agg_nb, *agg_bc;
    u32 tiles;
    // min tree levels 1.. (level 0 is br_depth[])
    i32 *lev[MinTree::MAXLEV];
    u64 lev_size[MinTree::MAXLEV];
    int nlev;
    S2State *st;
    u64 *tape;
    u8 *strings;
    u64 tape_cap, strings_cap;
    u64 tape_base, strings_base, msg_base;  // NDJSON shard: rebasing of every stored index (0 if unsharded)
    // byte-parallel string path (copy_strings): masks from stage 1 and what the string kernels derive from them
    StrView sv;           // base / lead / end / qm q st unit_h (null qm: path not used)
    u64 *em, *um;         // [chunks] emit mask, 'u' mask
    uint16_t *chunk_pre;  // [chunks] emitted bytes of the unit in front of the chunk
    u32 *unit_cnt;        // [units]  emitted bytes of the unit, then (k_str_scan) their exclusive prefix
    u64 units;
};

struct Agg {
    i32 d;
    u32 w, s, lb, nb, bc;
};
__device__ __forceinline__ Agg agg_combine(const Agg &a, const Agg &b) {
    return Agg{a.d + b.d, a.w + b.w, a.s + b.s, a.lb > b.lb ? a.lb : b.lb, a.nb + b.nb, a.bc + b.bc};
}
__device__ __forceinline__ Agg agg_shfl_up(const Agg &a, int delta) {
    return Agg{__shfl_up(a.d, delta, 64), __shfl_up(a.w, delta, 64), __shfl_up(a.s, delta, 64), __shfl_up(a.lb, delta, 64),
               __shfl_up(a.nb, delta, 64), __shfl_up(a.bc, delta, 64)};
}

__device__ __forceinline__ Agg token_agg(const S2Dev &p, u32 i) {
    const u8 k = p.kind[i];
    const bool last = i + 1 == p.n;
    const u8 nk = last ? (u8)K_BAD : p.kind[i + 1];
    Agg a;
    a.d = depth_delta(k);
    a.w = tape_words(k, nk, last);
    const u32 dl = p.dlen[i];
    a.s = (k == K_STRING && dl != DLEN_INVALID && (dl & DLEN_COPY)) ? (dl & ~DLEN_COPY) : 0u;
    a.lb = is_bracket(k) ? i + 1 : 0u;
    a.nb = (k == K_NL && !last && nk != K_NL) ? 1u : 0u;
    a.bc = is_bracket(k) ? 1u : 0u;
    return a;
}

// ---- string kernels (copy_strings): sj_strings.h, one 64-byte chunk per lane, one 4 KiB unit per wave ---------
__global__ __launch_bounds__(256) void k_str_masks(S2Dev p) {
    const u64 c = (u64)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if (c >= p.units * 64) return;  // whole waves
    u64 em, um;
    if (!str_chunk_masks(p.sv, c, &em, &um)) atomicOr(&p.st->err, 1u);
    p.em[c] = em;
    p.um[c] = um;
    const u32 n = (u32)popc64(em);
    u32 incl = n;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const u32 o = __shfl_up(incl, s, 64);
        if (lane >= s) incl += o;
    }
    p.chunk_pre[c] = (uint16_t)(incl - n);
    if (lane == 63) p.unit_cnt[c >> 6] = incl;
}

// one block of 1024 threads: exclusive scan of u32 data[n] in place (n padded to a multiple of 4 by the caller's
// allocation), 4096 elements per round with 16-byte coalesced accesses; returns the total (valid in every thread)
__device__ u64 block_exclusive_scan_u32(u32 *data, u64 n) {
    __shared__ u32 s_wave[16];
    __shared__ u64 s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (u64 start = 0; start < n; start += 4096) {
        const u64 i = start + (u64)threadIdx.x * 4;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (i + 3 < n) {
            v = *reinterpret_cast<const uint4 *>(data + i);
        } else {
            if (i < n) v.x = data[i];
            if (i + 1 < n) v.y = data[i + 1];
            if (i + 2 < n) v.z = data[i + 2];
        }
        const u32 t = v.x + v.y + v.z + v.w;
        u32 incl = t;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const u32 o = __shfl_up(incl, s, 64);
            if (lane >= s) incl += o;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        u32 before = 0;
        for (int w = 0; w < wave; w++) before += s_wave[w];
        const u64 carry = s_carry;
        const u32 ex = (u32)carry + before + incl - t;  // positions are < 2^32
        const uint4 o = make_uint4(ex, ex + v.x, ex + v.x + v.y, ex + v.x + v.y + v.z);
        if (i + 3 < n) {
            *reinterpret_cast<uint4 *>(data + i) = o;
        } else {
            if (i < n) data[i] = o.x;
            if (i + 1 < n) data[i + 1] = o.y;
            if (i + 2 < n) data[i + 2] = o.z;
        }
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + before + incl;
        __syncthreads();
    }
    return s_carry;
}

// one block: exclusive scan of the unit counts (in place) + Strings.B length
__global__ __launch_bounds__(1024) void k_str_scan(S2Dev p) {
    const u64 total = block_exclusive_scan_u32(p.unit_cnt, p.units);
    if (threadIdx.x == 0) p.st->strings_len_masks = total;
}

__global__ __launch_bounds__(256) void k_str_emit(S2Dev p) {
    __shared__ u32 s_in[4][16][64];       // the wave's 64 chunks, dword-major (bank = lane)
    __shared__ u8 s_out[4][4096 + 16];    // the unit's unescaped bytes
    const u64 c = (u64)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (c >= p.units * 64) return;
    const u64 unit = c >> 6;
    const u64 em = p.em[c];
    const u32 pre = p.chunk_pre[c];
    const u32 n = (u32)popc64(em);
    const u32 total = (u32)__shfl((int)(pre + n), 63, 64);
    if (total == 0) return;  // wave-uniform
    if (em != 0) {  // the chunk holds message bytes: its 64-byte line is readable
        const uint4 *src = reinterpret_cast<const uint4 *>(p.sv.base + c * 64);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 v = src[q];
            s_in[wave][4 * q + 0][lane] = v.x;
            s_in[wave][4 * q + 1][lane] = v.y;
            s_in[wave][4 * q + 2][lane] = v.z;
            s_in[wave][4 * q + 3][lane] = v.w;
        }
        const u8 *in8 = reinterpret_cast<const u8 *>(&s_in[wave][0][0]);
        str_chunk_emit(p.sv, c, em, p.um[c], c ? p.um[c - 1] : 0ull, &s_out[wave][pre],
                       [&](u32 q) { return in8[((q >> 2) * 64 + lane) * 4 + (q & 3)]; });
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const u64 g = (u64)p.unit_cnt[unit];  // exclusive prefix: Strings.B offset of the unit
    if (g + total > p.strings_cap) return;
    u8 *dst = p.strings + g;
    const u32 words = total >> 2;
    for (u32 i = lane; i < words; i += 64)  // unaligned 4-byte global stores are fine on gfx950
        *reinterpret_cast<u32 *>(dst + 4 * i) = *reinterpret_cast<const u32 *>(&s_out[wave][4 * i]);
    const u32 tail = words * 4 + lane;
    if (tail < total) dst[tail] = s_out[wave][tail];
}

// ---- kernel 1: token kinds + string lengths (parseStringSimdValidateOnly) ---------------------------------
__global__ __launch_bounds__(256) void k_string_measure(S2Dev p) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    const u8 kind = token_kind(p.msg[p.pos[i]], p.ndjson != 0);
    p.kind[i] = kind;
    u32 out = 0;
    if (kind == K_STRING && p.sv.qm) {  // every string is copied: offset and length come from the emit masks
        const u64 a0 = (u64)p.pos[i] + p.sv.lead + 1;
        const u64 a1 = (i + 1 < p.n ? (u64)p.pos[i + 1] : p.len) + p.sv.lead;
        out = (u32)(emitted_before(p.unit_cnt, p.chunk_pre, p.em, a1) - emitted_before(p.unit_cnt, p.chunk_pre, p.em, a0)) |
              DLEN_COPY;
    } else if (kind == K_STRING) {
        const MsgView mv{p.msg, p.len};
        u32 sl, dl;
        if (!string_walk(mv, p.pos[i], nullptr, &sl, &dl)) {
            out = DLEN_INVALID;
            atomicOr(&p.st->err, 1u);
        } else {
            out = dl | ((p.copy_strings || sl != dl) ? DLEN_COPY : 0u);
        }
    }
    p.dlen[i] = out;
}

// ---- kernels 3-5: device-wide scan of (depth, tape words, string bytes, last bracket, newline runs) -----
__device__ __forceinline__ Agg block_reduce(Agg v, Agg *lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const Agg o = agg_shfl_up(v, s);
        if (lane >= s) v = agg_combine(o, v);
    }
    if (lane == 63) lds[wave] = v;
    __syncthreads();
    Agg tot = lds[0];
    for (int w = 1; w < S2_BLOCK / 64; w++) tot = agg_combine(tot, lds[w]);
    return tot;
}

__global__ __launch_bounds__(S2_BLOCK) void k_scan_reduce(S2Dev p) {
    __shared__ Agg lds[S2_BLOCK / 64];
    const u32 base = blockIdx.x * S2_TILE + threadIdx.x * S2_ITEMS;
    Agg v{0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < S2_ITEMS; k++)
        if (base + k < p.n) v = agg_combine(v, token_agg(p, base + k));
    const Agg tot = block_reduce(v, lds);
    if (threadIdx.x == 0) {
        p.agg_d[blockIdx.x] = tot.d;
        p.agg_w[blockIdx.x] = tot.w;
        p.agg_s[blockIdx.x] = tot.s;
        p.agg_lb[blockIdx.x] = tot.lb;
        p.agg_nb[blockIdx.x] = tot.nb;
        p.agg_bc[blockIdx.x] = tot.bc;
    }
}

// one block: exclusive scan over the tile aggregates (in place) + totals
__global__ __launch_bounds__(1024) void k_scan_tiles(S2Dev p) {
    __shared__ Agg lds[16];
    __shared__ Agg carry_s;
    __shared__ unsigned long long words64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        carry_s = Agg{0, 0, 0, 0, 0, 0};
        words64 = 0;
    }
    __syncthreads();
    for (u32 start = 0; start < p.tiles; start += 1024) {
        const u32 t = start + threadIdx.x;
        Agg incl{0, 0, 0, 0, 0};
        if (t < p.tiles) incl = Agg{p.agg_d[t], p.agg_w[t], p.agg_s[t], p.agg_lb[t], p.agg_nb[t], p.agg_bc[t]};
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {  // inclusive scan inside the wave
            const Agg o = agg_shfl_up(incl, s);
            if (lane >= s) incl = agg_combine(o, incl);
        }
        if (lane == 63) lds[wave] = incl;
        __syncthreads();
        Agg before = carry_s;  // everything in front of this wave
        for (int w = 0; w < wave; w++) before = agg_combine(before, lds[w]);
        const Agg prev = agg_shfl_up(incl, 1);
        const Agg excl = lane > 0 ? agg_combine(before, prev) : before;
        if (t < p.tiles) {
            p.agg_d[t] = excl.d;
            p.agg_w[t] = excl.w;
            p.agg_s[t] = excl.s;
            p.agg_lb[t] = excl.lb;
            p.agg_nb[t] = excl.nb;
            p.agg_bc[t] = excl.bc;
        }
        __syncthreads();
        if (threadIdx.x == 1023) {
            const Agg total = agg_combine(before, incl);
            words64 += (unsigned long long)(u32)(total.w - carry_s.w);  // chunk sum < 2^32
            carry_s = total;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const Agg tot = carry_s;
        p.st->final_depth = tot.d;
        p.st->tape_len = words64 + 2ull;  // + opening root + closing root
        p.st->strings_len = tot.s;
        p.st->records = tot.nb;
        p.st->n_br = tot.bc;
        if (words64 + 2ull > 0xfffffff0ull) atomicOr(&p.st->err, 4u);
        if (tot.d != 0) atomicOr(&p.st->err, 1u);  // scopes still open at the end (succeed: :433-435)
    }
}

__global__ __launch_bounds__(S2_BLOCK) void k_scan_apply(S2Dev p) {
    __shared__ Agg lds[S2_BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 base = blockIdx.x * S2_TILE + threadIdx.x * S2_ITEMS;
    Agg item[S2_ITEMS];
    Agg v{0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < S2_ITEMS; k++) {
        item[k] = (base + k < p.n) ? token_agg(p, base + k) : Agg{0, 0, 0, 0, 0, 0};
        v = agg_combine(v, item[k]);
    }
    Agg incl = v;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const Agg o = agg_shfl_up(incl, s);
        if (lane >= s) incl = agg_combine(o, incl);
    }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    Agg run{p.agg_d[blockIdx.x], p.agg_w[blockIdx.x], p.agg_s[blockIdx.x], p.agg_lb[blockIdx.x], p.agg_nb[blockIdx.x],
            p.agg_bc[blockIdx.x]};
    for (int w = 0; w < wave; w++) run = agg_combine(run, lds[w]);
    {
        const Agg prev = agg_shfl_up(incl, 1);
        if (lane > 0) run = agg_combine(run, prev);
    }
    // run = exclusive prefix for this thread's first token
#pragma unroll
    for (int k = 0; k < S2_ITEMS; k++) {
        const u32 i = base + k;
        if (i >= p.n) break;
        const Agg a = item[k];
        p.tape_off[i] = run.w + 1u;  // word 0 is the opening root (write_tape(0,'r'), :172)
        p.str_off[i] = run.s;
        if (a.nb) {
            p.nlb[run.nb] = i;
            p.match[i] = run.nb;
        }
        const u32 c = run.bc;  // brackets in front of this token
        run = agg_combine(run, a);
        p.depth[i] = run.d;
        p.last_br[i] = run.lb;
        if (a.bc) {
            p.br_tok[c] = i;
            p.br_depth[c] = run.d;
        }
    }
}

// ---- kernel 6: one level of the 64-ary min tree over br_depth[] (one wave per group) -----------------------
// The number of brackets is only known on the device: the launcher sizes grids and level arrays for the
// worst case (every token a bracket) and the kernels derive the real level sizes from S2State::n_br.
__device__ __forceinline__ MinTree make_tree(const S2Dev &p) {
    MinTree mt;
    mt.lev[0] = p.br_depth;
    u64 sz = p.st->n_br;
    mt.size[0] = sz;
    mt.nlev = 1;
    while (sz > 64 && mt.nlev < MinTree::MAXLEV) {
        sz = (sz + 63) / 64;
        mt.lev[mt.nlev] = p.lev[mt.nlev];
        mt.size[mt.nlev] = sz;
        mt.nlev++;
    }
    return mt;
}
__global__ __launch_bounds__(256) void k_min_level(S2Dev p, int l) {
    const MinTree mt = make_tree(p);
    if (l >= mt.nlev) return;
    const int lane = threadIdx.x & 63;
    for (u64 g = (u64)blockIdx.x * 4 + (threadIdx.x >> 6); g < mt.size[l]; g += (u64)gridDim.x * 4) {
        const u64 k = g * 64 + lane;
        i32 v = k < mt.size[l - 1] ? mt.lev[l - 1][k] : 0x7fffffff;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            const i32 o = __shfl_xor(v, s, 64);
            v = o < v ? o : v;
        }
        if (lane == 0) p.lev[l][g] = v;
    }
}

__device__ __forceinline__ Tokens make_tokens(const S2Dev &p) {
    Tokens t{p.pos, p.n, p.kind, p.depth, p.tape_off, p.str_off, p.last_br, p.match, p.ctxb};
    t.tape_base = p.tape_base;
    t.strings_base = p.strings_base;
    t.msg_base = p.msg_base;
    return t;
}

// ---- kernel 7: bracket partners and resume contexts -------------------------------------------------------
__global__ __launch_bounds__(256) void k_brackets(S2Dev p) {
    const u32 n_br = p.st->n_br;
    const MinTree mt = make_tree(p);
    for (u32 c = blockIdx.x * 256 + threadIdx.x; c < n_br; c += gridDim.x * 256)  // compact bracket index
        if (is_close(p.kind[p.br_tok[c]])) bracket_resolve_compact(mt, p.br_tok, p.kind, c, p.match, p.ctxb);
}

// ---- kernel 8: grammar check + tape words of brackets, atoms, numbers and roots ----------------------------
// Numbers are the expensive tokens (a byte loop and a 128-bit multiply) and only ~10 % of all tokens: every
// block first handles everything else and queues its number tokens in LDS, then parses them with the lanes
// packed densely, so that a wave of commas does not pay for the one number among them.
static constexpr int EMIT_BLOCK = 1024;
__global__ __launch_bounds__(EMIT_BLOCK) void k_emit(S2Dev p) {
    __shared__ u32 s_num[EMIT_BLOCK];
    __shared__ u32 s_nb[256][9];  // 32-byte windows, 36-byte stride (bank-conflict free)
    __shared__ u32 s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const u32 i = blockIdx.x * EMIT_BLOCK + threadIdx.x;
    const Tokens t = make_tokens(p);
    const MsgView mv{p.msg, p.len};
    const u64 tape_len = p.st->tape_len;
    if (tape_len > p.tape_cap) return;  // cannot happen: the launcher sizes the tape for 2n+2 words
    bool bad = false;
    if (i < p.n) {
        // round 1: everything that only depends on i, issued together (one memory round trip)
        const u8 k = p.kind[i];
        const u8 pk = i > 0 ? p.kind[i - 1] : (u8)K_BAD, ppk = i > 1 ? p.kind[i - 2] : (u8)K_BAD;
        const u8 nk = i + 1 < p.n ? p.kind[i + 1] : (u8)K_BAD;
        const u32 lb = i > 0 ? p.last_br[i - 1] : 0u;
        const u32 o = p.tape_off[i], m = p.match[i], dl = p.dlen[i], so = p.str_off[i];
        // round 2: the last bracket in front (context) and the partner of a bracket
        const u8 bk = lb ? p.kind[lb - 1] : (u8)K_BAD, bc = lb ? p.ctxb[lb - 1] : (u8)CTX_ROOT;
        const bool br = is_bracket(k) && m < p.n;
        const u32 mo = br ? p.tape_off[m] : 0u;
        bad = grammar_violation_v(i, k, pk, ppk, gap_ctx_v(lb, bk, bc));
        switch (k) {
        case K_OPEN_OBJ:
        case K_OPEN_ARR:  // payload: tape index just after the matching close (annotate_previousloc, :336)
            p.tape[o] = ((u64)(k == K_OPEN_OBJ ? '{' : '[') << 56) | (br ? p.tape_base + mo + 1 : 0ull);
            break;
        case K_CLOSE_OBJ:
        case K_CLOSE_ARR:  // payload: tape index of the matching open (:335)
            p.tape[o] = ((u64)(k == K_CLOSE_OBJ ? '}' : ']') << 56) | (br ? p.tape_base + mo : 0ull);
            break;
        case K_TRUE:
        case K_FALSE:
        case K_NULL:
            p.tape[o] = (u64)(k == K_TRUE ? 't' : (k == K_FALSE ? 'f' : 'n')) << 56;
            bad |= !atom_valid(mv, p.pos[i], k);
            break;
        case K_NUM: {  // wave-aggregated append: one LDS atomic per wave
            const u64 act = __ballot(1);
            const int lane = threadIdx.x & 63;
            const int leader = (int)__builtin_ctzll(act);
            u32 base = 0;
            if (lane == leader) base = atomicAdd(&s_cnt, (u32)__builtin_popcountll(act));
            base = (u32)__shfl((int)base, leader, 64);
            s_num[base + (u32)__builtin_popcountll(act & ((1ull << lane) - 1))] = i;
            break;
        }
        case K_STRING:
            if (p.sv.qm && dl != DLEN_INVALID) {  // copy mode: the bytes are written by k_str_emit, only the tape words here
                p.tape[o] = ((u64)'"' << 56) | (STRINGBUFBIT + p.strings_base + so);
                p.tape[o + 1] = dl & ~DLEN_COPY;
            }
            break;
        case K_NL:
            if (i + 1 < p.n && nk != K_NL)
                emit_root(p.nlb, p.st->records, p.tape_off, (u32)tape_len, m + 1, p.tape, p.tape_base);
            break;
        default: break;
        }
        if (i == 0) emit_root(p.nlb, p.st->records, p.tape_off, (u32)tape_len, 0, p.tape, p.tape_base);
    }
    __syncthreads();
    // the queued numbers, 256 at a time: the first 32 bytes of each go to LDS (two unaligned 16-byte loads instead
    // of one dependent byte load per digit); longer numbers fall back to the message itself
    const u32 cnt = s_cnt;
    for (u32 j0 = 0; j0 < cnt; j0 += 256) {
        const u32 j = j0 + threadIdx.x;
        if (threadIdx.x < 256 && j < cnt) {
            const u32 q = s_num[j];
            const u32 at = p.pos[q];
            const u64 rest = p.len - at;
            u32 *w = s_nb[threadIdx.x];
            if (rest >= 32) {
                const uint4 a = *reinterpret_cast<const uint4 *>(p.msg + at), b = *reinterpret_cast<const uint4 *>(p.msg + at + 16);
                w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
                w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
            } else {
                u8 *wb = reinterpret_cast<u8 *>(w);
                for (u32 k = 0; k < (u32)rest; k++) wb[k] = p.msg[at + k];
            }
            u64 tag = 0, val = 0;
            u32 numlen = 0;
            const u32 avail = rest < 32 ? (u32)rest : 32u;
            int st = parse_number(reinterpret_cast<const u8 *>(w), avail, &tag, &val, &numlen);
            if (numlen == 32 && rest > 32) st = parse_number(p.msg + at, (u32)rest, &tag, &val, &numlen);
            if (st == NUM_FAIL) {
                bad = true;
            } else {
                const u32 o = p.tape_off[q];
                p.tape[o] = tag;
                p.tape[o + 1] = val;
                if (st == NUM_NEEDS_BIGNUM) p.bigq[atomicAdd(&p.st->bignum_count, 1u)] = q;
            }
        }
    }
    if (bad) atomicOr(&p.st->err, 1u);
}

// ---- kernel 9: strings (tape words + unescaped copy into Strings.B) -----------------------------------------
__global__ __launch_bounds__(256) void k_emit_strings(S2Dev p) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    if (p.kind[i] != K_STRING) return;
    const u32 dl = p.dlen[i];
    if (dl == DLEN_INVALID) return;
    if ((u64)p.str_off[i] + (dl & ~DLEN_COPY) > p.strings_cap) return;
    const Tokens t = make_tokens(p);
    const MsgView mv{p.msg, p.len};
    emit_string(t, mv, i, (dl & DLEN_COPY) != 0, dl & ~DLEN_COPY, p.tape, p.sv.qm ? nullptr : p.strings);
}

// ---- kernel 10: exact tie-break for >19-digit mantissas whose neighbours disagree ------------------------------
__global__ __launch_bounds__(64) void k_bignum(S2Dev p) {
    const u32 cnt = p.st->bignum_count;
    Big X, Y;
    for (u32 q = blockIdx.x * 64 + threadIdx.x; q < cnt; q += gridDim.x * 64) {
        const u32 i = p.bigq[q];
        const u32 at = p.pos[i];
        u64 tag, val;
        u32 numlen = 0;
        (void)parse_number(p.msg + at, (u32)(p.len - at), &tag, &val, &numlen);
        const u32 o = p.tape_off[i];
        const u64 cand = p.tape[o + 1];
        const u64 sign = cand & 0x8000000000000000ull;
        const u64 r = bignum_round(p.msg + at, numlen, cand & ~0x8000000000000000ull, X, Y);
        if (r == 0x7ff0000000000000ull) atomicOr(&p.st->err, 1u);  // strconv.ErrRange
        p.tape[o + 1] = r | sign;
    }
}

// ---- launcher -----------------------------------------------------------------------------------------------
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

size_t stage2_workspace_bytes(size_t n) {
    size_t b = sizeof(S2State) + 256;
    b += align_up(n, 256) * 2;                      // kind, ctxb
    b += align_up(n * 4, 256) * 11;                 // dlen depth tape_off str_off last_br match nlb bigq br_tok br_depth (+1)
    const size_t tiles = (n + S2_TILE - 1) / S2_TILE + 1;
    b += align_up(tiles * 4, 256) * 6;
    size_t lv = n;
    for (int l = 1; l < MinTree::MAXLEV; l++) {
        lv = (lv + 63) / 64;
        b += align_up(lv * 4, 256);
    }
    return b + 4096;
}

// carve the device view out of the workspace (deterministic: both phases rebuild the same view)
static S2Dev stage2_view(const void *d_msg, size_t len, const u32 *d_pos, size_t n, u32 flags, void *ws, u64 *d_tape,
                         size_t tape_cap, u8 *d_strings, size_t strings_cap, void *str_aux) {
    S2Dev p;
    char *w = reinterpret_cast<char *>(ws);
    auto carve = [&](size_t bytes) {
        char *r = w;
        w += align_up(bytes, 256);
        return r;
    };
    p.st = reinterpret_cast<S2State *>(carve(sizeof(S2State)));
    p.msg = reinterpret_cast<const u8 *>(d_msg);
    p.len = len;
    p.pos = d_pos;
    p.n = (u32)n;
    p.ndjson = flags & 1u;
    p.copy_strings = (flags >> 1) & 1u;
    p.kind = reinterpret_cast<u8 *>(carve(n));
    p.ctxb = reinterpret_cast<u8 *>(carve(n));
    p.dlen = reinterpret_cast<u32 *>(carve(n * 4));
    p.depth = reinterpret_cast<i32 *>(carve(n * 4));
    p.tape_off = reinterpret_cast<u32 *>(carve(n * 4));
    p.str_off = reinterpret_cast<u32 *>(carve(n * 4));
    p.last_br = reinterpret_cast<u32 *>(carve(n * 4));
    p.match = reinterpret_cast<u32 *>(carve(n * 4));
    p.nlb = reinterpret_cast<u32 *>(carve(n * 4));
    p.bigq = reinterpret_cast<u32 *>(carve(n * 4));
    p.br_tok = reinterpret_cast<u32 *>(carve(n * 4));
    p.br_depth = reinterpret_cast<i32 *>(carve(n * 4));
    p.tiles = (u32)((n + S2_TILE - 1) / S2_TILE);
    p.agg_d = reinterpret_cast<i32 *>(carve((size_t)p.tiles * 4));
    p.agg_w = reinterpret_cast<u32 *>(carve((size_t)p.tiles * 4));
    p.agg_s = reinterpret_cast<u32 *>(carve((size_t)p.tiles * 4));
    p.agg_lb = reinterpret_cast<u32 *>(carve((size_t)p.tiles * 4));
    p.agg_nb = reinterpret_cast<u32 *>(carve((size_t)p.tiles * 4));
    p.agg_bc = reinterpret_cast<u32 *>(carve((size_t)p.tiles * 4));
    p.nlev = 1;
    p.lev[0] = nullptr;
    p.lev_size[0] = n;
    {
        u64 sz = n;
        while (sz > 64 && p.nlev < MinTree::MAXLEV) {
            sz = (sz + 63) / 64;
            p.lev[p.nlev] = reinterpret_cast<i32 *>(carve(sz * 4));
            p.lev_size[p.nlev] = sz;
            p.nlev++;
        }
    }
    p.tape = d_tape;
    p.strings = d_strings;
    p.tape_cap = tape_cap;
    p.strings_cap = strings_cap;
    p.tape_base = p.strings_base = p.msg_base = 0;
    // byte-parallel strings: only when every string is copied and stage 1 left its masks
    const uintptr_t addr = reinterpret_cast<uintptr_t>(d_msg);
    p.sv.base = reinterpret_cast<const u8 *>(addr & ~(uintptr_t)63);
    p.sv.lead = addr & 63;
    p.sv.end = p.sv.lead + len;
    p.sv.qm = p.sv.q = p.sv.st = nullptr;
    p.sv.unit_h = nullptr;
    p.em = p.um = nullptr;
    p.chunk_pre = nullptr;
    p.unit_cnt = nullptr;
    p.units = 0;
    if (str_aux && p.copy_strings) {
        const StrAux a = str_aux_layout(str_aux, (size_t)p.sv.end);
        p.sv.qm = a.qm;
        p.sv.q = a.q;
        p.sv.st = a.st;
        p.sv.unit_h = a.unit_h;
        p.em = a.em;
        p.um = a.um;
        p.chunk_pre = a.chunk_pre;
        p.unit_cnt = a.unit_cnt;
        p.units = (p.sv.end + 4095) / 4096;  // units that hold message bytes (stage 1 wrote their masks)
    }
    return p;
}

// Phase 1: token kinds, string lengths and the device-wide scan.  Afterwards S2State holds tape_len /
// strings_len of this message (what an NDJSON shard exchanges with the other shards) and every token
// knows its depth and its tape / Strings.B offsets.
hipError_t stage2_launch_measure(const void *d_msg, size_t len, const u32 *d_pos, size_t n, u32 flags, void *ws,
                                 hipStream_t stream, void *str_aux) {
    const S2Dev p = stage2_view(d_msg, len, d_pos, n, flags, ws, nullptr, 0, nullptr, 0, str_aux);
    hipError_t e = hipMemsetAsync(p.st, 0, sizeof(S2State), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(p.match, 0, n * 4, stream);
    if (e != hipSuccess) return e;
    const u32 gb = (u32)((n + 255) / 256);
    if (p.sv.qm) {
        hipLaunchKernelGGL(k_str_masks, dim3((u32)((p.units * 64 + 255) / 256)), dim3(256), 0, stream, p);
        hipLaunchKernelGGL(k_str_scan, dim3(1), dim3(1024), 0, stream, p);
    }
    hipLaunchKernelGGL(k_string_measure, dim3(gb), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(k_scan_reduce, dim3(p.tiles), dim3(S2_BLOCK), 0, stream, p);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, stream, p);
    hipLaunchKernelGGL(k_scan_apply, dim3(p.tiles), dim3(S2_BLOCK), 0, stream, p);
    return hipGetLastError();
}

// Phase 2: bracket matching, grammar check, tape and Strings.B.  The three bases rebase every index the
// tape stores (tape positions, Strings.B offsets, Message offsets): 0 for a whole message, the exclusive
// prefix sums over the preceding shards for an NDJSON shard.
hipError_t stage2_launch_emit(const void *d_msg, size_t len, const u32 *d_pos, size_t n, u32 flags, void *ws, u64 *d_tape,
                              size_t tape_cap, u8 *d_strings, size_t strings_cap, u64 tape_base, u64 strings_base,
                              u64 msg_base, hipStream_t stream, void *str_aux) {
    S2Dev p = stage2_view(d_msg, len, d_pos, n, flags, ws, d_tape, tape_cap, d_strings, strings_cap, str_aux);
    p.tape_base = tape_base;
    p.strings_base = strings_base;
    p.msg_base = msg_base;
    const u32 gb = (u32)((n + 255) / 256);
    for (int l = 1; l < p.nlev; l++) {  // grid-stride: the kernels use the real bracket count
        const u64 want = (p.lev_size[l] + 3) / 4;
        hipLaunchKernelGGL(k_min_level, dim3((u32)(want < 2048 ? want : 2048)), dim3(256), 0, stream, p, l);
    }
    hipLaunchKernelGGL(k_brackets, dim3(gb < 8192 ? gb : 8192), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(k_emit, dim3((u32)((n + EMIT_BLOCK - 1) / EMIT_BLOCK)), dim3(EMIT_BLOCK), 0, stream, p);
    if (!p.sv.qm) hipLaunchKernelGGL(k_emit_strings, dim3(gb), dim3(256), 0, stream, p);
    if (p.sv.qm) hipLaunchKernelGGL(k_str_emit, dim3((u32)((p.units * 64 + 255) / 256)), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(k_bignum, dim3(64), dim3(64), 0, stream, p);
    return hipGetLastError();
}

hipError_t stage2_launch(const void *d_msg, size_t len, const u32 *d_pos, size_t n, u32 flags, void *ws, u64 *d_tape,
                         size_t tape_cap, u8 *d_strings, size_t strings_cap, hipStream_t stream) {
    hipError_t e = stage2_launch_measure(d_msg, len, d_pos, n, flags, ws, stream, nullptr);
    if (e != hipSuccess) return e;
    return stage2_launch_emit(d_msg, len, d_pos, n, flags, ws, d_tape, tape_cap, d_strings, strings_cap, 0, 0, 0, stream,
                              nullptr);
}

}  // namespace sj
