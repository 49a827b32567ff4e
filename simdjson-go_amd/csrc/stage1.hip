// stage1.hip -- stage 1 (structural index) as ONE single-pass kernel for gfx950.
//
// Replaces the reference's 64-byte loop _find_structural_bits_in_slice
// (find_structural_bits_amd64.s:49-155) and its Go driver findStructuralIndices
// (stage1_find_marks_amd64.go:41-148).
//
// Shape: one 64-byte chunk per lane, BLOCK lanes per tile (BLOCK*64 contiguous bytes).  The
// three cross-chunk dependencies of the reference loop are resolved like this:
//   * odd-backslash carry  -- constant per chunk unless the chunk is all backslashes, so the
//                             predecessor's trailing-run parity is taken from the neighbour
//                             lane (wave shuffle) or read back from memory (wave-first lane);
//   * in-string parity     -- XOR scan: ballot inside the wave, LDS across waves, and a
//                             decoupled look-back chain over per-tile descriptors across
//                             tiles (so the input is fetched from HBM exactly once);
//   * output offset        -- + scan of per-chunk structural counts, same three levels.
// The pseudo-structural predecessor bit needs only the class of the previous byte (see
// DESIGN.md "pseudo_pred without the quote state").
// Output: ABSOLUTE uint32 byte positions (the running sum of the reference's deltas).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "sj_chunk.h"
#include "sj_device.h"

namespace sj {

// ---- decoupled look-back over one 64-bit descriptor per tile ---------------------------
// descriptor = status(2) << 62 | value(62); the value IS the payload (single 8-byte granule,
// relaxed agent-scope accesses: MI355X guide, Guideline 16 form R2).
static constexpr u64 ST_AGG = 1ull << 62, ST_PREFIX = 2ull << 62, VAL_MASK = (1ull << 62) - 1;

__device__ __forceinline__ u64 desc_load(const u64 *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void desc_store(u64 *p, u64 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool XOR>
__device__ __forceinline__ u64 wave_reduce(u64 v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        u64 o = __shfl_xor(v, s, 64);
        v = XOR ? (v ^ o) : (v + o);
    }
    return v;
}

// Called by all 64 lanes of one wave.  Returns the exclusive prefix of `agg` over tiles < t.
template <bool XOR>
__device__ __forceinline__ u64 lookback(u64 *desc, u32 t, u64 agg, int lane) {
    if (t == 0) {
        if (lane == 0) desc_store(&desc[0], ST_PREFIX | (agg & VAL_MASK));
        return 0;
    }
    if (lane == 0) desc_store(&desc[t], ST_AGG | (agg & VAL_MASK));
    u64 acc = 0;
    long long j = (long long)t - 1;
    for (;;) {
        const long long idx = j - lane;
        const u64 d = idx >= 0 ? desc_load(&desc[idx]) : ST_PREFIX;  // identity before tile 0
        const u32 status = (u32)(d >> 62);
        const u64 invalid = __ballot(status == 0);
        const u64 prefixes = __ballot(status == 2);
        const int fp = prefixes ? ctz64(prefixes) : 64;
        const u64 need = fp >= 63 ? ~0ull : ((2ull << fp) - 1);
        if (invalid & need) {
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        u64 v = (lane <= fp) ? (d & VAL_MASK) : 0;
        v = wave_reduce<XOR>(v);
        acc = XOR ? (acc ^ v) : (acc + v);
        if (fp < 64) break;
        j -= 64;
    }
    if (lane == 0) desc_store(&desc[t], ST_PREFIX | ((XOR ? (acc ^ agg) : (acc + agg)) & VAL_MASK));
    return acc;
}

// ---- memory peeks for the first lane of a wave -----------------------------------------
// Parity of the backslash run that ends right before byte `p` (p > 0), i.e. the reference's
// prev_iter_ends_odd_backslash at a chunk boundary.
__device__ __noinline__ u32 peek_backslash_parity(const u8 *base, u64 lead, u64 p) {
    u32 n = 0;
    while (p > lead && base[p - 1] == '\\') {
        n++;
        p--;
    }
    return n & 1u;
}

// pseudo_pred carry-in from the previous byte only: whitespace, one of {}[]:, or
// an unescaped quote.
__device__ __noinline__ u32 peek_pseudo_pred(const u8 *base, u64 lead, u64 p) {
    if (p <= lead) return 1;  // stage1_find_marks_amd64.go:56: starts as 1
    const u8 b = base[p - 1];
    if (b == ' ' || b == '\t' || b == '\n' || b == '\r') return 1;
    if (b == '{' || b == '}' || b == '[' || b == ']' || b == ':' || b == ',') return 1;
    if (b == '"') return peek_backslash_parity(base, lead, p - 1) ^ 1u;
    return 0;
}

// ---- chunk load ------------------------------------------------------------------------
// `base` is 64-byte aligned; the message occupies [lead, lead+len) of it.  Bytes outside are
// replaced by 0x20, exactly like the reference's space-masked tail
// (find_structural_bits_amd64.s:134-155); leading pad bytes are whitespace as well, which
// leaves the initial pseudo_pred (=1) semantics untouched.
__device__ __noinline__ void load_chunk_edge(const u8 *base, u64 off, u64 lead, u64 end, u32 *w) {
    for (int j = 0; j < 16; j++) {
        u32 v = 0;
        for (int b = 0; b < 4; b++) {
            const u64 g = off + 4u * j + b;
            const u32 byte = (g >= lead && g < end) ? base[g] : 0x20u;
            v |= byte << (8 * b);
        }
        w[j] = v;
    }
}

__device__ __forceinline__ void load_chunk(const u8 *base, u64 off, u64 lead, u64 end, u32 (&w)[16]) {
    const bool interior = off >= lead && off + 64 <= end;
    if (__ballot(!interior) == 0) {  // wave-uniform fast path: 4 x global_load_dwordx4
        const uint4 *p = reinterpret_cast<const uint4 *>(base + off);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint4 v = p[k];
            w[4 * k + 0] = v.x;
            w[4 * k + 1] = v.y;
            w[4 * k + 2] = v.z;
            w[4 * k + 3] = v.w;
        }
    } else {
        u32 tmp[16];
        load_chunk_edge(base, off, lead, end, tmp);
#pragma unroll
        for (int j = 0; j < 16; j++) w[j] = tmp[j];
    }
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void stage1_kernel(const u8 *__restrict__ base, u64 lead, u64 len,
                                                       u32 ndjson, u32 *__restrict__ out_pos, u64 pos_cap,
                                                       Stage1State *__restrict__ st, u64 *__restrict__ desc_par,
                                                       u64 *__restrict__ desc_cnt, u32 num_tiles) {
    constexpr int WAVES = BLOCK / 64;
    __shared__ u32 s_tile;
    __shared__ u32 s_wave_par[WAVES];
    __shared__ u32 s_wave_cnt[WAVES];
    __shared__ u64 s_excl_par;
    __shared__ u64 s_excl_cnt;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // dynamic tile id: tiles are started in id order, so every predecessor in the look-back
    // chain is resident or finished (forward progress without any dispatch-order assumption).
    if (tid == 0) s_tile = atomicAdd(&st->tile_counter, 1u);
    __syncthreads();
    const u32 t = s_tile;

    const u64 end = lead + len;
    const u64 off = ((u64)t * BLOCK + tid) * 64;  // byte offset of this lane's chunk in `base`

    u32 w[16];
    load_chunk(base, off, lead, end, w);
    const Classes c = classify(w);

    // ---- backslash carry ---------------------------------------------------------------
    // parity of the run of backslashes at the END of this chunk: if the chunk is not all
    // backslashes this is the carry into the next chunk whatever our own carry-in is
    // (an all-backslash chunk passes its carry-in through: 64 is even).
    const bool all_bs = c.bs == ~0ull;
    const u32 trail_odd = all_bs ? 0u : ((u32)__builtin_clzll(~c.bs) & 1u);
    u32 carry_in = __shfl_up(trail_odd, 1, 64);
    const bool wave_has_all_bs = __ballot(all_bs) != 0;
    if (off == 0) carry_in = 0;
    else if (lane == 0 || wave_has_all_bs) carry_in = peek_backslash_parity(base, lead, off);
    u32 carry_out;
    const u64 odd_ends = odd_backslash_ends(c.bs, carry_in, carry_out);
    const u64 quote_bits = c.quote & ~odd_ends;

    // ---- in-string parity: wave (ballot) -> tile (LDS) -> global (look-back) --------------
    const u32 par = (u32)popc64(quote_bits) & 1u;
    const u64 par_ballot = __ballot(par != 0);
    const u64 lanes_below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const u32 par_in_wave = (u32)popc64(par_ballot & lanes_below) & 1u;
    if (lane == 0) s_wave_par[wave] = (u32)popc64(par_ballot) & 1u;
    u64 quote_mask = prefix_xor(quote_bits);
    __syncthreads();
    u32 par_before_wave = 0, tile_par = 0;
#pragma unroll
    for (int i = 0; i < WAVES; i++) {
        const u32 p = s_wave_par[i];
        tile_par ^= p;
        if (i < wave) par_before_wave ^= p;
    }
    if (wave == 0) {
        const u64 ex = lookback<true>(desc_par, t, tile_par, lane);
        if (lane == 0) s_excl_par = ex;
    }
    __syncthreads();
    const u32 g_par = (u32)s_excl_par & 1u;
    if ((g_par ^ par_before_wave ^ par_in_wave) & 1u) quote_mask = ~quote_mask;

    // unescaped control characters inside strings (find_quote_mask_and_bits_amd64.s:67-80)
    const bool err = (c.ctrl & quote_mask) != 0;
    if (__ballot(err) != 0 && lane == 0) atomicOr(&st->error, 1u);

    // ---- pseudo-structural predecessor -------------------------------------------------
    const u32 pp_out = (u32)(((c.structs | quote_bits | c.ws) >> 63) & 1u);
    u32 pp_in = __shfl_up(pp_out, 1, 64);
    if (lane == 0) pp_in = peek_pseudo_pred(base, lead, off);

    u64 s = finalize(c.structs, c.ws, quote_mask, quote_bits, pp_in);
    if (ndjson) s |= c.nl & ~quote_mask;

    // ---- output offsets ----------------------------------------------------------------
    const u32 n = (u32)popc64(s);
    u32 incl = n;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_wave_cnt[wave] = incl;
    __syncthreads();
    u32 cnt_before_wave = 0, tile_cnt = 0;
#pragma unroll
    for (int i = 0; i < WAVES; i++) {
        const u32 v = s_wave_cnt[i];
        tile_cnt += v;
        if (i < wave) cnt_before_wave += v;
    }
    if (wave == 0) {
        const u64 ex = lookback<false>(desc_cnt, t, tile_cnt, lane);
        if (lane == 0) s_excl_cnt = ex;
    }
    __syncthreads();
    u64 o = s_excl_cnt + cnt_before_wave + (incl - n);

    // ---- flatten (flatten_bits_amd64.s:26-60, absolute positions instead of deltas) -------
    const u32 pos0 = (u32)(off - lead);
    while (s) {
        const int b = ctz64(s);
        if (o < pos_cap) out_pos[o] = pos0 + (u32)b;
        o++;
        s &= s - 1;
    }

    if (t == num_tiles - 1 && tid == BLOCK - 1) {
        st->total = s_excl_cnt + tile_cnt;
        st->ends_in_quote = (g_par ^ tile_par) & 1u;
    }
}

// =============================================================================================
// v2: one combined look-back per tile.  Every chunk is finalized under BOTH hypotheses about the
// in-string state at the start of its wave-unit (the work is lane-local and cheap next to the
// transposition), so a tile can publish (parity, count|outside, count|inside) before it knows its
// own incoming parity, and a single look-back chain resolves parity and output offset together.
// A tile is BLOCK lanes x CH chunks (CH passes of BLOCK*64 contiguous bytes).
// =============================================================================================
// descriptor:  AGG    = 1<<62 | P<<61 | T1<<28 | T0          (T0/T1: 28 bits each)
//              PREFIX = 2<<62 | G_end<<61 | COUNT_end        (48 bits)
__device__ __forceinline__ u64 pack_agg(u32 P, u32 T0, u32 T1) {
    return ST_AGG | ((u64)(P & 1u) << 61) | ((u64)T1 << 28) | (u64)T0;
}
__device__ __forceinline__ u64 pack_prefix(u32 G, u64 count) { return ST_PREFIX | ((u64)(G & 1u) << 61) | count; }

__device__ __forceinline__ u32 wave_sum_u32(u32 v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
}

// all 64 lanes of one wave; returns (G, BASE) = in-string parity and structural count before tile t
__device__ __forceinline__ void lookback2(u64 *desc, u32 t, u32 P, u32 T0, u32 T1, int lane, u32 &G_out, u64 &BASE_out) {
    if (t == 0) {
        if (lane == 0) desc_store(&desc[0], pack_prefix(P, T0));
        G_out = 0;
        BASE_out = 0;
        return;
    }
    if (lane == 0) desc_store(&desc[t], pack_agg(P, T0, T1));
    // F = effect of the already-composed tiles (nearer to t): state(g) -> (g ^ Fp, + Ft[g])
    u32 Fp = 0;
    u64 Ft0 = 0, Ft1 = 0;
    long long j = (long long)t - 1;
    u32 G = 0;
    u64 BASE = 0;
    for (;;) {
        const long long idx = j - lane;
        const u64 d = idx >= 0 ? desc_load(&desc[idx]) : pack_prefix(0, 0);  // virtual prefix before tile 0
        const u32 status = (u32)(d >> 62);
        const u64 invalid = __ballot(status == 0);
        const u64 prefixes = __ballot(status == 2);
        const int fp = prefixes ? ctz64(prefixes) : 64;
        const u64 need = fp >= 63 ? ~0ull : ((2ull << fp) - 1);
        if (invalid & need) {
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        const bool isagg = lane < fp;  // lanes nearer than the first prefix hold aggregates
        const u32 p_l = isagg ? (u32)((d >> 61) & 1u) : 0u;
        const u32 t0_l = (u32)(d & 0x0fffffffu), t1_l = (u32)((d >> 28) & 0x0fffffffu);
        const u64 pb = __ballot(p_l != 0);
        // parity contributed by the aggregates between this lane and the far end of the window
        const u64 above = lane >= 63 ? 0ull : (~0ull << (lane + 1));
        const u32 par_above = (u32)popc64(pb & above) & 1u;
        const u32 PW = (u32)popc64(pb) & 1u;
        if (fp < 64) {
            const u64 dp = __shfl(d, fp, 64);
            const u32 Gfar = (u32)((dp >> 61) & 1u);
            const u64 Cfar = dp & 0x0000ffffffffffffull;
            const u32 gb = Gfar ^ par_above;
            const u32 sum = wave_sum_u32(isagg ? (gb ? t1_l : t0_l) : 0u);
            const u32 gw = Gfar ^ PW;  // state right before the already-composed part
            G = gw ^ Fp;
            BASE = Cfar + sum + (gw ? Ft1 : Ft0);
            break;
        }
        // 64 aggregates, no prefix: compose the window as a function of the unknown far state
        const u32 TW0 = wave_sum_u32(par_above ? t1_l : t0_l);
        const u32 TW1 = wave_sum_u32(par_above ? t0_l : t1_l);
        const u64 n0 = TW0 + (PW ? Ft1 : Ft0);
        const u64 n1 = TW1 + (PW ? Ft0 : Ft1);
        Ft0 = n0;
        Ft1 = n1;
        Fp ^= PW;
        j -= 64;
    }
    if (lane == 0) desc_store(&desc[t], pack_prefix(G ^ P, BASE + (G ? T1 : T0)));
    G_out = G;
    BASE_out = BASE;
}

template <int BLOCK, int CH>
__global__ __launch_bounds__(BLOCK) void stage1_kernel_v2(const u8 *__restrict__ base, u64 lead, u64 len, u32 ndjson,
                                                          u32 *__restrict__ out_pos, u64 pos_cap,
                                                          Stage1State *__restrict__ st, u64 *__restrict__ desc,
                                                          u32 num_tiles) {
    constexpr int WAVES = BLOCK / 64;
    constexpr int UNITS = WAVES * CH;  // unit u = pass * WAVES + wave, in byte order
    __shared__ u32 s_tile;
    __shared__ u32 s_par[UNITS];
    __shared__ u32 s_cnt[2][UNITS];
    __shared__ u32 s_G;
    __shared__ u64 s_BASE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(&st->tile_counter, 1u);
    __syncthreads();
    const u32 t = s_tile;
    const u64 end = lead + len;
    const u64 tile_off = (u64)t * (BLOCK * CH) * 64;

    u64 sA[CH], sB[CH];  // final structural masks if the wave-unit starts outside / inside a string
    u32 ex[CH];          // exclusive in-wave offsets: exA | exB << 16
    u32 eflags = 0;      // bit 2k: control char in string under A, bit 2k+1: under B

#pragma unroll
    for (int k = 0; k < CH; k++) {
        const u64 off = tile_off + ((u64)k * BLOCK + tid) * 64;
        u32 w[16];
        load_chunk(base, off, lead, end, w);
        const Classes c = classify(w);

        const bool all_bs = c.bs == ~0ull;
        const u32 trail_odd = all_bs ? 0u : ((u32)__builtin_clzll(~c.bs) & 1u);
        u32 carry_in = __shfl_up(trail_odd, 1, 64);
        const bool wave_has_all_bs = __ballot(all_bs) != 0;
        if (off == 0) carry_in = 0;
        else if (lane == 0 || wave_has_all_bs) carry_in = peek_backslash_parity(base, lead, off);
        u64 odd_ends = 0;
        if (__ballot(c.bs != 0 || carry_in != 0) != 0) {  // wave-uniform: most waves see no backslash at all
            u32 carry_out;
            odd_ends = odd_backslash_ends(c.bs, carry_in, carry_out);
        }
        const u64 quote_bits = c.quote & ~odd_ends;

        const u32 par = (u32)popc64(quote_bits) & 1u;
        const u64 par_ballot = __ballot(par != 0);
        const u64 lanes_below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        const bool in_wave = (popc64(par_ballot & lanes_below) & 1) != 0;
        u64 qm = prefix_xor(quote_bits);  // relative to the start of this wave-unit
        if (in_wave) qm = ~qm;

        const u32 pp_out = (u32)(((c.structs | quote_bits | c.ws) >> 63) & 1u);
        u32 pp_in = __shfl_up(pp_out, 1, 64);
        if (lane == 0) pp_in = peek_pseudo_pred(base, lead, off);

        u64 a = finalize(c.structs, c.ws, qm, quote_bits, pp_in);
        u64 b = finalize(c.structs, c.ws, ~qm, quote_bits, pp_in);
        if (ndjson) {
            a |= c.nl & ~qm;
            b |= c.nl & qm;
        }
        sA[k] = a;
        sB[k] = b;
        if (c.ctrl & qm) eflags |= 1u << (2 * k);
        if (c.ctrl & ~qm) eflags |= 2u << (2 * k);

        // in-wave inclusive scans of both counts at once (16-bit fields: a wave holds <= 4096 bits)
        const u32 n2 = (u32)popc64(a) | ((u32)popc64(b) << 16);
        u32 incl = n2;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u32 o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        ex[k] = incl - n2;
        if (lane == 63) {
            const int u = k * WAVES + wave;
            s_par[u] = (u32)popc64(par_ballot) & 1u;
            s_cnt[0][u] = incl & 0xffffu;
            s_cnt[1][u] = incl >> 16;
        }
    }
    __syncthreads();

    // per-unit parity prefix (relative to the tile start) and the tile aggregate for both incoming states
    u32 pre_mask = 0;  // bit u = parity of units < u
    u32 P = 0, T0 = 0, T1 = 0;
#pragma unroll
    for (int u = 0; u < UNITS; u++) {
        pre_mask |= P << u;
        T0 += s_cnt[P][u];
        T1 += s_cnt[P ^ 1u][u];
        P ^= s_par[u];
    }
    if (wave == 0) {
        u32 G;
        u64 BASE;
        lookback2(desc, t, P, T0, T1, lane, G, BASE);
        if (lane == 0) {
            s_G = G;
            s_BASE = BASE;
        }
    }
    __syncthreads();
    const u32 G = s_G;
    u64 unit_base = s_BASE;

    // ---- flatten (flatten_bits_amd64.s:26-60, absolute positions instead of deltas) -------
    bool err = false;
#pragma unroll
    for (int k = 0; k < CH; k++) {
#pragma unroll
        for (int wv = 0; wv < WAVES; wv++) {
            const int u = k * WAVES + wv;
            const u32 h = G ^ ((pre_mask >> u) & 1u);
            if (wv == wave) {
                u64 s = h ? sB[k] : sA[k];
                u64 o = unit_base + (h ? (ex[k] >> 16) : (ex[k] & 0xffffu));
                err |= ((eflags >> (2 * k + h)) & 1u) != 0;
                const u32 pos0 = (u32)(tile_off + ((u64)k * BLOCK + tid) * 64 - lead);
                while (s) {
                    const int bit = ctz64(s);
                    if (o < pos_cap) out_pos[o] = pos0 + (u32)bit;
                    o++;
                    s &= s - 1;
                }
            }
            unit_base += s_cnt[h][u];
        }
    }
    if (__ballot(err) != 0 && lane == 0) atomicOr(&st->error, 1u);
    if (t == num_tiles - 1 && tid == 0) {
        st->total = unit_base;
        st->ends_in_quote = (G ^ P) & 1u;
    }
}

// ---- launcher --------------------------------------------------------------------------
// Variant selection (A/B on hardware): SJHIP_S1_VARIANT = 0 (v1: 512 lanes, two look-backs),
// 1..4 = v2 with (BLOCK, CH) = (256,1) (256,2) (256,4) (512,2).  Default: S1_DEFAULT_VARIANT.
static constexpr int S1_DEFAULT_VARIANT = 3;

struct S1Variant {
    int block, ch;
};
static S1Variant s1_variant() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("SJHIP_S1_VARIANT");
        v = e ? atoi(e) : S1_DEFAULT_VARIANT;
        if (v < 0 || v > 4) v = S1_DEFAULT_VARIANT;
    }
    switch (v) {
    case 0: return {512, 0};
    case 1: return {256, 1};
    case 2: return {256, 2};
    case 3: return {256, 4};
    default: return {512, 2};
    }
}

static inline u32 stage1_tiles(size_t len, size_t lead) {
    const S1Variant v = s1_variant();
    const u64 tile_bytes = (u64)v.block * (v.ch ? v.ch : 1) * 64;
    const u64 span = (u64)lead + len;
    return (u32)((span + tile_bytes - 1) / tile_bytes);
}

size_t stage1_workspace_bytes(size_t len) {
    const size_t tiles = (len + 128) / (256 * 64) + 2;  // smallest tile of any variant
    return sizeof(Stage1State) + 2 * tiles * sizeof(u64);
}

// zero the Stage1State and the tile descriptors (must precede every launch)
hipError_t stage1_prepare(size_t len, size_t lead, void *ws, hipStream_t stream) {
    const u32 tiles = stage1_tiles(len, lead);
    return hipMemsetAsync(ws, 0, sizeof(Stage1State) + 2 * (size_t)tiles * sizeof(u64), stream);
}

// d_msg may be any device pointer; ws must hold stage1_workspace_bytes(len + 64) and be prepared.
hipError_t stage1_launch_prepared(const void *d_msg, size_t len, int ndjson, u32 *d_pos, size_t pos_cap, void *ws,
                                  hipStream_t stream) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(d_msg);
    const u8 *base = reinterpret_cast<const u8 *>(a & ~(uintptr_t)63);
    const u64 lead = a & 63;
    const u32 tiles = stage1_tiles(len, lead);
    Stage1State *st = reinterpret_cast<Stage1State *>(ws);
    u64 *desc_par = reinterpret_cast<u64 *>(st + 1);
    u64 *desc_cnt = desc_par + tiles;
    if (tiles == 0) return hipSuccess;
    const S1Variant v = s1_variant();
    const u32 nd = (u32)(ndjson != 0);
#define S1_V2(B, C)                                                                                              \
    hipLaunchKernelGGL((stage1_kernel_v2<B, C>), dim3(tiles), dim3(B), 0, stream, base, lead, (u64)len, nd, d_pos, \
                       (u64)pos_cap, st, desc_par, tiles)
    if (v.ch == 0)
        hipLaunchKernelGGL(stage1_kernel<512>, dim3(tiles), dim3(512), 0, stream, base, lead, (u64)len, nd, d_pos,
                           (u64)pos_cap, st, desc_par, desc_cnt, tiles);
    else if (v.block == 256 && v.ch == 1) S1_V2(256, 1);
    else if (v.block == 256 && v.ch == 2) S1_V2(256, 2);
    else if (v.block == 256 && v.ch == 4) S1_V2(256, 4);
    else S1_V2(512, 2);
#undef S1_V2
    return hipGetLastError();
}

hipError_t stage1_launch(const void *d_msg, size_t len, int ndjson, u32 *d_pos, size_t pos_cap, void *ws,
                         hipStream_t stream) {
    hipError_t e = stage1_prepare(len, reinterpret_cast<uintptr_t>(d_msg) & 63, ws, stream);
    if (e != hipSuccess) return e;
    return stage1_launch_prepared(d_msg, len, ndjson, d_pos, pos_cap, ws, stream);
}

}  // namespace sj
