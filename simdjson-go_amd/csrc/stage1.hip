// stage1.hip -- stage 1 (structural index) as ONE single-pass kernel for gfx950.
//
// Replaces the reference's 64-byte loop _find_structural_bits_in_slice
// (find_structural_bits_amd64.s:49-155) and its Go driver findStructuralIndices
// (stage1_find_marks_amd64.go:41-148).
//
// Shape: one 64-byte chunk per lane and pass, BLOCK lanes, CH passes per tile
// (BLOCK*CH*64 contiguous bytes).  The cross-chunk dependencies of the reference loop:
//   * odd-backslash carry  -- constant per chunk unless the chunk is all backslashes, so the
//                             predecessor's trailing-run parity comes from the neighbour lane
//                             (DPP wave shift); the first lane of a wave derives it from the 8
//                             bytes in front of the wave's 4 KiB unit (one scalar load);
//   * pseudo_pred carry    -- needs only the class of the previous byte, same route;
//   * in-string parity and output offset -- every chunk is finalized under BOTH hypotheses
//     about the in-string state at the start of its wave unit (lane-local, cheap next to the
//     bit-plane transposition), so a tile can publish (parity, count|outside, count|inside)
//     before it knows its own incoming state.  One decoupled look-back over 64-bit tile
//     descriptors then resolves parity and offset together; it is done by wave 0 of the block
//     over a window of 256 descriptors (4 per lane), so even with one tile per CU in flight it
//     finishes in one or two memory round trips.
// The input is fetched from HBM exactly once.
// Output: ABSOLUTE uint32 byte positions (the running sum of the reference's deltas).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "sj_bounds.h"
#include "sj_chunk.h"
#include "sj_device.h"
#include "sj_stage2.h"
#include "sj_strings.h"

namespace sj {

// optional extra outputs for the whole parse (all null for stage 1 alone): per 64-byte chunk the in-string mask
// relative to the unit start, the unescaped quotes and the escape starters; per 4 KiB unit the resolved state
struct S1Aux {
    // (Arr: sj_bounds.h -- plain pointers in the product build, bounds-checked views under -DSJ_DEBUG_BOUNDS)
    Arr<u64> qm, st;     // null unless the strings are handled byte-parallel (stage2.hip); the unescaped quotes are not written:
                         // they are where qm changes (sj_strings.h StrView::quotes)
    Arr<u8> unit_h;
    Arr<u64> unit_slow;  // per unit: chunks with an escaped character that no simple escape names (sj_strings.h)
    Arr<u8> kind;        // [pos_cap] kind of every structural (sj_stage2.h), written next to its position
    // round 5: no records.  Phase A counts the emitted bytes (the fast formula of sj_strings.h: in-string bytes that are
    // neither quotes nor escape starters; units with a \u escape are counted again by k_measure) and the opening quotes of
    // every unit under BOTH hypotheses about the state at its start (four popcounts and two wave sums), the flatten -- the
    // first place that knows the state -- picks the pair that applies and leaves unit_cnt / unit_str; k_str_emit derives the
    // emit mask of a chunk from the three masks it streams anyway (stage2.hip).  (The first half of the round let the
    // flatten read the masks back and write 16-byte records: 24 B written + 24 read + 16 written per chunk instead of 24
    // written; the reads came back through the fabric, not from the L2.)
    Arr<u32> unit_cnt;   // [units] emitted bytes of the unit
    Arr<u32> unit_str;   // [units] strings that begin in the unit (opening quotes)
    Arr<u32> tile_unit;  // [units + 1] the unit that holds token 4096 T: the positions are 32 bits wide and wrap in a message of
                         // more than 4 GiB -- the token kernels rebuild a tile's positions from the unit its first token lies in
    u64 *trace;        // TRACE builds only: TRACE_WORDS s_memtime stamps per (tile, wave)
    unsigned long long *host;  // pinned host memory or null: the host record (sj_device.h S1_HOST_*: three words)
    u64 *clean_desc;   // the descriptor set of the previous launch on this workspace and how many of them it used: zeroed by this
    u32 clean_tiles;   // launch for the next one (sj_device.h Stage1State; no ordering needed: nobody reads them any more)
    u32 par;           // epoch & 1: the control slot and the descriptor set of this launch
    bool want_flag;    // leave Stage1State::has_starter (WithCopyStrings(false): stage2.hip no_escapes)
    uint4 *zero2;      // null, or a second region the launch's last block zeroes: the stage-2 state and scan slots of THIS parse
    u64 zero2_quads;   // (every stage-2 kernel of the previous parse is through when this launch runs, none of this one has begun)
    u32 exp;           // SJ_EXP builds only: parts to leave out (A/B timing; results are wrong)
};
#if defined(SJ_DEBUG_BOUNDS)
#define S1_POS_VIEW(out_pos, cap) Arr<u32>((out_pos), (cap), A_S1_POS)
#else
#define S1_POS_VIEW(out_pos, cap) (out_pos)
#endif
#if defined(SJ_EXP)
#define SJ_S1EXP(a, b) ((((a).exp >> (b)) & 1u) != 0)
#else
#define SJ_S1EXP(a, b) false
#endif
// Start of a block: its slice of the cleaning this launch does for the next one (sj_device.h Stage1State) and of the stage-2
// state of this parse.  Order-free: the descriptors and the control slot belong to a launch that is over, the stage-2 region to
// kernels that have not begun.  (The first half of round 6 let "the last block" clean up behind a completion counter: an
// atomic per block and two dependent round trips at the very end of the kernel, where nothing hides them -- 256 MiB 0.0935 ->
// 0.0955 ms, profiles/r06_stage1_no_prepare_ab.txt -- and counting the blocks in front of their last flatten did not hide them either.)
__device__ __forceinline__ void launch_clean(Stage1State *st, const S1Aux &aux) {
    const u64 thr = (u64)blockIdx.x * blockDim.x + threadIdx.x, nthr = (u64)gridDim.x * blockDim.x;
    for (u64 i = thr; i < aux.clean_tiles; i += nthr) aux.clean_desc[i] = 0;
    for (u64 i = thr; i < aux.zero2_quads; i += nthr) aux.zero2[i] = make_uint4(0u, 0u, 0u, 0u);
    if (thr == 0) *reinterpret_cast<uint4 *>(&st->c[aux.par ^ 1u]) = make_uint4(0u, 0u, 0u, 0u);
}
// an error of the launch: control word (device) and host record (a store of a constant: any number of blocks may do it)
__device__ __forceinline__ void report_error(Stage1Ctrl *ctl, const S1Aux &aux, u32 bit) {
    atomicOr(&ctl->error, bit);
    if (aux.host) __hip_atomic_store(&aux.host[bit == 1u ? 1 : 2], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the block that flattens the last tile leaves the count: device word (stage 2 reads it) and word 0 of the host record
__device__ __forceinline__ void report_total(Stage1State *st, const S1Aux &aux, u64 total, u32 ends_in_quote, const u8 *msg, u64 len) {
    __hip_atomic_store(&st->total, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (aux.host) {
        const u64 word = S1_HOST_VALID | (ends_in_quote ? S1_HOST_IN_QUOTE : 0) | ((u64)(len ? msg[len - 1] : 0u) << S1_HOST_LAST_SHIFT) |
                         (total & S1_HOST_TOTAL_MASK);
        __hip_atomic_store(&aux.host[0], (unsigned long long)word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// per-phase timeline of a tile for every wave (profiling builds of the kernels, sjhip_stage1_trace):
//   0 phase A begins   1 phase A done (arrival)   2 serial section done (the wave that ran it; 0 otherwise)
//   3 state of the tile known (past the second barrier / the result flag)   4 flatten done   5 HW_ID
static constexpr int TRACE_WORDS = 8;

template <bool TRACE>
__device__ __forceinline__ void trace_put(u64 *trace, u32 tile, int waves, int wave, int lane, int k) {
    if (TRACE && lane == 0) trace[((u64)tile * waves + wave) * TRACE_WORDS + k] = __builtin_readcyclecounter();
}

// ---- tiles ---------------------------------------------------------------------------------
// A tile is UNITS = WAVES * CH wave units of 4 KiB.  A document rarely is a whole number of rounds of (blocks x tiles):
// configs[1] is 2053 tiles for 256 blocks, 8.02 rounds -- five blocks would run a ninth tile while 251 idle.  So only
// the whole rounds use full tiles; what is left is cut into at most one tile per block of `su` units each (the other
// units of such a tile are void: no loads, no classification), and a document smaller than one round is spread over
// as many blocks as it has units.  Tiles are still numbered, drawn and chained in byte order.
struct TileMap {
    u32 nf;  // tiles below nf hold UNITS units each
    u32 su;  // the tiles from nf on hold su units each (1 <= su <= UNITS)
    u32 nu;  // units of the message: ceil((lead + len) / 4096); a tile's units from nu on are void
};
static constexpr u64 VOID_UNIT = 1ull << 40;
template <int UNITS>
__device__ __forceinline__ u64 tile_unit(TileMap tm, u32 t, int local) {  // wave-uniform arguments
    const u64 u = t < tm.nf ? (u64)t * UNITS + (u64)local
                            : ((u32)local < tm.su ? (u64)tm.nf * UNITS + (u64)(t - tm.nf) * tm.su + (u64)local : VOID_UNIT);
    return u < tm.nu ? u : VOID_UNIT;
}

// ---- tile descriptors ------------------------------------------------------------------
// One naturally aligned 8-byte granule per tile, status and payload together, relaxed
// agent-scope accesses (MI355X guide, Guideline 16 form R2):
//   AGG    = 1<<62 | P<<61 | T1<<28 | T0     P: parity of unescaped quotes in the tile,
//                                            T0/T1: structural count if the tile starts
//                                            outside / inside a string (28 bits each)
//   PREFIX = 2<<62 | G<<61 | COUNT           state and count at the END of the tile (48 bits)
static constexpr u64 ST_AGG = 1ull << 62, ST_PREFIX = 2ull << 62;

__device__ __forceinline__ u64 desc_load(const u64 *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void desc_store(u64 *p, u64 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 pack_agg(u32 P, u32 T0, u32 T1) {
    return ST_AGG | ((u64)(P & 1u) << 61) | ((u64)T1 << 28) | (u64)T0;
}
__device__ __forceinline__ u64 pack_prefix(u32 G, u64 count) { return ST_PREFIX | ((u64)(G & 1u) << 61) | count; }

// ---- wave primitives (DPP: no LDS traffic) -----------------------------------------------
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ u32 dpp_or0(u32 src) {  // lanes without a source (or masked rows) read 0
    return (u32)__builtin_amdgcn_update_dpp(0, (int)src, CTRL, ROW_MASK, 0xf, true);
}
__device__ __forceinline__ u32 wave_incl_scan(u32 v) {  // inclusive + scan over the 64 lanes
    v += dpp_or0<0x111>(v);       // row_shr:1
    v += dpp_or0<0x112>(v);       // row_shr:2
    v += dpp_or0<0x114>(v);       // row_shr:4
    v += dpp_or0<0x118>(v);       // row_shr:8
    v += dpp_or0<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
    v += dpp_or0<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
    return v;
}
// value of the lane below; lane 0 receives `first`
__device__ __forceinline__ u32 wave_shift_up(u32 v, u32 first) {
    return (u32)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ u32 lane63(u32 v) { return (u32)__builtin_amdgcn_readlane((int)v, 63); }
__device__ __forceinline__ u32 uniform(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u32 lanes_below_popc(u64 mask) {  // popcount of mask bits below this lane
    return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
}

// ---- carries into the first chunk of a wave unit -------------------------------------------
// Parity of the backslash run that ends right before byte `p` (p > lead), i.e. the reference's
// prev_iter_ends_odd_backslash at a chunk boundary (find_odd_backslash_sequences_amd64.s:34-58).
__device__ __forceinline__ u32 peek_backslash_parity(const u8 *base, u64 lead, u64 p, u64 end = ~0ull) {
    u32 n = 0;
    if (p > end) return 0;  // (a chunk of the last unit that lies behind the message: blanks in front of it)
    while (p > lead && base[p - 1] == '\\') {
        n++;
        p--;
    }
    return n & 1u;
}

static constexpr u64 BS8 = 0x5c5c5c5c5c5c5c5cull, SP8 = 0x2020202020202020ull;

// carry_in of the chunk at `off` from the 8 bytes before it
__device__ __forceinline__ u32 carry_from_prev8(u64 prev8, const u8 *base, u64 lead, u64 off) {
    const u64 x = prev8 ^ BS8;
    if (x == 0) return peek_backslash_parity(base, lead, off);  // >= 8 backslashes: walk (rare)
    return ((u32)__builtin_clzll(x) >> 3) & 1u;
}
// pseudo_pred carry from the previous byte only: whitespace, one of {}[]:, or an
// unescaped quote.
__device__ __forceinline__ u32 pseudo_pred_from_prev8(u64 prev8, const u8 *base, u64 lead, u64 off) {
    const u32 b = (u32)(prev8 >> 56);
    if (b == ' ' || b == '\t' || b == '\n' || b == '\r') return 1;
    if (b == '{' || b == '}' || b == '[' || b == ']' || b == ':' || b == ',') return 1;
    if (b != '"') return 0;
    const u64 x = (prev8 ^ BS8) << 8;
    if (x == 0) return peek_backslash_parity(base, lead, off - 1) ^ 1u;
    return ((((u32)__builtin_clzll(x | 0xffu) >> 3) & 1u)) ^ 1u;
}

// ---- chunk load ------------------------------------------------------------------------
// `base` is 64-byte aligned; the message occupies [lead, lead+len) = [lead, end) of it.  A wave unit is 64
// consecutive chunks (4 KiB, one per lane): one scalar base + lane * 64.  In the first and the last unit of a
// message the bytes outside the message count as 0x20, exactly like the reference's space-masked tail
// (find_structural_bits_amd64.s:134-155); leading pad bytes are whitespace as well, which leaves the initial
// pseudo_pred (=1) semantics untouched.  (Rounds 1-5: a preparation kernel left blank-padded copies of the two units
// in the workspace.  Round 6: the wave that loads one of the two builds it in its registers, edge_unit_load below --
// a branch where the loads are issued, nothing in phase A: blanking the class masks there, a dozen 64-bit operations
// behind a uniform branch in the middle of the classification, cost 2-3 % at every size, profiles/r06_stage1_no_prepare_ab.txt.)
// a unit in flight: this lane's chunk and (every lane the same) the 8 message bytes in front of the unit, which the
// carries into its first chunk come from -- fetched with the chunk so that nothing waits for a scalar load later
// (measured -2.5 %).  One unit per wave is in flight; a second one (each pass loaded a whole tile ahead) measured
// 5-8 % SLOWER: the flatten's stores then queue behind twice as many loads.
struct UnitRegs {
    uint4 v[4];
    u64 prev8;
};
// The first or the last unit of the message (at most two units per launch).  A chunk inside the message is loaded, a chunk
// outside is blanks, and the (at most two) chunks the message covers only partly -- the one with its first byte when the
// message does not begin on a 64-byte boundary, the one with its last -- are fetched a byte per lane (only message bytes are
// touched), laid down in 64 bytes of LDS and picked up by the lane that owns the chunk.
__device__ __forceinline__ void edge_unit_load(const u8 *__restrict__ base, u64 lead, u64 end, u64 unit, int lane, UnitRegs &r,
                                               u8 *s_edge_w) {
    const u64 c0 = unit * 4096 + (u64)lane * 64;
    const uint4 blank = make_uint4(0x20202020u, 0x20202020u, 0x20202020u, 0x20202020u);
#pragma unroll
    for (int q = 0; q < 4; q++) r.v[q] = blank;
    if (c0 >= lead && c0 + 64 <= end) {
        const uint4 *p = reinterpret_cast<const uint4 *>(base + c0);
#pragma unroll
        for (int q = 0; q < 4; q++) r.v[q] = p[q];
    }
    const u64 cl = (end - 1) & ~63ull;  // the chunk that holds the last byte (len > 0)
    const bool head = unit == 0 && lead > 0;
    const bool tail = (end & 63) != 0 && (cl >> 12) == unit && !(head && cl == 0);  // (one chunk may hold both ends: once)
#pragma unroll 1
    for (int k = 0; k < 2; k++) {
        if (!(k == 0 ? head : tail)) continue;  // (uniform)
        const u64 pc = k == 0 ? 0ull : cl;
        const u64 a = pc + (u64)lane;
        u8 b = 0x20;
        if (a >= lead && a < end) b = base[a];
        s_edge_w[lane] = b;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (c0 == pc) {
            const uint4 *e = reinterpret_cast<const uint4 *>(s_edge_w);
#pragma unroll
            for (int q = 0; q < 4; q++) r.v[q] = e[q];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();  // (the window is read: the second round may write it)
    }
}
__device__ __forceinline__ void unit_issue(const u8 *__restrict__ base, u64 lead, u64 end, u64 unit, u32 nu, int lane, UnitRegs &r,
                                           u8 *s_edge_w) {
    if (unit != 0 && unit + 1 != nu) {  // (uniform)
        const uint4 *p = reinterpret_cast<const uint4 *>(base + unit * 4096) + lane * 4;
#pragma unroll
        for (int q = 0; q < 4; q++) r.v[q] = p[q];
    } else {
        edge_unit_load(base, lead, end, unit, lane, r, s_edge_w);
    }
    // (unit 0 has nothing in front of it: its own first bytes are read instead and not used)
    r.prev8 = *reinterpret_cast<const u64 *>(base + (unit ? unit * 4096 - 8 : 0));
}
// ---- the look-back (wave 0 of a block) -------------------------------------------------------
// Resumable look-back: one call = one window of 256 descriptors (4 per lane, nearest first).
struct LookBack {
    long long j;  // nearest descriptor not yet consumed
    u32 Fp;       // effect of the already-composed (nearer) tiles: state g -> (g ^ Fp, + Ft[g])
    u64 Ft0, Ft1;
};

__device__ __forceinline__ void lookback_load(const u64 *desc, long long j, int lane, u64 (&d)[4]) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const long long idx = j - 4 * lane - q;
        d[q] = idx >= 0 ? desc_load(&desc[idx]) : pack_prefix(0, 0);  // virtual prefix before tile 0
    }
}
// Evaluates the window d loaded at lb.j.  0: needed descriptors still invalid (reload the same window),
// 1: resolved (G, BASE valid), 2: 256 aggregates folded into lb, look further back (load at the new lb.j)
__device__ __forceinline__ int lookback_eval(const u64 (&d)[4], LookBack &lb, int lane, u32 &G, u64 &BASE) {
    // this lane's four descriptors, nearest first, as one function of the state at their far end
    u32 lp = 0, lt0 = 0, lt1 = 0, lG = 0;
    u64 lC = 0;
    bool linv = false, lpre = false;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const u32 status = (u32)(d[q] >> 62);
        if (!linv && !lpre) {
            if (status == 0) {
                linv = true;
            } else if (status == 2) {
                lpre = true;
                lG = (u32)((d[q] >> 61) & 1u);
                lC = d[q] & 0x0000ffffffffffffull;
            } else {
                const u32 P = (u32)((d[q] >> 61) & 1u);
                const u32 T0 = (u32)(d[q] & 0x0fffffffu), T1 = (u32)((d[q] >> 28) & 0x0fffffffu);
                const u32 a0 = T0 + (P ? lt1 : lt0), a1 = T1 + (P ? lt0 : lt1);  // this one first, then the nearer part
                lt0 = a0;
                lt1 = a1;
                lp ^= P;
            }
        }
    }
    const u64 inv_b = __ballot(linv), pre_b = __ballot(lpre);
    const int fp = pre_b ? ctz64(pre_b) : 64;  // first lane that holds a PREFIX
    const u64 below_fp = fp >= 64 ? ~0ull : ((1ull << fp) - 1);
    if (inv_b & below_fp) return 0;  // (the prefix lane itself met its prefix before any invalid one)
    if (lane > fp) {
        lp = 0;
        lt0 = 0;
        lt1 = 0;
    }
    const u64 pb = __ballot(lp != 0);
    const u64 above = lane >= 63 ? 0ull : (~0ull << (lane + 1));
    const u32 par_above = (u32)popc64(pb & above) & 1u;  // parity between this lane and the far end
    const u32 s0 = lane63(wave_incl_scan(par_above ? lt1 : lt0));
    const u32 s1 = lane63(wave_incl_scan(par_above ? lt0 : lt1));
    const u32 WP = (u32)popc64(pb) & 1u;
    if (fp < 64) {
        const u32 g = (u32)__builtin_amdgcn_readlane((int)lG, fp);
        const u64 c = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(lC >> 32), fp) << 32) |
                      (u32)__builtin_amdgcn_readlane((int)(u32)lC, fp);
        const u32 gw = g ^ WP;  // state right in front of the already-composed part
        G = gw ^ lb.Fp;
        BASE = c + (g ? s1 : s0) + (gw ? lb.Ft1 : lb.Ft0);
        return 1;
    }
    // 256 aggregates and no prefix: fold the window into F and look further back
    const u64 a0 = (u64)s0 + (WP ? lb.Ft1 : lb.Ft0);
    const u64 a1 = (u64)s1 + (WP ? lb.Ft0 : lb.Ft1);
    lb.Ft0 = a0;
    lb.Ft1 = a1;
    lb.Fp ^= WP;
    lb.j -= 256;
    return 2;
}

// ---- phase A: everything that does not need the state in front of the tile ----------------
// One pass = one 64-byte chunk per lane.  The two candidate structural masks of every chunk go to
// the wave's private LDS window m[pass][outside|inside][lane], the lane's inclusive structural counts
// to pre[pass][lane].  While pass k is being computed the loads of pass k+1 (or of pass 0 of the next tile) are in
// flight: they are issued at the start of pass k, before its classification.  TOP: the wave keeps the highest issue
// priority through the whole phase (wave 0 of the barrier kernels: it has the look-back to resolve afterwards).
// s_unit[u] = parity << 31 | ctrl-in-string(inside) << 27 | ctrl-in-string(outside) << 26 |
//             count(inside) << 13 | count(outside)          for unit u = pass * WAVES + wave.
__device__ __forceinline__ bool par_ballot_any(u64 m) { return __ballot(((u32)m | (u32)(m >> 32)) != 0) != 0; }
template <int BLOCK, int CH, bool NDJSON, bool AUX>
__device__ __forceinline__ void phase_a(const u8 *__restrict__ base, u64 lead, u64 end, TileMap tm, u32 t, u32 t_next,
                                        bool has_next, int lane, int wave, UnitRegs &pf, u64 *m, u32 *pre, u32 *s_unit,
                                        const S1Aux &aux, u64 (&kp)[CH][4], uint2 *s_ucnt,
                                        u32 &seen_st, u8 *s_edge_w, bool TOP = false) {
    constexpr int WAVES = BLOCK / 64;
    constexpr int UNITS = WAVES * CH;
#if defined(SJ_S1_ROLL)
#pragma unroll 1
#else
#pragma unroll
#endif
    for (int k = 0; k < CH; k++) {
        const u64 unit = tile_unit<UNITS>(tm, t, k * WAVES + wave);  // wave-uniform
        // the unit behind this one (next pass, or the first pass of the next tile): its loads are issued below
        const u64 unit_nx = k + 1 < CH ? tile_unit<UNITS>(tm, t, (k + 1) * WAVES + wave)
                                       : (has_next ? tile_unit<UNITS>(tm, t_next, wave) : VOID_UNIT);
        if (unit == VOID_UNIT) {  // a tail tile holds fewer units than a full one: nothing to classify
            m[(k * 2 + 0) * 64 + lane] = 0;
            m[(k * 2 + 1) * 64 + lane] = 0;
            pre[k * 64 + lane] = 0;
            if (lane == 0) s_unit[k * WAVES + wave] = 0;
            if (AUX && lane == 0) s_ucnt[k * WAVES + wave] = make_uint2(0u, 0u);
            kp[k][0] = kp[k][1] = kp[k][2] = kp[k][3] = 0;
            if (unit_nx != VOID_UNIT) unit_issue(base, lead, end, unit_nx, tm.nu, lane, pf, s_edge_w);
            continue;
        }
        const u64 unit_off = unit * 4096;
        u32 w[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            w[4 * q + 0] = pf.v[q].x;
            w[4 * q + 1] = pf.v[q].y;
            w[4 * q + 2] = pf.v[q].z;
            w[4 * q + 3] = pf.v[q].w;
        }
        u32 carry0 = 0, pp0 = 1;  // carries into lane 0 from the bytes in front of the unit
        if (unit != 0) {  // those 8 bytes are message bytes: lead < 64, and the unit begins in front of `end`
            const u64 prev8 = ((u64)uniform((u32)(pf.prev8 >> 32)) << 32) | uniform((u32)pf.prev8);
            carry0 = carry_from_prev8(prev8, base, lead, unit_off);
            pp0 = pseudo_pred_from_prev8(prev8, base, lead, unit_off);
        }

        // the next pass goes in flight before this one is classified: a whole pass of math to arrive in (the compiler
        // keeps the chunk in its own registers; issuing the loads only after classify(), into the registers the chunk
        // dies in, measured 1-2 % slower)
        if (unit_nx != VOID_UNIT) unit_issue(base, lead, end, unit_nx, tm.nu, lane, pf, s_edge_w);
        __builtin_amdgcn_sched_barrier(0);
        // issue priority by progress: the SIMD arbiter prefers its oldest wave, which then finishes a pass long before
        // the others and leaves the tail of every phase to one or two waves that cannot fill the pipe; with the waves
        // that are behind going first the four finish together (measured -4 %).  TOP: this wave stays in front (wave 0
        // of the barrier kernel: it has the look-back to do while the others are still in this phase).
        if (k == 0 || TOP) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(1);
        const Classes c = classify(w);
        if (AUX) {  // the kind of the token a byte would start, as four bit planes (flatten_tile looks them up per structural)
            kp[k][0] = c.kp[0];
            kp[k][1] = c.kp[1];
            kp[k][2] = NDJSON ? c.kp[2] | c.nl : c.kp[2];  // '\n' is a token (kind 12) only in NDJSON
            kp[k][3] = NDJSON ? c.kp[3] | c.nl : c.kp[3];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (TOP) __builtin_amdgcn_s_setprio(3);
        else if (k == 0) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(0);

        // ---- backslash carry: parity of the run of backslashes at the END of the previous chunk.
        // If that chunk is not all backslashes this does not depend on ITS carry-in.
        u64 quote_bits = c.quote;
        u64 starters = 0;  // backslashes that begin an escape sequence
        u64 nonsimple = 0; // escaped characters other than " \\ / b f n r t
        const u32 bs_any = (u32)c.bs | (u32)(c.bs >> 32);
        if (__ballot(bs_any != 0) != 0 || carry0 != 0) {  // wave-uniform: many waves see no backslash at all
            const bool all_bs = c.bs == ~0ull;
            const u32 trail_odd = all_bs ? 0u : ((u32)__builtin_clzll(~c.bs) & 1u);
            u32 carry_in = wave_shift_up(trail_odd, carry0);
            if (__ballot(all_bs) != 0 && lane != 0) carry_in = peek_backslash_parity(base, lead, unit_off + (u64)lane * 64, end);
            const u64 escaped = escaped_mask(c.bs, carry_in);
            quote_bits &= ~escaped;
            if (AUX) {
                starters = c.bs & ~escaped;
                nonsimple = escaped & ~c.esc1;
            }
        }

        // ---- in-string mask relative to the start of the wave unit
        const u32 par =
            ((u32)__builtin_popcount((u32)quote_bits) + (u32)__builtin_popcount((u32)(quote_bits >> 32))) & 1u;
        const u64 par_ballot = __ballot(par != 0);
        u64 qm = prefix_xor(quote_bits);
        if (lanes_below_popc(par_ballot) & 1u) qm = ~qm;

        // ---- pseudo-structural predecessor
        const u32 pp_out = ((u32)(c.structs >> 32) | (u32)(quote_bits >> 32) | (u32)(c.ws >> 32)) >> 31;
        const u32 pp_in = wave_shift_up(pp_out, pp0);

        u64 a = finalize(c.structs, c.ws, qm, quote_bits, pp_in);
        u64 b = finalize(c.structs, c.ws, ~qm, quote_bits, pp_in);
        if (NDJSON) {
            a = bitop3<(TA | (TB & ~TC))>(a, c.nl, qm);
            b = bitop3<(TA | (TB & TC))>(b, c.nl, qm);
        }
        m[(k * 2 + 0) * 64 + lane] = a;
        m[(k * 2 + 1) * 64 + lane] = b;
        // (AUX: the mask arrays are there -- stage1_launch refuses kinds without them -- and a unit that is not void begins in
        // front of `end`: no tests here, each one would be a branch in the middle of the pass)
        if (AUX && !SJ_S1EXP(aux, 16)) {  // whole parse: stage 2 unescapes the strings from these masks (stage2.hip)
            const u64 ci = unit * 64 + (u64)lane;
            aux.qm[ci] = qm;  // relative to the state at the start of the unit: resolved with aux.unit_h
            aux.st[ci] = starters;
            // (hypothesis-free: an escape outside a string makes the document invalid anyway)
            const u64 slow = __ballot(((u32)nonsimple | (u32)(nonsimple >> 32)) != 0);
            if (lane == 0) aux.unit_slow[unit] = slow;
        }
        if (AUX) {
            // every string copied: emitted bytes (sj_strings.h: in-string bytes that are neither quotes nor escape starters --
            // the fast formula; units with a \u escape are counted again by k_measure) and opening quotes of the unit under
            // hypothesis 0, and under either hypothesis together (the two sets are disjoint: the other one is the difference)
            uint2 tot = make_uint2(0u, 0u);
            if (!SJ_S1EXP(aux, 18)) {
                const u64 nqst = ~quote_bits & ~starters;
                const u32 cnt = wave_incl_scan((u32)popc64(qm & nqst) | ((u32)popc64(nqst) << 16));
                const u32 opn = wave_incl_scan((u32)popc64(qm & quote_bits) | ((u32)popc64(quote_bits) << 16));
                tot = make_uint2(lane63(cnt), lane63(opn));
            }
            if (lane == 0) s_ucnt[k * WAVES + wave] = tot;
        }
        // unescaped control characters inside strings (find_quote_mask_and_bits_amd64.s:67-80), per hypothesis
        const u64 in_a = c.ctrl & qm, in_b = c.ctrl & ~qm;
        const u32 bad = (__ballot(((u32)in_a | (u32)(in_a >> 32)) != 0) != 0 ? 1u : 0u) |
                        (__ballot(((u32)in_b | (u32)(in_b >> 32)) != 0) != 0 ? 2u : 0u);
        // unit totals of both counts at once (16-bit fields: a wave holds <= 4096 bits)
        const u32 incl = wave_incl_scan((u32)popc64(a) | ((u32)popc64(b) << 16));
        pre[k * 64 + lane] = incl;  // flatten needs the lane's offset inside the unit: kept instead of scanned again
        const u32 tot = lane63(incl);
        // (whole parse, bit 28: the unit holds an escape starter -- k_str_emit does not read the st masks of the others)
        const u32 has_st = AUX && __ballot(((u32)starters | (u32)(starters >> 32)) != 0) != 0 ? 1u : 0u;
        seen_st |= has_st;  // (wave-uniform)
        // (bit 29: the unit holds an unescaped quote -- the walks of the selective copy pass over units without one 64 at a time)
        const u32 has_q = AUX && par_ballot_any(quote_bits) ? 1u : 0u;
        if (lane == 0)
            s_unit[k * WAVES + wave] =
                (((u32)popc64(par_ballot) & 1u) << 31) | (has_q << 29) | (has_st << 28) | (bad << 26) | ((tot >> 16) << 13) | (tot & 0x1fffu);
    }
}

// tile aggregate for both incoming states, one unit per lane (called by one wave);
// pre_mask bit u = parity of the units in front of unit u
template <int UNITS>
__device__ __forceinline__ void tile_aggregate(const u32 *s_unit, int lane, u32 &P, u32 &T0, u32 &T1, u32 &pre_mask) {
    const u32 v = lane < UNITS ? s_unit[lane] : 0u;
    const u64 pbm = __ballot((v >> 31) != 0);
    const u32 pre = lanes_below_popc(pbm) & 1u;  // parity of the units in front of this one
    const u32 c0 = v & 0x1fffu, c1 = (v >> 13) & 0x1fffu;
    T0 = lane63(wave_incl_scan(pre ? c1 : c0));
    T1 = lane63(wave_incl_scan(pre ? c0 : c1));
    P = (u32)popc64(pbm) & 1u;
    pre_mask = (u32)__ballot(pre != 0);
}

// ---- flatten (flatten_bits_amd64.s:26-60, absolute positions instead of deltas) ------------
// Each wave expands its own units: lanes scatter their positions into the wave's LDS staging buffer, then the
// wave copies the buffer out with coalesced 256-byte stores.  A unit with more positions than the buffer holds
// (512: the window that held the masks; 1024 in the whole-parse kernel) takes several rounds.
// The whole-parse kernel (KIND) also writes the KIND of every token (sj_stage2.h) next to its position.  The kind is a
// function of the token's first byte, and phase A has the bit planes of that function for every chunk (Classes::kp):
// they wait in LDS (kpl, 32 bytes per chunk), the staged entries are 16-bit offsets inside the unit, and the copy-out
// looks the four plane bits of an entry up -- no load from the message.  (The first version gathered the byte of every
// structural from the message and translated it through a table: 17 us of 160 on configs[1], 100 us of 316 on
// configs[4], whose 78 M tokens are one gather each.)
template <int BLOCK, int CH, bool KIND>
__device__ __forceinline__ bool flatten_tile(TileMap tm, u64 *m, const u64 *kpl, const u32 *pre, const u32 *s_unit, u32 pre_mask, u32 G, u64 BASE, u32 t,
                                             u64 lead, int lane, int wave, SJ_ARR_PARAM(u32) out_pos, u64 pos_cap,
                                             u64 &tile_end, Arr<u8> unit_h, u64 len_, Arr<u8> kind_out, const S1Aux &aux,
                                             const uint2 *s_ucnt) {
    constexpr int WAVES = BLOCK / 64;
    constexpr int UNITS = WAVES * CH;
    // the window that held the masks (CH * 2 * 64 u64) stages the positions of a unit: 32-bit positions, or (KIND)
    // 16-bit offsets inside the unit; denser units take several rounds
    constexpr u32 CAP = KIND ? (u32)CH * 512u : (u32)CH * 256u;
    typedef typename std::conditional<KIND, uint16_t, u32>::type Staged;
    static_assert(UNITS <= 64, "one unit per lane in the prefix");
    // per-unit counts under the now known state, prefix over the units (lane u <-> unit u)
    const u32 v = lane < UNITS ? s_unit[lane] : 0u;
    const u32 hl = G ^ ((pre_mask >> (lane & 31)) & 1u);
    const u32 cl = lane < UNITS ? (hl ? ((v >> 13) & 0x1fffu) : (v & 0x1fffu)) : 0u;
    const u32 incl = wave_incl_scan(cl);
    tile_end = BASE + lane63(incl);
    const bool fits = tile_end <= pos_cap;  // uniform: no per-store capacity check needed

    const bool err = lane < UNITS && ((v >> (26 + hl)) & 1u) != 0;  // any lane: the caller ballots

    u64 sel[CH];
    u32 upto[CH];  // structurals of the unit up to and including this lane's chunk
#pragma unroll
    for (int k = 0; k < CH; k++) {
        const u32 h = G ^ ((pre_mask >> (k * WAVES + wave)) & 1u);
        sel[k] = m[(k * 2 + (int)h) * 64 + lane];
        const u32 both = pre[k * 64 + lane];
        upto[k] = h ? both >> 16 : both & 0xffffu;
        const u64 un = tile_unit<UNITS>(tm, t, k * WAVES + wave);
        if (unit_h && lane == 0 && un * 4096 < lead + len_) {  // (a void unit lies behind everything)
            // bit 0: the state at the start of the unit; bit 1: the unit holds an escape starter; bit 2: it holds an unescaped quote
            unit_h[un] = (u8)(h | (((s_unit[k * WAVES + wave] >> 28) & 3u) << 1));
            if (KIND) {  // whole parse: the unit's counts under the state that is now known
                const uint2 c = s_ucnt[k * WAVES + wave];
                aux.unit_cnt[un] = h ? (c.x >> 16) - (c.x & 0xffffu) : c.x & 0xffffu;
                aux.unit_str[un] = h ? (c.y >> 16) - (c.y & 0xffffu) : c.y & 0xffffu;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < CH; k++) {
        const int u = k * WAVES + wave;
        if (k == 0) __builtin_amdgcn_s_setprio(3);  // (see phase_a)
        else __builtin_amdgcn_s_setprio(1);
        const u32 C = (u32)__builtin_amdgcn_readlane((int)cl, u);
        const u64 g = BASE + ((u32)__builtin_amdgcn_readlane((int)incl, u) - C);
        if (KIND && lane == 0 && C != 0) {  // (a unit holds at most 4096 tokens: at most one 4096-token tile begins in it)
            const u64 T = (g + 4095) >> 12;
            if (T * 4096 < g + C) aux.tile_unit[T] = (u32)tile_unit<UNITS>(tm, t, u);
        }
        const u64 s = sel[k];
        const u32 lo0 = (u32)s, hi0 = (u32)(s >> 32);
        const u32 n = (u32)__builtin_popcount(lo0) + (u32)__builtin_popcount(hi0);
        const u32 loc = upto[k] - n;  // offset of this lane's first position inside the unit
        const u32 ubase = (u32)(tile_unit<UNITS>(tm, t, u) * 4096 - lead);  // (unused for a void unit: no bits)
        u32 pos0 = KIND ? (u32)lane * 64u : ubase + (u32)lane * 64u;        // what a lane stages: position, or offset in the unit
        __builtin_amdgcn_wave_barrier();  // the staging window is free: all masks are in registers / already copied out
        Staged *stage = reinterpret_cast<Staged *>(m);
        // the four kind planes of this pass, interleaved per 32-byte half of a chunk: one 16-byte LDS read per entry
        // (plane-major, four 4-byte reads: the copy-out issued 16 LDS reads per four entries)
        const uint4 *k128 = reinterpret_cast<const uint4 *>(kpl + (KIND ? k * 4 * 64 : 0));
        auto kind_of = [&](u32 e) -> u32 {  // e = chunk << 6 | bit
            const uint4 kk = k128[e >> 5];  // (chunk, half)
            const u32 b = e & 31u;
            return ((kk.x >> b) & 1u) | (((kk.y >> b) & 1u) << 1) | (((kk.z >> b) & 1u) << 2) | (((kk.w >> b) & 1u) << 3);
        };
        // copies the first cnt staged entries to out_pos[gd ...] (and their kinds to kind_out)
        auto copy_out = [&](u32 cnt, u64 gd) {
            if (KIND) {
                // four consecutive entries per lane: one 8-byte LDS read, sixteen plane words, one 16-byte store of the
                // positions and one 4-byte store of their kinds (neither is aligned to its size in memory: fine on gfx950)
                const u32 c4 = (fits ? cnt : 0u) & ~3u;
                for (u32 i = (u32)lane * 4u; i < c4; i += 256u) {
                    const uint2 e2 = *reinterpret_cast<const uint2 *>(stage + i);
                    const u32 e0 = e2.x & 0xffffu, e1 = e2.x >> 16, e3 = e2.y >> 16, e2v = e2.y & 0xffffu;
                    *reinterpret_cast<uint4 *>(arr_at(out_pos, gd + i, 4)) = make_uint4(ubase + e0, ubase + e1, ubase + e2v, ubase + e3);
                    if (!SJ_S1EXP(aux, 17)) *reinterpret_cast<u32 *>(arr_at(kind_out, gd + i, 4)) = kind_of(e0) | (kind_of(e1) << 8) | (kind_of(e2v) << 16) | (kind_of(e3) << 24);
                }
                for (u32 i = c4 + (u32)lane; i < cnt; i += 64) {  // the last <= 3 (or, without room for all, everything)
                    if (fits || gd + i < pos_cap) {
                        const u32 e = (u32)stage[i];
                        out_pos[gd + i] = ubase + e;
                        kind_out[gd + i] = (u8)kind_of(e);
                    }
                }
            } else if (fits) {
                // 16 bytes per lane (the positions are only 4-byte aligned in memory: fine on gfx950), then the rest
                const u32 c4 = cnt & ~3u;
                for (u32 i = (u32)lane * 4u; i < c4; i += 256u) {
#if defined(SJ_S1_PLAIN_STORE)  // (A/B)
                    *reinterpret_cast<uint4 *>(arr_at(out_pos, gd + i, 4)) = *reinterpret_cast<const uint4 *>(stage + i);
#else
                    // streaming stores: the positions are not read again by this kernel, and without them in the way the next
                    // pass over a cache-sized message finds more of it in the Infinity Cache (configs[1]: 0.1035 -> 0.094 ms)
                    nt_store4(arr_at(out_pos, gd + i, 4), *reinterpret_cast<const uint4 *>(stage + i));
#endif
                }
                if (c4 + (u32)lane < cnt) out_pos[gd + c4 + lane] = (u32)stage[c4 + lane];
            } else {
                for (u32 i = lane; i < cnt; i += 64)
                    if (gd + i < pos_cap) out_pos[gd + i] = (u32)stage[i];
            }
        };
        if (C <= CAP) {
            // two positions per iteration, the lowest and the highest set bit of the word (an odd last bit is
            // simply written twice to the same slot)
            const u32 nlo = (u32)__builtin_popcount(lo0);
            Staged *p = stage + loc, *q = p + nlo - 1;
            for (u32 x = lo0; x != 0;) {
                const u32 top = 31u - (u32)__builtin_clz(x);
                *p++ = (Staged)(pos0 + (u32)__builtin_ctz(x));
                *q-- = (Staged)(pos0 + top);
                x = bitop3<(TA & TB & ~TC)>(x, x - 1u, 1u << top);
            }
            pos0 += 32;
            p = stage + loc + nlo;
            q = p + (n - nlo) - 1;
            for (u32 x = hi0; x != 0;) {
                const u32 top = 31u - (u32)__builtin_clz(x);
                *p++ = (Staged)(pos0 + (u32)__builtin_ctz(x));
                *q-- = (Staged)(pos0 + top);
                x = bitop3<(TA & TB & ~TC)>(x, x - 1u, 1u << top);
            }
            __builtin_amdgcn_wave_barrier();
            copy_out(C, g);
        } else {
            u64 r = s;
            u32 l = loc;
            for (u32 r0 = 0; r0 < C; r0 += CAP) {
                const u32 lim = r0 + CAP;
                while (r != 0 && l < lim) {
                    stage[l - r0] = (Staged)(pos0 + (u32)ctz64(r));
                    r &= r - 1;
                    l++;
                }
                __builtin_amdgcn_wave_barrier();
                copy_out(C - r0 < CAP ? C - r0 : CAP, g + r0);
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    return err;
}

// ---- the kernel ---------------------------------------------------------------------------
// Persistent blocks draw tiles from a ticket counter: tiles are started in id order, so every
// predecessor in the look-back chain is resident or finished (forward progress without any
// dispatch-order assumption), and no block ever waits for the dispatcher.
template <int BLOCK, int CH, int WPE, bool NDJSON, bool AUX, bool TRACE = false>
__global__ __launch_bounds__(BLOCK, WPE) void stage1_kernel(const u8 *__restrict__ base, u64 lead, u64 len,
                                                                        u32 *__restrict__ out_pos,
                                                                        u64 pos_cap, Stage1State *__restrict__ st,
                                                                        u64 *__restrict__ desc, u32 num_tiles, TileMap tm, S1Aux aux) {
    constexpr int WAVES = BLOCK / 64;
    constexpr int UNITS = WAVES * CH;  // unit u = pass * WAVES + wave, in byte order
    static_assert(UNITS <= 32, "pre_mask is a u32");
    __shared__ u32 s_ticket[4];
    __shared__ u32 s_unit[3][UNITS];
    __shared__ u32 s_res2[2][4];  // look-back result of tile T(i) in slot i & 1: G, pre_mask, BASE (lo, hi)
    __shared__ u64 s_mask[2][WAVES][CH * 2 * 64];
    __shared__ u32 s_pre[2][WAVES][CH * 64];  // per chunk: inclusive structural counts of its unit, both hypotheses
    // whole parse: the kind planes of the tile that is flattened next (4 x u64 per chunk); a wave keeps the planes of the
    // tile it has just classified in registers until it has flattened the tile in front, then parks them here
    __shared__ __attribute__((aligned(16))) u64 s_kpl[AUX ? WAVES : 1][AUX ? CH * 4 * 64 : 2];
    __shared__ uint2 s_ucnt[3][AUX ? UNITS : 1];  // whole parse: string bytes / opening quotes per unit, both hypotheses (ring of s_unit)
    __shared__ __attribute__((aligned(16))) u8 s_edge[WAVES][64];  // unit_issue: the chunk that holds the message's first / last byte

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = (int)uniform((u32)tid >> 6);
    const u64 end = lead + len;
    u64 kp[CH][4];
    Stage1Ctrl *const ctl = &st->c[aux.par];
    launch_clean(st, aux);

    // The first tile of a block: its own index if the whole message is ONE round of tiles (num_tiles <= gridDim.x: every
    // tile in front of a block's tile then belongs to a block with a lower index, which the dispatcher started earlier, so
    // it is resident or through -- the look-back's forward-progress condition -- and nobody queues up on the ticket
    // counter in front of its first load: -3 us per launch, which matters for exactly these documents).  A message of
    // several rounds draws EVERY tile from the counter: tiles then start in id order whatever else runs on the device.
    // (Round 4 gave every block a static first tile; with a second kernel holding compute units a block could then draw
    // a later ticket and look back on tiles of blocks that had not been dispatched yet -- a bounded spin, i.e. a
    // spurious "internal error", with several >256-tile parses in flight on one device.)
    const bool one_round = num_tiles <= gridDim.x;  // (uniform over the grid)
    const u32 tk_base = one_round ? gridDim.x : 0u;
    // (round 6) the ticket of the SECOND tile is drawn together with the first: phase A of T(0) then sends T(1)'s first unit on
    // its way like every later phase A does for its successor, instead of behind the first barrier with the whole load
    // latency exposed (64 MiB: 0.0384 -> 0.0370 ms, same box, alternating; nothing at 256 MiB and 1 GiB)
    if (tid == 0) {
        s_ticket[0] = one_round ? blockIdx.x : atomicAdd(&ctl->tile_counter, 1u);
        s_ticket[1] = tk_base + atomicAdd(&ctl->tile_counter, 1u);
    }
    __syncthreads();
    const u32 t_first = uniform(s_ticket[0]);
    if (t_first >= num_tiles) return;
    UnitRegs pf;
    {
        const u64 un = tile_unit<UNITS>(tm, t_first, wave);
        if (un != VOID_UNIT) unit_issue(base, lead, end, un, tm.nu, lane, pf, s_edge[wave]);
    }

    // One loop, one copy of phase A and of the flatten in the instruction stream (a peeled first tile made every block
    // fetch ~9 KB more code cold at the start of every launch).  Two tiles in flight per block and one block barrier
    // per tile.  Iteration j: every wave runs phase A of T(j) (wave 0 at top priority, so it is through first); wave 0
    // then resolves the look-back of T(j-1), whose aggregate went out an iteration ago, while the other waves are
    // still in phase A, and leaves the next ticket; barrier; wave 0 aggregates and publishes T(j) (a few hundred cycles
    // into its flatten); everybody flattens T(j-1).  Rings: tickets 4 (T(k) in slot k & 3; iteration j writes T(j+2)
    // before the barrier -- iteration 0 also T(1) -- into slots whose tiles are done), look-back results 2, unit
    // state 3, masks 2 (private per wave).
    u32 t_prev = 0;                         // T(j-1): the tile that is flattened in iteration j
    u32 P0 = 0, T00 = 0, T01 = 0, pm0 = 0;  // its aggregates; meaningful in wave 0 only
    bool err = false;
    u32 seen_st = 0;                        // whole parse: a unit of this wave held an escape starter
    u32 eiq_last = 0;                       // lane 0 of wave 0: the state at the end of the message (the block of the last tile)
    for (u32 j = 0;; j++) {
        const bool first = j == 0;
        const u32 t_a = uniform(s_ticket[j & 3u]);                               // phase A runs on T(j)
        const u32 t_an = uniform(s_ticket[(j + 1u) & 3u]);  // T(j+1): its first unit is loaded behind T(j)'s last
        const bool has_a = t_a < num_tiles;
        const int ma = (int)(j & 1u), ua = (int)(j % 3u);        // mask / unit slot of T(j)
        const int mf = ma ^ 1, uf = ua == 0 ? 2 : ua - 1;        // ... of T(j-1)
        u32 tk2 = 0;  // the ticket drawn now (lane 0 of wave 0); it returns while phase A runs
        if (tid == 0 && has_a) tk2 = tk_base + atomicAdd(&ctl->tile_counter, 1u);
        if (has_a) {
            trace_put<TRACE>(aux.trace, t_a, WAVES, wave, lane, 0);
            phase_a<BLOCK, CH, NDJSON, AUX>(base, lead, end, tm, t_a, t_an, t_an < num_tiles, lane, wave, pf, s_mask[ma][wave],
                                            s_pre[ma][wave], s_unit[ua], aux, kp, s_ucnt[AUX ? ua : 0], seen_st, s_edge[wave], wave == 0 && !first);
            trace_put<TRACE>(aux.trace, t_a, WAVES, wave, lane, 1);
        }
        u32 *res = s_res2[j & 1u];
        if (wave == 0) {
            u32 G = 0;
            u64 BASE = 0;
            if (!first && t_prev != 0) {
                LookBack lb = {(long long)t_prev - 1, 0, 0, 0};
                u64 win[4];
                lookback_load(desc, lb.j, lane, win);
                u32 spins = 0;
                for (;;) {
                    const int r = lookback_eval(win, lb, lane, G, BASE);
                    if (r == 1) break;
                    if (r == 0) __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22)) {  // bounded: a bug must not hang the device
                        if (lane == 0) report_error(ctl, aux, 0x80000000u);
                        break;
                    }
                    lookback_load(desc, lb.j, lane, win);
                }
                if (lane == 0) desc_store(&desc[t_prev], pack_prefix(G ^ P0, BASE + (G ? T01 : T00)));
            }
            if (lane == 0) {
                if (!first) {
                    res[0] = G;
                    res[1] = pm0;
                    res[2] = (u32)BASE;
                    res[3] = (u32)(BASE >> 32);
                    if (t_prev == num_tiles - 1) {
                        eiq_last = (G ^ P0) & 1u;
                        __hip_atomic_store(&st->ends_in_quote, eiq_last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (has_a) s_ticket[(j + 2u) & 3u] = tk2;
            }
            if (!first) trace_put<TRACE>(aux.trace, t_prev, WAVES, wave, lane, 2);
        }
        __syncthreads();
        if (wave == 0 && has_a) {  // (only wave 0 keeps the aggregates: it resolves the tile in the next iteration)
            tile_aggregate<UNITS>(s_unit[ua], lane, P0, T00, T01, pm0);
            if (lane == 0) desc_store(&desc[t_a], t_a == 0 ? pack_prefix(P0, T00) : pack_agg(P0, T00, T01));
        }
        if (!first) {
            trace_put<TRACE>(aux.trace, t_prev, WAVES, wave, lane, 3);
            const u32 G = uniform(res[0]), pm = uniform(res[1]);
            const u64 BASE = ((u64)uniform(res[3]) << 32) | uniform(res[2]);
            u64 tile_end = 0;
            err |= flatten_tile<BLOCK, CH, AUX>(tm, s_mask[mf][wave], s_kpl[AUX ? wave : 0], s_pre[mf][wave], s_unit[uf], pm, G, BASE, t_prev, lead, lane, wave,
                                           S1_POS_VIEW(out_pos, pos_cap), pos_cap, tile_end, AUX ? aux.unit_h : Arr<u8>(nullptr), len,
                                           AUX ? aux.kind : Arr<u8>(nullptr), aux, s_ucnt[AUX ? uf : 0]);
            if (t_prev == num_tiles - 1 && tid == 0) report_total(st, aux, tile_end, eiq_last, base + lead, len);
            trace_put<TRACE>(aux.trace, t_prev, WAVES, wave, lane, 4);
            if (TRACE && lane == 0)
                aux.trace[((u64)t_prev * WAVES + wave) * TRACE_WORDS + 5] =
                    (u64)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | ((u64)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32);
            if (!has_a) break;
        }
        if (AUX && has_a) {  // the planes of T(j) wait in LDS for its flatten in the next iteration (this wave's window only)
#pragma unroll
            for (int k = 0; k < CH; k++) {  // (chunk, half) -> the four plane words of that half
                uint4 *dst = reinterpret_cast<uint4 *>(&s_kpl[wave][k * 4 * 64]) + lane * 2;
                dst[0] = make_uint4((u32)kp[k][0], (u32)kp[k][1], (u32)kp[k][2], (u32)kp[k][3]);
                dst[1] = make_uint4((u32)(kp[k][0] >> 32), (u32)(kp[k][1] >> 32), (u32)(kp[k][2] >> 32), (u32)(kp[k][3] >> 32));
            }
        }
        t_prev = t_a;
    }
    if (__ballot(err) != 0 && lane == 0) report_error(ctl, aux, 1u);
    if (AUX && aux.want_flag) {  // (uniform over the grid)
        // one look and at most one store per BLOCK (an atomicOr per wave -- 4096 of them on one word as the kernel ends -- cost
        // configs[1] 31 us: a word takes ~88 atomics per microsecond)
        const int any_st = __syncthreads_or((int)seen_st);
        if (any_st && tid == 0 && __hip_atomic_load(&ctl->has_starter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
            __hip_atomic_store(&ctl->has_starter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- the same tile pipeline without block barriers, DEPTH tiles in flight per block ------------------------------
// With the barrier kernel every block moves in step with the slowest block of its generation: the ~256 tiles in front
// of a tile are the ones the other blocks are working on at the same time, so a look-back cannot finish before the
// slowest of them has published its aggregate, and the flatten of the tile -- one barrier later -- waits for it
// (without look-back and barrier the same kernel runs a quarter faster).  Here the waves of a block meet only through
// LDS flags and the flatten of a tile lags its phase A by D = DEPTH - 1 tiles:
//   iteration i of a wave:   phase A of T(i+1)  ->  arrive  ->  [ need res(i+1-D) ]  ->  flatten T(i+1-D)
//   * arrival: +1 on an LDS counter; the wave that completes a tile aggregates it, publishes its AGG descriptor and
//     marks it ready;
//   * the serial work -- ticket T(i+3), look-back of the aggregated tiles in order, PREFIX descriptor, result record
//     res(j) = {state, unit parities, output base} and its flag -- is wave 0's: it runs phase A at top priority
//     (s_setprio), is through first, resolves what it must (the tile it is about to flatten) and tries the next
//     aggregated tile once without waiting; with DEPTH 3 that try is a whole round after the tile's aggregate was
//     published, and when a block in front is late nobody waits for it in this round.
// Rings: tickets 8 (T(j) in slot j & 7, published through s_tkn), unit state / arrival / aggregates 8, results 4,
// masks DEPTH (private per wave).  A wave can be DEPTH iterations ahead of the slowest wave of its block: res(j)
// needs every wave's arrival for T(j), which lies behind that wave's flatten of T(j-D-1).
template <int BLOCK, int CH, int DEPTH, int WPE, bool NDJSON, bool AUX, bool TRACE = false>
__global__ __launch_bounds__(BLOCK, WPE) void stage1_kernel_nb(const u8 *__restrict__ base, u64 lead, u64 len,
                                                               u32 *__restrict__ out_pos, u64 pos_cap,
                                                               Stage1State *__restrict__ st, u64 *__restrict__ desc,
                                                               u32 num_tiles, TileMap tm, S1Aux aux) {
    constexpr int WAVES = BLOCK / 64;
    constexpr int UNITS = WAVES * CH;
    constexpr u32 D = DEPTH - 1;
    static_assert(UNITS <= 32, "pre_mask is a u32");
    static_assert(DEPTH >= 2 && DEPTH <= 3, "rings of 8 and 4");
    static_assert(!AUX, "the whole parse runs the barrier kernel (one set of kind planes in flight)");
    __shared__ u32 s_tk[8];         // T(j) in slot j & 7
    __shared__ u32 s_tkn;           // T(0) .. T(s_tkn - 1) are in s_tk
    __shared__ u32 s_unit[8][UNITS];
    __shared__ u32 s_arrive[8];     // waves that have finished phase A of the tile in slot j & 7
    __shared__ u32 s_ready[8];      // j + 1 once T(j) is aggregated (s_agg) and its AGG descriptor published
    __shared__ u32 s_agg[8][4];     // P, T0, T1, pre_mask of that tile
    __shared__ u32 s_res[4][4];     // look-back result of T(j) in slot j & 3: G, pre_mask, BASE (lo, hi)
    __shared__ u32 s_resflag[4];    // j + 1 once s_res[j & 3] holds the result of T(j)
    __shared__ u64 s_mask[DEPTH][WAVES][CH * 2 * 64];
    __shared__ u32 s_pre[DEPTH][WAVES][CH * 64];
    __shared__ __attribute__((aligned(16))) u8 s_edge[WAVES][64];  // unit_issue

    const int tid = threadIdx.x;
    Stage1Ctrl *const ctl = &st->c[aux.par];
    launch_clean(st, aux);
    u32 eiq_last = 0;  // lane 0 of wave 0: the state at the end of the message (the block of the last tile)
    u64 kp[CH][4];  // (unused: plain stage 1)
    if (tid < 8) {
        s_arrive[tid] = 0;
        s_ready[tid] = 0;
        s_resflag[tid & 3] = 0;
    }
    const int lane = tid & 63;
    const int wave = (int)uniform((u32)tid >> 6);
    const u64 end = lead + len;

    // prologue (two block barriers, once per block), as in the barrier kernel
    if (tid == 0) s_tk[0] = atomicAdd(&ctl->tile_counter, 1u);
    __syncthreads();
    const u32 t_first = uniform(s_tk[0]);
    if (t_first >= num_tiles) {
        return;
    }
    UnitRegs pf;
    {
        const u64 un = tile_unit<UNITS>(tm, t_first, wave);
        if (un != VOID_UNIT) unit_issue(base, lead, end, un, tm.nu, lane, pf, s_edge[wave]);
    }
    if (tid == 0) {
        s_tk[1] = atomicAdd(&ctl->tile_counter, 1u);
        s_tk[2] = atomicAdd(&ctl->tile_counter, 1u);
        s_tkn = 3;
    }
    trace_put<TRACE>(aux.trace, t_first, WAVES, wave, lane, 0);
    u32 seen_st_nb = 0;  // (this kernel never runs the whole parse: unused)
    phase_a<BLOCK, CH, NDJSON, AUX>(base, lead, end, tm, t_first, 0, false, lane, wave, pf, s_mask[0][wave], s_pre[0][wave], s_unit[0],
                                    aux, kp, nullptr, seen_st_nb, s_edge[wave]);
    trace_put<TRACE>(aux.trace, t_first, WAVES, wave, lane, 1);
    __syncthreads();
    {
        const u32 t1 = uniform(s_tk[1]);
        if (t1 < num_tiles) {
            const u64 un = tile_unit<UNITS>(tm, t1, wave);
            if (un != VOID_UNIT) unit_issue(base, lead, end, un, tm.nu, lane, pf, s_edge[wave]);
        }
    }
    if (wave == 0) {
        u32 P, T0, T1, pm;
        tile_aggregate<UNITS>(s_unit[0], lane, P, T0, T1, pm);
        if (lane == 0) {
            desc_store(&desc[t_first], t_first == 0 ? pack_prefix(P, T0) : pack_agg(P, T0, T1));
            s_agg[0][0] = P;
            s_agg[0][1] = T0;
            s_agg[0][2] = T1;
            s_agg[0][3] = pm;
            __hip_atomic_store(&s_ready[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }

    // wave 0: the look-back of T(j).  wait: stay until it is resolved; otherwise one try, false if the tile is not
    // aggregated yet or a descriptor in front of it is still missing.
    auto resolve = [&](u32 j, bool wait) -> bool {
        const int uj = (int)(j & 7u);
        u32 spins = 0;
        while (__hip_atomic_load(&s_ready[uj], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != j + 1u) {
            if (!wait) return false;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 24)) {  // bounded: a bug must not hang the device
                if (lane == 0) report_error(ctl, aux, 0x80000000u);
                break;
            }
        }
        const u32 tj = uniform(s_tk[j & 7u]);
        const u32 P0 = uniform(s_agg[uj][0]), T00 = uniform(s_agg[uj][1]), T01 = uniform(s_agg[uj][2]),
                  pm0 = uniform(s_agg[uj][3]);
        u32 G = 0;
        u64 BASE = 0;
        if (tj != 0) {
            LookBack lb = {(long long)tj - 1, 0, 0, 0};
            u64 win[4];
            lookback_load(desc, lb.j, lane, win);
            spins = 0;
            for (;;) {
                const int r = lookback_eval(win, lb, lane, G, BASE);
                if (r == 1) break;
                if (r == 0) {
                    if (!wait) return false;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (++spins > (1u << 22)) {
                    if (lane == 0) report_error(ctl, aux, 0x80000000u);
                    break;
                }
                lookback_load(desc, lb.j, lane, win);
            }
            if (lane == 0) desc_store(&desc[tj], pack_prefix(G ^ P0, BASE + (G ? T01 : T00)));
        }
        if (lane == 0) {
            u32 *res = s_res[j & 3u];
            res[0] = G;
            res[1] = pm0;
            res[2] = (u32)BASE;
            res[3] = (u32)(BASE >> 32);
            if (tj == num_tiles - 1) {
                eiq_last = (G ^ P0) & 1u;
                __hip_atomic_store(&st->ends_in_quote, eiq_last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __hip_atomic_store(&s_resflag[j & 3u], j + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        trace_put<TRACE>(aux.trace, tj, WAVES, wave, lane, 2);
        return true;
    };

    u32 lb_next = 0;  // wave 0: the first tile whose look-back is not resolved yet
    bool err = false;
    for (u32 it = 0;; it++) {
        // tickets T(it+1) (phase A now) and T(it+2) (its first chunk is loaded at the end of this phase A)
        if (it != 0) {
            u32 spins = 0;
            while (__hip_atomic_load(&s_tkn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < it + 3u) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 24)) {
                    if (lane == 0) report_error(ctl, aux, 0x80000000u);
                    break;
                }
            }
        }
        const u32 ta = uniform(s_tk[(it + 1u) & 7u]), tn = uniform(s_tk[(it + 2u) & 7u]);
        const bool more = ta < num_tiles;
        u32 tk = 0xffffffffu;  // T(it+3): drawn now, it returns while phase A runs
        if (more && tid == 0) tk = atomicAdd(&ctl->tile_counter, 1u);
        if (more) {
            const int ma = (int)((it + 1u) % (u32)DEPTH), ua = (int)((it + 1u) & 7u);
            trace_put<TRACE>(aux.trace, ta, WAVES, wave, lane, 0);
            phase_a<BLOCK, CH, NDJSON, AUX>(base, lead, end, tm, ta, tn, tn < num_tiles, lane, wave, pf, s_mask[ma][wave], s_pre[ma][wave],
                                            s_unit[ua], aux, kp, nullptr, seen_st_nb, s_edge[wave], wave == 0);
            trace_put<TRACE>(aux.trace, ta, WAVES, wave, lane, 1);
            u32 arrived = 0;
            if (lane == 0) arrived = __hip_atomic_fetch_add(&s_arrive[ua], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (uniform(arrived) == (u32)WAVES - 1u) {  // the last wave of the tile
                u32 P1, T10, T11, pm1;
                tile_aggregate<UNITS>(s_unit[ua], lane, P1, T10, T11, pm1);
                if (lane == 0) {
                    desc_store(&desc[ta], pack_agg(P1, T10, T11));
                    s_arrive[ua] = 0;  // next used eight tiles from now
                    s_agg[ua][0] = P1;
                    s_agg[ua][1] = T10;
                    s_agg[ua][2] = T11;
                    s_agg[ua][3] = pm1;
                    __hip_atomic_store(&s_ready[ua], it + 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        if (wave == 0 && lane == 0) {  // (every iteration, so that nobody waits for a ticket that is never drawn)
            s_tk[(it + 3u) & 7u] = tk;
            __hip_atomic_store(&s_tkn, it + 4u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        const bool flat = it + 1u >= D;  // the tile to flatten in this iteration: T(it + 1 - D)
        const u32 f = it + 1u - D;
        const u32 tf = flat ? uniform(s_tk[f & 7u]) : 0u;
        if (flat && tf >= num_tiles) break;  // tickets are monotonic: nothing left in flight
        if (wave == 0) {
            if (flat)
                while (lb_next <= f) (void)resolve(lb_next++, true);
            // one try at the next aggregated tile, a round after its aggregate went out
            if (lb_next <= it && uniform(s_tk[lb_next & 7u]) < num_tiles && resolve(lb_next, false)) lb_next++;
        }
        if (!flat) continue;
        {
            u32 spins = 0;
            while (__hip_atomic_load(&s_resflag[f & 3u], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != f + 1u) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 24)) {
                    if (lane == 0) report_error(ctl, aux, 0x80000000u);
                    break;
                }
            }
        }
        const u32 *res = s_res[f & 3u];
        const int mf = (int)(f % (u32)DEPTH), uf = (int)(f & 7u);
        trace_put<TRACE>(aux.trace, tf, WAVES, wave, lane, 3);
        const u32 G = uniform(res[0]), pm = uniform(res[1]);
        const u64 BASE = ((u64)uniform(res[3]) << 32) | uniform(res[2]);
        u64 tile_end = 0;
        err |= flatten_tile<BLOCK, CH, false>(tm, s_mask[mf][wave], nullptr, s_pre[mf][wave], s_unit[uf], pm, G, BASE, tf, lead, lane,
                                              wave, S1_POS_VIEW(out_pos, pos_cap), pos_cap, tile_end, Arr<u8>(nullptr), len, Arr<u8>(nullptr), aux, nullptr);
        if (tf == num_tiles - 1 && tid == 0) report_total(st, aux, tile_end, eiq_last, base + lead, len);
        trace_put<TRACE>(aux.trace, tf, WAVES, wave, lane, 4);
        if (TRACE && lane == 0)
            aux.trace[((u64)tf * WAVES + wave) * TRACE_WORDS + 5] =
                (u64)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | ((u64)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32);
    }
    if (__ballot(err) != 0 && lane == 0) report_error(ctl, aux, 1u);
}

// ---- launcher --------------------------------------------------------------------------
// Tile shape (BLOCK lanes x CH passes), register budget (WPE = waves per SIMD the allocation must allow) and
// synchronisation scheme (nb: the barrier-free kernel).  SJHIP_S1_VARIANT selects alternatives for A/B runs on
// hardware; the variant can also be set per call (stage1_set_variant, used by the trace entry point).
static constexpr int S1_DEFAULT_VARIANT = 1;

struct S1Variant {
    int block, ch, wpe;
    int depth;  // 0: the barrier kernel; otherwise tiles in flight of the barrier-free kernel
};
static const S1Variant S1_VARIANTS[] = {{512, 2, 4, 0}, {1024, 2, 4, 0}, {768, 2, 3, 0},
                                        {1024, 2, 4, 2}, {1024, 2, 4, 3}, {1024, 1, 4, 0}};
static constexpr int S1_NVARIANTS = (int)(sizeof S1_VARIANTS / sizeof S1_VARIANTS[0]);
static int g_s1_variant = -1;
int stage1_set_variant(int v) {  // -1: back to SJHIP_S1_VARIANT / the default; returns the variant in effect
    if (v >= 0 && v < S1_NVARIANTS) {
        g_s1_variant = v;
    } else {
        const char *e = getenv("SJHIP_S1_VARIANT");
        g_s1_variant = e ? atoi(e) : S1_DEFAULT_VARIANT;
        if (g_s1_variant < 0 || g_s1_variant >= S1_NVARIANTS) g_s1_variant = S1_DEFAULT_VARIANT;
    }
    return g_s1_variant;
}
static S1Variant s1_variant() {
    if (g_s1_variant < 0) stage1_set_variant(-1);
    return S1_VARIANTS[g_s1_variant];
}

// persistent grid: as many blocks as fit on the device at once (more would only queue)
template <typename K>
static u32 grid_for(K kernel, int block, u32 tiles) {
    int dev = 0, cus = 256, per_cu = 1;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    static const int over = getenv("SJHIP_S1_BLOCKS_PER_CU") ? atoi(getenv("SJHIP_S1_BLOCKS_PER_CU")) : 0;
    if (over > 0) per_cu = over;
    const u64 cap = (u64)cus * (u64)per_cu;
    return (u32)(tiles < cap ? tiles : cap);
}

// Tile plan of a message: whole rounds of full tiles, then one short round of small tiles (see TileMap)
struct S1Plan {
    TileMap tm;
    u32 tiles;
};
static u32 s1_block_slots(const S1Variant &v) {  // blocks the device runs at once (the persistent grid)
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    }
    static const int over = getenv("SJHIP_S1_BLOCKS_PER_CU") ? atoi(getenv("SJHIP_S1_BLOCKS_PER_CU")) : 0;
    const int per_cu = over > 0 ? over : (v.block <= 512 ? 2 : 1);  // 16 waves per CU
    return (u32)cus * (u32)per_cu;
}
static S1Plan s1_plan(size_t len, size_t lead) {
    const S1Variant v = s1_variant();
    const u64 units_per_tile = (u64)(v.block / 64) * v.ch;
    const u64 units = ((u64)lead + len + 4095) / 4096;
    const u64 slots = s1_block_slots(v);
    static const bool no_tail = getenv("SJHIP_S1_NO_TAIL") != nullptr;  // A/B: full tiles only
    S1Plan p;
    u64 nf = units / (units_per_tile * slots) * slots;  // whole rounds
    u64 rest = units - nf * units_per_tile;
    u64 su = (rest + slots - 1) / slots;                // at most one small tile per block
    // (round 6: a handful of units behind whole rounds -- less than one per block -- is not worth a round of one-unit tiles, each a
    // full trip through the pipeline: a few blocks take one more full tile instead; 64 MiB 0.0402 -> 0.0384 ms, 256 MiB -1 %)
    if (no_tail || su >= units_per_tile || (nf > 0 && rest < slots)) {  // nearly a whole round anyway: full tiles
        nf += (rest + units_per_tile - 1) / units_per_tile;
        rest = 0;
        su = units_per_tile;
    }
    if (su == 0) su = 1;
    p.tm.nf = (u32)nf;
    p.tm.su = (u32)su;
    p.tm.nu = (u32)units;
    p.tiles = (u32)(nf + (rest + su - 1) / su);
    return p;
}

// workspace: Stage1State | descriptor set 0 | descriptor set 1 (each half of what is left; S1Ws in sj_device.h)
size_t stage1_workspace_bytes(size_t len) {
    const size_t tiles = (len + 128) / (256 * 2 * 64) + 2 + 2048;  // smallest tile of any variant + one round of small tiles
    return sizeof(Stage1State) + 2 * tiles * sizeof(u64) + 64;
}
static u64 *s1_desc_set(const S1Ws &ws, unsigned set) {
    const size_t half = (ws.bytes - sizeof(Stage1State)) / 16;  // descriptors per set
    return reinterpret_cast<u64 *>(reinterpret_cast<u8 *>(ws.p) + sizeof(Stage1State)) + (size_t)set * half;
}

// A workspace whose contents are unknown (just allocated), or a launch that is to stand alone (the timing and tracing entry
// points): everything zero, epoch 0.  zero2 / zero2_bytes: a second region to zero, or null
hipError_t stage1_prepare(S1Ws &ws, hipStream_t stream, void *zero2, size_t zero2_bytes) {
    hipError_t e = hipMemsetAsync(ws.p, 0, ws.bytes, stream);
    if (e == hipSuccess && zero2 && zero2_bytes) e = hipMemsetAsync(zero2, 0, zero2_bytes, stream);
    ws.epoch = 0;
    ws.prev_tiles = 0;
    return e;
}

// words of trace a launch of the current variant writes (sjhip_stage1_trace): tiles x waves x TRACE_WORDS
size_t stage1_trace_words(size_t len, size_t lead, unsigned *tiles_out, int *waves_out) {
    const S1Variant v = s1_variant();
    const u32 tiles = s1_plan(len, lead).tiles;
    if (tiles_out) *tiles_out = tiles;
    if (waves_out) *waves_out = v.block / 64;
    return (size_t)tiles * (size_t)(v.block / 64) * TRACE_WORDS;
}

// d_msg may be any device pointer; ws must hold stage1_workspace_bytes(len + 64) and be clean: zeroed once (stage1_prepare),
// then only used by launches of this function in stream order (sj_device.h Stage1State, S1Ws).
// d_trace (profiling only, plain stage 1 of a non-ND message): stage1_trace_words() zeroed u64.
hipError_t stage1_launch(const void *d_msg, size_t len, int ndjson, u32 *d_pos, size_t pos_cap, S1Ws &ws,
                         hipStream_t stream, void *aux_buf, u8 *d_kind, unsigned long long *h_state, void *zero2,
                         size_t zero2_bytes, unsigned long long *d_trace) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(d_msg);
    const u8 *base = reinterpret_cast<const u8 *>(a & ~(uintptr_t)63);
    const u64 lead = a & 63;
    const S1Plan plan = s1_plan(len, lead);
    const u32 tiles = plan.tiles;
    Stage1State *st = reinterpret_cast<Stage1State *>(ws.p);
    const unsigned par = ws.epoch & 1u;
    u64 *desc = s1_desc_set(ws, par);
    if ((size_t)tiles > (ws.bytes - sizeof(Stage1State)) / 16) return hipErrorInvalidValue;  // (the workspace is too small)
    if (tiles == 0) {  // (nothing is launched: the stage-2 state of an empty message is zeroed the plain way)
        return zero2 && zero2_bytes ? hipMemsetAsync(zero2, 0, zero2_bytes, stream) : hipSuccess;
    }
    const S1Variant v = s1_variant();
    const u32 nd = (u32)((ndjson & 1) != 0);
    S1Aux aux = {};
    aux.kind = nullptr;
    if (d_kind) aux.kind = SJ_ARR(d_kind, pos_cap, A_S1_KIND);
    aux.trace = reinterpret_cast<u64 *>(d_trace);
    aux.host = h_state;
    aux.want_flag = (ndjson & S1_WANT_STARTER_FLAG) != 0;
    aux.par = par;
    aux.clean_desc = s1_desc_set(ws, par ^ 1u);
    aux.clean_tiles = ws.prev_tiles;
    ws.prev_tiles = tiles;
    ws.epoch++;
    aux.zero2 = reinterpret_cast<uint4 *>(zero2);
    aux.zero2_quads = zero2 ? (u64)zero2_bytes / 16 : 0;  // (a multiple of 16 bytes, 16-byte aligned: stage2_zero_bytes)
#if defined(SJ_EXP)
    if (const char *e = getenv("SJHIP_EXP")) aux.exp = (u32)strtoul(e, nullptr, 0);
#endif
    if ((d_kind != nullptr) != (aux_buf != nullptr)) return hipErrorInvalidValue;  // (the whole-parse kernel writes kinds AND masks: no tests on the device)
    if (aux_buf) {
        const StrAux a = str_aux_layout(aux_buf, (size_t)lead + len);
        aux.qm = SJ_ARR(a.qm, a.chunks, A_S1_QM);
        aux.st = SJ_ARR(a.st, a.chunks, A_S1_ST);
        aux.unit_h = SJ_ARR(a.unit_h, a.units, A_S1_UNIT_H);
        aux.unit_slow = SJ_ARR(a.unit_slow, a.units, A_S1_UNIT_SLOW);
        aux.unit_cnt = SJ_ARR(a.unit_cnt, a.units, A_S1_UNIT_CNT);
        aux.unit_str = SJ_ARR(a.unit_str, a.units, A_S1_UNIT_STR);
        aux.tile_unit = SJ_ARR(a.tile_unit, a.units + 1, A_S1_TILE_UNIT);
    }
#define S1_LAUNCHK(K, B)                                                                                            \
    hipLaunchKernelGGL((K), dim3(grid_for(K, B, tiles)), dim3(B), 0, stream, base, lead, (u64)len, d_pos, (u64)pos_cap, \
                       st, desc, tiles, plan.tm, aux)
#define S1_LAUNCH(KERNEL, B, C, W)                                                    \
    do {                                                                              \
        const bool ax = aux_buf || d_kind;                                            \
        if (d_trace) {                                                                \
            if (nd || ax) return hipErrorInvalidValue;                                \
            S1_LAUNCHK((KERNEL<B, C, W, false, false, true>), B);                     \
        } else if (nd) {                                                              \
            if (ax) S1_LAUNCHK((KERNEL<B, C, W, true, true>), B);                     \
            else S1_LAUNCHK((KERNEL<B, C, W, true, false>), B);                       \
        } else {                                                                      \
            if (ax) S1_LAUNCHK((KERNEL<B, C, W, false, true>), B);                    \
            else S1_LAUNCHK((KERNEL<B, C, W, false, false>), B);                      \
        }                                                                             \
    } while (0)
#define S1_LAUNCH_NB(B, C, D, W)                                                            \
    do {                                                                                   \
        if (d_trace) {                                                                     \
            if (nd) return hipErrorInvalidValue;                                           \
            S1_LAUNCHK((stage1_kernel_nb<B, C, D, W, false, false, true>), B);             \
        } else if (nd) {                                                                   \
            S1_LAUNCHK((stage1_kernel_nb<B, C, D, W, true, false>), B);                    \
        } else {                                                                           \
            S1_LAUNCHK((stage1_kernel_nb<B, C, D, W, false, false>), B);                   \
        }                                                                                  \
    } while (0)
    if (aux_buf || d_kind) {  // the whole parse: the barrier kernel in its default shape (the variants are for plain stage 1)
        S1_LAUNCH(stage1_kernel, 1024, 2, 4);
    } else if (v.depth) {
        if (v.depth == 2) S1_LAUNCH_NB(1024, 2, 2, 4);
        else S1_LAUNCH_NB(1024, 2, 3, 4);
    } else {
        if (v.block == 1024 && v.ch == 1) S1_LAUNCH(stage1_kernel, 1024, 1, 4);
        else if (v.block == 1024) S1_LAUNCH(stage1_kernel, 1024, 2, 4);
        else if (v.block == 768) S1_LAUNCH(stage1_kernel, 768, 2, 3);
        else S1_LAUNCH(stage1_kernel, 512, 2, 4);
    }
#undef S1_LAUNCH
#undef S1_LAUNCH_NB
#undef S1_LAUNCHK
    return hipGetLastError();
}

// debug build (-DSJ_DEBUG_BOUNDS, sj_bounds.h): 1 and the record of the out-of-bounds accesses of the stage-1 kernels since the
// last call (cleared); 0 in the product build
int stage1_debug_bounds(unsigned *hits, unsigned *id, unsigned long long *index, unsigned long long *size) {
    *hits = *id = 0;
    *index = *size = 0;
#if defined(SJ_DEBUG_BOUNDS)
    BoundsHit h = {};
    if (hipMemcpyFromSymbol(&h, HIP_SYMBOL(g_bounds_hit), sizeof h) != hipSuccess) return 1;
    if (h.hits) {
        const BoundsHit zero = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bounds_hit), &zero, sizeof zero);
    }
    *hits = h.hits;
    *id = h.id;
    *index = h.index;
    *size = h.size;
    return 1;
#else
    return 0;
#endif
}

}  // namespace sj
