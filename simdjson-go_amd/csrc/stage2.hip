// stage2.hip -- stage 2 (tape build, string unescape, number parse) as follow-on kernels over the
// structural-index array produced by stage 1.  Replaces unifiedMachine
// (stage2_build_tape_amd64.go:160-446), parseString (:72-113), parse_string_amd64.s and
// parseNumber (parse_number.go:65-135).  The per-token logic lives in sj_stage2.h / sj_strings.h /
// sj_number.h / sj_bignum.h (host+device, replayed on the CPU by the test-suite); this file holds the
// kernels, the device-wide scan and the launcher.  No host synchronisation happens between the kernels.
//
// Launches of one parse, behind stage 1 (one token = one structural index), the same seven in both copy modes:
//   k_measure   two independent measuring passes in one launch.  String half: the units with a \u (or invalid) escape get
//               their emit masks escape by escape (GenUnit) as 16-byte records; WithCopyStrings(false): every unit -- the bytes
//               of the strings that hold an escape starter are selected 64 at a time and counted (sj_strings.h chunk_sel).
//               Token half: the kinds of 16 tokens per lane on bit planes (sj_tok16.h) -> one aggregate per 4096-token tile
//   k_scans     the exclusive scans (unit byte counts, unit string counts, tile aggregates) + totals: tape length,
//               Strings.B length, records, brackets
//   k_str_emit  Strings.B: emit mask, escaped characters and opening quotes of a chunk from stage 1's three masks; the
//               emitted (selected) bytes compacted through LDS; the Strings.B offset (and raw length) of the k-th string of
//               the message in soff[] / sinfo[]
//   k_s2_emit_planes  the token pass on bit planes: one packed block scan gives tape offsets, bracket ordinals and queue slots;
//               strings from soff[] / sinfo[] in order, atoms and short integers on the spot, other numbers to a global
//               queue; the tile's brackets matched inside the tile, the rest to the compact view (depth, tape offset, kind,
//               allowed-context set of the gap in front); newlines that separate records leave their tape offset
//   k_numbers   the queued numbers; in the same launch levels 1 and 2 of the 64-ary min tree over the bracket depths
//   k_min_upper the upper levels of the tree
//   k_br_match  previous-smaller-value queries over the compact bracket view for the brackets the tiles left open: partners'
//               tape words, the grammar check of every gap against the type of its container, root words
// plus k_emit_strings (the per-string fallback), k_bignum (on demand) and k_pack (small documents: the result straight into pinned host memory).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "sj_bignum.h"
#include "sj_bounds.h"
#include "sj_chunk.h"
#include "sj_device.h"
#include "sj_number.h"
#include "sj_stage2.h"
#include "sj_strings.h"
#include "sj_tok16.h"

namespace sj {

static constexpr int S2_TILE = 4096;  // tokens per tile (the packed scan form PAgg is sized for it)
static constexpr int RD_BLOCK = 256;  // k_measure
#if !defined(SJ_MEASURE_WAVES)
#define SJ_MEASURE_WAVES 7  // (72 registers, no scratch; 6: 77 registers; 8: 64 + 28 B of scratch -- measured: 7 is 4-6 % faster than 6, 8 no faster)
#endif
static constexpr u32 DLEN_INVALID = 0xffffffffu;
static constexpr u32 DLEN_COPY = 0x80000000u;

// The two device-wide scans that run over small arrays (unit byte counts, tile aggregates) are split over
// SCAN_SEGS blocks, because one CU moves only ~60 GB/s: a block reduces its contiguous segment, publishes the
// aggregate (payload words, drained store counter, flag), adds up the aggregates of the segments in front of it (one lane per
// predecessor; all blocks are resident, so the wait is bounded) and then writes its prefixes.
static constexpr int SCAN_SEGS = 32;
struct alignas(64) SegSlot {
    u32 flag;
    u32 am;
    i32 d;
    u32 nb, bc, pad;
    unsigned long long w, s;  // 64-bit sums: tape words / Strings.B bytes (unit scan: bytes in s)
};
static_assert(sizeof(SegSlot) == 64, "one line per segment");

struct alignas(32) TileAgg {
    Agg a;
    u32 pad[2];
};

// device view of all stage-2 arrays (carved out of one workspace by the launcher)
struct S2Dev {
    Arr<const u8> msg;   // (Arr: sj_bounds.h -- a plain pointer in the product build, bounds-checked under -DSJ_DEBUG_BOUNDS)
    u64 len;
    Arr<const u32> pos;
    u32 n;         // tokens -- or, with n_dev, an upper bound the arrays are sized for
    const unsigned long long *n_dev;  // null, or the token count on the device (Stage1State::total): the host has not
                                      // waited for stage 1 (small documents: one synchronisation per parse)
    u32 ndjson, copy_strings;
    const u32 *s1_has_starter;  // Stage1State::has_starter on the device, or null (see no_escapes below)
    Arr<const u8> kind;  // [n] token kinds (stage 1 writes them next to the positions)
    Arr<u32> dlen;     // [n] selective copy only: unescaped length | DLEN_COPY, or DLEN_INVALID
    Arr<u32> str_off;  // [n] selective copy only: Strings.B offset of a copied string
    Arr<u32> nl_off;   // [n] tape offset of the r-th record-separating newline
    Arr<uint4> numq;   // [n/2] (message offset lo, hi, tape offset, -) of the numbers k_numbers has to parse, in no particular order
    Arr<uint4> bigq;   // [n/2] ... of those that need the big-integer tie-break  (a number is followed by a token that is none:
    u32 numq_cap;      //       a document that parses holds at most n/2; entries beyond the room are dropped, the parse fails anyway)
    Arr<const u32> tile_unit;  // [units + 1] the unit that holds token 4096 T (stage 1): the 32-bit positions wrap in a message of more
                               // than 4 GiB; a tile's true positions = position of its first token (from its unit) + wrapped differences
    Arr<uint2> sinfo;  // [soff_cap] WithCopyStrings(false) on the masks, for the k-th string of the message: (Strings.B offset it has
                       // if it is copied, raw length of its content); behind the last one (length of Strings.B, 0)  (k_str_emit)
    Arr<u32> unit_tq;  // [units] ... and for a unit that ends inside a string: where that string's closing quote lies (k_measure)
    Arr<i32> br_depth; // [n] compact bracket view: depth after the c-th bracket (level 0 of the min tree)
    Arr<u32> br_off;   // [n]                       its tape offset
    Arr<u8> br_info;   // [n]                       kind | allowed contexts of the gap that ends with it << 4
    Arr<TileAgg> agg;  // [tiles] aggregates, then (k_s2_scan_tiles) exclusive prefixes
    u32 tiles;
    // min tree levels 1.. (level 0 is br_depth[])
    Arr<i32> lev[MinTree::MAXLEV];
    u64 lev_size[MinTree::MAXLEV];
    int nlev;
    S2State *st;
    SegSlot *seg_units, *seg_tiles, *seg_ustr;  // [SCAN_SEGS] segment aggregates of the multi-block scans (zeroed with st)
    int unit_segs, tile_segs;        // blocks of the unit scan / of the tile scan: SCAN_SEGS, or 1 where one block is through in
                                     // a single short round (<= 16 384 units, <= 1024 tiles): nothing to publish or wait for
    Arr<u64> tape;
    Arr<u8> strings;
    Arr<u8> keyflag;   // null, or [tape_cap / 2 + 8]: [tape offset of a string entry >> 1] = 1 iff the string is an object key
                       // (SJHIP_FLAG_KEY_FLAGS: marshal.hip reads them instead of recovering them from the token kinds)
    Arr<u8> str_out;   // where k_str_emit writes: `strings` (both copy modes: with WithCopyStrings(false) it only compacts the bytes of
                   // the strings that are copied)
    u64 tape_cap, strings_cap;
    u64 tape_base, strings_base, msg_base;  // NDJSON shard: rebasing of every stored index (0 if unsharded)
    // byte-parallel string path (copy_strings): masks from stage 1 and what the string kernels derive from them
    StrView sv;           // base / lead / end / qm q st unit_h (null qm: path not used)
    Arr<ChunkRec> rec;        // [chunks] emit mask + emitted bytes of the unit in front of the chunk (+ patch flag): only the units with
                              //          a \u or invalid escape (k_measure's general routine) have them
    Arr<u32> unit_cnt;        // [units]  emitted bytes of the unit, then (k_str_scan) their exclusive prefix
    Arr<u32> unit_str;        // [units]  strings that begin in the unit, then their exclusive prefix
    Arr<u32> soff;            // [soff_cap] every string copied: Strings.B offset of the k-th string of the message, and behind
    u32 soff_cap;             //          the last one the length of Strings.B (k_str_emit; the storage is dlen's and str_off's)
    Arr<u8> unit_copy;        // [units]  WithCopyStrings(false) only (else null): the states of the selective copy at the unit's ends
                              //          (USEL_*, k_measure -> k_str_emit)
    u64 units;
    u32 exp;  // SJ_EXP builds only: bit mask of parts to leave out (A/B timing of the kernels' parts; results are wrong)
};
#if defined(SJ_EXP)
#define SJ_EXPBIT(p, b) ((((p).exp >> (b)) & 1u) != 0)
#else
#define SJ_EXPBIT(p, b) false
#endif
// (clamped to the layout: a small document denser than the host assumed stays in bounds and is parsed again, parse_api.hip)
__device__ __forceinline__ u32 token_count(const S2Dev &p) {
    if (!p.n_dev) return p.n;
    const unsigned long long n = *p.n_dev;
    return n < p.n ? (u32)n : p.n;
}

// WithCopyStrings(false) of a message that holds no escape starter at all (round 6; parking-citations, most machine-written
// NDJSON): parseString copies a string only if unescaping changed it (parse_string_amd64.go:33-42), so NOTHING is copied, Strings.B
// is empty, every string word points into the message -- and the raw length of a string equals its unescaped length, i.e. the
// difference of two offsets of the full compaction (what copy mode computes from stage 1's unit counts without touching the
// message).  The selective machinery is skipped: no string half in k_measure (it visited every unit: 90 us on configs[4]),
// k_str_emit<true> only numbers the strings (soff[], 4 bytes each, no message read, no compaction), the emit pass reads 4 instead
// of 8 bytes per string.  The flag is stage 1's (one atomic per wave that saw a starter); stage 2 is queued before the host
// knows it, so every kernel asks on the device (grid-uniform).
__device__ __forceinline__ bool no_escapes(const S2Dev &p) {
    return !p.copy_strings && p.s1_has_starter != nullptr && *p.s1_has_starter == 0u;
}
// inclusive sum over the 64 lanes of a wave with DPP row shifts / broadcasts (six VALU instructions, no LDS crossbar)
__device__ __forceinline__ u32 wave_incl_sum(u32 v) {
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);  // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);  // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);  // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);  // row_bcast:15 -> rows 1, 3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);  // row_bcast:31 -> rows 2, 3
    return v;
}

// ---- string kernels (copy_strings): sj_strings.h, one 64-byte chunk per lane, one 4 KiB unit per wave ---------
// The general string routine of a unit, escape by escape instead of chunk by chunk.  sj_strings.h str_chunk_masks is the
// per-chunk statement (and what the host replay runs): a lane walks the escaped characters of its chunk one after the
// other, ~250 instructions and a handful of dependent loads each -- twitterescaped.json holds up to thirteen \u escapes
// per chunk, so a wave spent 57 us on one unit with most of its time in a single lane's serial chain.  Here the escaped
// characters of the unit's general chunks are listed in LDS and the lanes take them round robin: every escape is
// evaluated once, by any lane.  What an escape changes in the emit masks is order-independent: the fast formula has
// the bits of u,X,X,X,X set (they are plain in-string bytes) and a valid escape that emits n bytes CLEARS the bits
// j >= n -- of its own chunk or the next one -- with an LDS atomic AND; overlapping escapes only exist in documents that
// are rejected anyway.  Escapes whose 'u' lies in the last four bytes of the chunk in front of the unit reach into
// chunk 0: lane 0 lists them as foreign items (their owner reports their errors).
struct GenUnit {       // per wave
    u64 em[64];        // emit masks of the unit's chunks
    uint16_t list[2048 + 4];  // escaped characters: byte offset inside the unit; foreign items: 0x8000 | (0..3)
};
// le: escaped in-string characters of this lane's chunk that have to be evaluated (0: none); returns the wave's item count
__device__ __forceinline__ u32 gen_unit_list(GenUnit *gu, u64 le, u64 foreign, int lane) {
    const u32 n = (u32)popc64(le) + (lane == 0 ? (u32)popc64(foreign) : 0u);
    const u32 incl = wave_incl_sum(n);
    u32 o = incl - n;
    if (lane == 0)
        for (u64 f = foreign; f != 0; f &= f - 1) gu->list[o++] = (uint16_t)(GEN_FOREIGN | (u32)ctz64(f));
    for (u64 r = le; r != 0; r &= r - 1) gu->list[o++] = (uint16_t)((u32)lane * 64u + (u32)ctz64(r));
    return (u32)__shfl((int)incl, 63, 64);
}

// The unescaped quotes of this lane's chunk from the in-string masks of a whole unit, one chunk per lane, every lane active
// (sj_strings.h StrView::quotes: stage 1 does not store them -- a quote sits where the unit-relative mask changes; the bit in
// front of a chunk is the last bit of the lane below, 0 in front of lane 0)
__device__ __forceinline__ u64 wave_quotes(u64 qm) {
    const u32 prev_hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(qm >> 32), 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
    return qm ^ ((qm << 1) | (u64)(prev_hi >> 31));
}

// ---- WithCopyStrings(false): which strings are copied, decided 64 bytes at a time (sj_strings.h chunk_sel) ---------------
// The state at a unit's ends: does the string that is open at the start of unit u hold an escape starter in FRONT of the
// unit (the starters behind the last quote of the nearest unit in front that holds a quote, and all starters of the units
// in between), does the one open at its end hold one BEHIND it.  The whole wave walks; a string that stays open for more
// than SEL_WALK_CAP steps gives the document to the per-string path (S2_ERR_SERIAL_STRINGS, like a surrogate run that is too long).
static constexpr u32 SEL_WALK_CAP = 64;  // steps of 64 units: strings of up to 16 MiB
// Units without a quote lie wholly inside the open string: the walk passes over them 64 at a time on the unit flags stage 1
// leaves (unit_h: bit 1 "holds an escape starter", bit 2 "holds an unescaped quote"), and looks only at the masks of the
// one unit that holds the quote it is after.
__device__ __forceinline__ u32 sel_unit_in(const S2Dev &p, u64 u, int lane, bool &giveup) {
    u32 acc = 0;
    for (u32 step = 0; u > 0; step++) {
        if (step >= SEL_WALK_CAP) {
            giveup = true;
            return acc;
        }
        // lane l looks at unit u - 1 - l
        const bool have = (u64)lane < u;
        const u32 f = have ? (u32)p.sv.unit_h[u - 1 - (u64)lane] : 0u;
        const u64 qb = __ballot(have && (f & 4u)), sb = __ballot(have && (f & 2u));
        if (qb == 0) {  // 64 more units (or all that are left) inside the string
            acc |= sb != 0 ? 1u : 0u;
            if (u <= 64) break;
            u -= 64;
            continue;
        }
        const int near = __builtin_ctzll(qb);  // the nearest unit in front with a quote; the ones in between hold none
        acc |= (sb & ((1ull << near) - 1ull)) != 0 ? 1u : 0u;
        const u64 v = u - 1 - (u64)near;
        const u64 c = v * 64 + lane;
        const u64 q = wave_quotes(p.sv.qm[c]);
        const u64 st = (sb >> near) & 1ull ? p.sv.st[c] : 0ull;
        const u64 qc = __ballot(q != 0);
        if (qc != 0) {  // (always: the flag says so)
            const int L = 63 - __builtin_clzll(qc);  // the last chunk with a quote; behind its last quote the string is open
            bool m = lane > L && st != 0;
            if (lane == L) {
                const int hb = 63 - clz64(q);
                m = hb < 63 && (st >> (hb + 1)) != 0;
            }
            acc |= __ballot(m) != 0 ? 1u : 0u;
        }
        break;
    }
    return acc;
}
// (*tq: the aligned offset of that string's closing quote -- the first quote behind the unit)
__device__ __forceinline__ u32 sel_unit_out(const S2Dev &p, u64 u, int lane, bool &giveup, u32 *tq) {
    u32 acc = 0;
    u64 w = u + 1;  // the first unit behind
    for (u32 step = 0; w < p.units; step++) {
        if (step >= SEL_WALK_CAP) {
            giveup = true;
            return acc;
        }
        const bool have = w + (u64)lane < p.units;  // lane l looks at unit w + l
        const u32 f = have ? (u32)p.sv.unit_h[w + (u64)lane] : 0u;
        const u64 qb = __ballot(have && (f & 4u)), sb = __ballot(have && (f & 2u));
        if (qb == 0) {
            acc |= sb != 0 ? 1u : 0u;
            w += 64;
            continue;
        }
        const int near = __builtin_ctzll(qb);
        acc |= (sb & ((1ull << near) - 1ull)) != 0 ? 1u : 0u;
        const u64 v = w + (u64)near;
        const u64 c = v * 64 + lane;
        const u64 q = wave_quotes(p.sv.qm[c]);
        const u64 st = (sb >> near) & 1ull ? p.sv.st[c] : 0ull;
        const u64 qc = __ballot(q != 0);
        if (qc != 0) {
            const int L = __builtin_ctzll(qc);  // the first chunk with a quote: the closing quote of the open string
            bool m = lane < L && st != 0;
            const int lb = q ? ctz64(q) : 0;
            if (lane == L) m = lb > 0 && (st & ((1ull << lb) - 1ull)) != 0;
            *tq = (u32)((v * 64 + (u64)L) * 64) + (u32)__builtin_amdgcn_readlane(lb, L);
            acc |= __ballot(m) != 0 ? 1u : 0u;
        }
        break;
    }
    return acc;
}
// The selected bytes of this lane's chunk: the two scans over the wave's 64 chunks (function composition, sel_then) applied
// to the states at the unit's ends.
__device__ __forceinline__ u64 sel_wave_mask(const ChunkSel &cs, u32 fin, u32 gout, int lane) {
    // A scan element is x -> (x & a) | b on ONE bit of state: "b generates, a propagates" -- the carry chain of an addition.  With the
    // a and b bits of the 64 chunks in two 64-bit ballots, the state entering every chunk is the carry into its bit of b + (a | b) + fin
    // (carry out of bit i = b_i | c_i (a_i | b_i) = b_i | c_i a_i), i.e. sum ^ b ^ (a | b): four ballots and a dozen scalar
    // instructions for both directions (the backward chain on the bit-reversed words).  Until the end of round 6 these were two
    // Hillis-Steele scans by function composition (sj_strings.h sel_then; still what the host replay runs): twelve dependent
    // cross-lane shuffles per unit, in k_measure and again in k_str_emit.
    const u64 fa = __ballot((cs.fwd & 1u) != 0), fb = __ballot((cs.fwd & 2u) != 0);
    const u64 ga = brev64(__ballot((cs.bwd & 1u) != 0)), gb = brev64(__ballot((cs.bwd & 2u) != 0));
    const u64 fp = fa | fb, gp = ga | gb;
    const u64 fc = (fb + fp + (u64)(fin & 1u)) ^ fb ^ fp;            // bit i: the state at the start of chunk i
    const u64 gc = brev64((gb + gp + (u64)(gout & 1u)) ^ gb ^ gp);   // bit i: the state at the end of chunk i (coming from behind)
    (void)lane;
    return chunk_sel_mask(cs, (u32)(fc >> lane) & 1u, (u32)(gc >> lane) & 1u);
}
// unit flags of the selective copy (S2Dev::unit_copy): the states at the unit's ends, left by k_measure for k_str_emit
static constexpr u8 USEL_IN = 1u, USEL_OUT = 2u;

// SEL = WithCopyStrings(false).
template <bool SEL>
__device__ __forceinline__ void str_masks_body(const S2Dev &p, u32 block, u32 nblocks, GenUnit *gu) {
    // Without touching the message (sj_strings.h str_chunk_masks_fast is the per-chunk statement): a chunk takes the
    // general routine only if it, or the chunk in front of it, holds an escaped character that no simple escape names.
    // Persistent waves: unit wave_id, wave_id + waves, ...
    const int lane = threadIdx.x & 63;
    const u64 nwaves = (u64)nblocks * 4;
    u64 unit = (u64)block * 4 + (u64)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (unit >= p.units) return;  // whole waves
    // Every string copied: stage 1 has left the count of every unit and k_str_emit derives the emit masks itself; what is
    // left for this pass are the units that hold an escaped character no simple escape names -- a \u, whose bytes may
    // reach into the next chunk, or an invalid escape -- which unit_slow flags: their emit masks are computed here, escape
    // by escape, and left as records.  WithCopyStrings(false): every unit is visited -- the bytes of the strings that are
    // copied (the ones with an escape starter) are counted per unit, and the states at the unit's ends are left for k_str_emit.
    for (;; unit += nwaves) {
        if (unit >= p.units) return;
        const u64 sw = p.sv.unit_slow[unit], sp = unit ? p.sv.unit_slow[unit - 1] : 0ull;
        const bool slow = sw != 0 || (sp >> 63) != 0;  // (wave-uniform)
        if (!SEL && !slow) continue;
        const u64 c = unit * 64 + lane;
        const u32 uh = p.sv.unit_h[unit], uhp = unit ? (u32)p.sv.unit_h[unit - 1] : 0u;
        const u32 h = uh & 1u;
        if (SEL && !slow && !(uh & 2u)) {  // (uniform) no escape starter in the unit: nothing of it is copied unless a string that
            // crosses one of its ends holds a starter elsewhere -- decided on the unit flags alone, no mask of this unit is read
            // (parking-citations: every unit).  "Open at the start" = the state stage 1 left (a superset of the predicate
            // below -- it also holds when byte 0 closes the string, where the answer changes nothing); "open at the end" = the
            // state at the start of the next unit, the same bit the masks give.
            const bool open_in = h != 0, open_out = unit + 1 < p.units && (p.sv.unit_h[unit + 1] & 1u) != 0;
            bool giveup = false;
            const u32 fin = open_in ? sel_unit_in(p, unit, lane, giveup) : 0u;
            u32 tq = 0;
            const u32 gout = open_out ? sel_unit_out(p, unit, lane, giveup, &tq) : 0u;
            if (!fin && !gout) {
                if (lane == 0) {
                    p.unit_cnt[unit] = 0;
                    p.unit_copy[unit] = 0;
                    p.unit_tq[unit] = tq;
                    if (giveup) atomicOr(&p.st->err, S2_ERR_SERIAL_STRINGS);
                }
                continue;
            }
        }
        const u64 qm = p.sv.qm[c], q = wave_quotes(qm);
        const u64 st = (uh & 2u) ? p.sv.st[c] : 0ull;
        const u64 stp = (c && ((lane ? uh : uhp) & 2u)) ? p.sv.st[c - 1] >> 63 : 0ull;
        const ChunkFast cf = chunk_fast(qm, q, st, stp << 63, h);
        const u64 e = cf.esc;  // escaped characters inside strings
        u64 em = cf.em;
        if (slow) {  // (wave-uniform)
            const bool prev_slow = lane ? ((sw >> (lane - 1)) & 1u) != 0 : (sp >> 63) != 0;
            u32 flags = e != 0 ? CHUNK_SLOW : 0u;
            const bool own_slow = ((sw >> lane) & 1u) != 0 && e != 0;
            const bool gen = own_slow || prev_slow;  // == str_chunk_needs_general(c)
            if (__ballot(gen) != 0) {  // (wave-uniform) the unit takes the general routine
                // escaped in-string characters in the last four bytes of the chunk in front (they may reach into this one)
                u64 pe = (u64)__shfl_up((long long)e, 1, 64) >> 60;
                if (lane == 0) pe = c ? (p.sv.esc(c - 1) & p.sv.sm(c - 1)) >> 60 : 0ull;
                gu->em[lane] = em;
                const u32 items = gen_unit_list(gu, own_slow ? e : 0ull, (lane == 0 && gen) ? pe : 0ull, lane);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const u64 u0 = unit * 4096;
                bool bad = false, over = false;
                for (u32 j = (u32)lane; j < items; j += 64)
                    gen_item_masks(p.sv, u0, gu->list[j], &bad, &over,
                                   [&](u32 pos) { atomicAnd(&gu->em[pos >> 6], ~(1ull << (pos & 63))); });
                if (bad) atomicOr(&p.st->err, 1u);
                if (over) atomicOr(&p.st->err, S2_ERR_SERIAL_STRINGS);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                if (gen) {
                    em = gu->em[lane];
                    flags = (e != 0 || pe != 0) ? (CHUNK_SLOW | CHUNK_GENERAL) : 0u;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();  // the lists are read: the next unit may write them
            }
            const u32 n = (u32)popc64(em);
            const u32 incl = wave_incl_sum(n);
            p.rec[c] = ChunkRec{em, (incl - n) | flags, 0u};
            if (!SEL && lane == 63) p.unit_cnt[unit] = incl;
        }
        if (SEL) {
            const ChunkSel cs = chunk_sel(qm, q, st, h);
            bool giveup = false;
            // (a string is open at the start of the unit / at its end: only then do the neighbours matter)
            const bool open_in = (u32)__builtin_amdgcn_readlane((int)(u32)(cs.in & ~cs.oq & 1ull), 0) != 0;
            const bool open_out = unit + 1 < p.units && (u32)__builtin_amdgcn_readlane((int)(u32)(cs.in >> 63), 63) != 0;
            const u32 fin = open_in ? sel_unit_in(p, unit, lane, giveup) : 0u;
            u32 tq = 0;
            const u32 gout = open_out ? sel_unit_out(p, unit, lane, giveup, &tq) : 0u;
            u32 total = 0;
            if ((uh & 2u) || fin || gout) {  // (uniform) without a starter in reach nothing of the unit is copied
                const u64 sel = sel_wave_mask(cs, fin, gout, lane);
                total = (u32)__shfl((int)wave_incl_sum((u32)popc64(em & sel)), 63, 64);
            }
            if (lane == 0) {
                p.unit_cnt[unit] = total;
                p.unit_copy[unit] = (u8)((fin ? USEL_IN : 0u) | (gout ? USEL_OUT : 0u));
                p.unit_tq[unit] = tq;
                if (giveup) atomicOr(&p.st->err, S2_ERR_SERIAL_STRINGS);
            }
        }
    }
}

// ---- wave scans of full-width aggregates (DPP) ------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Agg agg_dpp(const Agg &v) {  // lanes without a source read the identity
    return Agg{__builtin_amdgcn_update_dpp(0, v.d, CTRL, ROW_MASK, 0xf, false),
               (u32)__builtin_amdgcn_update_dpp(0, (int)v.w, CTRL, ROW_MASK, 0xf, false),
               (u32)__builtin_amdgcn_update_dpp(0, (int)v.s, CTRL, ROW_MASK, 0xf, false),
               (u32)__builtin_amdgcn_update_dpp(0, (int)v.nb, CTRL, ROW_MASK, 0xf, false),
               (u32)__builtin_amdgcn_update_dpp(0, (int)v.bc, CTRL, ROW_MASK, 0xf, false),
               (u32)__builtin_amdgcn_update_dpp((int)AM_ALL, (int)v.am, CTRL, ROW_MASK, 0xf, false)};
}
__device__ __forceinline__ Agg agg_readlane(const Agg &v, int l) {
    return Agg{__builtin_amdgcn_readlane(v.d, l),
               (u32)__builtin_amdgcn_readlane((int)v.w, l),
               (u32)__builtin_amdgcn_readlane((int)v.s, l),
               (u32)__builtin_amdgcn_readlane((int)v.nb, l),
               (u32)__builtin_amdgcn_readlane((int)v.bc, l),
               (u32)__builtin_amdgcn_readlane((int)v.am, l)};
}
__device__ __forceinline__ Agg agg_row_scan(Agg v) {  // inclusive inside rows of 16 lanes
    v = agg_combine(agg_dpp<0x111, 0xf>(v), v);
    v = agg_combine(agg_dpp<0x112, 0xf>(v), v);
    v = agg_combine(agg_dpp<0x114, 0xf>(v), v);
    v = agg_combine(agg_dpp<0x118, 0xf>(v), v);
    return v;
}
__device__ __forceinline__ Agg agg_wave_inclusive(Agg v) {
    v = agg_row_scan(v);
    v = agg_combine(agg_dpp<0x142, 0xa>(v), v);  // row_bcast:15 -> rows 1, 3
    v = agg_combine(agg_dpp<0x143, 0xc>(v), v);  // row_bcast:31 -> rows 2, 3
    return v;
}

// ---- segment publish / look-back shared by the two multi-block scans -------------------------------------------
struct SegSum {
    Agg a;                    // 32-bit fields (wrap like the scan itself)
    unsigned long long w, s;  // true 64-bit sums
};
__device__ __forceinline__ void seg_publish(SegSlot *slot, const SegSum &v) {
    __hip_atomic_store(&slot->am, v.a.am, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&slot->d, v.a.d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&slot->nb, v.a.nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&slot->bc, v.a.bc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&slot->w, v.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&slot->s, v.s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // Payload before flag (MI355X guide, "handoff-flag": sc1 payload -> drained vmcnt -> sc1 flag): every word above is
    // an agent-scope store (sc1: performed at the memory side, not kept in this XCD's L2), so once the wave's store
    // counter has drained they are visible to every CU, and only then is the flag store issued.  The wait is inline
    // asm on purpose: the compiler neither emits it for relaxed stores nor may it drop it.  The reader polls the flag
    // and issues its (sc1, L1-bypassing) payload loads after the flag load has returned.  An agent-scope release
    // fence instead would write the whole L2 of this XCD back (~100 us when every block of a large kernel does it).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(&slot->flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// called by one whole wave of block `seg`: the ordered sum of segments 0 .. seg-1 (identity for seg 0)
__device__ __forceinline__ SegSum seg_lookback(SegSlot *slots, int seg, int lane, S2State *st) {
    SegSum r;
    r.a = agg_identity();
    r.w = r.s = 0;
    static_assert(SCAN_SEGS <= 64, "one lane per predecessor");
    if (lane < seg) {
        u32 spins = 0;
        while (__hip_atomic_load(&slots[lane].flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(16);  // ~0.4 us between polls: the line is also the target of the publishers' stores
            if (++spins > (1u << 20)) {  // bounded: a bug must not hang the device
                atomicOr(&st->err, 8u);
                break;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the flag load has returned before the payload loads are issued
        r.a.am = __hip_atomic_load(&slots[lane].am, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r.a.d = __hip_atomic_load(&slots[lane].d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r.a.nb = __hip_atomic_load(&slots[lane].nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r.a.bc = __hip_atomic_load(&slots[lane].bc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r.w = __hip_atomic_load(&slots[lane].w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r.s = __hip_atomic_load(&slots[lane].s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r.a.w = (u32)r.w;
        r.a.s = (u32)r.s;
    }
    r.a = agg_readlane(agg_wave_inclusive(r.a), 63);  // lanes >= seg hold the identity
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) {
        r.w += (unsigned long long)__shfl_xor((long long)r.w, sh, 64);
        r.s += (unsigned long long)__shfl_xor((long long)r.s, sh, 64);
    }
    return r;
}
// contiguous share of `n` items for segment `seg`, in whole multiples of `quantum`
__device__ __forceinline__ void seg_range(u64 n, u64 quantum, int seg, int segs, u64 &lo, u64 &hi) {
    const u64 per = ((n + (u64)segs - 1) / (u64)segs + quantum - 1) / quantum * quantum;
    lo = (u64)seg * per;
    hi = lo + per;
    if (lo > n) lo = n;
    if (hi > n) hi = n;
}

// ---- exclusive scan of the unit byte counts (in place) + Strings.B length: SCAN_SEGS blocks of 1024 threads ------
// 16 units per thread and round with 16-byte accesses.
__device__ __forceinline__ void unit_round_load(Arr<const u32> data, u64 i, u64 hi, u32 (&v)[16]) {
    if (i + 15 < hi) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 x = *reinterpret_cast<const uint4 *>(arr_at(data, i + 4 * q, 4));
            v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = i + q < hi ? data[i + q] : 0u;
    }
}
// (STR: the units' string counts instead -- every string copied: k_str_emit numbers the strings of the message with them)
template <bool STR>
__device__ __forceinline__ void str_scan_body(const S2Dev &p, int seg) {
    __shared__ u32 s_wave[2][16];
    __shared__ unsigned long long s_sum[16], s_prefix;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Arr<u32> data = STR ? p.unit_str : p.unit_cnt;
    SegSlot *const slots = STR ? p.seg_ustr : p.seg_units;
    u64 lo, hi;
    seg_range(p.units, 1024, seg, p.unit_segs, lo, hi);
    // pass 1: the segment's byte count
    unsigned long long mine = 0;
    for (u64 start = lo; start < hi; start += 16384) {
        u32 v[16];
        unit_round_load(data, start + (u64)tid * 16, hi, v);
#pragma unroll
        for (int q = 0; q < 16; q++) mine += v[q];
    }
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) mine += (unsigned long long)__shfl_xor((long long)mine, sh, 64);
    if (lane == 0) s_sum[wave] = mine;
    __syncthreads();
    if (wave == 0) {
        unsigned long long tot = 0;
        for (int w = 0; w < 16; w++) tot += s_sum[w];
        SegSum own;
        own.a = agg_identity();
        own.w = 0;
        own.s = tot;
        if (lane == 0) seg_publish(&slots[seg], own);
        const SegSum before = seg_lookback(slots, seg, lane, p.st);
        if (lane == 0) {
            s_prefix = before.s;
            if (seg == p.unit_segs - 1) {
                if (STR) p.st->n_strings = (u32)(before.s + tot);
                else {
                    // (no_escapes: the counts are the full compaction's -- they only measure the strings; nothing is copied)
                    p.st->strings_len_masks = no_escapes(p) ? 0ull : before.s + tot;
                    if (before.s + tot > 0xfffffff0ull) atomicOr(&p.st->err, 4u);  // (Strings.B offsets are 32 bits wide)
                }
            }
        }
    }
    __syncthreads();
    // pass 2: exclusive prefixes in place
    u64 carry = s_prefix;  // the same in every thread
    int buf = 0;
    for (u64 start = lo; start < hi; start += 16384, buf ^= 1) {
        const u64 i = start + (u64)tid * 16;
        u32 v[16];
        unit_round_load(data, i, hi, v);
        u32 t = 0;
#pragma unroll
        for (int q = 0; q < 16; q++) t += v[q];
        u32 incl = t;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const u32 o = __shfl_up(incl, sft, 64);
            if (lane >= sft) incl += o;
        }
        if (lane == 63) s_wave[buf][wave] = incl;
        __syncthreads();  // one barrier per round: the wave totals alternate between two buffers
        u32 before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const u32 x = s_wave[buf][w];
            before += w < wave ? x : 0u;
            total += x;
        }
        u32 run = (u32)carry + before + incl - t;  // positions are < 2^32
        if (i + 15 < hi) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint4 o;
                o.x = run; run += v[4 * q];
                o.y = run; run += v[4 * q + 1];
                o.z = run; run += v[4 * q + 2];
                o.w = run; run += v[4 * q + 3];
                *reinterpret_cast<uint4 *>(arr_at(data, i + 4 * q, 4)) = o;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                if (i + q < hi) data[i + q] = run;
                run += v[q];
            }
        }
        carry += total;
    }
}

// selector of v_perm_b32 that moves the bytes named by the 4-bit mask `nib` to the low end of a dword (zeros above)
constexpr u32 compress_selector(u32 nib) {
    u32 sel = 0x0c0c0c0cu;
    int o = 0;
    for (u32 b = 0; b < 4; b++)
        if ((nib >> b) & 1u) {
            sel = (sel & ~(0xffu << (8 * o))) | (b << (8 * o));
            o++;
        }
    return sel;
}
struct SelLut {
    u32 v[16];
};
constexpr SelLut make_sel_lut() {
    SelLut t{};
    for (u32 x = 0; x < 16; x++) t.v[x] = compress_selector(x);
    return t;
}
__constant__ SelLut c_sel = make_sel_lut();
__constant__ EscapeLut c_esc = make_escape_lut();

// Pass 2 of the string path: one 4 KiB unit per wave, one 64-byte chunk per lane.  A chunk is compacted eight
// bytes at a time: two v_perm_b32 squeeze the emitted bytes together and aligned 8-byte LDS atomics merge them into the
// wave's window (below).  Chunks with escapes first patch the translated bytes into their LDS copy of the chunk (sj_strings.h).
// Waves are persistent and software-pipelined: a wave walks over units wave_id, wave_id + waves, ...; while it compacts
// unit i its loads of the 64-byte chunks of unit i+1 and of the masks (records) of unit i+2 are in flight (one unit per
// block left two dependent memory round trips -- record, then chunk -- exposed in front of every 4 KiB).
//
// There are no records for ordinary units (second half of round 5).  The wave streams the three masks stage 1 left (24 B
// per chunk, read here for the first and only time) and derives what a record held -- emit mask, escaped characters, the
// chunk's offset inside the unit (one packed wave scan) -- and, new, the OPENING QUOTES: string number k of the message (k-th
// opening quote = k-th string token) starts at Strings.B offset soff[k] = E(position of its quote), and since Strings.B is
// the concatenation of the strings its unescaped length is soff[k + 1] - soff[k].  The wave writes soff[] for the quotes of
// its unit (unit_str: exclusive prefix of the units' string counts, k_scans); k_s2_emit_planes reads it in order and does
// not touch masks or records any more (it gathered two 16-byte records per string).  Only units with a \u (or invalid)
// escape -- whose emit masks k_measure computes escape by escape -- come with records (unit_slow, the predicate of
// str_masks_body).
// SEL (WithCopyStrings(false)): the emit mask is restricted to the bytes of the strings that are copied -- the ones that
// hold an escape starter (sj_strings.h chunk_sel; the states at the unit's ends come from k_measure) -- so Strings.B is
// again a plain compaction and nothing else is read or written: chunks without a selected byte are not loaded.  A string
// that is not copied needs its raw length: the wave also leaves the position of every CLOSING quote under the number of
// its string (scq[]).  (Rounds 3-4 compacted every string into a scratch buffer, measured every string token with two record
// gathers and copied the changed strings out of the scratch: k_str_measure + k_emit_strings, 165 us on configs[1].)
template <bool SEL>
__device__ __forceinline__ void str_emit_body(const S2Dev &p) {
    // One LDS window per wave, used twice: chunks with escapes park their dwords there (dword-major: bank = lane)
    // to patch bytes, and once every lane has its (patched) chunk back in registers the window receives the
    // unit's unescaped bytes.
    __shared__ __attribute__((aligned(16))) u8 s_io[4][4096 + 16];
    __shared__ GenUnit s_gu[4];  // (only .list is used here)
    __shared__ u32 s_sel[16];
    __shared__ u8 s_esc[256];
    if (threadIdx.x < 16) s_sel[threadIdx.x] = c_sel.v[threadIdx.x];
    s_esc[threadIdx.x] = c_esc.v[threadIdx.x];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (uniform: the unit's scalars stay in SGPRs)
    const bool NE = SEL && no_escapes(p);  // nothing is copied: the strings are only numbered (offsets of the full compaction in soff[])
    // (SJ_EXP bit 20: the units from the last to the first; every comparison below is on u64 and a unit in front of 0 is "behind the end")
    const bool REV = SJ_EXPBIT(p, 20);
    const u64 nwaves = REV ? (u64)0 - (u64)gridDim.x * 4 : (u64)gridDim.x * 4;
    u64 unit = REV ? p.units - 1 - ((u64)blockIdx.x * 4 + wave) : (u64)blockIdx.x * 4 + wave;
    struct Raw {   // what is requested two units ahead
        u64 qm, st;          // the masks of the lane's chunk
        u32 stp_hi;          // ... and the upper half of st of the chunk in front (bit 31: its last byte starts an escape)
        ChunkRec r;          // record units
        u32 g, sb;           // exclusive prefixes of the unit: Strings.B offset, string ordinal
        u32 h, uf;           // state at the start of the unit | it holds a starter << 1; SEL: the states at its ends (USEL_*)
        bool rec, live, last;
    };
    struct Unit {  // what the wave works with
        u64 em, esc;      // emit mask; escaped characters that are emitted
        u32 pre_raw;      // ChunkRec::pre: emitted bytes of the unit in front of the chunk | CHUNK_SLOW | CHUNK_GENERAL
        u32 g;
    };
    // The per-unit scalars (uniform loads) run one unit ahead of the vector loads they steer: requested and needed in the
    // same place they cost a scalar-memory round trip in front of every unit's loads.
    struct Sc {
        u32 flags;     // 1: record unit, 2: the unit holds a starter, 4: the unit in front does, 8: state at the start of the unit
        u32 g, sb;     // exclusive prefixes of the unit: Strings.B offset, string ordinal
        u32 uf, tq;    // SEL: the states at the unit's ends (USEL_*), the closing quote of the string open at its end
        bool live, last;
    };
    auto load_sc = [&](u64 u) {
        Sc c;
        c.flags = c.g = c.sb = c.uf = c.tq = 0;
        c.live = u < p.units;
        c.last = u + 1 == p.units;
        if (!c.live) return c;
        const u64 sw = p.sv.unit_slow[u], sp = u ? p.sv.unit_slow[u - 1] : 0ull;
        const u32 uh = p.sv.unit_h[u], uhp = u ? (u32)p.sv.unit_h[u - 1] : 0u;
        // (record units: the ones str_masks_body has done escape by escape)
        c.flags = ((sw != 0 || (sp >> 63) != 0) ? 1u : 0u) | (uh & 2u) | ((uhp & 2u) << 1) | ((uh & 1u) << 3);
        c.sb = p.unit_str[u];
        c.g = p.unit_cnt[u];
        if (SEL && !NE) {  // (k_measure's: not written when the message holds no escape)
            c.uf = p.unit_copy[u];
            c.tq = p.unit_tq[u];
        }
        return c;
    };
    auto load_raw = [&](u64 u, const Sc &sc) {
        Raw x;
        x.qm = x.st = 0;
        x.stp_hi = 0;
        x.r = ChunkRec{0, 0, 0};
        x.g = sc.g;
        x.sb = sc.sb;
        x.h = ((sc.flags >> 3) & 1u) | (sc.flags & 2u);
        x.uf = sc.uf;
        x.rec = (sc.flags & 1u) != 0;
        x.live = sc.live;
        x.last = sc.last;
        if (!x.live) return x;
        const u64 c = u * 64 + lane;
        x.qm = p.sv.qm[c];
        // (the st masks of a unit without an escape starter are zero and stay unread: parking-citations has none at all)
        if (sc.flags & 2u) x.st = p.sv.st[c];
        if (c && (sc.flags & (lane ? 2u : 4u))) x.stp_hi = reinterpret_cast<const u32 *>(arr_at(p.sv.st, c - 1, 1))[1];
        if (x.rec) x.r = p.rec[c];
        return x;
    };
    // The offsets of the unit's strings (SEL: and the positions of its closing quotes) are staged in the wave's escape list
    // (free between two units' patches) and leave with coalesced stores; a lane writing the offsets of its chunk's strings
    // straight to memory -- up to eight scattered 4-byte stores per unit and lane -- made this kernel 30-50 % slower
    // (154 instead of 116 us on configs[1])
    u32 *const s_so = reinterpret_cast<u32 *>(&s_gu[wave].list[0]);
    uint2 *const s_so2 = reinterpret_cast<uint2 *>(&s_gu[wave].list[0]);
    constexpr u32 SO_CAP = SEL ? 512 : 1024;  // (GenUnit::list holds 2052 16-bit entries = 1026 words = 513 pairs)
    auto convert = [&](const Raw &x, u64 u, u32 tq) {
        Unit v;
        v.em = x.r.em;
        v.pre_raw = x.r.pre;
        v.esc = 0;
        v.g = x.g;
        const u32 h = x.h & 1u;
        const u64 xq = wave_quotes(x.qm);  // (a unit that is not live holds zeros)
        const ChunkFast f = chunk_fast(x.qm, xq, x.st, (u64)x.stp_hi << 32, h);
        u32 flags;
        if (x.rec) {  // (uniform) emit mask and flags from k_measure's general routine
            v.esc = ((x.st << 1) | (u64)(x.stp_hi >> 31)) & v.em;
            flags = v.pre_raw & ~CHUNK_PRE_MASK;
        } else {
            v.em = f.em;
            v.esc = f.esc;
            flags = f.esc != 0 ? CHUNK_SLOW : 0u;
        }
        u64 cq = 0;
        if (SEL && !NE) {  // only the bytes of strings that hold an escape starter
            u64 sel = 0;
            if ((x.h & 2u) || x.uf)  // (uniform)
                sel = sel_wave_mask(chunk_sel(x.qm, xq, x.st, h), (x.uf & USEL_IN) ? 1u : 0u, (x.uf & USEL_OUT) ? 1u : 0u, lane);
            v.em &= sel;
            v.esc &= sel;
            // (the flags stay: a chunk without an escaped character of its own may still receive the bytes of a \u escape that
            // begins in the chunk in front -- it must park its bytes for the patch like in the other mode)
            cq = xq & ~f.oq;
        }
        const u32 n = (u32)popc64(v.em);
        const u32 ns = (u32)popc64(f.oq);
        const u32 incl = wave_incl_sum(n | (ns << 16));
        const u32 tot = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        const u32 pre = (incl & 0xffffu) - n, spre = (incl >> 16) - ns, stotal = tot >> 16;
        v.pre_raw = pre | flags;
        // the strings that begin in this chunk: their number in the message and their Strings.B offset; SEL: and the raw length
        // of their content -- the distance to their closing quote: in this chunk, or (the last string of the chunk only) in the
        // next chunk of the unit that holds a closing quote, or (the last string of the unit only) where k_measure found it
        if ((stotal != 0 || x.last) && x.live) {  // (uniform)
            const bool staged = stotal <= SO_CAP;  // (uniform)
            u32 far = 0;  // SEL: aligned offset of the first closing quote behind this chunk
            if (SEL && !NE) {
                const u64 cqb = __ballot(cq != 0);
                const u64 above = lane < 63 ? cqb >> (lane + 1) : 0ull;
                const int L = above ? lane + 1 + (int)ctz64(above) : lane;  // the next chunk of the unit with a closing quote
                const u32 fb = (u32)__shfl(cq ? (int)ctz64(cq) : 0, L, 64);
                far = above ? (u32)((u * 64 + (u64)L) * 64) + fb : tq;
            }
            u32 i = spre;
            for (u64 r = f.oq; r != 0; r &= r - 1, i++) {
                const u32 b = (u32)ctz64(r);
                const u32 val = x.g + pre + (u32)popc64(v.em & ((1ull << b) - 1ull));
                if (SEL && NE) {
                    if (staged) s_so[i] = val;
                    else if (x.sb + i < p.soff_cap) p.soff[x.sb + i] = val;
                } else if (SEL) {
                    const u64 cqa = b < 63 ? cq & (~0ull << (b + 1)) : 0ull;  // closing quotes of the chunk behind the opening one
                    const u32 open_at = (u32)((u * 64 + (u64)lane) * 64) + b;
                    const u32 raw = (cqa ? (u32)((u * 64 + (u64)lane) * 64) + (u32)ctz64(cqa) : far) - open_at - 1u;
                    if (staged) s_so2[i] = make_uint2(val, raw);
                    else if (x.sb + i < p.soff_cap) p.sinfo[x.sb + i] = make_uint2(val, raw);
                } else {
                    if (staged) s_so[i] = val;
                    else if (x.sb + i < p.soff_cap) p.soff[x.sb + i] = val;
                }
            }
            if (staged) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                for (u32 j = (u32)lane; j < stotal; j += 64) {
                    if (x.sb + j >= p.soff_cap) continue;
                    if (SEL && !NE) p.sinfo[x.sb + j] = s_so2[j];
                    else p.soff[x.sb + j] = s_so[j];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();  // (the next user of the list waits for these reads)
            }
            // (behind the last string: the end of Strings.B, so that every length is a difference)
            if (x.last && lane == 63 && x.sb + stotal < p.soff_cap) {
                if (SEL && !NE) p.sinfo[x.sb + stotal] = make_uint2(x.g + (tot & 0xffffu), 0u);
                else p.soff[x.sb + stotal] = x.g + (tot & 0xffffu);
            }
        }
        if (NE) {  // the unit's bytes stay where they are: nothing to load, patch, compact or store
            v.em = 0;
            v.esc = 0;
            v.pre_raw = 0;
        }
        return v;
    };
    auto load_chunk = [&](u64 u, u64 em, u32 (&w)[16]) {
        // The chunk must hold message bytes: then its 64-byte line is readable.  An emit mask alone does not say so: a
        // document that ends inside a string (a stage-1 error) is "in a string" through the blanks stage 1 pads its last
        // unit with, up to 4 KiB behind the message -- found by the bounds-checked build (sj_bounds.h); in the product
        // build those reads went past the end of the caller's buffer.
        if (u < p.units && em != 0 && (u * 64 + lane) * 64 < p.sv.end && !SJ_EXPBIT(p, 5)) {
            const uint4 *src = reinterpret_cast<const uint4 *>(arr_at(p.sv.base, (u * 64 + lane) * 64, 64));
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint4 v = src[q];
                w[4 * q + 0] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
            }
        }
    };
    const Sc sc1 = load_sc(unit), sc2 = load_sc(unit + nwaves);
    Sc sc3 = load_sc(unit + 2 * nwaves);
    Unit cur = convert(load_raw(unit, sc1), unit, sc1.tq), nxt = convert(load_raw(unit + nwaves, sc2), unit + nwaves, sc2.tq);
    __syncthreads();
    if (unit >= p.units) return;
    u8 *in8 = &s_io[wave][0];
    u32 *in32 = reinterpret_cast<u32 *>(in8);
    auto byte_ix = [&](u32 q) { return ((q >> 2) * 64 + lane) * 4 + (q & 3); };
    u32 w[16], w_n[16];
    load_chunk(unit, cur.em, w);
    for (;;) {
        const u64 c = unit * 64 + lane;
        const u64 next = unit + nwaves;
        const bool more = next < p.units;  // wave-uniform
        const Raw nn = load_raw(next + nwaves, sc3);   // two units ahead (its scalars were requested an iteration ago)
        load_chunk(next, nxt.em, w_n);                 // one unit ahead (its masks were requested an iteration ago)
        const Sc sc4 = load_sc(next + 2 * nwaves);     // the scalars of the unit three ahead
        const u64 em = cur.em;
        const u32 pre_raw = cur.pre_raw;
        const u32 pre = pre_raw & CHUNK_PRE_MASK;
        const bool patched = (pre_raw & CHUNK_SLOW) != 0, general = (pre_raw & CHUNK_GENERAL) != 0;
        const u32 n = (u32)popc64(em);
        const u32 total = (u32)__shfl((int)(pre + n), 63, 64);
        const u64 g = (u64)cur.g;  // exclusive prefix: Strings.B offset of the unit
        if (total != 0) {  // wave-uniform
            const bool mine = em != 0 && patched;
            // A unit whose escapes are all simple ones (\" \\ \/ \b \f \n \r \t: twitter.json's URLs) is not parked any more
            // (round 6): an escaped character is ONE byte in, one byte out, so it is translated where the compaction put it, in
            // the output window below -- the 16 + 16 LDS instructions that parked a chunk and took it back, issued by the whole
            // wave if a single lane had an escape, and two waits, are gone for those units.  Only a unit with a \u (or invalid)
            // escape, whose translated bytes differ in number and may fall into the neighbour chunk, takes the parked form.
            const bool any_general = __ballot(mine && general) != 0;  // (wave-uniform)
            if (mine && any_general) {
#pragma unroll
                for (int q = 0; q < 16; q++) in32[q * 64 + lane] = w[q];
            }
            if (any_general) {
                // The general patch of a unit, escape by escape (see GenUnit / k_measure): the emitted escaped characters
                // of the unit's general chunks -- simple escapes and the 'u' of \u escapes that emit bytes -- are listed
                // and the lanes take them round robin; the translated bytes go into the parked copy of the chunk they
                // fall into, this one or the next (sj_strings.h str_chunk_patch is the per-chunk statement).
                u64 le = 0, foreign = 0;
                if (mine && general) le = cur.esc;
                if (lane == 0 && general && c > 0)  // escapes of the chunk in front whose bytes reach into chunk 0
                    foreign = ((p.sv.esc(c - 1) & p.rec[c - 1].em) >> 60) & 0xfull;
                GenUnit *gu = &s_gu[wave];
                const u32 items = gen_unit_list(gu, le, foreign, lane);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const u64 u0 = unit * 4096;
                for (u32 j = (u32)lane; j < items; j += 64)
                    gen_item_patch(p.sv, u0, gu->list[j], [&](u32 pos, u8 v) {  // into the dword-major window
                        const u32 q = pos & 63u, L = pos >> 6;
                        in8[((q >> 2) * 64 + L) * 4 + (q & 3)] = v;
                    });
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
            if (mine && any_general) {
                if (!general) {  // simple escapes only: translated in place, nothing is read from the message
                    for (u64 r = cur.esc; r != 0; r &= r - 1) {
                        const u32 ix = byte_ix((u32)ctz64(r));
                        in8[ix] = s_esc[in8[ix]];
                    }
                }
#pragma unroll
                for (int q = 0; q < 16; q++) w[q] = in32[q * 64 + lane];
            }
            if (any_general) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();  // every lane has its chunk in registers: the window turns into the output
            }
            // The window is cleared and the lanes OR their bytes in at their byte offsets with ALIGNED 8-byte LDS atomics:
            // eight message bytes at a time are squeezed together (two v_perm_b32), shifted to where they belong inside an
            // aligned 8-byte slot and its successor, and ds_or_b64 merges them with what the neighbour lanes put there.
            // (Unaligned ds_write_b32 did the placement before: the LDS executes those a lane at a time -- 19 LDS cycles
            // per instruction on average, the LDS 92 % busy, 42 % of the wave-cycles stalled on LDS issue.)
            {
                uint4 *z = reinterpret_cast<uint4 *>(&s_io[wave][0]) + lane * 4;
                const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
                z[0] = zero; z[1] = zero; z[2] = zero; z[3] = zero;
                if (lane == 0) reinterpret_cast<uint4 *>(&s_io[wave][4096])[0] = zero;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (em != 0) {
                u32 o = pre;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const u32 nib0 = (u32)(em >> (8 * q)) & 15u, nib1 = (u32)(em >> (8 * q + 4)) & 15u;
                    const u32 cd0 = __builtin_amdgcn_perm(0u, w[2 * q], s_sel[nib0]);
                    const u32 cd1 = __builtin_amdgcn_perm(0u, w[2 * q + 1], s_sel[nib1]);
                    const u32 n0 = (u32)__builtin_popcount(nib0), n1 = (u32)__builtin_popcount(nib1);
                    const u64 piece = (u64)cd0 | ((u64)cd1 << (8u * n0));
                    const u32 sh = 8u * (o & 7u);
                    const u64 lo = piece << sh, hi = sh ? piece >> (64u - sh) : 0ull;
                    unsigned long long *slot = reinterpret_cast<unsigned long long *>(&s_io[wave][o & ~7u]);
                    if (lo != 0) atomicOr(slot, (unsigned long long)lo);      // (a zero changes nothing)
                    if (hi != 0) atomicOr(slot + 1, (unsigned long long)hi);
                    o += n0 + n1;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (!any_general && __ballot(mine) != 0) {  // (wave-uniform) simple escapes: the escaped character where it landed
                if (mine) {
                    // of the eight simple escapes only b f n r t change the byte (\" \\ \/ stand for themselves -- every URL of
                    // twitter.json): one LDS read per escape, the test and the translation in registers (a 4-bit table in a
                    // constant: (c - 'b') / 2 -> 8, 12, 10, 13, 9), a write only where something changes
                    for (u64 r = cur.esc; r != 0; r &= r - 1) {
                        const u32 b = (u32)ctz64(r);
                        u8 *at = &s_io[wave][pre + (u32)popc64(em & ((1ull << b) - 1ull))];
                        const u32 d = (u32)*at - (u32)'b';
                        if (d < 19u && ((0x51011u >> d) & 1u)) *at = (u8)((0x9D0A000C08ull >> ((d >> 1) * 4u)) & 15u);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
            if (g + total <= p.strings_cap && !SJ_EXPBIT(p, 6)) {
                u8 *dst = arr_at(p.str_out, g, total);
                const u32 q16 = total >> 4;
                for (u32 i = lane; i < q16; i += 64) {  // 16 bytes per lane; Strings.B offsets are byte-granular: unaligned stores are fine on gfx950
                    *reinterpret_cast<uint4 *>(dst + 16 * i) = *reinterpret_cast<const uint4 *>(&s_io[wave][16 * i]);
                }
                const u32 tail = q16 * 16 + lane;
                if (tail < total) dst[tail] = s_io[wave][tail];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();  // the window is read out: the next unit may write it
        }
        if (!more) break;
        unit = next;
        cur = nxt;
        nxt = convert(nn, next + nwaves, sc3.tq);
        sc3 = sc4;
#pragma unroll
        for (int q = 0; q < 16; q++) w[q] = w_n[q];
    }
}

// (two kernels for the two modes: WithCopyStrings(false) is compiled for four waves per SIMD -- 128 registers, 16 bytes of
// scratch -- where the compiler's own choice was 145 registers and three: 139 -> 121-125 us on configs[4], 151-164 -> 137 us on
// configs[1]; the same bound on the copy-mode kernel, which sits at 125 registers anyway, measured 1-4 % slower)
template <bool SEL>
__global__ void k_str_emit(S2Dev p);
template <>
__global__ __launch_bounds__(256) void k_str_emit<false>(S2Dev p) { str_emit_body<false>(p); }
template <>
__global__ __launch_bounds__(256, 4) void k_str_emit<true>(S2Dev p) { str_emit_body<true>(p); }

// ---- the token scan ----------------------------------------------------------------------------------------
// Inside a tile the scan runs on the packed form PAgg (sj_stage2.h); wave scans use DPP row shifts / broadcasts
// (no LDS traffic).  S = false drops the Strings.B byte count (every string copied: the emit masks place them).
template <bool S>
__device__ __forceinline__ PAgg pagg_comb(const PAgg &a, const PAgg &b) {  // a in front of b
    return PAgg{a.x + b.x, a.y + b.y, am_combine(a.z, b.z), S ? a.s + b.s : 0u};
}
template <bool S, int CTRL, int ROW_MASK>
__device__ __forceinline__ PAgg pagg_dpp(const PAgg &v) {  // lanes without a source read the identity
    return PAgg{(u32)__builtin_amdgcn_update_dpp(0, (int)v.x, CTRL, ROW_MASK, 0xf, false),
                (u32)__builtin_amdgcn_update_dpp(0, (int)v.y, CTRL, ROW_MASK, 0xf, false),
                (u32)__builtin_amdgcn_update_dpp((int)AM_ALL, (int)v.z, CTRL, ROW_MASK, 0xf, false),
                S ? (u32)__builtin_amdgcn_update_dpp(0, (int)v.s, CTRL, ROW_MASK, 0xf, false) : 0u};
}
template <bool S>
__device__ __forceinline__ PAgg pagg_wave_inclusive(PAgg v) {
    v = pagg_comb<S>(pagg_dpp<S, 0x111, 0xf>(v), v);  // row_shr:1
    v = pagg_comb<S>(pagg_dpp<S, 0x112, 0xf>(v), v);  // row_shr:2
    v = pagg_comb<S>(pagg_dpp<S, 0x114, 0xf>(v), v);  // row_shr:4
    v = pagg_comb<S>(pagg_dpp<S, 0x118, 0xf>(v), v);  // row_shr:8
    v = pagg_comb<S>(pagg_dpp<S, 0x142, 0xa>(v), v);  // row_bcast:15 -> rows 1, 3
    v = pagg_comb<S>(pagg_dpp<S, 0x143, 0xc>(v), v);  // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ PAgg pagg_readlane(const PAgg &v, int l) {
    return PAgg{(u32)__builtin_amdgcn_readlane((int)v.x, l), (u32)__builtin_amdgcn_readlane((int)v.y, l),
                (u32)__builtin_amdgcn_readlane((int)v.z, l), (u32)__builtin_amdgcn_readlane((int)v.s, l)};
}
// Block-level exclusive prefix of one value per thread (WAVES <= 16 waves); returns the exclusive prefix of the
// calling thread and, in `total`, the sum over the block.  s_w: WAVES entries of LDS.
template <bool S, int WAVES>
__device__ __forceinline__ PAgg pagg_block_exclusive(const PAgg &mine, PAgg *s_w, int lane, int wave, PAgg &total) {
    const PAgg incl = pagg_wave_inclusive<S>(mine);
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    // every wave scans the wave totals in its first row
    PAgg t = lane < WAVES ? s_w[lane] : pagg_identity();
    t = pagg_comb<S>(pagg_dpp<S, 0x111, 0xf>(t), t);
    t = pagg_comb<S>(pagg_dpp<S, 0x112, 0xf>(t), t);
    t = pagg_comb<S>(pagg_dpp<S, 0x114, 0xf>(t), t);
    t = pagg_comb<S>(pagg_dpp<S, 0x118, 0xf>(t), t);
    total = pagg_readlane(t, WAVES - 1);
    PAgg before = pagg_identity();
    if (wave > 0) before = pagg_readlane(t, (wave - 1) & 15);
    return pagg_comb<S>(before, pagg_dpp<S, 0x138, 0xf>(incl));  // wave_shr:1: the lane in front, identity in lane 0
}

// Both measuring passes in one launch (they are independent and neither fills the device on its own): blocks below
// `mblocks` turn the string masks of stage 1 into emit masks and unit counts, the others reduce the token kinds of a tile
// to its scan aggregate.
__device__ __forceinline__ void s2_reduce_planes(const S2Dev &p, u32 block);
__global__ __launch_bounds__(RD_BLOCK, SJ_MEASURE_WAVES) void k_measure(S2Dev p, u32 mblocks);

// ---- pass 2: exclusive scan over the tile aggregates (in place) + totals: SCAN_SEGS blocks ---------------------
// Inside a segment every thread owns K consecutive tiles (K <= 32 per round), so a block scans once per round
// and the aggregates are read with four independent loads in flight per thread.
__device__ __forceinline__ Agg tile_chunk_sum(const S2Dev &p, u32 first, u32 K, u32 hi) {
    Agg acc = agg_identity();
    for (u32 j = 0; j < K; j += 4) {
        Agg a[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const u32 t = first + j + q;
            a[q] = (j + q < K && t < hi) ? p.agg[t].a : agg_identity();
        }
#pragma unroll
        for (int q = 0; q < 4; q++) acc = agg_combine(acc, a[q]);
    }
    return acc;
}
__device__ __forceinline__ void scan_tiles_body(const S2Dev &p, int seg) {
    __shared__ Agg s_w[16];
    __shared__ unsigned long long s_w64[16], s_s64[16];
    __shared__ SegSum s_before;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 n = token_count(p), tiles = (u32)(((u64)n + S2_TILE - 1) / S2_TILE);
    u64 lo64, hi64;
    seg_range(tiles, 64, seg, p.tile_segs, lo64, hi64);
    const u32 lo = (u32)lo64, hi = (u32)hi64;
    // pass 1: the segment's aggregate (32-bit fields wrap; the two sizes are also summed in 64 bits)
    Agg seg_acc = agg_identity();  // meaningful in every thread after the loop
    unsigned long long w64 = 0, s64 = 0;
    for (u32 start = lo; start < hi; start += 1024u * 32u) {
        const u32 left = hi - start;
        const u32 K = left >= 1024u * 32u ? 32u : (left + 1023u) / 1024u;
        const Agg acc = tile_chunk_sum(p, start + (u32)tid * K, K, hi);
        unsigned long long ws = acc.w, bs = acc.s;
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) {
            ws += (unsigned long long)__shfl_xor((long long)ws, sh, 64);
            bs += (unsigned long long)__shfl_xor((long long)bs, sh, 64);
        }
        const Agg incl = agg_wave_inclusive(acc);
        if (lane == 63) {
            s_w[wave] = incl;
            s_w64[wave] = ws;
            s_s64[wave] = bs;
        }
        __syncthreads();
        Agg t = lane < 16 ? s_w[lane] : agg_identity();
        seg_acc = agg_combine(seg_acc, agg_readlane(agg_row_scan(t), 15));
        for (int w = 0; w < 16; w++) {
            w64 += s_w64[w];
            s64 += s_s64[w];
        }
        __syncthreads();
    }
    if (wave == 0) {
        SegSum own;
        own.a = seg_acc;
        own.w = w64;
        own.s = s64;
        if (lane == 0) seg_publish(&p.seg_tiles[seg], own);
        const SegSum before = seg_lookback(p.seg_tiles, seg, lane, p.st);
        if (lane == 0) {
            s_before = before;
            if (seg == p.tile_segs - 1) {
                const Agg tot = agg_combine(before.a, seg_acc);
                const unsigned long long words64 = before.w + w64, bytes64 = before.s + s64;
                p.st->final_depth = tot.d;
                p.st->tape_len = words64 + 2ull;  // + opening root + closing root
                p.st->strings_len = bytes64;      // (every string copied: strings_len_masks, left by the unit scan, counts)
                p.st->records = tot.nb;
                p.st->n_br = tot.bc;
                // the gap behind the last bracket (empty if the last token is a bracket, as in every accepted document)
                p.st->tail_mask = n == 0 || is_bracket(p.kind[n - 1]) ? AM_ALL : am_value(tot.am);
                if (words64 + 2ull > 0xfffffff0ull || bytes64 > 0xfffffff0ull) atomicOr(&p.st->err, 4u);
                if (tot.d != 0) atomicOr(&p.st->err, 1u);  // scopes still open at the end (succeed: :433-435)
                // tokens behind the last bracket lie at depth 0: the root context must allow them
                if (!context_allowed(p.st->tail_mask, CTX_ROOT)) atomicOr(&p.st->err, 1u);
            }
        }
    }
    __syncthreads();
    // pass 2: exclusive prefixes in place
    Agg carry = s_before.a;
    for (u32 start = lo; start < hi; start += 1024u * 32u) {
        const u32 left = hi - start;
        const u32 K = left >= 1024u * 32u ? 32u : (left + 1023u) / 1024u;
        const u32 first = start + (u32)tid * K;
        const Agg acc = tile_chunk_sum(p, first, K, hi);
        const Agg incl = agg_wave_inclusive(acc);
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        Agg t = lane < 16 ? s_w[lane] : agg_identity();
        t = agg_row_scan(t);
        const Agg round_total = agg_readlane(t, 15);
        Agg run = carry;
        if (wave > 0) run = agg_combine(run, agg_readlane(t, (wave - 1) & 15));
        run = agg_combine(run, agg_dpp<0x138, 0xf>(incl));  // wave_shr:1: the lane in front, identity in lane 0
        for (u32 j = 0; j < K; j += 4) {
            Agg a[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const u32 t2 = first + j + q;
                a[q] = (j + q < K && t2 < hi) ? p.agg[t2].a : agg_identity();
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const u32 t2 = first + j + q;
                if (j + q < K && t2 < hi) p.agg[t2].a = run;
                run = agg_combine(run, a[q]);
            }
        }
        carry = agg_combine(carry, round_total);
        __syncthreads();
    }
}

// Both small scans in one launch: blocks 0 .. SCAN_SEGS-1 scan the unit byte counts (every string copied), the others
// the tile aggregates (as two launches each cost its ~10 us of fixed latency).
__global__ __launch_bounds__(1024) void k_scans(S2Dev p) {
    if (p.sv.qm) {
        const int b = (int)blockIdx.x, us = p.unit_segs, ss = p.unit_str ? p.unit_segs : 0;
        if (b < us) str_scan_body<false>(p, b);
        else if (b < us + ss) str_scan_body<true>(p, b - us);
        else scan_tiles_body(p, b - us - ss);
    } else {
        scan_tiles_body(p, (int)blockIdx.x);
    }
}

__device__ __forceinline__ int top_bit(u64 m) { return 63 - __builtin_clzll(m); }  // m != 0

// ==== round 5: the token pass on bit planes, sixteen tokens per lane (sj_tok16.h) ====================================
// k_measure's token half and the emit pass in their plane form.  A tile is still 4096 tokens (the packed scan form PAgg and
// the tile aggregates are unchanged); a block is 256 threads x 16 tokens.  What changed against the per-token kernels
// above (kept as variant 0 for A/B runs, SJHIP_S2_VARIANT):
//   * the scan element of a lane comes from ~150 boolean instructions on 19-bit windows of the kind planes instead of 16
//     table look-ups (sj_tok16.h); per-token work is left only for tokens that write something;
//   * queue slots (strings, scalars) and the tile's bracket list are ORDERED: their indices come out of the same block
//     scan as the tape offsets -- no LDS atomics;
//   * brackets are matched INSIDE the tile: the tile's brackets wait in LDS (one packed word each: tape offset, depth,
//     kind, gap set), a wave answers the previous-smaller-value questions of 64 of them with ballots over the list, and a
//     pair whose two ends lie in the tile is written here, next to the other words of its stretch of the tape (the
//     device-wide matcher wrote every pair as two scattered 8-byte stores into lines that had long left the L2: 72 MB of
//     sector writes and 84 MB of reads for 17 MB of words on configs[1], 95 + 28 MB on configs[4]).  Only brackets whose
//     container starts in front of the tile stay "live" in the compact view (sj_tok16.h BR_DONE): k_br_match skips the rest.
// (the reduce runs 256 lanes of sixteen tokens; the emit pass is compiled for sixteen and for eight tokens per lane --
// 256 or 512 threads per tile -- SJHIP_S2_ITEMS)
static constexpr int TK_BLOCK = 256, TK_ITEMS = 16, TK_WAVES = TK_BLOCK / 64;
static_assert(TK_BLOCK * TK_ITEMS == S2_TILE, "256 lanes of sixteen tokens are one tile");

struct TileLane {
    Lane16 m;
    u32 base;  // token index of the lane's first token
};
// The sixteen kinds of this thread (K_NL behind the end of the message, like token_pelement's sentinel), the neighbours'
// kinds through s_edge (TK_BLOCK + 2 words: [1 + tid] = kind 14 | kind 15 << 8 | kind 0 << 16 of thread tid), the masks.
// Contains one __syncthreads().
template <int ITEMS>
__device__ __forceinline__ TileLane tile_lane(const S2Dev &p, u32 t0, u32 n, int tid, u32 *s_edge) {
    constexpr int BLK = S2_TILE / ITEMS;
    TileLane r;
    r.base = t0 + (u32)tid * ITEMS;
    constexpr u32 NL4 = 0x01010101u * K_NL;
    u32 d[4] = {NL4, NL4, ITEMS == 16 ? NL4 : 0u, ITEMS == 16 ? NL4 : 0u};
    if (r.base + ITEMS <= n) {
        if (ITEMS == 16) {
            const uint4 kv = *reinterpret_cast<const uint4 *>(arr_at(p.kind, r.base, 16));
            d[0] = kv.x; d[1] = kv.y; d[2] = kv.z; d[3] = kv.w;
        } else {
            const uint2 kv = *reinterpret_cast<const uint2 *>(arr_at(p.kind, r.base, 8));
            d[0] = kv.x; d[1] = kv.y;
        }
    } else if (r.base < n) {
#pragma unroll
        for (int j = 0; j < ITEMS; j++)  // (constant indices: the words stay in registers)
            if (r.base + (u32)j < n) d[j >> 2] = (d[j >> 2] & ~(0xffu << (8 * (j & 3)))) | ((u32)p.kind[r.base + j] << (8 * (j & 3)));
    }
    s_edge[1 + tid] = (d[ITEMS / 4 - 1] >> 16) | ((d[0] & 0xffu) << 16);
    if (tid == 0) s_edge[0] = t0 == 0 ? (u32)K_NONE | ((u32)K_NONE << 8) : (u32)p.kind[t0 - 2] | ((u32)p.kind[t0 - 1] << 8);
    if (tid == 1) s_edge[1 + BLK] = ((u64)t0 + S2_TILE < n ? (u32)p.kind[t0 + S2_TILE] : (u32)K_NL) << 16;
    __syncthreads();
    const u32 prev2 = s_edge[tid] & 0xffffu, next1 = (s_edge[tid + 2] >> 16) & 0xffu;
    const u32 cnt = n > r.base ? (n - r.base < (u32)ITEMS ? n - r.base : (u32)ITEMS) : 0u;
    r.m = lane16_masks<ITEMS>(planes16(d[0], d[1], d[2], d[3]), prev2, next1, (1u << cnt) - 1u, r.base == 0 && cnt != 0);
    return r;
}

// ---- pass 1 on planes: tile aggregates ------------------------------------------------------------------------------
__device__ __forceinline__ void s2_reduce_planes(const S2Dev &p, u32 block) {
    __shared__ u32 s_edge[TK_BLOCK + 2];
    __shared__ PAgg s_w[TK_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 n = token_count(p);
    if ((u64)block * S2_TILE >= n) return;  // (the grid is sized for the upper bound)
    const u32 t0 = block * S2_TILE;
    const TileLane tl = tile_lane<TK_ITEMS>(p, t0, n, tid, s_edge);
    const Lane16 &m = tl.m;
    const u32 base = tl.base;
    // Strings.B offsets: with stage 1's masks (both copy modes) they are a property of the message -- k_str_emit leaves the
    // offset of the k-th string in soff[k], and what goes through the scan is the NUMBER of strings; without the masks (the
    // per-string fallback, S2_ERR_SERIAL_STRINGS) every string is walked here and the bytes to copy go through the scan
    u32 sbytes = 0;
    if (!p.sv.qm) {
        const MsgView mv{p.msg, p.len};
        for (u32 r = m.str; r != 0; r &= r - 1) {
            const u32 k = (u32)__builtin_ctz(r);
            u32 sl, dl, out;
            if (!string_walk(mv, p.pos[base + k], nullptr, &sl, &dl)) {
                out = DLEN_INVALID;
                atomicOr(&p.st->err, 1u);
            } else {
                const bool cp = p.copy_strings || sl != dl;
                out = dl | (cp ? DLEN_COPY : 0u);
                sbytes += cp ? dl : 0u;
            }
            p.dlen[base + k] = out;
        }
    }
    // (the tile's first string is then string number Agg::s of the message)
    if (p.sv.qm) sbytes = popc32(m.str);
    PAgg acc = lane16_pagg(m);
    acc.s = sbytes;
    const PAgg incl = pagg_wave_inclusive<true>(acc);
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    if (tid == 0) {
        PAgg tot = s_w[0];
        for (int w = 1; w < TK_WAVES; w++) tot = pagg_comb<true>(tot, s_w[w]);
        p.agg[block].a = pagg_unpack(tot);
    }
}

__global__ __launch_bounds__(RD_BLOCK, SJ_MEASURE_WAVES) void k_measure(S2Dev p, u32 mblocks) {
    __shared__ GenUnit s_gu[RD_BLOCK / 64];
    if (blockIdx.x < mblocks) {
        if (p.copy_strings) str_masks_body<false>(p, blockIdx.x, mblocks, &s_gu[threadIdx.x >> 6]);
        else if (!no_escapes(p)) str_masks_body<true>(p, blockIdx.x, mblocks, &s_gu[threadIdx.x >> 6]);
    }
    else s2_reduce_planes(p, blockIdx.x - mblocks);
}

// ---- pass 3 on planes -----------------------------------------------------------------------------------------------
// queue entry of a string / scalar: token index inside the tile | tape offset inside the tile << 12 | (string: object key
// << 25) | (scalar: kind - 8 << 26)
// MODE 0: no masks (the per-string fallback: lengths from the walks of the token reduce, both copy modes); 1: every string
// copied, offsets from soff[]; 2: WithCopyStrings(false) on the masks -- soff[] and the closing quotes scq[]
// WIDE: a message of 4 GiB or more (the 32-bit positions wrap: every tile rebuilds its true offsets, below)
template <int MODE, int ITEMS, bool WIDE>
__global__ __launch_bounds__(S2_TILE / ITEMS, ITEMS == 8 ? 8 : 4) void k_s2_emit_planes(S2Dev p) {
    constexpr bool MASKS = MODE != 0;
    constexpr int BLK = S2_TILE / ITEMS, WAVES = BLK / 64;
    __shared__ __attribute__((aligned(16))) u32 s_pos[S2_TILE + 4];
    // strings [0, S) | the tile's brackets [SB, SB + B) | atoms and numbers [SB + B, SB + B + D): a tile has 4096 tokens.
    // MASKS: no string queue -- [0, S] holds the Strings.B offsets of the tile's S strings and of the string behind them
    // (soff[] / sinfo[].x, left by k_str_emit in message order = token order), SB = S + 1; MODE 2: the raw lengths of the
    // tile's strings (sinfo[].y) behind everything else, from SC on -- a string is followed by a token that is none, so
    // 2 S + B + D fits the 4096 + 8 entries in every document that parses; where it does not the lengths are read from
    // memory.  (A lane reading the entries of its own strings
    // straight from memory -- nine sparsely populated load instructions per wave -- was measured 3 % / 6 % slower over
    // the whole parse of configs[1] / configs[4] than this staging with full-width loads.)
    __shared__ u32 s_q[S2_TILE + 8];
    __shared__ u32 s_edge[BLK + 2];
    __shared__ PAgg s_w[WAVES];
    __shared__ u32 s_wc[WAVES];
    __shared__ i32 s_gmin[S2_TILE / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 n = token_count(p);
    if ((u64)blockIdx.x * S2_TILE >= n) return;  // (the grid is sized for the upper bound)
    // (SJ_EXP bit 21: the tiles from the last to the first -- the order of a sweep decides what the Infinity Cache still holds of the sweep in front)
    const u32 bid = SJ_EXPBIT(p, 21) ? (u32)(((u64)n + S2_TILE - 1) / S2_TILE) - 1u - blockIdx.x : blockIdx.x;
    const u32 t0 = bid * S2_TILE;
    const u64 tape_len = p.st->tape_len;
    if (tape_len > p.tape_cap) return;  // cannot happen: the launcher sizes the tape for 2n+2 words
    const Agg tp = p.agg[bid].a;  // prefix of the tile (needed behind the scan: requested first)
    // The positions are 32 bits wide: in a message of more than 4 GiB they are the true offsets modulo 2^32.  A tile keeps them
    // relative to the true offset of its first token -- which follows from the unit that token lies in (stage 1's tile_unit) --
    // and adds the 64-bit base where a byte of the message is addressed (tokens of a tile less than 4 GiB apart: else the parse fails)
    // (a message below 4 GiB: the base is 0 at compile time -- as a run-time case the look-up and the additions cost the parse of
    // configs[4] 1-2 %)
    u64 tbase = 0;
    if (MASKS && WIDE) {
        const u32 first = p.pos[t0];
        tbase = (u64)p.tile_unit[bid] * 4096u + (((u64)first + p.sv.lead) & 4095u) - p.sv.lead;
    }
    const u32 endpos = (u32)(p.len - tbase);
    if (MASKS && WIDE && tid == 0) {  // the tile's offsets are 32-bit differences from its first token: a tile that spans 4 GiB or more cannot be rebuilt
        const u64 nb = (u64)t0 + S2_TILE < n
                           ? (u64)p.tile_unit[bid + 1] * 4096u + (((u64)p.pos[t0 + S2_TILE] + p.sv.lead) & 4095u) - p.sv.lead
                           : p.len;
        if (nb - tbase >= (1ull << 32)) atomicOr(&p.st->err, 4u);
    }
    {
        const u32 base = t0 + (u32)tid * ITEMS;
        u32 pp[ITEMS];
        if (base + ITEMS <= n) {
#pragma unroll
            for (int q = 0; q < ITEMS / 4; q++) {
                const uint4 a = *reinterpret_cast<const uint4 *>(arr_at(p.pos, base + 4 * q, 4));
                pp[4 * q] = a.x; pp[4 * q + 1] = a.y; pp[4 * q + 2] = a.z; pp[4 * q + 3] = a.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < ITEMS; k++) pp[k] = base + k < n ? p.pos[base + k] : endpos;
        }
        if (MASKS) {
#pragma unroll
            for (int k = 0; k < ITEMS; k++) pp[k] = base + k < n ? pp[k] - (u32)tbase : endpos;  // (wrapped differences)
        }
#pragma unroll
        for (int q = 0; q < ITEMS / 4; q++)
            *reinterpret_cast<uint4 *>(&s_pos[tid * ITEMS + 4 * q]) = make_uint4(pp[4 * q], pp[4 * q + 1], pp[4 * q + 2], pp[4 * q + 3]);
        if (tid == 2) s_pos[S2_TILE] = (u64)t0 + S2_TILE < n ? p.pos[t0 + S2_TILE] - (u32)tbase : endpos;
    }
    const TileLane tl = tile_lane<ITEMS>(p, t0, n, tid, s_edge);  // (holds the block barrier behind the stores above)
    const Lane16 &m = tl.m;
    const u32 base = tl.base;
    const MsgView mv{p.msg, p.len};
    // selective copy: the measured lengths of the lane's strings -> Strings.B bytes of the lane, and (below) every copied
    // string's offset
    u32 dv[ITEMS];
    u32 sbytes = 0;
    if (!MASKS) {
        if (base + ITEMS <= n) {
#pragma unroll
            for (int q = 0; q < ITEMS / 4; q++) {
                const uint4 x = *reinterpret_cast<const uint4 *>(arr_at(p.dlen, base + 4 * q, 4));
                dv[4 * q] = x.x; dv[4 * q + 1] = x.y; dv[4 * q + 2] = x.z; dv[4 * q + 3] = x.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < ITEMS; k++) dv[k] = base + k < n ? p.dlen[base + k] : 0u;
        }
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            dv[k] = (((m.str >> k) & 1u) && dv[k] != DLEN_INVALID && (dv[k] & DLEN_COPY)) ? (dv[k] & ~DLEN_COPY) : 0u;
            sbytes += dv[k];
        }
    }
    // ---- the scan inside the tile: tape words, brackets, opens, records, context function (+ Strings.B bytes), and the
    // queue slots: strings | scalars << 13
    PAgg mine = lane16_pagg(m);
    const u32 cnts = lane16_counts(m);
    mine.s = MASKS ? cnts : sbytes;
    PAgg total;
    const PAgg ex = pagg_block_exclusive<true, WAVES>(mine, s_w, lane, wave, total);
    u32 cex = ex.s, ctot = total.s;
    if (!MASKS) {  // the scan's s field carries bytes: the slots get a scan of their own
        const u32 incl = wave_incl_sum(cnts);
        if (lane == 63) s_wc[wave] = incl;
        __syncthreads();
        cex = incl - cnts;
        ctot = 0;
#pragma unroll
        for (int w = 0; w < WAVES; w++) {
            const u32 x = s_wc[w];
            cex += w < wave ? x : 0u;
            ctot += x;
        }
    }
    const u32 S = ctot & 0x1fffu, D = ctot >> 13, B = total.x >> 14;
    const u32 T0 = tp.w + 1u;  // tape offset of the tile's first word (word 0 is the opening root, write_tape(0,'r'), :172)
    const u32 lane_w = ex.x & 0x3fffu, lane_bc = ex.x >> 14, lane_open = ex.y & 0x1fffu, lane_nb = ex.y >> 13;
    bool bad = lane16_illegal(m);  // a token that is legal in no context at all
    const u32 SB = MASKS ? S + 1u : S, SC = SB + B + D;
    // MODE 2 on a message without an escape starter (no_escapes): no string is copied, k_str_emit left the offsets of the FULL
    // compaction in soff[] -- a string's length is the difference of two of them, as in MODE 1 -- and no raw lengths
    const bool NE = MODE == 2 && no_escapes(p);        // (grid-uniform)
    const bool y_staged = !NE && SC + S <= (u32)S2_TILE + 8u;  // (block-uniform)
    if (MASKS && !SJ_EXPBIT(p, 9)) {  // the tile's first string is string number tp.s of the message (k_measure counted them through the scan)
        for (u32 j = (u32)tid; j <= S; j += BLK) {
            const bool in = tp.s + j < p.soff_cap;
            if (MODE == 2 && !NE) {
                const uint2 e = in ? p.sinfo[tp.s + j] : make_uint2(0u, 0u);
                s_q[j] = e.x;
                if (y_staged && j < S) s_q[SC + j] = e.y;
            } else {
                s_q[j] = in ? p.soff[tp.s + j] : 0u;
            }
        }
    }
    if (!MASKS) {
        u32 slot = cex & 0x1fffu, run = ex.s;
        for (u32 r = m.str; r != 0; r &= r - 1) {  // strings: worked on densely below
            const u32 j = (u32)__builtin_ctz(r), idx = (u32)tid * ITEMS + j;
            s_q[slot++] = idx | ((lane_w + lane16_words_before(m, j)) << 12) | (((m.keystr >> j) & 1u) << 25);
            // the Strings.B offset of the string inside the tile, in the slot of the token's own position
            s_pos[idx] = run;  // (the dense pass reads the position back from memory: no LDS of its own)
            u32 c = 0;
#pragma unroll
            for (int k = 0; k < ITEMS; k++) c = (u32)k == j ? dv[k] : c;
            run += c;
        }
    }
    {
        u32 slot = SB + B + (cex >> 13);
        for (u32 r = m.num | m.atom; r != 0; r &= r - 1) {
            const u32 j = (u32)__builtin_ctz(r), idx = (u32)tid * ITEMS + j;
            const u32 kd = ((m.num >> j) & 1u) ? (u32)K_NUM : (u32)lane16_atom_kind(m, j);
            s_q[slot++] = idx | ((lane_w + lane16_words_before(m, j)) << 12) | ((kd - 8u) << 26);
        }
    }
    {
        u32 slot = SB + lane_bc;
        const u32 am_in = am_combine(tp.am, ex.z);
        for (u32 r = m.br; r != 0; r &= r - 1) {
            const u32 j = (u32)__builtin_ctz(r), upto = (2u << j) - 1u;
            const i32 drel = (i32)(2u * (lane_open + popc32(m.open & upto))) - (i32)(lane_bc + popc32(m.br & upto));  // behind the bracket
            s_q[slot++] = tbr_pack(lane_w + lane16_words_before(m, j), drel, lane16_bracket_kind(m, j), lane16_gap_set(m, j, am_in));
        }
    }
    for (u32 r = m.nlr; r != 0; r &= r - 1) {  // record-separating newlines leave their tape offset (query.hip)
        const u32 j = (u32)__builtin_ctz(r);
        p.nl_off[tp.nb + lane_nb + popc32(m.nlr & ((1u << j) - 1u))] = T0 + lane_w + lane16_words_before(m, j);
    }
    __syncthreads();  // the queues and the bracket list are complete
    // ---- strings on the masks: a string ends where the next one begins (Strings.B is the concatenation of the strings that
    // are copied); both tape words in one 16-byte store (the tape is only 8-byte aligned: fine on gfx950).  MODE 2: a string
    // of length 0 there is not copied (it holds no escape: every escape emits a byte) -- the tape points into the message
    // and the length is the raw length of its content (stage2_build_tape_amd64.go:90-109)
    if (MASKS) {
        u32 ks = cex & 0x1fffu;
        for (u32 r = m.str; r != 0; r &= r - 1, ks++) {
            const u32 j = (u32)__builtin_ctz(r), lo = lane_w + lane16_words_before(m, j);
            const u32 so = s_q[ks], se = s_q[ks + 1];
            u64 w0 = string_word(true, p.strings_base + so, 0), w1 = (u64)(se - so);
            if (MODE == 2 && (NE || se == so)) {
                w0 = string_word(false, 0, p.msg_base + tbase + s_pos[(u32)tid * ITEMS + j] + 1);
                if (!NE) w1 = (u64)(y_staged ? s_q[SC + ks] : (tp.s + ks < p.soff_cap ? p.sinfo[tp.s + ks].y : 0u));
            }
            if (!SJ_EXPBIT(p, 8)) *reinterpret_cast<uint4 *>(arr_at(p.tape, T0 + lo, 2)) = make_uint4((u32)w0, (u32)(w0 >> 32), (u32)w1, (u32)(w1 >> 32));
            if (p.keyflag) p.keyflag[(T0 + lo) >> 1] = (u8)((m.keystr >> j) & 1u);
        }
    }
    // ---- the per-string fallback: the lengths the walks of the token reduce left; a string that is copied (every string, or
    // with WithCopyStrings(false) one that unescaping changed) points into Strings.B -- k_emit_strings walks it again and
    // writes its bytes at the offset left in str_off --, the others point into the message (parseString,
    // stage2_build_tape_amd64.go:90-109)
    if (!MASKS) {
        for (u32 j = (u32)tid; j < S; j += BLK) {
            const u32 v = s_q[j], idx = v & 0xfffu, lo = (v >> 12) & 0x1fffu;
            const u32 dlw = p.dlen[t0 + idx];
            const u32 at = p.pos[t0 + idx], so = tp.s + s_pos[idx];
            if (dlw != DLEN_INVALID) {
                const bool cp = (dlw & DLEN_COPY) != 0;
                const u64 w0 = string_word(cp, p.strings_base + so, p.msg_base + at + 1), w1 = dlw & ~DLEN_COPY;
                *reinterpret_cast<uint4 *>(arr_at(p.tape, T0 + lo, 2)) = make_uint4((u32)w0, (u32)(w0 >> 32), (u32)w1, (u32)(w1 >> 32));
                if (p.keyflag) p.keyflag[(T0 + lo) >> 1] = (u8)((v >> 25) & 1u);
                p.str_off[t0 + idx] = so;
            }
        }
    }
    // ---- atoms: validated from the 8 message bytes at the token; numbers: a plain integer of up to 18 digits is parsed
    // here (sj_number.h parse_int_fast); only the others -- floats, long integers -- move to the global queue (k_numbers; the
    // order does not matter): a wave draws the slots of its queued numbers with one atomic (round 4 queued every number and
    // marked the parsed ones: 10 MB of entries on configs[1] that k_numbers read only to skip them)
    for (u32 j = (u32)tid; j < (SJ_EXPBIT(p, 10) ? 0u : D); j += BLK) {
        const u32 v = s_q[SB + B + j], idx = v & 0xfffu, o = T0 + ((v >> 12) & 0x1fffu);
        const u64 at = SJ_EXPBIT(p, 11) ? (u64)(tid & 7) * 8u : tbase + s_pos[idx];
        const u8 ak = (u8)(8u + ((v >> 26) & 3u));
        bool slow = false;
        if (ak == K_NUM) {
            u64 iv = 0;
            const bool fast = parse_int_fast(load8_guarded(mv, at), load8_guarded(mv, at + 8), load8_guarded(mv, at + 16), &iv);
            if (fast) {
                const u64 tw = (u64)'l' << 56;
                *reinterpret_cast<uint4 *>(arr_at(p.tape, o, 2)) = make_uint4((u32)tw, (u32)(tw >> 32), (u32)iv, (u32)(iv >> 32));
            }
            slow = !fast;
        } else {
            bad |= !atom_valid_word(load8_guarded(mv, at), p.len - at, ak);
            p.tape[o] = atom_word(ak);
        }
        const u64 mq = __ballot(slow);  // (the lanes still in the loop)
        if (mq != 0) {
            const int leader = __builtin_ctzll(mq);
            u32 qb = 0;
            if (lane == leader) qb = atomicAdd(&p.st->num_count, (u32)__popcll(mq));
            qb = (u32)__shfl((int)qb, leader, 64);
            const u32 slot = qb + (u32)__popcll(mq & ((1ull << lane) - 1ull));
            if (slow && slot < p.numq_cap) p.numq[slot] = make_uint4((u32)at, (u32)(at >> 32), o, 0u);
        }
    }
    // ---- brackets: matched inside the tile (sj_stage2.h bracket_resolve is the per-bracket statement, k_br_match the
    // device-wide form).  One rule for every bracket: the container of the gap in front of it -- the partner of a close, the
    // parent of an open -- is the bracket behind the last one in front with depth <= (depth in front - 1).  Depths here are
    // relative to the tile's start; a question that no bracket of the tile answers stays for k_br_match.
    if (!SJ_EXPBIT(p, 12)) {
        const u32 G = (B + 63u) / 64u;
        for (u32 g = (u32)wave; g < G; g += WAVES) {  // the minimum depth of every group of 64
            const u32 c = g * 64u + (u32)lane;
            i32 v = c < B ? tbr_depth(s_q[SB + c]) : 0x7fffffff;
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) {
                const i32 o = __shfl_xor(v, sft, 64);
                v = o < v ? o : v;
            }
            if (lane == 0) s_gmin[g] = v;
        }
        __syncthreads();
        const u64 lt = lane ? (~0ull >> (64 - lane)) : 0ull;  // lanes below this one
        for (u32 g = (u32)wave; g < G; g += WAVES) {  // wave-uniform
            const u32 c = g * 64u + (u32)lane;
            const bool valid = c < B;
            const u32 e = valid ? s_q[SB + c] : 0u;
            const i32 drel = valid ? tbr_depth(e) : 0x7fffffff;
            const u8 kd = tbr_kind(e);
            const u32 gap = tbr_gap(e), oc = T0 + tbr_off(e);
            const bool close = is_close(kd);
            const i32 q = close ? drel : drel - 2;  // depth in front of the bracket - 1 (relative)
            const bool root = valid && tp.d + q < 0;  // nothing is open in front of it: the root context
            const bool need = valid && !root;
            i32 res = -1;      // index (inside the tile) of the bracket in front of the partner / parent
            bool pend = need;  // not answered yet
            // inside the group: one ballot per distinct q (the last bracket in front with depth == q, see k_br_match)
            for (u64 pm = __ballot(pend); pm != 0;) {
                const i32 v = __builtin_amdgcn_readlane(q, __builtin_ctzll(pm));
                const u64 at = __ballot(drel == v) & lt;
                const bool mine = need && q == v;
                if (mine && at != 0) {
                    res = (i32)(g * 64u) + top_bit(at);
                    pend = false;
                }
                pm &= ~__ballot(mine);
            }
            // the groups in front, nearest first -- only those that hold a depth as low as the lowest question left
            u64 pm = __ballot(pend);
            if (pm != 0 && g > 0) {
                i32 qmax = pend ? q : (i32)0x80000000;
#pragma unroll
                for (int sft = 32; sft >= 1; sft >>= 1) {
                    const i32 o = __shfl_xor(qmax, sft, 64);
                    qmax = o > qmax ? o : qmax;
                }
                u64 cand = __ballot((u32)lane < g && s_gmin[(u32)lane < g ? lane : 0] <= qmax);
                while (cand != 0 && pm != 0) {
                    const u32 gp = (u32)top_bit(cand);
                    cand &= ~(1ull << gp);
                    const i32 dprev = tbr_depth(s_q[SB + gp * 64u + (u32)lane]);  // (a group in front is full)
                    for (u64 pp = pm; pp != 0;) {
                        const i32 v = __builtin_amdgcn_readlane(q, __builtin_ctzll(pp));
                        const u64 at = __ballot(dprev <= v);
                        const bool mine = pend && q == v;
                        if (mine && at != 0) {
                            res = (i32)(gp * 64u) + top_bit(at);
                            pend = false;
                        }
                        pp &= ~__ballot(mine);
                    }
                    pm = __ballot(pend);
                }
            }
            if (!valid) continue;
            bool done = false;
            if (root) {  // (a close is rejected here: it needs OBJ / ARR)
                bad |= !context_allowed(gap, CTX_ROOT);
                done = true;
            } else if (res >= 0) {
                const u32 ej = s_q[SB + (u32)res + 1u];  // partner (close) / parent (open)
                const u8 jk = tbr_kind(ej);
                bad |= !context_allowed(gap, jk == K_OPEN_OBJ ? (u8)CTX_OBJ : (u8)CTX_ARR);
                done = true;
                if (close) {  // payloads: annotate_previousloc (stage2_build_tape_amd64.go:335-336)
                    const u32 oj = T0 + tbr_off(ej);
                    const u64 wc = ((u64)(kd == K_CLOSE_OBJ ? '}' : ']') << 56) | (p.tape_base + oj);
                    const u64 wj = ((u64)(jk == K_OPEN_OBJ ? '{' : '[') << 56) | (p.tape_base + oc + 1);
                    if (tp.d + drel == 0) {  // a record: the root word in front of its open bracket and the one behind its close bracket
                        const u64 ro = ((u64)'r' << 56) | (p.tape_base + oc + 2), rc = ((u64)'r' << 56) | (p.tape_base + oj - 1);
                        *reinterpret_cast<uint4 *>(arr_at(p.tape, (u64)oj - 1, 2)) = make_uint4((u32)ro, (u32)(ro >> 32), (u32)wj, (u32)(wj >> 32));
                        *reinterpret_cast<uint4 *>(arr_at(p.tape, oc, 2)) = make_uint4((u32)wc, (u32)(wc >> 32), (u32)rc, (u32)(rc >> 32));
                    } else {
                        p.tape[oc] = wc;
                        p.tape[oj] = wj;
                    }
                }
            }
            // the compact bracket view of the whole message: every bracket keeps its depth (the matcher's questions pass
            // over it), only the live ones ask and write there
            const u32 cg = tp.bc + c;
            if (SJ_EXPBIT(p, 13)) continue;
            p.br_depth[cg] = tp.d + drel;
            p.br_off[cg] = oc;
            p.br_info[cg] = (u8)(kd | (gap << 4) | (done ? BR_DONE : 0u));
        }
    }
    if (__syncthreads_or(bad ? 1 : 0) && tid == 0) atomicOr(&p.st->err, 1u);
}

// ---- numbers (parseNumber, parse_number.go:65-135): one queued number per lane ------------------------------------
// The first 32 bytes of each number go to LDS (two unaligned 16-byte loads instead of one dependent byte load
// per digit); longer numbers fall back to the message itself.
// The same launch builds levels 1 and 2 of the min tree over the bracket depths in its blocks from `nblocks` on (both
// only need the emit pass's output; as launches of their own they cost 25 us): a block takes 4096 depths -- one level-2
// entry -- at a time, one wave per 64 of them (a level-1 entry), and folds the 64 minima through LDS.
__device__ __forceinline__ MinTree make_tree(const S2Dev &p);
// bid / nb: this block's index among the nb blocks that share the role (several roles run in one launch)
__device__ __forceinline__ void tree12_body(const S2Dev &p, u32 bid, u32 nb) {
    __shared__ i32 s_min[4];
    {
        const MinTree mt = make_tree(p);
        if (mt.nlev < 2) return;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const u64 n2 = (mt.size[1] + 63) / 64;  // level-2 entries (also when the tree has no level 2: one pass)
        for (u64 b = bid; b < n2; b += nb) {
            i32 wmin = 0x7fffffff;
            i32 dv[16];  // the sixteen groups of the wave: all loads first (one round trip instead of sixteen)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const u64 k = (b * 64 + (u64)wave * 16 + i) * 64 + lane;
                dv[i] = k < mt.size[0] ? p.br_depth[k] : 0x7fffffff;
            }
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const u64 g = b * 64 + (u64)wave * 16 + i;  // (wave-uniform)
                i32 v = dv[i];
#pragma unroll
                for (int sft = 32; sft >= 1; sft >>= 1) {
                    const i32 o = __shfl_xor(v, sft, 64);
                    v = o < v ? o : v;
                }
                if (lane == 0 && g < mt.size[1]) p.lev[1][g] = v;
                wmin = v < wmin ? v : wmin;
            }
            if (mt.nlev > 2) {  // (block-uniform)
                __syncthreads();
                if (lane == 0) s_min[wave] = wmin;
                __syncthreads();
                if (threadIdx.x == 0) {
                    i32 m = s_min[0];
                    for (int w = 1; w < 4; w++) m = s_min[w] < m ? s_min[w] : m;
                    p.lev[2][b] = m;
                }
            }
        }
    }
}
__device__ __forceinline__ void numbers_body(const S2Dev &p, u32 bid, u32 nblocks) {
    __shared__ u32 s_nb[256][9];  // 32-byte windows, 36-byte stride (bank-conflict free)
    u32 cnt = p.st->num_count;
    bool bad = cnt > p.numq_cap;  // (more numbers than a document that parses can hold: entries were dropped)
    if (bad) cnt = p.numq_cap;
    for (u32 j = bid * 256 + threadIdx.x; j < cnt; j += nblocks * 256) {
        const uint4 q4 = p.numq[j];
        const u64 at = ((u64)q4.y << 32) | q4.x;
        const uint2 q = make_uint2(q4.x, q4.z);  // (.y: the tape offset)
        const u64 rest = p.len - at;
        u32 *w = s_nb[threadIdx.x];
        if (rest >= 32) {
            const uint4 a = *reinterpret_cast<const uint4 *>(arr_at(p.msg, at, 16)), b = *reinterpret_cast<const uint4 *>(arr_at(p.msg, at + 16, 16));
            w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
            w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
        } else {
            u8 *wb = reinterpret_cast<u8 *>(w);
            for (u32 k = 0; k < (u32)rest; k++) wb[k] = p.msg[at + k];
        }
        u64 tag = 0, val = 0;
        u32 numlen = 0;
        const int st = parse_number_head32(reinterpret_cast<const u8 *>(w), arr_at(p.msg, at, rest), rest, &tag, &val, &numlen);
        if (st == NUM_FAIL) {
            bad = true;
        } else {
            p.tape[q.y] = tag;
            p.tape[q.y + 1] = val;
            if (st == NUM_NEEDS_BIGNUM) {
                const u32 slot = atomicAdd(&p.st->bignum_count, 1u);
                if (slot < p.numq_cap) p.bigq[slot] = q4;
            }
        }
    }
    if (bad) atomicOr(&p.st->err, 1u);
}
__global__ __launch_bounds__(256) void k_numbers(S2Dev p, u32 nblocks) {
    if (blockIdx.x >= nblocks) tree12_body(p, blockIdx.x - nblocks, gridDim.x - nblocks);
    else numbers_body(p, blockIdx.x, nblocks);
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
__device__ __forceinline__ void min_level_groups(const S2Dev &p, const MinTree &mt, int l, u64 first, u64 stride, int lane) {
    for (u64 g = first; g < mt.size[l]; g += stride) {  // one wave per group of 64 entries of the level below
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
// levels 3.. (n_br / 262144 entries and fewer): one block, level after level
__global__ __launch_bounds__(1024) void k_min_upper(S2Dev p) {
    const MinTree mt = make_tree(p);
    for (int l = 3; l < mt.nlev; l++) {
        min_level_groups(p, mt, l, (u64)(threadIdx.x >> 6), 16, threadIdx.x & 63);
        __threadfence_block();
        __syncthreads();
    }
}

// ---- bracket partners, contexts and the grammar check of every gap ----------------------------------------------
// Every bracket asks one previous-smaller-value question over the compact view (sj_stage2.h: bracket_resolve is the
// per-bracket statement): a close for its partner, an open for its parent.  Because consecutive depths differ
// by one, "the last bracket in front with depth <= q" is the last one with depth == q as long as the brackets
// in between are deeper, so a wave answers the 64 questions of one group with ballots over its own 64 depths
// (no memory traffic, no serial walk); what is not answered inside the group is looked up in the group in
// front and then through the min tree, the whole wave loading 64 entries per step.

// last k < `idx` with lev[L][k] <= v, resolved down to level 0 (wave-cooperative, uniform arguments); -1: none
__device__ i64 wave_psv_tree(const MinTree &mt, int L, u64 idx, i32 v, int lane) {
    u64 h = 0;
    for (;;) {
        if (idx == 0) return -1;
        const u64 hb = ((idx - 1) >> 6) << 6;
        const u64 e = hb + (u64)lane;
        const i32 val = e < idx ? mt.lev[L][e] : 0x7fffffff;
        const u64 m = __ballot(val <= v);
        if (m) {
            h = hb + (u64)top_bit(m);
            break;
        }
        if (hb == 0 || L + 1 >= mt.nlev) return -1;
        idx = hb >> 6;
        L++;
    }
    while (L > 0) {
        L--;
        const u64 e = (h << 6) + (u64)lane;
        const i32 val = e < mt.size[L] ? mt.lev[L][e] : 0x7fffffff;
        const u64 m = __ballot(val <= v);  // not empty: the parent entry is the minimum of these
        h = (h << 6) + (u64)top_bit(m);
    }
    return (i64)h;
}

// One rule for every bracket: the container that owns the gap in front of it -- its partner if it closes, its parent if
// it opens -- is the bracket behind the last one in front with depth <= (depth in front - 1); the type of that
// container (the root context if the depth in front is not positive) must be in the set of contexts the gap allows.
// A close writes both tape words of its pair; a pair at depth 0 is a record (or the document) and also writes the root
// words around it: the open-root word in front of it points behind its close-root word and vice versa (startContinue
// :196-221, succeed :428-442) -- the same lines of the tape, one 16-byte store each.
__device__ __forceinline__ void br_match_body(const S2Dev &p, u32 bid, u32 nb) {
    const u32 n_br = p.st->n_br;
    const MinTree mt = make_tree(p);
    const int lane = threadIdx.x & 63;
    const u64 lt = lane ? (~0ull >> (64 - lane)) : 0ull;  // lanes below this one
    const u32 waves = nb * 4;
    const bool store = p.st->tape_len <= p.tape_cap;  // (cannot fail: the launcher sizes the tape for 2n+2 words)
    bool bad = false;
    // everything that does not depend on an answer is requested together -- the group's depths, kinds and tape offsets,
    // and the depths of the group in front (where most questions that leave the group end) -- and one group ahead:
    // the kernel is bound by the latency of its dependent loads (3 M VALU instructions in 160 k cycles), a wave works
    // on group g while the loads of group g + waves are in flight
    struct Group {
        i32 dep, pd;
        u32 oc;
        u8 info;
    };
    auto load_group = [&](u32 g) {
        Group r{0x7fffffff, 0x7fffffff, 0u, (u8)K_BAD};
        const u32 c = g * 64 + (u32)lane;
        if ((u64)g * 64 < n_br) {
            if (c < n_br) {
                r.dep = p.br_depth[c];
                r.info = p.br_info[c];
                r.oc = p.br_off[c];
            }
            if (g > 0) r.pd = p.br_depth[(u64)(g - 1) * 64 + lane];
        }
        return r;
    };
    u32 g = bid * 4 + (threadIdx.x >> 6);
    Group nx = load_group(g);
    for (; (u64)g * 64 < n_br; g += waves) {  // wave-uniform
        const u32 c = g * 64 + (u32)lane;
        const Group cur = nx;
        // (a bracket whose container lies in its own tile was resolved there, k_s2_emit_planes: it keeps its depth for the
        // questions that pass over it and asks nothing itself)
        const bool valid = c < n_br && !(cur.info & BR_DONE);
        nx = load_group(g + waves);
        const i32 dep = cur.dep, pd = cur.pd;
        const u8 info = cur.info;
        const u32 oc = cur.oc;
        const u8 kd = info & 15u;
        const bool close = is_close(kd);
        const i32 q = close ? dep : dep - 2;  // depth in front of the bracket - 1
        const bool need = valid && q >= 0;
        i64 res = -1;      // compact index of the bracket in front of the partner / parent
        bool pend = need;  // not answered yet
        // inside the group: one ballot per distinct q
        for (u64 pm = __ballot(pend); pm != 0;) {
            const i32 v = __builtin_amdgcn_readlane(q, __builtin_ctzll(pm));
            const u64 at = __ballot(dep == v) & lt;
            const bool mine = need && q == v;
            if (mine && at != 0) {
                res = (i64)g * 64 + top_bit(at);
                pend = false;
            }
            pm &= ~__ballot(mine);
        }
        // the group in front, then the tree: one question per distinct q that is still open
        u64 pm = __ballot(pend);
        if (pm != 0 && g > 0) {
            while (pm != 0) {
                const i32 v = __builtin_amdgcn_readlane(q, __builtin_ctzll(pm));
                const u64 at = __ballot(pd <= v);
                const i64 k = at ? (i64)(g - 1) * 64 + top_bit(at)
                                 : (mt.nlev > 1 ? wave_psv_tree(mt, 1, (u64)(g - 1), v, lane) : -1);
                const bool mine = pend && q == v;
                if (mine) {
                    res = k;
                    pend = false;
                }
                pm &= ~__ballot(mine);
            }
        }
        if (!valid) continue;
        const u32 gap = (u32)(info >> 4) & 7u;  // contexts the gap that ends with this bracket allows
        if (q < 0) {  // nothing is open in front of it: the root context (a close needs OBJ / ARR and is rejected)
            bad |= !context_allowed(gap, CTX_ROOT);
            continue;
        }
        const u32 j = (u32)(res + 1);  // partner (close) / parent (open); bracket 0 if nothing was found
        const u8 jk = (u8)(p.br_info[j] & 15u);
        bad |= !context_allowed(gap, jk == K_OPEN_OBJ ? (u8)CTX_OBJ : (u8)CTX_ARR);
        if (close && store) {  // payloads: annotate_previousloc (stage2_build_tape_amd64.go:335-336)
            const u32 oj = p.br_off[j];
            const u64 wc = ((u64)(kd == K_CLOSE_OBJ ? '}' : ']') << 56) | (p.tape_base + oj);
            const u64 wj = ((u64)(jk == K_OPEN_OBJ ? '{' : '[') << 56) | (p.tape_base + oc + 1);
            if (dep == 0) {  // a record: the root word in front of its open bracket and the one behind its close bracket
                const u64 ro = ((u64)'r' << 56) | (p.tape_base + oc + 2), rc = ((u64)'r' << 56) | (p.tape_base + oj - 1);
                *reinterpret_cast<uint4 *>(arr_at(p.tape, (u64)oj - 1, 2)) = make_uint4((u32)ro, (u32)(ro >> 32), (u32)wj, (u32)(wj >> 32));
                *reinterpret_cast<uint4 *>(arr_at(p.tape, oc, 2)) = make_uint4((u32)wc, (u32)(wc >> 32), (u32)rc, (u32)(rc >> 32));
            } else {
                p.tape[oc] = wc;
                p.tape[oj] = wj;
            }
        }
    }
    if (bad) atomicOr(&p.st->err, 1u);
}
__global__ __launch_bounds__(256) void k_br_match(S2Dev p) { br_match_body(p, blockIdx.x, gridDim.x); }
// ---- the per-string fallback (no masks): every string that is copied is walked again, now writing (string_walk) -------
__global__ __launch_bounds__(256) void k_emit_strings(S2Dev p) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= token_count(p)) return;
    if (p.kind[i] != K_STRING) return;
    const u32 dl = p.dlen[i];
    if (dl == DLEN_INVALID || !(dl & DLEN_COPY)) return;
    const u32 so = p.str_off[i];
    if ((u64)so + (dl & ~DLEN_COPY) > p.strings_cap) return;
    const MsgView mv{p.msg, p.len};
    u32 sl, dl2;
    string_walk(mv, p.pos[i], arr_at(p.strings, so, dl & ~DLEN_COPY), &sl, &dl2);
}

// ---- exact tie-break for >19-digit mantissas whose neighbours disagree --------------------------------------------
__global__ __launch_bounds__(64) void k_bignum(S2Dev p) {
    u32 cnt = p.st->bignum_count;
    if (cnt > p.numq_cap) cnt = p.numq_cap;
    Big X, Y;
    for (u32 q = blockIdx.x * 64 + threadIdx.x; q < cnt; q += gridDim.x * 64) {
        const uint4 e = p.bigq[q];
        const u64 at = ((u64)e.y << 32) | e.x;
        const u32 o = e.z;
        u64 tag, val;
        u32 numlen = 0;
        const u64 room = p.len - at;
        (void)parse_number(arr_at(p.msg, at, room), (u32)(room < 0x7fffffffu ? room : 0x7fffffffu), &tag, &val, &numlen);
        const u64 cand = p.tape[o + 1];
        const u64 sign = cand & 0x8000000000000000ull;
        const u64 r = bignum_round(arr_at(p.msg, at, numlen), numlen, cand & ~0x8000000000000000ull, X, Y);
        if (r == 0x7ff0000000000000ull) atomicOr(&p.st->err, 1u);  // strconv.ErrRange
        p.tape[o + 1] = r | sign;
    }
}

// ---- launcher -----------------------------------------------------------------------------------------------
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// grid of the kernels whose waves walk over their work items: as many 256-thread blocks as the device holds at once
// (8 per CU), fewer if there is less work
template <typename K>
static u32 persistent_blocks(K kernel, u64 want) {
    int dev = 0, cus = 256, per_cu = 4;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    const u64 cap = (u64)cus * (u64)per_cu;
    return (u32)(want < 1 ? 1 : (want < cap ? want : cap));
}

// the zeroed region: S2State and the segment slots of the two scans (zeroed by stage 1's preparation kernel)
size_t stage2_zero_bytes() { return (sizeof(S2State) + 3 * SCAN_SEGS * sizeof(SegSlot) + 255) / 256 * 256; }

size_t stage2_workspace_bytes(size_t n) {
    size_t b = 256;
    b += align_up(n + 16, 256);                     // br_info
    b += align_up(n * 4, 256) * 5;                  // dlen str_off nl_off br_depth br_off
    b += align_up((n / 2 + 8) * 16, 256) * 2;       // numq bigq
    b += align_up((n / 2 + 8) * 16, 256);           // sinfo (WithCopyStrings(false): 8 bytes per string)
    const size_t tiles = (n + S2_TILE - 1) / S2_TILE + 1;
    b += align_up(tiles * sizeof(TileAgg), 256);
    size_t lv = n;
    for (int l = 1; l < MinTree::MAXLEV; l++) {
        lv = (lv + 63) / 64;
        b += align_up(lv * 4, 256);
    }
    return b + 4096;
}

// carve the device view out of the workspaces (deterministic: both phases rebuild the same view)
static S2Dev stage2_view(const S2Args &a) {
    S2Dev p;
    const size_t n = a.n;
    char *w = reinterpret_cast<char *>(a.ws);
    auto carve = [&](size_t bytes) {
        char *r = w;
        w += align_up(bytes, 256);
        return r;
    };
    p.st = reinterpret_cast<S2State *>(a.ws_zero);
    p.seg_units = reinterpret_cast<SegSlot *>(reinterpret_cast<char *>(a.ws_zero) + sizeof(S2State));
    p.seg_tiles = p.seg_units + SCAN_SEGS;
    p.seg_ustr = p.seg_tiles + SCAN_SEGS;
    p.msg = SJ_ARR(reinterpret_cast<const u8 *>(a.d_msg), a.len, A_MSG);
    p.len = a.len;
    p.pos = SJ_ARR(a.d_pos, n, A_POS);
    p.n = (u32)n;
    p.n_dev = a.n_dev;
    p.ndjson = a.flags & 1u;
    p.copy_strings = (a.flags >> 1) & 1u;
    p.s1_has_starter = a.s1_has_starter;
    p.kind = SJ_ARR(a.d_kind, n, A_KIND);
    p.br_info = SJ_ARR(reinterpret_cast<u8 *>(carve(n + 16)), n + 16, A_BR_INFO);
    p.dlen = SJ_ARR(reinterpret_cast<u32 *>(carve(n * 4)), n, A_DLEN);
    p.str_off = SJ_ARR(reinterpret_cast<u32 *>(carve(n * 4)), n, A_STR_OFF);
    p.nl_off = SJ_ARR(reinterpret_cast<u32 *>(carve(n * 4)), n, A_NL_OFF);
    p.numq_cap = (u32)(n / 2 + 8);
    p.numq = SJ_ARR(reinterpret_cast<uint4 *>(carve((size_t)p.numq_cap * 16)), p.numq_cap, A_NUMQ);
    p.bigq = SJ_ARR(reinterpret_cast<uint4 *>(carve((size_t)p.numq_cap * 16)), p.numq_cap, A_BIGQ);
    p.tile_unit = nullptr;
    u32 *const scq_mem = reinterpret_cast<u32 *>(carve(((size_t)n / 2 + 8) * 16));  // >= 4 (n + 2) bytes
    p.br_depth = SJ_ARR(reinterpret_cast<i32 *>(carve(n * 4)), n, A_BR_DEPTH);
    p.br_off = SJ_ARR(reinterpret_cast<u32 *>(carve(n * 4)), n, A_BR_OFF);
    p.tiles = (u32)((n + S2_TILE - 1) / S2_TILE);
    {
        const u64 us = ((u64)a.len + 64 + 4095) / 4096;
        // (one CU moves ~60 GB/s: a single block only where its share is a few tens of kilobytes -- measured: one block
        // for configs[4]'s 19 500 tile aggregates took 73 us, 32 blocks 18)
        p.unit_segs = us <= 16384 ? 1 : SCAN_SEGS;
        p.tile_segs = p.tiles <= 1024 ? 1 : SCAN_SEGS;
    }
    p.agg = SJ_ARR(reinterpret_cast<TileAgg *>(carve((size_t)(p.tiles + 1) * sizeof(TileAgg))), p.tiles + 1, A_AGG);
    p.nlev = 1;
    p.lev[0] = nullptr;
    p.lev_size[0] = n;
    {
        u64 sz = n;
        while (sz > 64 && p.nlev < MinTree::MAXLEV) {
            sz = (sz + 63) / 64;
            p.lev[p.nlev] = SJ_ARR(reinterpret_cast<i32 *>(carve(sz * 4)), sz, A_LEV);
            p.lev_size[p.nlev] = sz;
            p.nlev++;
        }
    }
    p.tape = SJ_ARR(a.d_tape, a.tape_cap, A_TAPE);
    p.strings = SJ_ARR(a.d_strings, a.strings_cap, A_STRINGS);
    p.keyflag = nullptr;
    if (a.d_keyflag) p.keyflag = SJ_ARR(a.d_keyflag, a.tape_cap / 2 + 8, A_KEYFLAG);
    p.tape_cap = a.tape_cap;
    p.strings_cap = a.strings_cap;
    p.tape_base = a.tape_base;
    p.strings_base = a.strings_base;
    p.msg_base = a.msg_base;
    // byte-parallel strings: only when every string is copied and stage 1 left its masks
    const uintptr_t addr = reinterpret_cast<uintptr_t>(a.d_msg);
    p.sv.lead = addr & 63;
    p.sv.end = p.sv.lead + a.len;
    // (k_str_emit reads whole 64-byte chunks: the arenas and the callers' device buffers carry that much slack)
    p.sv.base = SJ_ARR(reinterpret_cast<const u8 *>(addr & ~(uintptr_t)63), (p.sv.end + 63) / 64 * 64, A_SV_BASE);
    p.sv.qm = p.sv.st = nullptr;
    p.sv.unit_h = nullptr;
    p.sv.unit_slow = nullptr;
    p.rec = nullptr;
    p.unit_cnt = nullptr;
    p.unit_str = nullptr;
    p.soff = nullptr;
    p.sinfo = nullptr;
    p.unit_tq = nullptr;
    p.soff_cap = 0;
    p.unit_copy = nullptr;
    p.units = 0;
    p.exp = 0;
#if defined(SJ_EXP)
    if (const char *e = getenv("SJHIP_EXP")) p.exp = (u32)strtoul(e, nullptr, 0);
#endif
    p.str_out = p.strings;
    if (a.str_aux) {
        const StrAux x = str_aux_layout(a.str_aux, (size_t)p.sv.end);
        p.sv.qm = SJ_ARR((const u64 *)x.qm, x.chunks, A_SV_QM);
        p.sv.st = SJ_ARR((const u64 *)x.st, x.chunks, A_SV_ST);
        p.sv.unit_h = SJ_ARR((const u8 *)x.unit_h, x.units, A_SV_UNIT_H);
        p.sv.unit_slow = SJ_ARR((const u64 *)x.unit_slow, x.units, A_SV_UNIT_SLOW);
        p.rec = SJ_ARR(reinterpret_cast<ChunkRec *>(x.rec), x.chunks, A_REC);
        p.unit_cnt = SJ_ARR(x.unit_cnt, x.units, A_UNIT_CNT);
        // the strings of the message are numbered, their Strings.B offsets go where the per-string fallback keeps its lengths
        // (dlen and str_off are adjacent: 2 x align_up(4n, 256) >= 4 (n + 2) bytes)
        p.unit_str = SJ_ARR(x.unit_str, x.units, A_UNIT_STR);
        p.tile_unit = SJ_ARR((const u32 *)x.tile_unit, x.units + 1, A_TILE_UNIT);
        p.soff_cap = (u32)(n + 2 < 0xffffffffull ? n + 2 : 0xffffffffull);
        p.soff = SJ_ARR(arr_raw(p.dlen), p.soff_cap, A_SOFF);
        if (!p.copy_strings) {  // the states at the units' ends (k_measure), the closing quotes
            p.unit_copy = SJ_ARR(x.unit_copy, x.units, A_UNIT_COPY);
            p.sinfo = SJ_ARR(reinterpret_cast<uint2 *>(scq_mem), p.soff_cap, A_STRQ);  // (8 n + 128 bytes)
            p.unit_tq = SJ_ARR(x.unit_tq, x.units, A_UNIT_TQ);
        }
        p.units = (p.sv.end + 4095) / 4096;  // units that hold message bytes (stage 1 wrote their masks)
    }
    return p;
}

// nl_off (tape offsets of the record-separating root pairs) of the last run in workspace `ws`: query.hip
void stage2_records_view(void *ws, size_t n_tokens, const uint32_t **nl_off) {
    S2Args a = {};
    a.n = n_tokens;
    a.ws = ws;
    const S2Dev p = stage2_view(a);
    *nl_off = arr_raw(p.nl_off);
}

// Phase 1: string masks, tile aggregates and the two device-wide scans.  Afterwards S2State holds tape_len /
// strings_len of this message (what an NDJSON shard exchanges with the other shards).  The zeroed region must be zero.
hipError_t stage2_launch_measure(const S2Args &a) {
    const S2Dev p = stage2_view(a);
    if (a.n == 0) return hipSuccess;
    {
        // (WithCopyStrings(false) visits every unit in the string half: all of the device's slots; otherwise only the units with
        // a \u escape are worked on: half of them)
        u32 mblocks = p.sv.qm ? persistent_blocks(k_measure, (p.units + 3) / 4) : 0;
        if (p.sv.qm && p.copy_strings) mblocks = mblocks / 2 + 1;
        hipLaunchKernelGGL(k_measure, dim3(mblocks + p.tiles), dim3(RD_BLOCK), 0, a.stream, p, mblocks);
    }
    hipLaunchKernelGGL(k_scans, dim3(p.sv.qm ? p.unit_segs * (p.unit_str ? 2 : 1) + p.tile_segs : p.tile_segs), dim3(1024), 0, a.stream, p);
    return hipGetLastError();
}

// Phase 2: tape words, bracket matching with the grammar check and the root words, Strings.B.  The three bases
// rebase every index the tape stores (tape positions, Strings.B offsets, Message offsets): 0 for a whole message,
// the exclusive prefix sums over the preceding shards for an NDJSON shard.
hipError_t stage2_launch_emit(const S2Args &a) {
    const S2Dev p = stage2_view(a);
    const size_t n = a.n;
    if (n == 0) return hipSuccess;
    const u32 gb = (u32)((n + 255) / 256);
    // The string bytes run in front of the tape kernels: every string copied -- k_str_emit leaves the Strings.B offset of
    // every string of the message (soff[]) for k_s2_emit_planes; WithCopyStrings(false) -- the compaction of the units that
    // hold strings to be copied goes to the scratch buffer k_emit_strings takes them from.  (Rounds 3 and 4 could also run
    // k_str_emit on a second stream beside the tape kernels, SJHIP_S2_OVERLAP: -2 % / +3 % on the two workloads, never the
    // default; gone with the records the tape kernels read then.)
    const int mode = !p.sv.qm ? 0 : (p.copy_strings ? 1 : 2);
    if (mode == 1) hipLaunchKernelGGL(k_str_emit<false>, dim3(persistent_blocks(k_str_emit<false>, (p.units + 3) / 4)), dim3(256), 0, a.stream, p);
    else if (mode == 2) hipLaunchKernelGGL(k_str_emit<true>, dim3(persistent_blocks(k_str_emit<true>, (p.units + 3) / 4)), dim3(256), 0, a.stream, p);
    {
        static const int items = getenv("SJHIP_S2_ITEMS") ? atoi(getenv("SJHIP_S2_ITEMS")) : 8;  // tokens per lane (A/B)
        const bool wide = mode != 0 && p.len >= (1ull << 32);
#define SJ_EMIT(M, I)                                                                                                     \
    do {                                                                                                                  \
        if (wide) hipLaunchKernelGGL((k_s2_emit_planes<M, I, (M) != 0>), dim3(p.tiles), dim3(S2_TILE / I), 0, a.stream, p); \
        else hipLaunchKernelGGL((k_s2_emit_planes<M, I, false>), dim3(p.tiles), dim3(S2_TILE / I), 0, a.stream, p);       \
    } while (0)
        if (items == 16) {
            if (mode == 0) SJ_EMIT(0, 16);
            else if (mode == 1) SJ_EMIT(1, 16);
            else SJ_EMIT(2, 16);
        } else {
            if (mode == 0) SJ_EMIT(0, 8);
            else if (mode == 1) SJ_EMIT(1, 8);
            else SJ_EMIT(2, 8);
        }
#undef SJ_EMIT
    }
    {  // numbers, and beside them levels 1 and 2 of the min tree (grid-stride: the kernel uses the real bracket count).
        // (Measured and dropped in round 4: the numbers beside the bracket matcher instead -- both wait on dependent
        // loads, but one launch of the two took 97 + 15 us for the tree levels where the two launches take 50 + 49.)
        const u32 nblocks = gb < 2048 ? gb : 2048;
        const u64 want = p.nlev > 1 ? (p.lev_size[1] + 63) / 64 : 0;  // one block per 4096 depths at a time
        const u32 lblocks = (u32)(want < 2048 ? want : 2048);
        hipLaunchKernelGGL(k_numbers, dim3(nblocks + lblocks), dim3(256), 0, a.stream, p, nblocks);
    }
    if (mode == 0) hipLaunchKernelGGL(k_emit_strings, dim3(gb), dim3(256), 0, a.stream, p);
    if (p.nlev > 3) hipLaunchKernelGGL(k_min_upper, dim3(1), dim3(1024), 0, a.stream, p);
    hipLaunchKernelGGL(k_br_match, dim3(gb < 2048 ? gb : 2048), dim3(256), 0, a.stream, p);  // (8 waves per SIMD resident)
    return hipGetLastError();
}

// ---- small documents: state + tape + Strings.B straight into pinned host memory -------------------------------------
__global__ __launch_bounds__(256) void k_pack(const S2State *st, const u64 *tape, const u8 *strings, u32 masks, u8 *dst, u64 cap) {
    const S2State s = *st;
    const u64 tl = s.tape_len, sl = masks ? s.strings_len_masks : s.strings_len;
    const bool ok = s.err == 0 && s.bignum_count == 0 && STAGE2_PACK_HEAD + 8 * tl + sl <= cap;
    const u32 tid = blockIdx.x * 256 + threadIdx.x, nthr = gridDim.x * 256;
    if (tid == 0) {
        *reinterpret_cast<S2State *>(dst) = s;
        *reinterpret_cast<u64 *>(dst + 64) = ok ? 1ull : 0ull;
    }
    if (!ok) return;
    uint4 *dt = reinterpret_cast<uint4 *>(dst + STAGE2_PACK_HEAD);
    const uint4 *st4 = reinterpret_cast<const uint4 *>(tape);  // (the arena is 16-byte aligned)
    for (u64 i = tid; i < tl / 2; i += nthr) dt[i] = st4[i];
    if ((tl & 1) && tid == 0) reinterpret_cast<u64 *>(dst + STAGE2_PACK_HEAD)[tl - 1] = tape[tl - 1];
    u8 *ds = dst + STAGE2_PACK_HEAD + 8 * tl;  // 8-byte aligned
    const u64 s8 = sl / 8;
    for (u64 i = tid; i < s8; i += nthr) reinterpret_cast<u64 *>(ds)[i] = reinterpret_cast<const u64 *>(strings)[i];
    if (tid < (sl & 7)) ds[s8 * 8 + tid] = strings[s8 * 8 + tid];
}
hipError_t stage2_launch_pack(const S2Args &a, void *h_dst, size_t cap) {
    hipLaunchKernelGGL(k_pack, dim3(128), dim3(256), 0, a.stream, (const S2State *)a.ws_zero, (const u64 *)a.d_tape, (const u8 *)a.d_strings,
                       a.str_aux ? 1u : 0u, (u8 *)h_dst, (u64)cap);
    return hipGetLastError();
}

// ---- the 64-byte state to pinned host memory: one wave instead of a copy command (the runtime's blit kernel for 64 bytes
// takes 4.2-4.4 us in every trace of the round; this one is a load, eight 8-byte stores and the end of a kernel) ----------
__global__ __launch_bounds__(64) void k_state_out(const unsigned long long *st, unsigned long long *dst) {
    if (threadIdx.x < sizeof(S2State) / 8)
        __hip_atomic_store(&dst[threadIdx.x], st[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t stage2_launch_state_out(const S2Args &a, void *h_dst) {
    hipLaunchKernelGGL(k_state_out, dim3(1), dim3(64), 0, a.stream, (const unsigned long long *)a.ws_zero, (unsigned long long *)h_dst);
    return hipGetLastError();
}

// ---- debug build (-DSJ_DEBUG_BOUNDS, sj_bounds.h): the record of the first out-of-bounds access, read and cleared -------
// returns 0 in the product build; 1 in the debug build with *hits = accesses that were out of bounds since the last call
int stage2_debug_bounds(unsigned *hits, unsigned *id, unsigned long long *index, unsigned long long *size) {
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
#if defined(SJ_DEBUG_BOUNDS)
__global__ void k_bounds_selftest(Arr<u32> a, u32 *out) {
    u32 v = a[3];               // in bounds
    v += a[16];                 // one behind the end: recorded, redirected to a[15]
    v += *arr_at(a, 14, 4);     // elements 14..17: recorded, redirected to 12..15
    *out = v;
}
#endif
// the checker checked: -1 in the product build, else the number of violations a kernel with two deliberate ones recorded
int stage2_debug_bounds_selftest() {
#if defined(SJ_DEBUG_BOUNDS)
    u32 *d = nullptr, host[17];
    for (u32 k = 0; k < 17; k++) host[k] = k;
    if (hipMalloc((void **)&d, sizeof host) != hipSuccess) return -2;
    (void)hipMemcpy(d, host, sizeof host, hipMemcpyHostToDevice);
    unsigned hits, id;
    unsigned long long index, size;
    (void)stage2_debug_bounds(&hits, &id, &index, &size);  // clear
    hipLaunchKernelGGL(k_bounds_selftest, dim3(1), dim3(1), 0, 0, Arr<u32>(d, 16, A_SELFTEST), d + 16);
    (void)hipDeviceSynchronize();
    u32 out = 0;
    (void)hipMemcpy(&out, d + 16, 4, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    (void)stage2_debug_bounds(&hits, &id, &index, &size);
    if (out != 3 + 15 + 12 || id != A_SELFTEST || index != 16 || size != 16) return -3;
    return (int)hits;
#else
    return -1;
#endif
}

// exact tie-break of the queued numbers (S2State::bignum_count != 0 after the emit phase: rare)
hipError_t stage2_launch_bignum(const S2Args &a) {
    const S2Dev p = stage2_view(a);
    hipLaunchKernelGGL(k_bignum, dim3(64), dim3(64), 0, a.stream, p);
    return hipGetLastError();
}

}  // namespace sj
