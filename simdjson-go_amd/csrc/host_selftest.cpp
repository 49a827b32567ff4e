// host_selftest.cpp -- CPU replay of the per-chunk device arithmetic in sj_chunk.h.
//
// NOT part of the product path: built only by the test-suite (g++, no HIP) so that the
// lane-local stage-1 math and the "peek" carry rules of stage1.hip can be checked against
// the oracle without a GPU.  The chunk loop below plays the role of the lanes; every
// cross-chunk input is derived exactly the way the kernel derives it (memory peeks and a
// running parity / count), never from the reference-style carried scalars.
#include <stdlib.h>
#include <stdint.h>
#include <string.h>

#include "sj_chunk.h"

using namespace sj;

static u32 peek_backslash_parity(const u8 *msg, u64 p) {
    u32 n = 0;
    while (p > 0 && msg[p - 1] == '\\') {
        n++;
        p--;
    }
    return n & 1u;
}

static u32 peek_pseudo_pred(const u8 *msg, u64 p) {
    if (p == 0) return 1;
    const u8 b = msg[p - 1];
    if (b == ' ' || b == '\t' || b == '\n' || b == '\r') return 1;
    if (b == '{' || b == '}' || b == '[' || b == ']' || b == ':' || b == ',') return 1;
    if (b == '"') return peek_backslash_parity(msg, p - 1) ^ 1u;
    return 0;
}

// optional per-chunk outputs (what the stage-1 kernel leaves for the string kernels; absolute in-string mask,
// i.e. unit state 0)
static uint64_t *g_qm = nullptr, *g_q = nullptr, *g_st = nullptr, *g_slow = nullptr;

extern "C" int sj_selftest_stage1(const uint8_t *msg, size_t len, int ndjson, uint32_t *pos_out, size_t cap,
                                  size_t *n_out, uint32_t *error, uint32_t *ends_in_quote) {
    u32 par = 0;
    size_t n = 0;
    u32 err = 0;
    for (u64 off = 0; off < len; off += 64) {
        u8 chunk[64];
        memset(chunk, 0x20, 64);
        memcpy(chunk, msg + off, len - off < 64 ? len - off : 64);
        u32 w[16];
        memcpy(w, chunk, 64);
        const Classes c = classify(w);
        const u32 carry_in = off == 0 ? 0 : peek_backslash_parity(msg, off);
        const u64 escaped = escaped_mask(c.bs, carry_in);
        const u64 quote_bits = c.quote & ~escaped;  // what the kernel runs
        {
            u32 carry_out;  // the reference's formulation must agree at the quotes
            if ((c.quote & ~odd_backslash_ends(c.bs, carry_in, carry_out)) != quote_bits) err |= 0x100;
        }
        u64 quote_mask = prefix_xor(quote_bits);
        if (par) quote_mask = ~quote_mask;
        par ^= (u32)popc64(quote_bits) & 1u;
        if (g_qm) {
            g_qm[off >> 6] = quote_mask;
            g_q[off >> 6] = quote_bits;
            g_st[off >> 6] = c.bs & ~escaped;
            if ((escaped & ~c.esc1) != 0) g_slow[off >> 12] |= 1ull << ((off >> 6) & 63);
        }
        if (c.ctrl & quote_mask) err = 1;
        const u32 pp_in = peek_pseudo_pred(msg, off);
        u64 s = finalize(c.structs, c.ws, quote_mask, quote_bits, pp_in);
        if (ndjson) s |= c.nl & ~quote_mask;
        while (s) {
            const int b = ctz64(s);
            if (n < cap) pos_out[n] = (u32)(off + b);
            n++;
            s &= s - 1;
        }
    }
    *n_out = n;
    *error = err;
    *ends_in_quote = par;
    return 0;
}

// raw class masks of one chunk, for the per-routine KAT replay
extern "C" void sj_selftest_classify(const uint8_t *in64, uint64_t *out6) {
    u32 w[16];
    memcpy(w, in64, 64);
    const Classes c = classify(w);
    // the butterfly transposition classify() uses against the one-plane-at-a-time dot4 form
    u64 pl[8];
    transpose_planes(w, pl);
    const u64 ref[8] = {plane_of<0>(w), plane_of<1>(w), plane_of<2>(w), plane_of<3>(w),
                        plane_of<4>(w), plane_of<5>(w), plane_of<6>(w), plane_of<7>(w)};
    for (int k = 0; k < 8; k++)
        if (pl[k] != ref[k]) abort();
    out6[0] = c.bs;
    out6[1] = c.quote;
    out6[2] = c.structs;
    out6[3] = c.ws;
    out6[4] = c.ctrl;
    out6[5] = c.nl;
    out6[6] = c.esc1;
    for (int j = 0; j < 4; j++) out6[7 + j] = c.kp[j];
}
extern "C" uint64_t sj_selftest_odd_backslash(uint64_t bs, uint32_t carry_in, uint32_t *carry_out) {
    return odd_backslash_ends(bs, carry_in, *carry_out);
}
extern "C" uint64_t sj_selftest_prefix_xor(uint64_t x) { return prefix_xor(x); }
extern "C" uint64_t sj_selftest_finalize(uint64_t st, uint64_t ws, uint64_t qm, uint64_t qb, uint32_t pp) {
    return finalize(st, ws, qm, qb, pp);
}

// ---- number parsing replay (sj_number.h + sj_bignum.h) ----------------------------------------
#include "sj_bignum.h"
#include "sj_number.h"

extern "C" int sj_selftest_parse_number(const uint8_t *buf, size_t len, uint64_t *tag, uint64_t *val, int *used_bignum) {
    u32 numlen = 0;
    *used_bignum = 0;
    u8 head[32] = {0};  // like k_numbers: the first 32 bytes are parsed from a copy
    for (size_t k = 0; k < 32 && k < len; k++) head[k] = buf[k];
    int st = parse_number_head32(head, buf, len, tag, val, &numlen);
    if (st == NUM_NEEDS_BIGNUM) {
        static Big X, Y;
        *used_bignum = 1;
        const u64 sign = *val & 0x8000000000000000ull;
        const u64 r = bignum_round(buf, numlen, *val & ~0x8000000000000000ull, X, Y);
        if (r == 0x7ff0000000000000ull) return 0;
        *val = r | sign;
        st = NUM_OK;
    }
    return st;
}

// the integer fast path of k_s2_emit on its own: 1 and *val if it takes the number at buf[0..len), else 0
extern "C" int sj_selftest_int_fast(const uint8_t *buf, size_t len, uint64_t *val) {
    u64 w[3] = {0, 0, 0};
    for (size_t k = 0; k < 24 && k < len; k++) w[k >> 3] |= (u64)buf[k] << (8 * (k & 7));
    return parse_int_fast(w[0], w[1], w[2], val) ? 1 : 0;
}

// ---- whole parse replay: stage 1 + the data-parallel stage 2 of sj_stage2.h ---------------------
// Every "kernel" of stage2.hip is a plain loop over the same per-token functions here; the scans
// are sequential sums.  Used by the CPU test-suite to check the algorithm against the oracle.
#include <stdlib.h>

#include <vector>

#include "sj_host.h"
#include "sj_stage2.h"
#include "sj_ftoa.h"
#include "sj_planes.h"
#include "sj_tok16.h"
#include "sj_strings.h"

extern "C" void sj_selftest_trim(const uint8_t *msg, size_t len, size_t *off, size_t *out_len) {
    trim_space(msg, len, off, out_len);
}

static int selftest_parse(const uint8_t *msg0, size_t len0, uint32_t flags, uint64_t tape_base, uint64_t strings_base,
                          uint64_t msg_base, uint64_t **tape_out, size_t *tape_len, uint8_t **strings_out,
                          size_t *strings_len, size_t *msg_off, size_t *msg_len);
extern "C" int sj_selftest_parse(const uint8_t *msg0, size_t len0, uint32_t flags, uint64_t **tape_out, size_t *tape_len,
                                 uint8_t **strings_out, size_t *strings_len, size_t *msg_off, size_t *msg_len) {
    return selftest_parse(msg0, len0, flags, 0, 0, 0, tape_out, tape_len, strings_out, strings_len, msg_off, msg_len);
}
// one NDJSON shard of a larger document: the tape is emitted with rebased indices (sj_stage2.h, Tokens::*_base)
extern "C" int sj_selftest_parse_shard(const uint8_t *msg0, size_t len0, uint32_t flags, uint64_t tape_base,
                                       uint64_t strings_base, uint64_t msg_base, uint64_t **tape_out, size_t *tape_len,
                                       uint8_t **strings_out, size_t *strings_len, size_t *msg_off, size_t *msg_len) {
    return selftest_parse(msg0, len0, flags, tape_base, strings_base, msg_base, tape_out, tape_len, strings_out,
                          strings_len, msg_off, msg_len);
}
static int selftest_parse(const uint8_t *msg0, size_t len0, uint32_t flags, uint64_t tape_base, uint64_t strings_base,
                          uint64_t msg_base, uint64_t **tape_out, size_t *tape_len, uint8_t **strings_out,
                          size_t *strings_len, size_t *msg_off, size_t *msg_len) {
    const bool ndjson = flags & 1;
    const bool copy_all = flags & 2;
    bool copy = copy_all;       // every string copied through the emit masks (byte-parallel path)
    bool masks = true;          // the emit masks are computed (both modes); false after a surrogate-walk overflow
    bool force_copy = false;    // every string copied, but through the per-string walks (after an overflow, parse_api.hip)
    size_t off, len;
    trim_space(msg0, len0, &off, &len);
    *msg_off = off;
    *msg_len = len;
    *tape_out = nullptr;
    *strings_out = nullptr;
    *tape_len = *strings_len = 0;
    const u8 *msg = msg0 + off;
    // stage 1
    std::vector<u32> pos(len + 64);
    size_t n = 0;
    u32 err = 0, inq = 0;
    const size_t units = (len + 4095) / 4096 + 1, chunks = units * 64;
    std::vector<u64> v_qm(chunks, 0), v_q(chunks, 0), v_st(chunks, 0), v_em(chunks, 0), v_um(chunks, 0);
    std::vector<u8> v_h(units, 0);
    std::vector<u64> v_slow(units, 0);
    std::vector<u32> v_flags(chunks, 0);
    std::vector<uint16_t> v_pre(chunks, 0);
    std::vector<ChunkRec> v_rec(chunks);
    std::vector<u32> v_ucnt(units, 0), v_ucount(units, 0);
    g_qm = v_qm.data();
    g_q = v_q.data();
    g_st = v_st.data();
    g_slow = v_slow.data();
    sj_selftest_stage1(msg, len, ndjson, pos.data(), pos.size(), &n, &err, &inq);
    g_qm = g_q = g_st = g_slow = nullptr;
    if (len == 0 || err || inq || n == 0 || !(msg[len - 1] == '}' || msg[len - 1] == ']')) return 1;
    const MsgView mv{msg, len};
    // stage 2: the kernels of stage2.hip as loops, in launch order
    static constexpr KindLut KLUT = make_kind_lut();
    static constexpr ElementLut ELUT = make_element_lut();
    u32 bad = 0;
    // k_str_masks / k_str_scan (every string copied)
    // the device form of the in-string mask: relative to the start of its 4 KiB unit, with the unit's state in unit_h (the
    // replay above keeps the absolute mask in v_qm)
    std::vector<u64> v_qrel(chunks, 0);
    for (size_t u = 0; u < units; u++) {
        const u8 h = u ? (u8)(v_qm[u * 64 - 1] >> 63) : (u8)0;
        v_h[u] = h;
        for (size_t c = u * 64; c < (u + 1) * 64; c++) v_qrel[c] = h ? ~v_qm[c] : v_qm[c];
    }
    const StrView sv{msg, 0, len, v_qrel.data(), v_st.data(), v_h.data(), v_slow.data()};
    const size_t used_units = (len + 4095) / 4096;
    // the device does not store the unescaped quotes: they are where the unit-relative in-string mask changes (StrView::quotes)
    for (size_t c = 0; c < used_units * 64; c++)
        if (sv.quotes(c) != v_q[c]) return 97;
    u64 masks_total = 0;
    if (masks) {
        for (size_t c = 0; c < used_units * 64; c++)
        {
            bool overflow;
            if (!str_chunk_masks_fast(sv, c, &v_em[c], &v_flags[c], &overflow)) bad = 1;
            if (overflow) force_copy = true;
            {  // the fast path must agree with the general routine wherever it decides on its own
                u64 em2, um2;
                bool esc2, ov2;
                const bool ok2 = str_chunk_masks(sv, c, &em2, &um2, &esc2, &ov2);
                if (!ov2 && !str_chunk_needs_general(sv, c) && (!ok2 || em2 != v_em[c])) return 96;
                if ((v_flags[c] & CHUNK_GENERAL) && esc2 != str_chunk_has_escapes(sv, c)) return 96;
            }
        }
        if (force_copy) {  // S2_ERR_SERIAL_STRINGS: nothing of the mask pass is a verdict
            copy = false;
            masks = false;
            force_copy = copy_all;  // (selective copy stays selective: the walks decide per string)
            bad = 0;
        }
    }
    if (masks) {
        // The escape-by-escape form of the general routine (what k_measure runs: stage2.hip GenUnit, sj_strings.h
        // gen_item_masks) must give the verdict, the emit masks and the flags of the per-chunk form above.
        u32 bad2 = 0;
        for (size_t u = 0; u < used_units; u++) {
            u64 em2[64], e_[64], pe_[64];
            bool gen[64], own[64], any = false;
            for (u32 l = 0; l < 64; l++) {
                const u64 c = u * 64 + l, smc = sv.sm(c);
                e_[l] = sv.esc(c) & smc;
                pe_[l] = c ? (sv.esc(c - 1) & sv.sm(c - 1)) >> 60 : 0ull;
                em2[l] = smc & ~v_st[c];
                own[l] = sv.nonsimple(c) && e_[l] != 0;
                gen[l] = own[l] || (c > 0 && sv.nonsimple(c - 1));
                any |= gen[l];
            }
            bool b2 = false, o2 = false;
            if (any) {
                std::vector<u32> items;
                if (gen[0])
                    for (u64 f = pe_[0]; f != 0; f &= f - 1) items.push_back(GEN_FOREIGN | (u32)ctz64(f));
                for (u32 l = 0; l < 64; l++)
                    if (own[l])
                        for (u64 r = e_[l]; r != 0; r &= r - 1) items.push_back(l * 64 + (u32)ctz64(r));
                for (size_t j = items.size(); j-- > 0;)  // (any order: here the reverse of the listing)
                    gen_item_masks(sv, (u64)u * 4096, items[j], &b2, &o2, [&](u32 pos) { em2[pos >> 6] &= ~(1ull << (pos & 63)); });
            }
            if (o2) return 93;  // (an overflow would have set force_copy above)
            bad2 |= b2 ? 1u : 0u;
            if (bad || b2) continue;  // a rejected document: the masks are not compared
            for (u32 l = 0; l < 64; l++) {
                const u64 c = u * 64 + l;
                const u32 f2 = gen[l] ? ((e_[l] != 0 || pe_[l] != 0) ? (CHUNK_SLOW | CHUNK_GENERAL) : 0u) : (e_[l] != 0 ? CHUNK_SLOW : 0u);
                if (em2[l] != v_em[c] || f2 != v_flags[c]) return 93;
            }
        }
        if ((bad2 != 0) != (bad != 0)) return 93;
    }
    if (masks) {
        for (size_t u = 0; u < used_units; u++) {
            u32 run = 0;
            for (size_t c = u * 64; c < u * 64 + 64; c++) {
                v_pre[c] = (uint16_t)run;
                v_rec[c] = ChunkRec{v_em[c], run | v_flags[c], 0u};
                run += (u32)popc64(v_em[c]);
            }
            v_ucnt[u] = (u32)masks_total;
            v_ucount[u] = run;  // (what the measuring phase sees: the unit scan has not run yet)
            for (size_t c = u * 64; c < u * 64 + 64; c++) v_rec[c].abs = (u32)masks_total + v_pre[c];  // k_str_emit
            masks_total += run;
        }
    }
    // Every string copied, second half of round 5: stage 1 keeps no records.  Its phase A counts the emitted bytes and the
    // opening quotes of every unit under both hypotheses about the state at the start of the unit (stage1.hip phase_a), the
    // flatten picks the pair that applies; k_str_emit derives emit mask, escaped characters and opening quotes of a chunk from
    // the three masks (sj_strings.h chunk_fast) and leaves the Strings.B offset of the k-th string of the message in soff[k]
    // (+ the length of Strings.B behind the last one).  The replay's masks are absolute (state 0 at every unit start), so
    // the other hypothesis is exercised on the complemented mask: both must give what the record path gives.
    std::vector<u32> v_soff;
    if (masks && !bad) {
        for (size_t u = 0; u < used_units; u++) {
            const bool slow = v_slow[u] != 0 || (u > 0 && (v_slow[u - 1] >> 63) != 0);  // k_measure's general routine has the unit
            u32 nstr = 0;
            for (u32 hyp = 0; hyp < 2; hyp++) {
                u32 cA = 0, cAB = 0, oA = 0, oAB = 0;
                for (size_t c = u * 64; c < u * 64 + 64; c++) {
                    const u64 qmr = hyp ? ~v_qm[c] : v_qm[c];  // the mask relative to a unit that starts inside a string
                    const u64 nqst = ~v_q[c] & ~v_st[c];
                    cA += (u32)popc64(qmr & nqst);
                    cAB += (u32)popc64(nqst);
                    oA += (u32)popc64(qmr & v_q[c]);
                    oAB += (u32)popc64(v_q[c]);
                    const ChunkFast f = chunk_fast(qmr, v_q[c], v_st[c], c ? v_st[c - 1] : 0ull, hyp);
                    if (!slow && (f.em != v_em[c] || (f.esc != 0 ? CHUNK_SLOW : 0u) != v_flags[c])) return 87;
                    if (f.oq != (v_q[c] & v_qm[c])) return 87;
                    if (slow && (((v_st[c] << 1) | (c ? v_st[c - 1] >> 63 : 0ull)) & v_em[c]) != (sv.esc(c) & v_em[c])) return 87;
                    if (hyp == 0)
                        for (u64 r = f.oq; r != 0; r &= r - 1) {
                            v_soff.push_back(v_ucnt[u] + v_pre[c] + (u32)popc64(v_em[c] & ((1ull << ctz64(r)) - 1ull)));
                            nstr++;
                        }
                }
                const u32 ucnt = hyp ? cAB - cA : cA, ustr = hyp ? oAB - oA : oA;
                if (!slow && ucnt != v_ucount[u]) return 87;
                if (ustr != nstr) return 87;
            }
        }
        v_soff.push_back((u32)masks_total);
    }
    // k_s2_reduce: kinds, string lengths (selective copy), elements
    std::vector<u8> kind(n);
    std::vector<u32> dlen(n, 0), copied(n, 0);
    std::vector<u8> needcopy(n, 0), strbad(n, 0);
    for (size_t i = 0; i < n; i++) {
        u8 k = KLUT.v[msg[pos[i]]];
        if (k == K_NL && !ndjson) k = K_BAD;
        kind[i] = k;
        if (k != token_kind(msg[pos[i]], ndjson)) return 99;  // the table must agree with the switch
        if (!copy && k == K_STRING) {
            u32 sl, dl;
            if (!string_walk(mv, pos[i], nullptr, &sl, &dl)) {
                bad = 1;
                strbad[i] = 1;
            } else {
                dlen[i] = dl;
                needcopy[i] = force_copy || sl != dl;
                copied[i] = needcopy[i] ? dl : 0u;
            }
        }
    }
    auto kind_at = [&](size_t i, long d) -> u8 {
        const long j = (long)i + d;
        return (j < 0 || j >= (long)n) ? (u8)K_BAD : kind[(size_t)j];
    };
    // The scan element of 64 tokens at once from the bit planes of their kinds (sj_planes.h) against the per-token
    // statement: every group's aggregate, every token's tape offset and bracket ordinal inside its group, and the
    // allowed contexts of the gap of every bracket
    std::vector<Agg> elem(n);  // (token_element once per token: the plane check and the scan below share it)
    for (size_t i = 0; i < n; i++) elem[i] = token_element((u32)i, (u32)n, kind[i], kind_at(i, -1), kind_at(i, -2), kind_at(i, 1), 0u);
    {
        Agg prefix = agg_identity();
        for (size_t g0 = 0; g0 < n; g0 += 64) {
            const u32 cnt = (u32)(n - g0 < 64 ? n - g0 : 64);
            const KindPlanes kp = kind_planes(kind.data() + g0, cnt);
            {  // the multiplication form of the transposition
                u32 w16[16] = {0};
                memcpy(w16, kind.data() + g0, cnt);
                const KindPlanes k2 = kind_planes_words(w16);
                if (k2.b0 != kp.b0 || k2.b1 != kp.b1 || k2.b2 != kp.b2 || k2.b3 != kp.b3) return 91;
            }
            const u64 valid = cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull);
            const GroupCarry cy{g0 >= 2 ? kind[g0 - 2] : (u8)K_NONE, g0 >= 1 ? kind[g0 - 1] : (u8)K_NONE,
                                g0 + 64 < n ? kind[g0 + 64] : (u8)K_NL};
            const GroupMasks gm = group_masks(kp, valid, cy, g0 == 0);
            Agg fold = agg_identity();
            for (u32 j = 0; j < cnt; j++) {
                const size_t i = g0 + j;
                const Agg &e = elem[i];
                if (group_words_before(gm, j) != fold.w || group_brackets_before(gm, j) != fold.bc) return 91;
                if (is_bracket(kind[i]) && group_gap_set(gm, j, prefix.am) != gap_mask(agg_combine(prefix, fold), e)) return 91;
                fold = agg_combine(fold, e);
            }
            const Agg ga = group_aggregate(gm);
            if (ga.d != fold.d || ga.w != fold.w || ga.nb != fold.nb || ga.bc != fold.bc || ga.am != fold.am) return 91;
            prefix = agg_combine(prefix, fold);
        }
    }
    // The same element for SIXTEEN tokens per lane from 19-bit windows of the kind planes (sj_tok16.h: what k_measure and
    // k_s2_emit run since round 5): the transposition, every lane's aggregate, every writing token's tape offset, every
    // bracket's kind and gap set, every atom's kind
    {
        Agg prefix = agg_identity();
        for (size_t g0 = 0; g0 < n; g0 += 16) {
            const u32 cnt = (u32)(n - g0 < 16 ? n - g0 : 16);
            u8 kb[16];
            for (u32 j = 0; j < 16; j++) kb[j] = j < cnt ? kind[g0 + j] : (u8)K_NL;
            u32 w4[4];
            memcpy(w4, kb, 16);
            const Planes16 pl = planes16(w4[0], w4[1], w4[2], w4[3]);
            for (u32 pb = 0; pb < 4; pb++) {
                u32 want = 0;
                for (u32 j = 0; j < 16; j++) want |= (u32)((kb[j] >> pb) & 1u) << j;
                const u32 got = pb & 1 ? ((pb & 2 ? pl.p23 : pl.p01) >> 16) : ((pb & 2 ? pl.p23 : pl.p01) & 0xffffu);
                if (got != want) return 89;
            }
            const u32 prev2 = (u32)(g0 >= 2 ? kind[g0 - 2] : (u8)K_NONE) | ((u32)(g0 >= 1 ? kind[g0 - 1] : (u8)K_NONE) << 8);
            const u32 next1 = g0 + 16 < n ? kind[g0 + 16] : (u32)K_NL;
            const u32 valid = cnt == 16 ? 0xffffu : ((1u << cnt) - 1u);
            const Lane16 lm = lane16_masks(pl, prev2, next1, valid, g0 == 0);
            for (u32 h = 0; h < 2; h++) {  // the eight-token form of the lane on either half: the same masks
                const size_t h0 = g0 + 8 * h;
                if (h0 >= n) break;
                const u32 c8 = (u32)(n - h0 < 8 ? n - h0 : 8);
                const Planes16 p8 = planes16(w4[2 * h], w4[2 * h + 1], 0u, 0u);
                const u32 pv = (u32)(h0 >= 2 ? kind[h0 - 2] : (u8)K_NONE) | ((u32)(h0 >= 1 ? kind[h0 - 1] : (u8)K_NONE) << 8);
                const Lane16 l8 = lane16_masks<8>(p8, pv, h0 + 8 < n ? kind[h0 + 8] : (u32)K_NL, (1u << c8) - 1u, h0 == 0);
                const u32 sh = 8 * h, mk = 0xffu;
                if (l8.w1 != ((lm.w1 >> sh) & mk) || l8.w2 != ((lm.w2 >> sh) & mk) || l8.br != ((lm.br >> sh) & mk) || l8.open != ((lm.open >> sh) & mk) ||
                    l8.nlr != ((lm.nlr >> sh) & mk) || (l8.a_root & mk) != ((lm.a_root >> sh) & mk) || (l8.a_obj & mk) != ((lm.a_obj >> sh) & mk) ||
                    (l8.a_arr & mk) != ((lm.a_arr >> sh) & mk) || l8.gap_start != ((lm.gap_start >> sh) & mk) || l8.str != ((lm.str >> sh) & mk) ||
                    l8.keystr != ((lm.keystr >> sh) & mk) || l8.num != ((lm.num >> sh) & mk) || l8.atom != ((lm.atom >> sh) & mk) ||
                    (l8.b0 & mk) != ((lm.b0 >> sh) & mk) || (l8.b1 & mk) != ((lm.b1 >> sh) & mk) || (l8.a_root >> 8) != 0xffu)
                    return 88;
            }
            Agg fold = agg_identity();
            bool illegal = false;
            for (u32 j = 0; j < cnt; j++) {
                const size_t i = g0 + j;
                const Agg &e = elem[i];
                const u8 k = kind[i];
                if (e.w && lane16_words_before(lm, j) != fold.w) return 89;
                if (((lm.br >> j) & 1u) != (is_bracket(k) ? 1u : 0u) || ((lm.str >> j) & 1u) != (k == K_STRING ? 1u : 0u) ||
                    ((lm.num >> j) & 1u) != (k == K_NUM ? 1u : 0u) || ((lm.atom >> j) & 1u) != ((k == K_TRUE || k == K_FALSE || k == K_NULL) ? 1u : 0u) ||
                    ((lm.nlr >> j) & 1u) != e.nb)
                    return 89;
                if (((lm.keystr >> j) & 1u) != ((k == K_STRING && kind_at(i, 1) == K_COLON) ? 1u : 0u)) return 89;
                if (is_bracket(k)) {
                    if (lane16_bracket_kind(lm, j) != k) return 89;
                    if (lane16_gap_set(lm, j, prefix.am) != gap_mask(agg_combine(prefix, fold), e)) return 89;
                    const u32 e32 = tbr_pack(fold.w, fold.d + e.d, k, 5u);
                    if (tbr_off(e32) != fold.w || tbr_depth(e32) != fold.d + e.d || tbr_kind(e32) != k || tbr_gap(e32) != 5u) return 89;
                }
                if (((lm.atom >> j) & 1u) && lane16_atom_kind(lm, j) != k) return 89;
                illegal |= am_value(e.am) == 0;
                fold = agg_combine(fold, e);
            }
            const Agg ga = pagg_unpack(lane16_pagg(lm));
            if (ga.d != fold.d || ga.w != fold.w || ga.nb != fold.nb || ga.bc != fold.bc || ga.am != fold.am) return 89;
            if (lane16_illegal(lm) != illegal) return 89;
            u32 ns = 0, nd = 0;
            for (u32 j = 0; j < cnt; j++) {
                ns += kind[g0 + j] == K_STRING;
                nd += kind[g0 + j] >= K_NUM && kind[g0 + j] <= K_NULL;
            }
            if (lane16_counts(lm) != (ns | (nd << 13))) return 89;
            prefix = agg_combine(prefix, fold);
        }
    }
    // k_s2_scan_tiles + k_s2_emit: the scan as a sequential sum
    std::vector<i32> br_depth;
    std::vector<u32> br_off, nl_off;
    std::vector<u8> br_info;
    std::vector<u32> toff(n), soff(n, 0);
    Agg run = agg_identity();
    u64 words = 0, sbytes = 0;
    bool bad_ctx = false;  // a token that is legal in no context at all
    for (size_t i = 0; i < n; i++) {
        Agg e = elem[i];
        e.s = copied[i];
        {  // the packed table form the kernels use must be the same element
            const PAgg pe = token_pelement(ELUT.v, kind_window(kind.data(), (u32)i, (u32)n), copied[i]), want = pagg_pack(e);
            if (pe.x != want.x || pe.y != want.y || pe.z != want.z || pe.s != want.s) return 97;
        }
        toff[i] = run.w + 1u;
        soff[i] = run.s;
        if (am_value(e.am) == 0) {
            bad = 1;
            bad_ctx = true;
        }
        if (is_bracket(kind[i])) {
            br_depth.push_back(run.d + e.d);
            br_off.push_back(toff[i]);
            br_info.push_back((u8)(kind[i] | (gap_mask(run, e) << 4)));
        }
        if (e.nb) nl_off.push_back(toff[i]);
        words += e.w;
        sbytes += e.s;
        run = agg_combine(run, e);
    }
    // WithCopyStrings(false), byte-parallel (sj_strings.h chunk_sel): the bytes of the strings that hold an escape starter,
    // chunk by chunk; F / G across chunks as the kernels compute them -- a scan inside the unit, the state at the unit's
    // ends from the quotes and starters of the neighbouring units (sel_unit_in / sel_unit_out below) -- against one
    // sequential pass over the whole message; then per string: copied or not, Strings.B offset, both lengths, against the
    // per-string walks above.
    std::vector<u64> v_sel(used_units * 64, 0);
    if (masks && !copy && !bad) {
        const size_t nch = used_units * 64;
        std::vector<ChunkSel> cs(nch);
        for (size_t c = 0; c < nch; c++) cs[c] = chunk_sel(v_qm[c], v_q[c], v_st[c], 0u);
        // ground truth of the two scans
        std::vector<u32> F(nch + 1, 0), G(nch + 1, 0);  // F[c]: state at the start of chunk c; G[c]: state at the END of chunk c - 1
        for (size_t c = 0; c < nch; c++) F[c + 1] = sel_apply(cs[c].fwd, F[c]);
        for (size_t c = nch; c-- > 0;) G[c] = sel_apply(cs[c].bwd, G[c + 1]);
        // the kernels' way: the state at a unit's start from the units in front (the last quote of the nearest unit that
        // holds one, the starters behind it), at its end from the units behind (the first quote, the starters in front of it)
        for (size_t u = 0; u < used_units; u++) {
            u32 fin = 0, gout = 0;
            const bool open_in = (cs[u * 64].in & ~cs[u * 64].oq & 1ull) != 0;
            if (open_in)
                for (size_t v = u; v-- > 0;) {
                    bool quote = false;
                    for (size_t c = v * 64 + 64; c-- > v * 64;) {
                        if (v_q[c] != 0) {
                            const int hb = 63 - clz64(v_q[c]);
                            if (hb < 63 && (v_st[c] >> (hb + 1)) != 0) fin = 1;
                            quote = true;
                            break;
                        }
                        if (v_st[c] != 0) fin = 1;
                    }
                    if (quote) break;
                }
            const bool open_out = u + 1 < used_units && (cs[u * 64 + 63].in >> 63) != 0;
            if (open_out)
                for (size_t v = u + 1; v < used_units; v++) {
                    bool quote = false;
                    for (size_t c = v * 64; c < v * 64 + 64; c++) {
                        if (v_q[c] != 0) {
                            const int lb = ctz64(v_q[c]);
                            if (lb > 0 && (v_st[c] & ((1ull << lb) - 1ull)) != 0) gout = 1;
                            quote = true;
                            break;
                        }
                        if (v_st[c] != 0) gout = 1;
                    }
                    if (quote) break;
                }
            if ((open_in ? F[u * 64] : 0u) != fin) return 85;
            if ((open_out ? G[u * 64 + 64] : 0u) != gout) return 85;
            // inside the unit: the composed functions of the chunks in front / behind applied to the unit's state
            u32 fc = 1u;  // identity
            for (u32 l = 0; l < 64; l++) {
                if (sel_apply(fc, fin) != (l == 0 && !open_in ? 0u : F[u * 64 + l]) && (cs[u * 64 + l].hr != 0)) return 85;
                fc = sel_then(fc, cs[u * 64 + l].fwd);
            }
            u32 gc = 1u;
            for (u32 l = 64; l-- > 0;) {
                if (sel_apply(gc, gout) != G[u * 64 + l + 1] && (cs[u * 64 + l].tr != 0)) return 85;
                gc = sel_then(gc, cs[u * 64 + l].bwd);
            }
        }
        for (size_t c = 0; c < nch; c++) v_sel[c] = chunk_sel_mask(cs[c], F[c], G[c + 1]);
        // per string: k-th opening quote, k-th closing quote
        std::vector<u32> so2, cq2;
        u64 runb = 0;
        for (size_t c = 0; c < nch; c++) {
            const u64 emc = v_em[c] & v_sel[c];
            for (u64 r = cs[c].oq; r != 0; r &= r - 1) so2.push_back((u32)(runb + popc64(emc & ((1ull << ctz64(r)) - 1ull))));
            for (u64 r = cs[c].cq; r != 0; r &= r - 1) cq2.push_back((u32)(c * 64 + ctz64(r)));
            runb += (u64)popc64(emc);
        }
        so2.push_back((u32)runb);
        size_t k = 0;
        for (size_t i = 0; i < n; i++) {
            if (kind[i] != K_STRING || strbad[i]) {
                if (kind[i] == K_STRING) k++;
                continue;
            }
            if (k + 1 >= so2.size() || k >= cq2.size()) return 84;
            const bool ch = so2[k + 1] != so2[k];
            if (ch != (needcopy[i] != 0)) return 84;
            if (ch ? (so2[k] != soff[i] || so2[k + 1] - so2[k] != dlen[i]) : (cq2[k] - pos[i] - 1 != dlen[i])) return 84;
            k++;
        }
        if (runb != sbytes) return 84;
    }
    // The same pass organised the way a kernel on bit planes would run it (sj_planes.h): a tile of 4096 tokens per wave,
    // 64 tokens per lane; a lane knows the masks of its group, the exclusive prefix over the lanes in front of it (and
    // the tiles in front of the tile) and derives from those alone the tape offset of every token, the compact index,
    // depth, tape offset and gap set of every bracket and the ordinal of every record-separating newline.
    {
        Agg tile_prefix = agg_identity();
        bool bad_planes = false;
        for (size_t t0 = 0; t0 < n; t0 += 4096) {
            Agg ex = tile_prefix;
            for (u32 l = 0; l < 64; l++) {
                const size_t g0 = t0 + 64 * (size_t)l;
                if (g0 >= n) break;
                const u32 cnt = (u32)(n - g0 < 64 ? n - g0 : 64);
                const u64 valid = cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull);
                const GroupCarry cy{g0 >= 2 ? kind[g0 - 2] : (u8)K_NONE, g0 >= 1 ? kind[g0 - 1] : (u8)K_NONE,
                                    g0 + 64 < n ? kind[g0 + 64] : (u8)K_NL};
                const GroupMasks gm = group_masks(kind_planes(kind.data() + g0, cnt), valid, cy, g0 == 0);
                if ((~(gm.a_root | gm.a_obj | gm.a_arr) & valid) != 0) bad_planes = true;
                for (u64 r = gm.w1 | gm.br | gm.nlr; r != 0; r &= r - 1) {  // the tokens that write something
                    const u32 j = (u32)ctz64(r);
                    const size_t i = g0 + j;
                    const u32 o = ex.w + 1u + group_words_before(gm, j);
                    if (o != toff[i]) return 90;
                    if ((gm.br >> j) & 1u) {
                        const u32 c = ex.bc + group_brackets_before(gm, j);
                        const u64 upto = j == 63 ? ~0ull : ((1ull << (j + 1)) - 1ull);
                        const i32 d = ex.d + 2 * (i32)popc64(gm.open & upto) - (i32)popc64(gm.br & upto);
                        if (c >= br_off.size() || br_off[c] != o || br_depth[c] != d ||
                            br_info[c] != (u8)(kind[i] | (group_gap_set(gm, j, ex.am) << 4)))
                            return 90;
                    }
                    if ((gm.nlr >> j) & 1u) {
                        const u32 r_ord = ex.nb + (u32)popc64(gm.nlr & (j ? (~0ull >> (64 - j)) : 0ull));
                        if (r_ord >= nl_off.size() || nl_off[r_ord] != o) return 90;
                    }
                }
                ex = agg_combine(ex, group_aggregate(gm));
            }
            tile_prefix = ex;
        }
        if (bad_planes != bad_ctx) return 90;
        if (tile_prefix.d != run.d || tile_prefix.w != run.w || tile_prefix.nb != run.nb || tile_prefix.bc != run.bc || tile_prefix.am != run.am)
            return 90;
    }
    if (copy) sbytes = masks_total;
    const u32 tlen = (u32)words + 2;  // + opening and closing root
    if (run.d != 0) bad = 1;
    const u32 tail_mask = is_bracket(kind[n - 1]) ? AM_ALL : am_value(run.am);
    if (!context_allowed(tail_mask, CTX_ROOT)) bad = 1;  // k_scans: tokens behind the last bracket lie at depth 0
    u64 *tape = (u64 *)calloc(tlen + 2, sizeof(u64));
    u8 *strs = (u8 *)malloc(sbytes + 64);
    size_t str_ord = 0;
    for (size_t i = 0; i < n; i++) {
        const u8 k = kind[i];
        if (k == K_TRUE || k == K_FALSE || k == K_NULL) {
            tape[toff[i]] = atom_word(k);
            if (!atom_valid(mv, pos[i], k)) bad = 1;
        } else if (k == K_STRING && copy) {
            const u64 a0 = (u64)pos[i] + 1, a1 = i + 1 < n ? pos[i + 1] : len;
            const u64 so = emitted_before(v_ucnt.data(), v_rec.data(), a0);
            const u64 se = emitted_before(v_ucnt.data(), v_rec.data(), a1);
            if (so != emitted_before_abs(v_rec.data(), a0) || se != emitted_before_abs(v_rec.data(), a1)) return 95;
            if (!bad) {  // k_s2_emit_planes: the string's number in the message -> soff[]
                if (str_ord + 1 >= v_soff.size() || v_soff[str_ord] != so || v_soff[str_ord + 1] - v_soff[str_ord] != se - so) return 86;
            }
            str_ord++;
            tape[toff[i]] = string_word(true, strings_base + so, 0);
            tape[toff[i] + 1] = se - so;
        } else if (k == K_STRING && !strbad[i]) {
            tape[toff[i]] = string_word(needcopy[i], strings_base + soff[i], msg_base + pos[i] + 1);
            tape[toff[i] + 1] = dlen[i];
            if (needcopy[i]) {  // k_emit_strings
                u32 sl, dl;
                string_walk(mv, pos[i], strs + soff[i], &sl, &dl);
            }
        } else if (k == K_NUM) {
            u64 tag, val;
            int ub;
            const bool okn = sj_selftest_parse_number(msg + pos[i], len - pos[i], &tag, &val, &ub) != 0;
            {  // the integer fast path of k_s2_emit decides nothing the general routine decides differently
                u64 fv = 0;
                if (parse_int_fast(load8_guarded(mv, pos[i]), load8_guarded(mv, (u64)pos[i] + 8), load8_guarded(mv, (u64)pos[i] + 16), &fv) &&
                    (!okn || tag != ((u64)'l' << 56) || val != fv))
                    return 98;
            }
            if (!okn) bad = 1;
            else {
                tape[toff[i]] = tag;
                tape[toff[i] + 1] = val;
            }
        }
    }
    // k_min_level + k_br_match: min tree over the compact bracket view; partners, contexts, gap check, root words
    MinTree mt;
    std::vector<std::vector<i32>> levels;
    mt.lev[0] = br_depth.data();
    mt.size[0] = br_depth.size();
    mt.nlev = 1;
    while (mt.size[mt.nlev - 1] > 64) {
        const u64 ps = mt.size[mt.nlev - 1], ns = (ps + 63) / 64;
        levels.emplace_back(ns);
        for (u64 g = 0; g < ns; g++) {
            i32 mn = 0x7fffffff;
            for (u64 k = g * 64; k < ps && k < g * 64 + 64; k++) mn = mt.lev[mt.nlev - 1][k] < mn ? mt.lev[mt.nlev - 1][k] : mn;
            levels.back()[g] = mn;
        }
        mt.lev[mt.nlev] = levels.back().data();
        mt.size[mt.nlev] = ns;
        mt.nlev++;
    }
    const u32 n_br = (u32)br_off.size();
    for (u32 c = 0; c < n_br; c++)  // k_br_match: container of every gap, pair words, root words
        if (!bracket_resolve(mt, br_off.data(), br_info.data(), c, tape_base, tape)) bad = 1;
    if (n_br == 0) bad = 1;  // unreachable: token 0 must be an open bracket
    std::vector<u8> full;  // selective copy: k_str_emit's compaction of ALL strings (the device's scratch buffer)
    if (masks && !copy) full.resize(masks_total + 64);
    if (masks && !bad) {  // k_str_emit: patch the escapes of a chunk, then keep the bytes its emit mask names
        u8 *const sink = copy ? strs : full.data();
        std::vector<u8> ubuf(4096), selbuf;
        for (size_t c = 0; c < used_units * 64; c++) {
            if ((c & 63) == 0) {
                // the escape-by-escape form of the general patch (k_str_emit: gen_item_patch) for the whole unit, to be
                // compared with the per-chunk form below
                const size_t u = c >> 6;
                for (u32 q = 0; q < 4096; q++) ubuf[q] = sv.at(u * 4096 + q);
                std::vector<u32> items;
                bool any = false;
                for (u32 l = 0; l < 64; l++) any |= v_em[c + l] != 0 && (v_flags[c + l] & CHUNK_GENERAL) != 0;
                if (any) {
                    if ((v_flags[c] & CHUNK_GENERAL) && c > 0)
                        for (u64 f = ((sv.esc(c - 1) & v_em[c - 1]) >> 60) & 0xfull; f != 0; f &= f - 1)
                            items.push_back(GEN_FOREIGN | (u32)ctz64(f));
                    for (u32 l = 0; l < 64; l++)
                        if (v_em[c + l] != 0 && (v_flags[c + l] & CHUNK_GENERAL))
                            for (u64 r = sv.esc(c + l) & v_em[c + l]; r != 0; r &= r - 1) items.push_back(l * 64 + (u32)ctz64(r));
                    for (size_t j = items.size(); j-- > 0;)
                        gen_item_patch(sv, (u64)u * 4096, items[j], [&](u32 pos, u8 v) { ubuf[pos] = v; });
                }
            }
            if (v_em[c] == 0) continue;
            u8 chunk[64];
            for (u32 q = 0; q < 64; q++) chunk[q] = sv.at(c * 64 + q);
            if (v_flags[c] & CHUNK_GENERAL) {
                str_chunk_patch(sv, c, [&](u32 q, u8 v) { chunk[q] = v; });
                for (u64 r = v_em[c]; r != 0; r &= r - 1)  // every emitted byte: both forms agree
                    if (chunk[ctz64(r)] != ubuf[(c & 63) * 64 + ctz64(r)]) return 92;
            } else if (v_flags[c] & CHUNK_SLOW) {
                str_chunk_patch_simple(sv, c, v_em[c], [&](u32 q) { return chunk[q]; }, [&](u32 q, u8 v) { chunk[q] = v; });
            }
            u8 *dst = sink + v_ucnt[c >> 6] + v_pre[c];
            for (u64 r = v_em[c]; r != 0; r &= r - 1) *dst++ = chunk[ctz64(r)];
            if (!copy)  // the byte-parallel selective copy keeps only the bytes of strings that hold an escape
                for (u64 r = v_em[c] & v_sel[c]; r != 0; r &= r - 1) selbuf.push_back(chunk[ctz64(r)]);
        }
        if (!copy && (selbuf.size() != sbytes || (sbytes && memcmp(selbuf.data(), strs, sbytes) != 0))) return 83;
        if (!copy)  // k_emit_strings: the strings that changed are taken from the compaction -- the walk's bytes
            for (size_t i = 0; i < n; i++)
                if (kind[i] == K_STRING && needcopy[i] && !strbad[i] &&
                    memcmp(full.data() + emitted_before(v_ucnt.data(), v_rec.data(), (u64)pos[i] + 1), strs + soff[i], dlen[i]) != 0)
                    return 94;
    }
    if (bad) {
        free(tape);
        free(strs);
        return 2;
    }
    *tape_out = tape;
    *tape_len = tlen;
    *strings_out = strs;
    *strings_len = sbytes;
    return 0;
}
extern "C" void sj_selftest_free(void *p) { free(p); }

// number formatting of the tape -> JSON text path (sj_ftoa.h): appendFloat / AppendInt / AppendUint
// the text of a float64; also checked here: exactly the bytes of the text are written (k_ms_tile formats straight into the
// tile's window, where the next entry's text follows) and the length-only form agrees -- 0x80000000 is set otherwise
extern "C" unsigned sj_selftest_format_float(uint64_t bits, uint8_t *out32) {
    uint8_t tmp[40];
    memset(tmp, 0xee, sizeof tmp);
    const unsigned n = format_float(bits, tmp);
    unsigned bad = n > 25 || float_text_len(bits) != n ? 0x80000000u : 0u;
    for (unsigned k = n; k < sizeof tmp && n <= 25; k++)
        if (tmp[k] != 0xee) bad = 0x80000000u;
    memcpy(out32, tmp, 32);
    return n | bad;
}
extern "C" unsigned sj_selftest_format_int(uint64_t raw, int is_unsigned, uint8_t *out24) {
    return is_unsigned ? format_uint(raw, out24) : format_int(raw, out24);
}

// batch_api.hip: the SWAR newline -> carriage return of the packing kernel against the byte loop, every byte value in
// every lane over a background of newlines and of other bytes
extern "C" int sj_selftest_newlines_to_cr(void) {
    for (u32 bg = 0; bg < 4; bg++) {
        const u32 back = bg == 0 ? 0x0a0a0a0au : bg == 1 ? 0x00000000u : bg == 2 ? 0xffffffffu : 0x0b090d8au;
        for (u32 lane = 0; lane < 4; lane++)
            for (u32 b = 0; b < 256; b++) {
                const u32 w = (back & ~(0xffu << (8 * lane))) | (b << (8 * lane));
                u32 want = 0;
                for (u32 k = 0; k < 4; k++) {
                    const u32 x = (w >> (8 * k)) & 0xffu;
                    want |= (x == 0x0au ? 0x0du : x) << (8 * k);
                }
                if (newlines_to_cr(w) != want) return 1;
            }
    }
    return 0;
}
