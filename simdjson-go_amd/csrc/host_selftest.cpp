// host_selftest.cpp -- CPU replay of the per-chunk device arithmetic in sj_chunk.h.
//
// NOT part of the product path: built only by the test-suite (g++, no HIP) so that the
// lane-local stage-1 math and the "peek" carry rules of stage1.hip can be checked against
// the oracle without a GPU.  The chunk loop below plays the role of the lanes; every
// cross-chunk input is derived exactly the way the kernel derives it (memory peeks and a
// running parity / count), never from the reference-style carried scalars.
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
        u32 carry_out;
        const u64 odd_ends = odd_backslash_ends(c.bs, carry_in, carry_out);
        const u64 quote_bits = c.quote & ~odd_ends;
        u64 quote_mask = prefix_xor(quote_bits);
        if (par) quote_mask = ~quote_mask;
        par ^= (u32)popc64(quote_bits) & 1u;
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
    out6[0] = c.bs;
    out6[1] = c.quote;
    out6[2] = c.structs;
    out6[3] = c.ws;
    out6[4] = c.ctrl;
    out6[5] = c.nl;
}
extern "C" uint64_t sj_selftest_odd_backslash(uint64_t bs, uint32_t carry_in, uint32_t *carry_out) {
    return odd_backslash_ends(bs, carry_in, *carry_out);
}
extern "C" uint64_t sj_selftest_prefix_xor(uint64_t x) { return prefix_xor(x); }
extern "C" uint64_t sj_selftest_finalize(uint64_t st, uint64_t ws, uint64_t qm, uint64_t qb, uint32_t pp) {
    return finalize(st, ws, qm, qb, pp);
}
