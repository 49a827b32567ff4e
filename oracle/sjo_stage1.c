/*
 * sjo_stage1.c -- ORACLE (test infrastructure only, see sjo.h).
 * Scalar restatement of the reference's stage 1 ("find structural bits").
 * Each function cites the reference routine it follows.
 */
#include "sjo.h"

#include <stdlib.h>
#include <string.h>

/* byte-compare -> 64-bit mask, bit j <-> in[j] (VPCMPEQB+VPMOVMSKB pairs in the reference) */
static uint64_t eq_mask(const uint8_t *in, uint8_t c) {
    uint64_t m = 0;
    for (int j = 0; j < 64; j++)
        if (in[j] == c) m |= 1ULL << j;
    return m;
}

/* find_odd_backslash_sequences_amd64.s:24-61 (macro FIND_ODD_BACKSLASH_SEQUENCES :34-58) */
uint64_t sjo_find_odd_backslash_sequences(const uint8_t *in, uint64_t *prev_iter_ends_odd_backslash) {
    const uint64_t even_bits = 0x5555555555555555ULL;
    const uint64_t odd_bits = 0xaaaaaaaaaaaaaaaaULL;
    uint64_t bs_bits = eq_mask(in, '\\');
    uint64_t start_edges = bs_bits & ~(bs_bits << 1);
    uint64_t prev = *prev_iter_ends_odd_backslash;
    /* flip lowest if we have an odd-length run at the end of the prior iteration */
    uint64_t even_start_mask = even_bits ^ prev;
    uint64_t even_starts = start_edges & even_start_mask;
    uint64_t odd_starts = start_edges & (odd_bits ^ prev); /* == start_edges & ~even_start_mask on bits!=0 */
    uint64_t even_carries = bs_bits + even_starts;
    uint64_t odd_carries = bs_bits + odd_starts;
    /* carry-out of the odd add: run of backslashes reaches the end with odd length */
    uint64_t iter_ends_odd_backslash = odd_carries < bs_bits ? 1 : 0;
    odd_carries |= prev; /* push in bit zero as a potential end if we had an odd-numbered run at the end of the previous iteration */
    *prev_iter_ends_odd_backslash = iter_ends_odd_backslash;
    uint64_t even_carry_ends = even_carries & ~bs_bits;
    uint64_t odd_carry_ends = odd_carries & ~bs_bits;
    uint64_t even_start_odd_end = even_carry_ends & odd_bits;
    uint64_t odd_start_even_end = odd_carry_ends & even_bits;
    return even_start_odd_end | odd_start_even_end;
}

/* carry-less multiply by all-ones == prefix XOR (VPCLMULQDQ at find_quote_mask_and_bits_amd64.s:65) */
static uint64_t prefix_xor(uint64_t x) {
    x ^= x << 1;
    x ^= x << 2;
    x ^= x << 4;
    x ^= x << 8;
    x ^= x << 16;
    x ^= x << 32;
    return x;
}

/* find_quote_mask_and_bits_amd64.s:49-84 */
uint64_t sjo_find_quote_mask_and_bits(const uint8_t *in, uint64_t odd_ends,
                                      uint64_t *prev_iter_inside_quote, uint64_t *quote_bits,
                                      uint64_t *error_mask) {
    uint64_t qb = eq_mask(in, '"') & ~odd_ends;
    *quote_bits = qb;
    uint64_t quote_mask = prefix_xor(qb) ^ *prev_iter_inside_quote;
    /* unescaped characters (< 0x20) within strings: (in ^ 0x80) <s 0xa0 (:67-78) */
    uint64_t unescaped = 0;
    for (int j = 0; j < 64; j++)
        if (in[j] <= 0x1f) unescaped |= 1ULL << j;
    *error_mask |= unescaped & quote_mask;
    *prev_iter_inside_quote = (uint64_t)((int64_t)quote_mask >> 63); /* SARQ $63 (:81) */
    return quote_mask;
}

/* find_whitespace_and_structurals_amd64.s:62-103; nibble LUTs at :6-45 */
void sjo_find_whitespace_and_structurals(const uint8_t *in, uint64_t *whitespace, uint64_t *structurals) {
    static const uint8_t low_nibble_mask[16] = {16, 0, 0, 0, 0, 0, 0, 0, 0, 8, 12, 1, 2, 9, 0, 0};
    static const uint8_t high_nibble_mask[16] = {8, 0, 18, 4, 0, 1, 0, 1, 0, 0, 0, 3, 2, 1, 0, 0};
    uint64_t ws = 0, st = 0;
    for (int j = 0; j < 64; j++) {
        uint8_t b = in[j];
        /* VPSHUFB zeroes lanes whose index byte has bit 7 set; the high nibble is taken
         * from (b >> 4) & 0x7f, i.e. bytes >= 0x80 index entries 8..15 of the high table
         * (all of 0,0,0,3,2,1,0,0 would be wrong) -- the reference shifts as 64-bit lanes and
         * masks with 0x7f, so the index keeps bit 3 of the high nibble; entries 8..15 are
         * only reachable for b >= 0x80, and for those the LOW lookup index (b itself) has
         * bit 7 set => low lookup is 0 => V == 0.  Net: bytes >= 0x80 are neither. */
        uint8_t lo = (b & 0x80) ? 0 : low_nibble_mask[b & 0x0f];
        uint8_t hi = high_nibble_mask[(b >> 4) & 0x0f];
        uint8_t v = lo & hi;
        if (v & 0x07) st |= 1ULL << j;
        if (v & 0x18) ws |= 1ULL << j;
    }
    *whitespace = ws;
    *structurals = st;
}

/* finalize_structurals_amd64.s:19-36 */
uint64_t sjo_finalize_structurals(uint64_t structurals, uint64_t whitespace, uint64_t quote_mask,
                                  uint64_t quote_bits, uint64_t *prev_iter_ends_pseudo_pred) {
    /* mask off anything inside quotes */
    structurals &= ~quote_mask;
    /* add the real quote bits back into our bitmask as well */
    structurals |= quote_bits;
    uint64_t pseudo_pred = structurals | whitespace;
    uint64_t shifted_pseudo_pred = (pseudo_pred << 1) | *prev_iter_ends_pseudo_pred;
    *prev_iter_ends_pseudo_pred = pseudo_pred >> 63;
    uint64_t pseudo_structurals = shifted_pseudo_pred & (~whitespace) & (~quote_mask);
    structurals |= pseudo_structurals;
    /* now, we've used our close quotes all we need to: eliminate them */
    structurals &= ~(quote_bits & ~quote_mask);
    return structurals;
}

/* find_newline_delimiters_amd64.s:16-28 */
uint64_t sjo_find_newline_delimiters(const uint8_t *in, uint64_t quote_mask) {
    return eq_mask(in, 0x0a) & ~quote_mask;
}

/* flatten_bits_amd64.s:26-60 */
void sjo_flatten_bits_incremental(uint32_t *base, int *base_index, uint64_t mask, uint64_t *carried,
                                  uint64_t *position) {
    uint64_t shifts = 0;
    int idx = *base_index;
    int first = 1;
    while (mask != 0) {
        uint64_t zeros = (uint64_t)__builtin_ctzll(mask);
        if (first) {
            /* two shifts because (63+1) exceeds a 6-bit shift count (:36-38) */
            mask >>= 1;
            mask >>= zeros;
            zeros += 1;
            shifts += zeros;
            zeros += *carried;
            *carried = 0;
            first = 0;
        } else {
            zeros += 1;
            mask = zeros >= 64 ? 0 : mask >> zeros; /* x86 SHRQ masks the count to 6 bits; zeros<=64 only if mask had just bit 63 -> handled below */
            shifts += zeros;
        }
        base[idx++] = (uint32_t)zeros;
        *position += zeros;
    }
    *base_index = idx;
    *carried += 64 - shifts;
}

/* single-chunk fused routine: find_structural_bits_amd64.s:3-36 */
uint64_t sjo_find_structural_bits(const uint8_t *in, uint64_t *prev_iter_ends_odd_backslash,
                                  uint64_t *prev_iter_inside_quote, uint64_t *error_mask,
                                  uint64_t *prev_iter_ends_pseudo_pred) {
    uint64_t quote_bits = 0, whitespace = 0, structurals = 0;
    uint64_t odd_ends = sjo_find_odd_backslash_sequences(in, prev_iter_ends_odd_backslash);
    uint64_t quote_mask =
        sjo_find_quote_mask_and_bits(in, odd_ends, prev_iter_inside_quote, &quote_bits, error_mask);
    sjo_find_whitespace_and_structurals(in, &whitespace, &structurals);
    return sjo_finalize_structurals(structurals, whitespace, quote_mask, quote_bits,
                                    prev_iter_ends_pseudo_pred);
}

/* one loop body of find_structural_bits_amd64.s:56-115 */
static void slice_chunk(const uint8_t *chunk, uint64_t *peob, uint64_t *piq, uint64_t *em, uint64_t *pepp,
                        uint32_t *indexes, int *index, uint64_t *carried, uint64_t *position,
                        uint64_t ndjson) {
    uint64_t quote_bits = 0, whitespace = 0, structurals = 0;
    uint64_t odd_ends = sjo_find_odd_backslash_sequences(chunk, peob);
    uint64_t quote_mask = sjo_find_quote_mask_and_bits(chunk, odd_ends, piq, &quote_bits, em);
    sjo_find_whitespace_and_structurals(chunk, &whitespace, &structurals);
    uint64_t s = sjo_finalize_structurals(structurals, whitespace, quote_mask, quote_bits, pepp);
    if (ndjson) s |= sjo_find_newline_delimiters(chunk, quote_mask); /* :91-96 */
    sjo_flatten_bits_incremental(indexes, index, s, carried, position);
}

/* _find_structural_bits_in_slice: find_structural_bits_amd64.s:49-155 (+ Go wrapper
 * find_subroutines_amd64.go:151-173, indexes_len = indexSizeWithSafetyBuffer) */
uint64_t sjo_find_structural_bits_in_slice(const uint8_t *buf, uint64_t len,
                                           uint64_t *prev_iter_ends_odd_backslash,
                                           uint64_t *prev_iter_inside_quote, uint64_t *error_mask,
                                           uint64_t *prev_iter_ends_pseudo_pred, uint32_t *indexes,
                                           int *index, uint64_t *carried, uint64_t *position,
                                           uint64_t ndjson) {
    if (len == 0) return 0; /* Go wrapper :157-159 */
    uint64_t ax = 0;
    uint64_t cx = len & ~63ULL;
    while (ax < cx) {
        slice_chunk(buf + ax, prev_iter_ends_odd_backslash, prev_iter_inside_quote, error_mask,
                    prev_iter_ends_pseudo_pred, indexes, index, carried, position, ndjson);
        ax += 64;
        if (*index >= SJO_INDEX_SIZE_SAFE) return ax; /* :111-112 */
    }
    /* check_partial_load (:124-128): mask the remaining (<64) bytes with whitespace (:134-155) */
    uint64_t rem = len & 63;
    if (rem != 0) {
        uint8_t chunk[64];
        memset(chunk, 0x20, sizeof chunk);
        memcpy(chunk, buf + ax, rem);
        slice_chunk(chunk, prev_iter_ends_odd_backslash, prev_iter_inside_quote, error_mask,
                    prev_iter_ends_pseudo_pred, indexes, index, carried, position, ndjson);
        ax += rem;
    }
    return ax;
}

static int json_markup(uint8_t b) { /* stage1_find_marks_amd64.go:28-39 */
    return b == '{' || b == '}' || b == '[' || b == ']' || b == ',' || b == ':';
}

/* findStructuralIndices: stage1_find_marks_amd64.go:41-148.  The channel hand-off of
 * index buffers is replaced by appending absolute positions to pos_out. */
int sjo_find_structural_indices(const uint8_t *msg, size_t len, int ndjson, uint32_t *pos_out,
                                size_t pos_cap, size_t *n_out) {
    const uint8_t *buf = msg;
    uint64_t buflen = len;
    uint64_t prev_iter_ends_odd_backslash = 0;
    uint64_t prev_iter_inside_quote = 0;
    uint64_t prev_iter_ends_pseudo_pred = 1;
    uint64_t error_mask = 0;
    size_t index_total = 0;
    uint64_t carried = 0;
    uint64_t position = ~0ULL;
    uint64_t stripped_index = ~0ULL;
    uint64_t base = 0;          /* bytes of msg consumed by earlier iterations */
    uint64_t abs_prev = ~0ULL;  /* absolute position of the last index delivered (−1 initially) */
    uint32_t *indexes = (uint32_t *)malloc(sizeof(uint32_t) * SJO_INDEX_SIZE);
    size_t n = 0;
    int overflow = 0;

    while (buflen > 0) {
        int length = 0;
        if (stripped_index != ~0ULL) {
            position += stripped_index;
            indexes[0] = (uint32_t)stripped_index;
            length = 1;
            stripped_index = ~0ULL;
        }
        uint64_t processed = sjo_find_structural_bits_in_slice(
            buf, buflen & ~63ULL, &prev_iter_ends_odd_backslash, &prev_iter_inside_quote, &error_mask,
            &prev_iter_ends_pseudo_pred, indexes, &length, &carried, &position, (uint64_t)ndjson);
        if (buflen - processed <= 64) {
            uint8_t padded[128];
            memset(padded, 0, sizeof padded);
            uint64_t padded_bytes = buflen - processed;
            memcpy(padded, buf + processed, padded_bytes);
            processed += sjo_find_structural_bits_in_slice(
                padded, padded_bytes, &prev_iter_ends_odd_backslash, &prev_iter_inside_quote,
                &error_mask, &prev_iter_ends_pseudo_pred, indexes, &length, &carried, &position,
                (uint64_t)ndjson);
        }
        if (length == 0) { /* :115-118 */
            error_mask = ~0ULL;
            break;
        }
        if (buflen == processed) { /* :120-129 */
            if (prev_iter_inside_quote != 0 || position >= buflen ||
                !(buf[position] == '}' || buf[position] == ']')) {
                error_mask = ~0ULL;
                break;
            }
        } else if (!json_markup(buf[position])) { /* :130-136 */
            stripped_index = indexes[length - 1];
            position -= stripped_index;
            length -= 1;
        }
        /* pj.indexChans <- index (:138): deliver as absolute positions */
        for (int i = 0; i < length; i++) {
            abs_prev += indexes[i];
            if (n < pos_cap) pos_out[n] = (uint32_t)abs_prev;
            else overflow = 1;
            n++;
        }
        index_total += (size_t)length;
        buf += processed;
        buflen -= processed;
        position -= processed;
        base += processed;
    }
    (void)base;
    free(indexes);
    *n_out = n;
    if (overflow) return 0;
    return error_mask == 0 && index_total > 0;
}
