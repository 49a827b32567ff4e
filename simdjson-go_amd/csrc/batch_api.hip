// batch_api.hip -- many documents in one launch set: sjhip_parse_batch / sjhip_parse_batch_device.
//
// The reference parses many small documents with one goroutine per Parse() (benchmarks_test.go:60-75).  A GPU parse has
// a fixed cost per launch set (single documents), so the batched form packs the documents into ONE device
// message and parses it as one ND document:
//   * every document is trimmed like Parse() trims it (bytes.TrimSpace, parse_json_amd64.go:55); an empty one fails the
//     batch with the stage-1 code, as Parse() of it would;
//   * the documents are laid out one behind the other, separated by '\n';
//   * a raw '\n' INSIDE a document becomes '\r': outside strings both are whitespace, inside a string both are the
//     same stage-1 error (a control character, find_quote_mask_and_bits_amd64.s:67-80) -- and "[1]\n[2]", two roots in
//     a single document, stays the stage-2 error it is in Parse() instead of turning into two records;
//   * every document must end like Parse() demands of a message (last structural '}' or ']',
//     stage1_find_marks_amd64.go:115-129): in the packed message only the last document would meet that rule.
// The result is what ParseND of that message returns: document i is root i of the tape (Iter.Advance walks them), string
// words point into one Strings.B.  One invalid document fails the whole batch (ParseND semantics).  Needs
// SJHIP_FLAG_COPY_STRINGS: without it string words would point into the packed message, which the caller never sees.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../include/sjhip.h"
#include "sj_chunk.h"
#include "sj_ctx.h"
#include "sj_host.h"

namespace {

struct DocDesc {
    uint64_t src;  // offset of the (trimmed) document in the source buffer
    uint64_t dst;  // offset in the packed message
    uint64_t len;
};

// The document that holds byte `p` of the packed message (or whose separator it is): the last one with dst <= p
__device__ __forceinline__ uint32_t doc_of(const DocDesc *__restrict__ docs, uint32_t n_docs, uint64_t p) {
    uint32_t lo = 0, hi = n_docs;  // docs[lo].dst <= p < docs[hi].dst
    while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (docs[mid].dst <= p) lo = mid;
        else hi = mid;
    }
    return lo;
}
// Parse() of a document fails in stage 1 unless its last structural is '}' or ']' (stage1_find_marks_amd64.go:115-129)
// -- a scalar, a truncated document, an empty or all-whitespace one.  Inside a batch only the last document meets that
// check, so every document is held to it here: the last byte that is not JSON whitespace must close a container.
__device__ __forceinline__ bool doc_end_ok(const uint8_t *__restrict__ s, uint64_t len) {
    while (len != 0) {
        const uint8_t b = s[len - 1];
        if (b == ' ' || b == '\t' || b == '\n' || b == '\r') len--;
        else return b == '}' || b == ']';
    }
    return false;
}

// One block per 4 KiB of the PACKED message (a grid over the longest document times the number of documents launches
// billions of empty blocks for one 100 MB document among a million small ones): a thread owns 16 packed bytes, finds
// its document with a binary search over the descriptors and walks on from there -- copy with '\n' -> '\r', the
// separator behind every document but the last, and the end check of the documents whose last byte it owns.
// IN_PLACE: the documents already lie at their packed offsets (host packing), only translation and separators.
template <bool IN_PLACE>
__global__ __launch_bounds__(256) void k_batch_pack(const uint8_t *__restrict__ src, const DocDesc *__restrict__ docs,
                                                    uint8_t *__restrict__ dst, uint32_t n_docs, uint64_t total,
                                                    unsigned int *bad_host) {
    const uint64_t p0 = (uint64_t)blockIdx.x * 4096 + (uint64_t)threadIdx.x * 16;
    if (p0 >= total) return;
    const uint64_t p1 = p0 + 16 < total ? p0 + 16 : total;
    uint32_t k = doc_of(docs, n_docs, p0);
    DocDesc d = docs[k];
    bool bad = false;
    if (p1 <= d.dst + d.len && p0 + 16 == p1) {
        // sixteen bytes of one document (the source is byte-aligned at best: unaligned 16-byte loads are fine on gfx950)
        uint4 v = *reinterpret_cast<const uint4 *>(IN_PLACE ? dst + p0 : src + d.src + (p0 - d.dst));
        v.x = sj::newlines_to_cr(v.x); v.y = sj::newlines_to_cr(v.y); v.z = sj::newlines_to_cr(v.z); v.w = sj::newlines_to_cr(v.w);
        *reinterpret_cast<uint4 *>(dst + p0) = v;
        if (!IN_PLACE && p1 == d.dst + d.len) bad = !doc_end_ok(src + d.src, d.len);
    } else {
        uint64_t next_dst = k + 1 < n_docs ? docs[k + 1].dst : ~0ull;
        for (uint64_t p = p0; p < p1; p++) {
            while (next_dst <= p) {
                d = docs[++k];
                next_dst = k + 1 < n_docs ? docs[k + 1].dst : ~0ull;
            }
            const uint64_t off = p - d.dst;
            if (off < d.len) {
                const uint8_t b = IN_PLACE ? dst[p] : src[d.src + off];
                dst[p] = b == '\n' ? (uint8_t)'\r' : b;
                if (!IN_PLACE && off + 1 == d.len) bad |= !doc_end_ok(src + d.src, d.len);
            } else {
                dst[p] = '\n';  // the separator behind document k
            }
        }
    }
    if (bad) __hip_atomic_store(bad_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

static int batch_common(sjhip_ctx *ctx, size_t n, uint32_t flags, size_t *tape_len, size_t *strings_len) {
    if (tape_len) *tape_len = 0;
    if (strings_len) *strings_len = 0;
    if (!ctx) return SJHIP_ERR_ARG;
    ctx->tape_len = ctx->strings_len = 0;
    if (!(flags & SJHIP_FLAG_COPY_STRINGS)) {
        sj::ctx_set_error(ctx, "sjhip_parse_batch needs SJHIP_FLAG_COPY_STRINGS (string words would point into the packed message)");
        return SJHIP_ERR_ARG;
    }
    if (n == 0) return SJHIP_ERR_STAGE1;  // like Parse of an empty message
    if (n > 0xfffffff0ull) {
        sj::ctx_set_error(ctx, "sjhip_parse_batch: too many documents in one call");
        return SJHIP_ERR_TOOBIG;
    }
    return SJHIP_OK;
}

// descriptors -> device (the context's query arena is free between parses)
static int upload_docs(sjhip_ctx *ctx, const std::vector<DocDesc> &docs, const DocDesc **d_docs) {
    int rc = sj::arena_reserve(ctx, ctx->d_q, docs.size() * sizeof(DocDesc));
    if (rc) return rc;
    if (hipMemcpyAsync(ctx->d_q.p, docs.data(), docs.size() * sizeof(DocDesc), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {  // (the vector goes away with the caller's frame)
        sj::ctx_set_error(ctx, "sjhip_parse_batch: descriptor upload failed");
        return SJHIP_ERR_HIP;
    }
    *d_docs = (const DocDesc *)ctx->d_q.p;
    return SJHIP_OK;
}

int sjhip_parse_batch(sjhip_ctx *ctx, const uint8_t *const *msgs, const size_t *lens, size_t n, uint32_t flags,
                      size_t *tape_len, size_t *strings_len) {
    int rc = batch_common(ctx, n, flags, tape_len, strings_len);
    if (rc) return rc;
    if (!msgs || !lens) return SJHIP_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return SJHIP_ERR_HIP;
    std::vector<DocDesc> docs(n);
    std::vector<size_t> host_off(n);
    uint64_t total = 0;
    for (size_t k = 0; k < n; k++) {
        size_t off = 0, ln = 0;
        if (lens[k] && msgs[k]) sj::trim_space(msgs[k], lens[k], &off, &ln);
        if (ln == 0) return SJHIP_ERR_STAGE1;
        // Parse() of this document alone fails in stage 1 unless its last structural closes a container
        // (stage1_find_marks_amd64.go:115-129); inside the packed message only the last document meets that check
        if (msgs[k][off + ln - 1] != '}' && msgs[k][off + ln - 1] != ']') return SJHIP_ERR_STAGE1;
        host_off[k] = off;
        docs[k] = DocDesc{0, total, ln};
        total += ln + (k + 1 < n ? 1 : 0);
    }
    if (total > 0xffffffc0ull) {
        sj::ctx_set_error(ctx, "sjhip_parse_batch: the packed message exceeds 4 GiB");
        return SJHIP_ERR_TOOBIG;
    }
    rc = sj::arena_reserve(ctx, ctx->d_msg, total + 128);
    if (rc) return rc;
    // H2D.  A copy command costs microseconds whatever it moves, so consecutive small documents are gathered in a pinned
    // block at their packed offsets (the separator bytes between them are written by k_batch_pack<true>) and travel as one
    // copy; a document of STAGE_DOC bytes or more goes straight from the caller's buffer.
    constexpr size_t STAGE_DOC = 64 << 10, STAGE_CAP = 32 << 20;
    if (!ctx->h_stage && sj::pinned_alloc((void **)&ctx->h_stage, STAGE_CAP) == hipSuccess) ctx->h_stage_cap = STAGE_CAP;
    if (!ctx->h_stage) (void)hipGetLastError();  // (no pinned memory: every document is copied on its own)
    auto h2d = [&](uint64_t dst, const void *src, size_t bytes) {
        return hipMemcpyAsync((uint8_t *)ctx->d_msg.p + dst, src, bytes, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
    };
    bool ok = true;
    uint64_t run_dst = 0;   // packed offset of the staged run
    size_t run_len = 0, stage_used = 0;  // bytes of the open run; bytes of the block that copies in flight may still read
    auto flush = [&]() {
        if (run_len) ok = ok && h2d(run_dst, ctx->h_stage + stage_used, run_len);
        stage_used += run_len;
        run_len = 0;
    };
    for (size_t k = 0; k < n && ok; k++) {
        const uint8_t *src = msgs[k] + host_off[k];
        if (!ctx->h_stage || docs[k].len >= STAGE_DOC) {
            flush();
            ok = ok && h2d(docs[k].dst, src, docs[k].len);
            continue;
        }
        const size_t gap = run_len ? (size_t)(docs[k].dst - (run_dst + run_len)) : 0;  // the separator in front (1 byte)
        if (stage_used + run_len + gap + docs[k].len > ctx->h_stage_cap) {  // the block is full: wait for its copies
            flush();
            ok = ok && hipStreamSynchronize(ctx->stream) == hipSuccess;
            stage_used = 0;
        }
        if (run_len == 0) run_dst = docs[k].dst;
        else run_len += (size_t)(docs[k].dst - (run_dst + run_len));
        memcpy(ctx->h_stage + stage_used + run_len, src, docs[k].len);
        run_len += docs[k].len;
    }
    flush();
    if (!ok) {
        sj::ctx_set_error(ctx, "sjhip_parse_batch: H2D of the documents failed");
        return SJHIP_ERR_HIP;
    }
    const DocDesc *d_docs = nullptr;
    rc = upload_docs(ctx, docs, &d_docs);
    if (rc) return rc;
    hipLaunchKernelGGL(k_batch_pack<true>, dim3((unsigned)((total + 4095) / 4096)), dim3(256), 0, ctx->stream, (const uint8_t *)nullptr, d_docs,
                       (uint8_t *)ctx->d_msg.p, (uint32_t)n, (uint64_t)total, (unsigned int *)nullptr);
    if (hipGetLastError() != hipSuccess) return SJHIP_ERR_HIP;
    const size_t last = n - 1;
    uint8_t last_byte = msgs[last][host_off[last] + docs[last].len - 1];
    if (last_byte == '\n') last_byte = '\r';  // (cannot happen: trimmed)
    return sj::parse_packed(ctx, (size_t)total, flags | SJHIP_FLAG_NDJSON, last_byte, 1, tape_len, strings_len);
}

int sjhip_parse_batch_device(sjhip_ctx *ctx, const void *d_buf, const size_t *offs, const size_t *lens, size_t n,
                             uint32_t flags, size_t *tape_len, size_t *strings_len) {
    int rc = batch_common(ctx, n, flags, tape_len, strings_len);
    if (rc) return rc;
    if (!d_buf || !offs || !lens) return SJHIP_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return SJHIP_ERR_HIP;
    // the documents are on the device: they are taken as they are (no trimming -- leading / trailing JSON whitespace of a
    // document is whitespace of its record); a document whose last non-whitespace byte does not close a container --
    // an empty or all-whitespace one, a scalar, a truncated one -- fails the batch with the stage-1 code like Parse()
    // of it would (checked on the device while the documents are packed)
    std::vector<DocDesc> docs(n);
    uint64_t total = 0;
    for (size_t k = 0; k < n; k++) {
        if (lens[k] == 0) return SJHIP_ERR_STAGE1;
        docs[k] = DocDesc{offs[k], total, lens[k]};
        total += lens[k] + (k + 1 < n ? 1 : 0);
    }
    if (total > 0xffffffc0ull) {
        sj::ctx_set_error(ctx, "sjhip_parse_batch_device: the packed message exceeds 4 GiB");
        return SJHIP_ERR_TOOBIG;
    }
    rc = sj::arena_reserve(ctx, ctx->d_msg, total + 128);
    if (rc) return rc;
    const DocDesc *d_docs = nullptr;
    rc = upload_docs(ctx, docs, &d_docs);
    if (rc) return rc;
    // the end check of every document (doc_end_ok) reports through a word of the context's pinned scratch block; the
    // parse below synchronises the stream, after which the word is final
    volatile unsigned int *bad = (volatile unsigned int *)(ctx->h_scratch + 1024);
    *bad = 0;
    hipLaunchKernelGGL(k_batch_pack<false>, dim3((unsigned)((total + 4095) / 4096)), dim3(256), 0, ctx->stream, (const uint8_t *)d_buf, d_docs,
                       (uint8_t *)ctx->d_msg.p, (uint32_t)n, (uint64_t)total, (unsigned int *)bad);
    if (hipGetLastError() != hipSuccess) return SJHIP_ERR_HIP;
    rc = sj::parse_packed(ctx, (size_t)total, flags | SJHIP_FLAG_NDJSON, 0, 0, tape_len, strings_len);
    if (rc != SJHIP_OK && rc != SJHIP_ERR_STAGE1 && rc != SJHIP_ERR_STAGE2) return rc;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return SJHIP_ERR_HIP;  // (a parse that failed early may not have waited)
    if (*bad) {  // a document that Parse() rejects in stage 1: that code wins (parse_json_amd64.go:97-105,123-126)
        ctx->tape_len = ctx->strings_len = 0;
        ctx->q_valid = ctx->r_valid = ctx->pack_valid = 0;
        if (tape_len) *tape_len = 0;
        if (strings_len) *strings_len = 0;
        return SJHIP_ERR_STAGE1;
    }
    return rc;
}
