// batch_api.hip -- many documents in one launch set: sjhip_parse_batch / sjhip_parse_batch_device.
//
// The reference parses many small documents with one goroutine per Parse() (benchmarks_test.go:60-75).  A GPU parse has
// a fixed cost per launch set (single documents), so the batched form packs the documents into ONE device
// message and parses it as one ND document:
//   * every document is trimmed like Parse() trims it (bytes.TrimSpace, parse_json_amd64.go:55); an empty one fails the
//     batch with the stage-1 code, as Parse() of it would;
//   * the documents are laid out one behind the other, separated by '\n';
//   * a raw '\n' INSIDE a document becomes '\r': outside strings both are whitespace, inside a string both are the
//     same stage-1 error (a control character, find_quote_mask_and_bits_amd64.s:67-80) -- and "1\n2", two roots in a
//     single document, stays the stage-2 error it is in Parse() instead of turning into two records.
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

// one block row per document (blockIdx.y), 4 KiB per block: copy with '\n' -> '\r'; the block that holds the end of
// a document writes the separator behind it
__global__ __launch_bounds__(256) void k_batch_pack(const uint8_t *__restrict__ src, const DocDesc *__restrict__ docs,
                                                    uint8_t *__restrict__ dst, uint32_t n_docs) {
    const uint32_t doc = blockIdx.z * 65535u + blockIdx.y;
    if (doc >= n_docs) return;
    const DocDesc d = docs[doc];
    const uint64_t first = (uint64_t)blockIdx.x * 4096;
    if (first >= d.len) return;
    const uint8_t *s = src + d.src;
    uint8_t *o = dst + d.dst;
    const uint64_t end = first + 4096 < d.len ? first + 4096 : d.len;
    // 16 bytes per thread where source and destination allow it, bytes otherwise
    const bool aligned = (((uintptr_t)(s + first) | (uintptr_t)(o + first)) & 15u) == 0;
    const uint64_t i = first + (uint64_t)threadIdx.x * 16;
    if (aligned && i + 16 <= end) {
        uint4 v = *reinterpret_cast<const uint4 *>(s + i);
        v.x = sj::newlines_to_cr(v.x); v.y = sj::newlines_to_cr(v.y); v.z = sj::newlines_to_cr(v.z); v.w = sj::newlines_to_cr(v.w);
        *reinterpret_cast<uint4 *>(o + i) = v;
    } else {
        for (uint64_t k = i; k < i + 16 && k < end; k++) {
            const uint8_t b = s[k];
            o[k] = b == '\n' ? (uint8_t)'\r' : b;
        }
    }
    if (end == d.len && threadIdx.x == 0 && doc + 1 < n_docs) o[d.len] = '\n';
}

// in place: the documents are already where they belong (host packing): only the translation and the separators
__global__ __launch_bounds__(256) void k_batch_fix(uint8_t *__restrict__ msg, const DocDesc *__restrict__ docs, uint32_t n_docs) {
    const uint32_t doc = blockIdx.z * 65535u + blockIdx.y;
    if (doc >= n_docs) return;
    const DocDesc d = docs[doc];
    const uint64_t first = (uint64_t)blockIdx.x * 4096;
    if (first >= d.len) return;
    uint8_t *o = msg + d.dst;
    const uint64_t end = first + 4096 < d.len ? first + 4096 : d.len;
    const uint64_t i = first + (uint64_t)threadIdx.x * 16;
    for (uint64_t k = i; k < i + 16 && k < end; k++)
        if (o[k] == '\n') o[k] = '\r';
    if (end == d.len && threadIdx.x == 0 && doc + 1 < n_docs) o[d.len] = '\n';
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
    if (n > 65535ull * 65535ull) {
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
    uint64_t total = 0, longest = 0;
    for (size_t k = 0; k < n; k++) {
        size_t off = 0, ln = 0;
        if (lens[k] && msgs[k]) sj::trim_space(msgs[k], lens[k], &off, &ln);
        if (ln == 0) return SJHIP_ERR_STAGE1;
        host_off[k] = off;
        docs[k] = DocDesc{0, total, ln};
        total += ln + (k + 1 < n ? 1 : 0);
        if (ln > longest) longest = ln;
    }
    if (total > 0xffffffc0ull) {
        sj::ctx_set_error(ctx, "sjhip_parse_batch: the packed message exceeds 4 GiB");
        return SJHIP_ERR_TOOBIG;
    }
    rc = sj::arena_reserve(ctx, ctx->d_msg, total + 128);
    if (rc) return rc;
    // H2D.  A copy command costs microseconds whatever it moves, so consecutive small documents are gathered in a pinned
    // block at their packed offsets (the separator bytes between them are written by k_batch_fix) and travel as one
    // copy; a document of STAGE_DOC bytes or more goes straight from the caller's buffer.
    constexpr size_t STAGE_DOC = 64 << 10, STAGE_CAP = 32 << 20;
    if (!ctx->h_stage && hipHostMalloc((void **)&ctx->h_stage, STAGE_CAP, hipHostMallocDefault) == hipSuccess) ctx->h_stage_cap = STAGE_CAP;
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
    const dim3 grid((unsigned)((longest + 4095) / 4096), (unsigned)(n < 65535 ? n : 65535), (unsigned)((n + 65534) / 65535));
    hipLaunchKernelGGL(k_batch_fix, grid, dim3(256), 0, ctx->stream, (uint8_t *)ctx->d_msg.p, d_docs, (uint32_t)n);
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
    // the documents are on the device: they are taken as they are (no trimming -- leading / trailing whitespace of a
    // document is whitespace of its record) except that an empty document is refused
    std::vector<DocDesc> docs(n);
    uint64_t total = 0, longest = 0;
    for (size_t k = 0; k < n; k++) {
        if (lens[k] == 0) return SJHIP_ERR_STAGE1;
        docs[k] = DocDesc{offs[k], total, lens[k]};
        total += lens[k] + (k + 1 < n ? 1 : 0);
        if (lens[k] > longest) longest = lens[k];
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
    const dim3 grid((unsigned)((longest + 4095) / 4096), (unsigned)(n < 65535 ? n : 65535), (unsigned)((n + 65534) / 65535));
    hipLaunchKernelGGL(k_batch_pack, grid, dim3(256), 0, ctx->stream, (const uint8_t *)d_buf, d_docs, (uint8_t *)ctx->d_msg.p, (uint32_t)n);
    if (hipGetLastError() != hipSuccess) return SJHIP_ERR_HIP;
    return sj::parse_packed(ctx, (size_t)total, flags | SJHIP_FLAG_NDJSON, 0, 0, tape_len, strings_len);
}
