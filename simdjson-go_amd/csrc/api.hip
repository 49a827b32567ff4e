// api.hip -- the C ABI of libsjhip (include/sjhip.h): context, stage-1 entry points and the
// per-routine known-answer kernels.  Whole-parse entry points live in parse_api.hip.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/sjhip.h"
#include "sj_chunk.h"
#include "sj_ctx.h"
#include "sj_device.h"

using namespace sj;

// ---------------------------------------------------------------------------------------------
int sjhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sjhip_supported(void) {
    const int n = sjhip_device_count();
    for (int d = 0; d < n; d++) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, d) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) return 1;
    }
    return 0;
}

void sj::ctx_set_error(sjhip_ctx *ctx, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
    va_end(ap);
}

int sj::ctx_hip_fail(sjhip_ctx *ctx, hipError_t e, const char *what) {
    ctx_set_error(ctx, "%s: %s", what, hipGetErrorString(e));
    return SJHIP_ERR_HIP;
}

// grow-only device arena (the buffers are recycled across calls like the reference's `reuse`)
int sj::arena_reserve(sjhip_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap) return SJHIP_OK;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 8 + 4096;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // (the runtime keeps the error until somebody asks: the next launch check would)
        b.p = nullptr;
        ctx_set_error(ctx, "hipMalloc of %zu bytes for a device arena: %s", want, hipGetErrorString(e));
        return SJHIP_ERR_HIP;
    }
    b.cap = want;
    b.gen++;
    return SJHIP_OK;
}

sjhip_ctx *sjhip_ctx_create(int device) {
    if (device < 0 || device >= sjhip_device_count()) return nullptr;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    sjhip_ctx *ctx = new sjhip_ctx();
    ctx->device = device;
    ctx->err[0] = 0;
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return nullptr;
    }
    ctx->stream = ctx->own_stream;
    if (sj::pinned_alloc((void **)&ctx->h_scratch, 4096) != hipSuccess) {
        (void)hipStreamDestroy(ctx->own_stream);
        delete ctx;
        return nullptr;
    }
    (void)hipEventCreate(&ctx->ev0);
    (void)hipEventCreate(&ctx->ev1);
    return ctx;
}

// every device arena of a context
#define SJ_CTX_ARENAS(ctx)                                                                                                   \
    {&(ctx)->d_msg, &(ctx)->d_pos, &(ctx)->d_ws, &(ctx)->d_kat, &(ctx)->d_tape, &(ctx)->d_strings, &(ctx)->d_s2, &(ctx)->d_s2z, \
     &(ctx)->d_aux, &(ctx)->d_scol, &(ctx)->d_stab, &(ctx)->d_q, &(ctx)->d_qtape, &(ctx)->d_qstrings,       \
     &(ctx)->d_keyflag}

size_t sjhip_ctx_device_bytes(const sjhip_ctx *ctx) {
    if (!ctx) return 0;
    const DevBuf *bufs[] = SJ_CTX_ARENAS(ctx);
    size_t total = 0;
    for (const DevBuf *b : bufs) total += b->cap;
    return total + sj::nd_big_device_bytes(ctx);
}

static void invalidate_result(sjhip_ctx *ctx);

int sjhip_ctx_trim(sjhip_ctx *ctx) {
    if (!ctx) return SJHIP_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
        return ctx_hip_fail(ctx, hipGetLastError(), "sjhip_ctx_trim");
    invalidate_result(ctx);
    ctx->kf_valid = 0;
    ctx->tape_len = ctx->strings_len = 0;
    DevBuf *bufs[] = SJ_CTX_ARENAS(ctx);
    for (DevBuf *b : bufs) {
        if (b->p) (void)hipFree(b->p);
        b->p = nullptr;
        b->cap = 0;
        b->gen++;
    }
    ctx->p_kind = nullptr;
    ctx->p_aux = nullptr;
    ctx->p_msg = nullptr;
    sj::release_nd_big(ctx);
    (void)hipSetDevice(ctx->device);
    if (ctx->h_pack) (void)hipHostFree(ctx->h_pack);
    if (ctx->h_view) (void)hipHostFree(ctx->h_view);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->h_in) (void)hipHostFree(ctx->h_in);
    ctx->h_pack = ctx->h_view = ctx->h_stage = ctx->h_in = nullptr;
    ctx->h_view_cap = ctx->h_stage_cap = ctx->h_in_cap = 0;
    return SJHIP_OK;
}

void sjhip_ctx_destroy(sjhip_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    DevBuf *bufs[] = SJ_CTX_ARENAS(ctx);
    for (DevBuf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    sj::release_nd_big(ctx);
    (void)hipSetDevice(ctx->device);
    if (ctx->h_scratch) (void)hipHostFree(ctx->h_scratch);
    if (ctx->h_pack) (void)hipHostFree(ctx->h_pack);
    if (ctx->h_view) (void)hipHostFree(ctx->h_view);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->h_in) (void)hipHostFree(ctx->h_in);
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

uint8_t *sjhip_input_block(sjhip_ctx *ctx, size_t bytes) {
    if (!ctx) return nullptr;
    if (bytes <= ctx->h_in_cap && ctx->h_in) return ctx->h_in;
    if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
    (void)hipStreamSynchronize(ctx->stream);  // (a copy out of the old block may still be queued)
    if (ctx->h_in) (void)hipHostFree(ctx->h_in);
    ctx->h_in = nullptr;
    ctx->h_in_cap = 0;
    const size_t cap = (bytes + bytes / 4 + 4096) & ~(size_t)4095;
    const hipError_t e = sj::pinned_alloc((void **)&ctx->h_in, cap);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        ctx->h_in = nullptr;
        ctx_set_error(ctx, "hipHostMalloc of %zu bytes for the input block: %s", cap, hipGetErrorString(e));
        return nullptr;
    }
    ctx->h_in_cap = cap;
    return ctx->h_in;
}

const char *sjhip_last_error(const sjhip_ctx *ctx) { return ctx ? ctx->err : "no context"; }

int sjhip_ctx_set_stream(sjhip_ctx *ctx, void *hip_stream) {
    if (!ctx) return SJHIP_ERR_ARG;
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return SJHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// stage 1
// ---------------------------------------------------------------------------------------------
#define HIPCHK(call, what)                                   \
    do {                                                     \
        hipError_t e_ = (call);                              \
        if (e_ != hipSuccess) return ctx_hip_fail(ctx, e_, what); \
    } while (0)

// Reads back the Stage1State and applies the reference's end-of-document verdict
// (stage1_find_marks_amd64.go:115-129,147).  `last_byte` is msg[len-1].
static int stage1_verdict(const Stage1State &st, size_t len, uint8_t last_byte) {
    if (len == 0) return 0;
    if (st.error) return 0;          // error_mask != 0
    if (st.total == 0) return 0;     // indexTotal == 0
    if (st.ends_in_quote) return 0;  // prev_iter_inside_quote != 0
    // the last structural must be '}' or ']'.  The message is TrimSpace'd, so its last byte is not
    // whitespace: outside a string that byte is a structural exactly when it is one of {}[]:, and
    // otherwise it belongs to a token whose first byte (the last structural) is not a bracket.
    return last_byte == '}' || last_byte == ']';
}

// A stage-1-only call re-uses (and may re-allocate) the arenas a device-resident parse result lives in -- the message
// copy, the positions with the token kinds behind them: queries, the serializer and MarshalJSON must not run on what
// is left (they return SJHIP_ERR_ARG until the next parse).
static void invalidate_result(sjhip_ctx *ctx) {
    ctx->q_valid = ctx->r_valid = ctx->ser_valid = ctx->ms_valid = 0;
    ctx->pending = 0;
    ctx->q_tape_len = ctx->q_strings_len = 0;
    ctx->f_valid = 0;
    ctx->pack_valid = 0;  // (sjhip_fetch goes back to the device copies, which a stage-1 call does not touch)
}

// stage 1 in two halves: enqueue (workspace, launch; the last block of the kernel leaves the packed result -- count,
// flags, the last message byte -- in one word of pinned host memory: no copy kernels behind the launch) and, once the
// stream has been synchronised, collect (the reference's end-of-document verdict).  (Polling the word instead of
// synchronising -- going on while the kernel's caches are written back -- measured no gain for the whole parse and
// would hand positions to other streams before they are visible there.)
// Stage 1 runs without a preparation kernel: every launch zeroes, for the next one, the control slot and the descriptor set
// the one before it used (sj_device.h Stage1State).  A workspace that has just been allocated is zeroed here, once.
static int stage1_workspace_clean(sjhip_ctx *ctx) {
    if (ctx->d_ws.p && ctx->ws_clean_gen != ctx->d_ws.gen) {
        ctx->s1ws.p = ctx->d_ws.p;
        ctx->s1ws.bytes = ctx->d_ws.cap;
        HIPCHK(stage1_prepare(ctx->s1ws, ctx->stream), "stage-1 workspace memset");
        ctx->ws_clean_gen = ctx->d_ws.gen;
    }
    return SJHIP_OK;
}

int sj::stage1_enqueue(sjhip_ctx *ctx, const void *d_msg, size_t len, int ndjson, void *d_pos, size_t pos_cap, void *str_aux,
                       uint8_t *d_kind, void *zero2, size_t zero2_bytes, unsigned long long *host_rec) {
    if (!host_rec) host_rec = (unsigned long long *)ctx->h_scratch;
    // (plain stage 1 hands out 32-bit positions: up to 4 GiB - 64; the whole parse -- str_aux -- lets them wrap, parse_api.hip)
    if (len >= 0xffffffc0ull && !str_aux) {
        ctx_set_error(ctx, "message too long for uint32 positions");
        return SJHIP_ERR_TOOBIG;
    }
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    int rc = arena_reserve(ctx, ctx->d_ws, stage1_workspace_bytes(len + 64));
    if (rc) return rc;
    rc = stage1_workspace_clean(ctx);
    if (rc) return rc;
    if (len > 0) {
        for (int k = 0; k < S1_HOST_WORDS; k++) ((volatile unsigned long long *)host_rec)[k] = 0;
        ctx->s1_par = ctx->s1ws.epoch & 1u;
        HIPCHK(stage1_launch(d_msg, len, ndjson, (uint32_t *)d_pos, pos_cap, ctx->s1ws, ctx->stream, str_aux, d_kind,
                             host_rec, zero2, zero2_bytes),
               "stage1 launch");
    }
    return SJHIP_OK;
}

int sj::stage1_collect(sjhip_ctx *ctx, size_t len, uint8_t last_byte, int have_last, size_t *n, int *ok,
                       const unsigned long long *host_rec) {
    if (!host_rec) host_rec = (const unsigned long long *)ctx->h_scratch;
    Stage1State hs_v;
    Stage1State *hs = &hs_v;
    memset(hs, 0, sizeof *hs);
    if (len > 0) {
        const unsigned long long word = *(const volatile unsigned long long *)host_rec;
        if (!(word & S1_HOST_VALID)) {
            ctx_set_error(ctx, "stage-1 kernel left no result");
            return SJHIP_ERR_HIP;
        }
        hs->total = word & S1_HOST_TOTAL_MASK;
        const volatile unsigned long long *hw = (const volatile unsigned long long *)host_rec;
        hs->error = (hw[1] ? 1u : 0u) | (hw[2] ? 0x80000000u : 0u);
        hs->ends_in_quote = (word & S1_HOST_IN_QUOTE) ? 1u : 0u;
        hs->last_byte = (uint32_t)((word >> S1_HOST_LAST_SHIFT) & 0xffu);
    }
    if (!have_last) last_byte = (uint8_t)hs->last_byte;
    ctx->s1 = *hs;
    if (hs->error & 0x80000000u) {  // a bounded spin loop of the kernel ran out: internal error, never a verdict
        ctx_set_error(ctx, "stage-1 kernel aborted (internal synchronisation timeout)");
        return SJHIP_ERR_HIP;
    }
    {  // debug build: an out-of-bounds store of the stage-1 kernel fails the call, whatever its verdict
        unsigned hits = 0, id = 0;
        unsigned long long index = 0, size = 0;
        if (stage1_debug_bounds(&hits, &id, &index, &size) && hits) {
            ctx_set_error(ctx, "bounds check: %u out-of-bounds accesses in stage 1, the first to array %u (sj_bounds.h ArrId) at element %llu of %llu",
                          hits, id, index, size);
            return SJHIP_ERR_HIP;
        }
    }
    *n = (size_t)hs->total;
    *ok = stage1_verdict(*hs, len, last_byte);
    return SJHIP_OK;
}

int sj::stage1_run_device(sjhip_ctx *ctx, const void *d_msg, size_t len, int ndjson, void *d_pos, size_t pos_cap,
                          uint8_t last_byte, int have_last, size_t *n, int *ok, void *str_aux, uint8_t *d_kind, void *zero2,
                          size_t zero2_bytes) {
    int rc = stage1_enqueue(ctx, d_msg, len, ndjson, d_pos, pos_cap, str_aux, d_kind, zero2, zero2_bytes);
    if (rc) return rc;
    if (len > 0) HIPCHK(hipStreamSynchronize(ctx->stream), "stage1 sync");
    return stage1_collect(ctx, len, last_byte, have_last, n, ok);
}

int sjhip_stage1_device(sjhip_ctx *ctx, const void *d_msg, size_t len, int ndjson, void *d_pos, size_t pos_cap,
                        size_t *n, int *ok) {
    if (!ctx || !n || !ok) return SJHIP_ERR_ARG;
    invalidate_result(ctx);
    return stage1_run_device(ctx, d_msg, len, ndjson != 0, d_pos, pos_cap, 0, 0, n, ok);
}

// Queued form: launches behind one another on the context's stream, no synchronisation of their own; every launch has its
// own record in pinned host memory (the upper half of h_scratch: SJHIP_STAGE1_QUEUE_SLOTS records of 32 bytes).
static inline unsigned long long *s1q_record(sjhip_ctx *ctx, int slot) {
    return (unsigned long long *)(ctx->h_scratch + 2048 + (size_t)slot * 32);
}
int sjhip_stage1_device_queue(sjhip_ctx *ctx, const void *d_msg, size_t len, int ndjson, void *d_pos, size_t pos_cap, int slot) {
    if (!ctx || slot < 0 || slot >= SJHIP_STAGE1_QUEUE_SLOTS) return SJHIP_ERR_ARG;
    static_assert(SJHIP_STAGE1_QUEUE_SLOTS * 32 <= 2048 && S1_HOST_WORDS * 8 <= 32, "the records fill the upper half of h_scratch");
    invalidate_result(ctx);
    if (len == 0) {  // (no launch: the record says so)
        unsigned long long *r = s1q_record(ctx, slot);
        r[0] = r[1] = r[2] = 0;
        return SJHIP_OK;
    }
    return stage1_enqueue(ctx, d_msg, len, ndjson != 0, d_pos, pos_cap, nullptr, nullptr, nullptr, 0, s1q_record(ctx, slot));
}
int sjhip_stage1_device_wait(sjhip_ctx *ctx) {
    if (!ctx) return SJHIP_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    HIPCHK(hipStreamSynchronize(ctx->stream), "stage1 sync");
    return SJHIP_OK;
}
int sjhip_stage1_device_result(sjhip_ctx *ctx, int slot, size_t len, size_t *n, int *ok) {
    if (!ctx || !n || !ok || slot < 0 || slot >= SJHIP_STAGE1_QUEUE_SLOTS) return SJHIP_ERR_ARG;
    return stage1_collect(ctx, len, 0, 0, n, ok, s1q_record(ctx, slot));
}

int sjhip_stage1(sjhip_ctx *ctx, const uint8_t *msg, size_t len, int ndjson, uint32_t *pos_out, size_t pos_cap,
                 size_t *n, int *ok) {
    if (!ctx || !n || !ok) return SJHIP_ERR_ARG;
    invalidate_result(ctx);
    if (len >= 0xffffffc0ull) {  // before anything is allocated or copied
        ctx_set_error(ctx, "message too long for uint32 positions (4 GiB - 64)");
        return SJHIP_ERR_TOOBIG;
    }
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    int rc = arena_reserve(ctx, ctx->d_msg, len + 128);
    if (rc) return rc;
    rc = arena_reserve(ctx, ctx->d_pos, (pos_cap + 16) * sizeof(uint32_t));
    if (rc) return rc;
    if (len) HIPCHK(hipMemcpyAsync(ctx->d_msg.p, msg, len, hipMemcpyHostToDevice, ctx->stream), "H2D msg");
    rc = stage1_run_device(ctx, ctx->d_msg.p, len, ndjson != 0, ctx->d_pos.p, pos_cap, len ? msg[len - 1] : 0, 1, n, ok);
    if (rc) return rc;
    const size_t ncopy = *n < pos_cap ? *n : pos_cap;
    if (ncopy && pos_out) {
        HIPCHK(hipMemcpyAsync(pos_out, ctx->d_pos.p, ncopy * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream),
               "D2H positions");
        HIPCHK(hipStreamSynchronize(ctx->stream), "sync");
    }
    return SJHIP_OK;
}

int sjhip_stage1_time(sjhip_ctx *ctx, const void *d_msg, size_t len, int ndjson, void *d_pos, size_t pos_cap,
                      int iters, float *ms_per_launch) {
    if (!ctx || iters <= 0 || !ms_per_launch) return SJHIP_ERR_ARG;
    invalidate_result(ctx);
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    int rc = arena_reserve(ctx, ctx->d_ws, stage1_workspace_bytes(len + 64));
    if (rc) return rc;
    rc = stage1_workspace_clean(ctx);
    if (rc) return rc;
    // The launches of a parse, one behind the other on the context's workspace (each cleans up for the next: there is nothing
    // else to a launch), each bracketed by a pair of events.
    float total = 0.f;
    for (int i = 0; i < iters; i++) {
        HIPCHK(hipEventRecord(ctx->ev0, ctx->stream), "event");
        HIPCHK(stage1_launch(d_msg, len, ndjson != 0, (uint32_t *)d_pos, pos_cap, ctx->s1ws, ctx->stream), "stage1 launch");
        HIPCHK(hipEventRecord(ctx->ev1, ctx->stream), "event");
        HIPCHK(hipEventSynchronize(ctx->ev1), "event sync");
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1), "elapsed");
        total += ms;
    }
    *ms_per_launch = total / (float)iters;
    return SJHIP_OK;
}

int sjhip_stage1_set_variant(int variant) { return stage1_set_variant(variant); }

int sjhip_stage1_trace(sjhip_ctx *ctx, const void *d_msg, size_t len, void *d_pos, size_t pos_cap, uint64_t *trace_out,
                       size_t trace_cap_words, unsigned *tiles, int *waves, int *words) {
    if (!ctx || !trace_out || !tiles || !waves || !words) return SJHIP_ERR_ARG;
    invalidate_result(ctx);
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    const size_t lead = (size_t)(reinterpret_cast<uintptr_t>(d_msg) & 63);
    const size_t nw = stage1_trace_words(len, lead, tiles, waves);
    *words = 8;
    if (nw > trace_cap_words) {
        ctx_set_error(ctx, "trace needs %zu words", nw);
        return SJHIP_ERR_ARG;
    }
    int rc = arena_reserve(ctx, ctx->d_ws, stage1_workspace_bytes(len + 64));
    if (rc) return rc;
    rc = arena_reserve(ctx, ctx->d_kat, nw * sizeof(uint64_t));
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(ctx->d_kat.p, 0, nw * sizeof(uint64_t), ctx->stream), "trace memset");
    rc = stage1_workspace_clean(ctx);
    if (rc) return rc;
    HIPCHK(stage1_launch(d_msg, len, 0, (uint32_t *)d_pos, pos_cap, ctx->s1ws, ctx->stream, nullptr, nullptr, nullptr, nullptr, 0,
                         (unsigned long long *)ctx->d_kat.p),
           "stage1 launch (trace)");
    HIPCHK(hipMemcpyAsync(trace_out, ctx->d_kat.p, nw * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream), "D2H trace");
    HIPCHK(hipStreamSynchronize(ctx->stream), "trace sync");
    return SJHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// per-routine KAT kernels: a single lane runs the lane-local device functions of sj_chunk.h that stage1_kernel is
// made of (classify, prefix_xor, finalize).  Two routines exist in the kernel in another form than the reference's:
//   * odd backslashes: stage1_kernel uses escaped_mask() with the carry taken from the neighbour chunk's trailing
//     run; the KAT evaluates the reference form odd_backslash_ends() AND the kernel form on the device and the entry
//     point fails if they disagree on any non-backslash byte or on the carry;
//   * flatten: stage1_kernel writes absolute positions through flatten_tile (LDS staging, coalesced copy-out); the
//     KAT entry point runs the real kernel on a 64-byte message whose structural mask is the given mask and turns
//     the absolute positions into the reference's deltas on the host (the carried / position bookkeeping is transport).
// ---------------------------------------------------------------------------------------------
struct KatIO {
    uint8_t in[64];
    uint64_t a[8];    // scalar inputs
    uint64_t out[8];  // scalar outputs
};

enum { KAT_ODD_BS, KAT_QUOTE, KAT_WS, KAT_FINALIZE, KAT_NEWLINE };

__global__ void kat_kernel(KatIO *io, int op) {
    if (threadIdx.x != 0) return;
    u32 w[16];
    for (int j = 0; j < 16; j++)
        w[j] = (u32)io->in[4 * j] | ((u32)io->in[4 * j + 1] << 8) | ((u32)io->in[4 * j + 2] << 16) |
               ((u32)io->in[4 * j + 3] << 24);
    const Classes c = classify(w);
    switch (op) {
    case KAT_ODD_BS: {
        u32 co;
        io->out[0] = odd_backslash_ends(c.bs, (u32)io->a[0], co);
        io->out[1] = co;
        // the form stage1_kernel runs (stage1.hip phase_a): escaped characters, carry from the trailing run
        const u64 escaped = escaped_mask(c.bs, (u32)io->a[0]);
        io->out[2] = escaped & ~c.bs;                         // must equal out[0]
        io->out[3] = c.quote & ~escaped;                      // the kernel's quote_bits
        io->out[4] = c.quote & ~io->out[0];                   // the reference's quote_bits
        const bool all_bs = c.bs == ~0ull;
        io->out[5] = all_bs ? io->a[0] : ((u32)__builtin_clzll(~c.bs) & 1u);  // the kernel's carry into the next chunk
        break;
    }
    case KAT_QUOTE: {  // a0 = odd_ends, a1 = prev_inside_quote, a2 = error_mask (accumulated)
        const u64 qb = c.quote & ~io->a[0];
        const u64 qm = prefix_xor(qb) ^ io->a[1];
        io->out[0] = qm;
        io->out[1] = qb;
        io->out[2] = io->a[2] | (c.ctrl & qm);
        io->out[3] = (u64)((long long)qm >> 63);
        break;
    }
    case KAT_WS:
        io->out[0] = c.ws;
        io->out[1] = c.structs;
        break;
    case KAT_FINALIZE: {  // a0 structurals a1 whitespace a2 quote_mask a3 quote_bits a4 pseudo_pred
        io->out[0] = finalize(io->a[0], io->a[1], io->a[2], io->a[3], (u32)io->a[4]);
        io->out[1] = (((io->a[0] & ~io->a[2]) | io->a[3] | io->a[1]) >> 63) & 1;
        break;
    }
    case KAT_NEWLINE:
        io->out[0] = c.nl & ~io->a[0];
        break;
    }
}

static int kat_run(sjhip_ctx *ctx, KatIO &h, int op) {
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    int rc = arena_reserve(ctx, ctx->d_kat, sizeof(KatIO));
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(ctx->d_kat.p, &h, sizeof h, hipMemcpyHostToDevice, ctx->stream), "H2D kat");
    hipLaunchKernelGGL(kat_kernel, dim3(1), dim3(64), 0, ctx->stream, (KatIO *)ctx->d_kat.p, op);
    HIPCHK(hipGetLastError(), "kat launch");
    HIPCHK(hipMemcpyAsync(&h, ctx->d_kat.p, sizeof h, hipMemcpyDeviceToHost, ctx->stream), "D2H kat");
    HIPCHK(hipStreamSynchronize(ctx->stream), "kat sync");
    return SJHIP_OK;
}

int sjhip_find_odd_backslash_sequences(sjhip_ctx *ctx, const uint8_t in[64], uint64_t *prev, uint64_t *odd_ends) {
    if (!ctx) return SJHIP_ERR_ARG;
    KatIO h;
    memset(&h, 0, sizeof h);
    memcpy(h.in, in, 64);
    h.a[0] = *prev;
    int rc = kat_run(ctx, h, KAT_ODD_BS);
    if (rc) return rc;
    if (h.out[2] != h.out[0] || h.out[3] != h.out[4] || h.out[5] != h.out[1]) {
        ctx_set_error(ctx, "escaped_mask (kernel form) disagrees with odd_backslash_ends (reference form)");
        return SJHIP_ERR_HIP;
    }
    *odd_ends = h.out[0];
    *prev = h.out[1];
    return SJHIP_OK;
}

int sjhip_find_quote_mask_and_bits(sjhip_ctx *ctx, const uint8_t in[64], uint64_t odd_ends, uint64_t *prev_inside,
                                   uint64_t *quote_bits, uint64_t *error_mask, uint64_t *quote_mask) {
    if (!ctx) return SJHIP_ERR_ARG;
    KatIO h;
    memset(&h, 0, sizeof h);
    memcpy(h.in, in, 64);
    h.a[0] = odd_ends;
    h.a[1] = *prev_inside;
    h.a[2] = *error_mask;
    int rc = kat_run(ctx, h, KAT_QUOTE);
    if (rc) return rc;
    *quote_mask = h.out[0];
    *quote_bits = h.out[1];
    *error_mask = h.out[2];
    *prev_inside = h.out[3];
    return SJHIP_OK;
}

int sjhip_find_whitespace_and_structurals(sjhip_ctx *ctx, const uint8_t in[64], uint64_t *whitespace,
                                          uint64_t *structurals) {
    if (!ctx) return SJHIP_ERR_ARG;
    KatIO h;
    memset(&h, 0, sizeof h);
    memcpy(h.in, in, 64);
    int rc = kat_run(ctx, h, KAT_WS);
    if (rc) return rc;
    *whitespace = h.out[0];
    *structurals = h.out[1];
    return SJHIP_OK;
}

int sjhip_finalize_structurals(sjhip_ctx *ctx, uint64_t structurals, uint64_t whitespace, uint64_t quote_mask,
                               uint64_t quote_bits, uint64_t *pseudo_pred, uint64_t *out) {
    if (!ctx) return SJHIP_ERR_ARG;
    KatIO h;
    memset(&h, 0, sizeof h);
    h.a[0] = structurals;
    h.a[1] = whitespace;
    h.a[2] = quote_mask;
    h.a[3] = quote_bits;
    h.a[4] = *pseudo_pred;
    int rc = kat_run(ctx, h, KAT_FINALIZE);
    if (rc) return rc;
    *out = h.out[0];
    *pseudo_pred = h.out[1];
    return SJHIP_OK;
}

int sjhip_find_newline_delimiters(sjhip_ctx *ctx, const uint8_t in[64], uint64_t quote_mask, uint64_t *mask) {
    if (!ctx) return SJHIP_ERR_ARG;
    KatIO h;
    memset(&h, 0, sizeof h);
    memcpy(h.in, in, 64);
    h.a[0] = quote_mask;
    int rc = kat_run(ctx, h, KAT_NEWLINE);
    if (rc) return rc;
    *mask = h.out[0];
    return SJHIP_OK;
}

int sjhip_flatten_bits_incremental(sjhip_ctx *ctx, uint32_t *base, int *base_index, uint64_t mask, uint64_t *carried,
                                   uint64_t *position) {
    if (!ctx) return SJHIP_ERR_ARG;
    // ':' at every set bit, blanks elsewhere: the structural mask of this 64-byte message is `mask` (the reference's
    // own whitespace-padding test builds its inputs the same way, find_subroutines_amd64_test.go:381-421)
    uint8_t msg[64];
    for (int j = 0; j < 64; j++) msg[j] = ((mask >> j) & 1u) ? ':' : ' ';
    uint32_t pos[64];
    size_t n = 0;
    int ok = 0;
    int rc = sjhip_stage1(ctx, msg, 64, 0, pos, 64, &n, &ok);  // stage1_kernel -> flatten_tile
    if (rc) return rc;
    if (n != (size_t)__builtin_popcountll(mask)) {
        ctx_set_error(ctx, "flatten: %zu positions for a mask with %d bits", n, __builtin_popcountll(mask));
        return SJHIP_ERR_HIP;
    }
    // flatten_bits_amd64.s:26-60: deltas relative to the last index, `carried` = distance from it to the end of
    // the previous mask
    uint64_t last = *position;
    const uint64_t start = last + *carried + 1;  // absolute position of bit 0 of this mask
    for (size_t i = 0; i < n; i++) {
        const uint64_t abs = start + pos[i];
        base[(*base_index)++] = (uint32_t)(abs - last);
        last = abs;
    }
    *carried = (start + 63) - last;
    *position = last;
    return SJHIP_OK;
}
