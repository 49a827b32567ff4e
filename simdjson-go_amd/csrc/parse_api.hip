// parse_api.hip -- whole-parse entry points of the C ABI: sjhip_parse / sjhip_parse_device /
// sjhip_fetch.  Mirrors parseMessage (parse_json_amd64.go:52-127): TrimSpace, stage 1, stage 2,
// and the reference's error precedence (a stage-1 failure is reported even when stage 2 would
// fail as well, :97-105 and :123-126).
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/sjhip.h"
#include "sj_ctx.h"
#include "sj_device.h"
#include "sj_host.h"

using namespace sj;

#define HIPCHK(call, what)                                        \
    do {                                                          \
        hipError_t e_ = (call);                                   \
        if (e_ != hipSuccess) return ctx_hip_fail(ctx, e_, what); \
    } while (0)

// Phase 1 (parse_begin): stage 1, then stage 2 up to the device-wide scans; a shard reads back the sizes.
// Phase 2 (parse_finish): the rest of stage 2 with the rebasing offsets, then the verdict.
// documents up to this size are parsed with one host synchronisation (SJHIP_SMALL_BYTES overrides; 0 turns it off)
static constexpr size_t PACK_BYTES = (size_t)2 << 20;  // pinned block for the results of small documents (sj_ctx.h h_pack)

static constexpr int PARSE_AGAIN_SYNCHRONOUS = -1000;  // internal: parse_finish -> parse_on_device

static size_t small_document_bytes() {
    static const size_t v = [] {
        const char *e = getenv("SJHIP_SMALL_BYTES");
        return e ? (size_t)strtoull(e, nullptr, 0) : (size_t)4 << 20;
    }();
    return v;
}

static S2Args s2_args(sjhip_ctx *ctx, uint64_t tape_base, uint64_t strings_base, uint64_t msg_base) {
    S2Args a = {};
    a.d_msg = ctx->p_msg;
    a.len = ctx->p_len;
    a.d_pos = (const uint32_t *)ctx->d_pos.p;
    a.d_kind = ctx->p_kind;
    a.n = ctx->p_nlay;
    a.n_dev = ctx->p_deferred ? (const unsigned long long *)((const char *)ctx->d_ws.p + offsetof(Stage1State, total)) : nullptr;
    a.flags = ctx->p_flags;
    // (stage 1 of this parse ran in the context's stage-1 workspace, whose first line is its state)
    a.s1_has_starter = ctx->p_aux ? (const uint32_t *)((const char *)ctx->d_ws.p + offsetof(Stage1State, c) + ctx->s1_par * sizeof(Stage1Ctrl) + offsetof(Stage1Ctrl, has_starter)) : nullptr;
    a.ws_zero = ctx->d_s2z.p;
    a.ws = ctx->d_s2.p;
    a.d_tape = (uint64_t *)ctx->d_tape.p;
    a.tape_cap = 2 * ctx->p_nlay + 2;
    a.d_strings = (uint8_t *)ctx->d_strings.p;
    a.strings_cap = ctx->p_len + 64;
    a.tape_base = tape_base;
    a.strings_base = strings_base;
    a.msg_base = msg_base;
    a.str_aux = ctx->p_aux;
    a.d_keyflag = (ctx->p_flags & SJHIP_FLAG_KEY_FLAGS) ? (uint8_t *)ctx->d_keyflag.p : nullptr;
    a.stream = ctx->stream;
    return a;
}

// An ND message of more than 4 GiB - 128 bytes is cut into shards of SJHIP_ND_SHARD_BYTES (default 1 GiB) at record boundaries --
// SJHIP_ND_LIMIT_BYTES moves the threshold (tests: the sharded path on documents of a few megabytes).  A single document does
// not shard; since round 5 it may be longer than 4 GiB all the same (the reference's index stream is deltas for exactly that,
// README.md:567-569): the 32-bit positions of stage 1 wrap, and every 4096-token tile of the token kernels rebuilds its true
// offsets from the unit its first token lies in (stage2.hip k_s2_emit_planes).  What stays 32 bits wide: the number of tokens,
// the tape length in words, the length of Strings.B, the distance between two neighbouring tokens -- a document beyond one of
// those is SJHIP_ERR_TOOBIG / a failed parse, as is a document beyond SINGLE_LIMIT.
static constexpr size_t ND_LIMIT = 0xffffffc0ull - 64;
static constexpr size_t TOKEN_LIMIT = 0xfffffff0ull;   // tokens (structural indexes) of one context's parse: 32-bit token indexes in stage 2
static constexpr size_t SINGLE_LIMIT = (size_t)1 << 38;  // 256 GiB: the positions' workspace alone is 5 bytes per message byte
static size_t env_bytes(const char *name, size_t dflt) {
    const char *e = getenv(name);
    const size_t v = e ? (size_t)strtoull(e, nullptr, 0) : 0;
    return v ? v : dflt;
}
static bool nd_too_big(size_t len, uint32_t flags) {
    if (!(flags & SJHIP_FLAG_NDJSON)) return false;
    return len > (len > (1u << 20) ? env_bytes("SJHIP_ND_LIMIT_BYTES", ND_LIMIT) : ND_LIMIT);
}
static size_t nd_shard_bytes() {
    const size_t v = env_bytes("SJHIP_ND_SHARD_BYTES", (size_t)1 << 30);
    return v < ND_LIMIT ? v : ND_LIMIT;
}

static int parse_begin(sjhip_ctx *ctx, const void *d_msg, size_t len, uint32_t flags, uint8_t last_byte, int have_last,
                       size_t *tape_len, size_t *strings_len) {
    ctx->tape_len = ctx->strings_len = 0;
    ctx->big_valid = 0;
    ctx->pending = 0;
    ctx->pack_valid = 0;
    ctx->q_valid = 0;
    ctx->r_valid = 0;
    ctx->kf_valid = 0;
    ctx->ser_valid = 0;
    ctx->ms_valid = 0;
    ctx->f_valid = 0;
    if (len == 0) return SJHIP_ERR_STAGE1;  // indexTotal == 0 (stage1_find_marks_amd64.go:147)
    if (len > SINGLE_LIMIT) {  // (an ND message beyond 4 GiB went to parse_nd_big)
        ctx_set_error(ctx, "document of %zu bytes: one context parses up to %zu", len, SINGLE_LIMIT);
        return SJHIP_ERR_TOOBIG;
    }
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    // One position (4 B) and one kind (1 B) per message byte is the worst case (every byte a structural): 5 bytes per
    // input byte of a grow-only, recycled arena -- 1.3 GB for configs[1] on a 288 GB device -- instead of guessing a
    // density and running stage 1 a second time on documents denser than the guess ("[[[[..." / "[1,1,1,...").
    const size_t pos_cap = (len + 63) / 64 * 64 + 64;
    size_t n = 0;
    int ok = 0;
    // (WithCopyStrings(false): stage 1 also says whether the message holds an escape starter at all, stage2.hip no_escapes)
    const int s1_mode = ((flags & SJHIP_FLAG_NDJSON) ? 1 : 0) | ((flags & SJHIP_FLAG_COPY_STRINGS) ? 0 : S1_WANT_STARTER_FLAG);
    // stage 1 leaves its string masks for the byte-parallel unescape.  Every string copied (the reference's default):
    // Strings.B is the compaction of all string bytes; WithCopyStrings(false): the compaction of the bytes of the strings
    // that hold an escape (stage2.hip k_str_emit) -- in both modes written once, in place
    void *aux = nullptr;
    {
        int rc = arena_reserve(ctx, ctx->d_aux, str_aux_bytes(len + 64));
        if (rc) return rc;
        aux = ctx->d_aux.p;
    }
    {
        // the state and the scan slots of stage 2: zeroed by stage 1's preparation kernel (no memset launch of their own)
        int rc = arena_reserve(ctx, ctx->d_s2z, stage2_zero_bytes());
        if (rc) return rc;
        // positions, then the token kinds stage 1 writes next to them (1 byte each, 256-byte aligned)
        rc = arena_reserve(ctx, ctx->d_pos, (pos_cap + 64) * (sizeof(uint32_t) + 1));
        if (rc) return rc;
        ctx->p_kind = (uint8_t *)ctx->d_pos.p + (pos_cap + 64) * sizeof(uint32_t);
        // A small document is not worth a host round trip in the middle: stage 2 is queued behind stage 1 with arenas
        // and grids sized for the upper bound (a token is at least one byte), the kernels read the token count on the
        // device, and stage 1's verdict is taken after the one synchronisation at the end.  (A shard needs its sizes first.)
        // ... and round 5: neither is a LARGE document once the context has seen the token density of its kind -- the layout
        // is then the density of the context's last parse plus a quarter (a benchmark loop, the blocks of a stream and a
        // pool's contexts parse the same kind of data again and again); the first large parse of a context, and a document
        // denser than the one before it, take the synchronous path.  configs[1] 0.531 -> 0.515 ms, configs[4] 0.849 -> 0.840 ms
        // (same box): the host round trip between stage 1 and stage 2 and the launch gap behind it.
        const bool small = len <= small_document_bytes();
        const bool known = ctx->p_density_q != 0 && small_document_bytes() != 0;
        // (round 6: a shard's phase 1 as well -- its sizes need one synchronisation, not a second one in front for stage 1's count)
        ctx->p_deferred = (small || known) && !ctx->p_no_defer;
        ctx->p_collected = 0;
        if (ctx->p_deferred && !small) {
            rc = stage1_enqueue(ctx, d_msg, len, s1_mode, ctx->d_pos.p, pos_cap, aux, ctx->p_kind,
                                ctx->d_s2z.p, stage2_zero_bytes());
            // tokens per KiB, + 1/16 -- and, where the context's arenas already hold more (the same document again: exactly its
            // count), as much as they hold: a layout that needs no allocation and fails for as few documents as possible
            const size_t est = (size_t)(((unsigned __int128)len * ctx->p_density_q * 17 / 16) >> 10) + 4096;
            size_t fit = ctx->d_tape.cap / 16 > 2 ? ctx->d_tape.cap / 16 - 2 : 0;  // (2 n + 2 tape words)
            if ((flags & SJHIP_FLAG_KEY_FLAGS) && ctx->d_keyflag.cap < fit + 9) fit = ctx->d_keyflag.cap > 9 ? ctx->d_keyflag.cap - 9 : 0;
            if (fit > 0 && stage2_workspace_bytes(fit) > ctx->d_s2.cap) {  // the largest n whose workspace fits
                size_t lo = 0, hi = fit;
                while (hi - lo > 1) {
                    const size_t mid = lo + (hi - lo) / 2;
                    if (stage2_workspace_bytes(mid) <= ctx->d_s2.cap) lo = mid;
                    else hi = mid;
                }
                fit = lo;
            }
            if (fit > 4 * est) fit = 4 * est;  // (grids are sized for the layout: not for a much larger document of the past)
            n = est > fit ? est : fit;
            if (n > len) n = len;
            ok = 1;
            ctx->p_last = last_byte;
            ctx->p_have_last = have_last;
        } else if (ctx->p_deferred) {
            rc = stage1_enqueue(ctx, d_msg, len, s1_mode, ctx->d_pos.p, pos_cap, aux, ctx->p_kind,
                                ctx->d_s2z.p, stage2_zero_bytes());
            // The stage-2 arrays are laid out for one token per four bytes (the densest fixture, marine_ik, has 0.22):
            // ~14 B of arena per message byte instead of 57.  The kernels clamp the device-side count to this layout
            // (stage2.hip token_count), so a denser document stays in bounds and is parsed again the synchronous way
            // (parse_on_device) once stage 1's real count is known.
            n = (!ctx->p_dense && len / 4 + 4096 < len) ? len / 4 + 4096 : len;
            ok = 1;
            ctx->p_last = last_byte;
            ctx->p_have_last = have_last;
        } else {
            rc = stage1_run_device(ctx, d_msg, len, s1_mode, ctx->d_pos.p, pos_cap, last_byte,
                                   have_last, &n, &ok, aux, ctx->p_kind, ctx->d_s2z.p, stage2_zero_bytes());
        }
        if (rc) return rc;
    }
    if (!ok) return SJHIP_ERR_STAGE1;
    if (n >= TOKEN_LIMIT) {  // (stage 1 counts in 40 bits; the token kernels index tokens with 32)
        ctx_set_error(ctx, "document of %zu tokens: one context parses fewer than 2^32", n);
        return SJHIP_ERR_TOOBIG;
    }
    int rc = arena_reserve(ctx, ctx->d_s2, stage2_workspace_bytes(n));
    if (rc) return rc;
    rc = arena_reserve(ctx, ctx->d_tape, (2 * n + 2) * sizeof(uint64_t));
    if (rc) return rc;
    rc = arena_reserve(ctx, ctx->d_strings, len + 64);
    if (rc) return rc;
    if (flags & SJHIP_FLAG_KEY_FLAGS) {
        rc = arena_reserve(ctx, ctx->d_keyflag, n + 1 + 8);  // tape_cap / 2 + 8
        if (rc) return rc;
    }
    ctx->p_aux = aux;
    ctx->pending = 1;
    ctx->p_msg = d_msg;
    ctx->p_len = len;
    ctx->p_n = n;
    ctx->p_nlay = n;
    ctx->p_flags = flags;
    HIPCHK(stage2_launch_measure(s2_args(ctx, 0, 0, 0)), "stage2 launch (measure)");
    if (tape_len || strings_len) {  // only the sharded path needs the sizes before phase 2
        S2State *hs = (S2State *)(ctx->h_scratch + 512);
        HIPCHK(hipMemcpyAsync(hs, ctx->d_s2z.p, sizeof(S2State), hipMemcpyDeviceToHost, ctx->stream), "D2H stage2 sizes");
        HIPCHK(hipStreamSynchronize(ctx->stream), "stage2 sizes sync");
        if (ctx->p_deferred) {  // stage 1's verdict and count arrived with the same synchronisation (its last block wrote them to pinned memory)
            size_t n1 = 0;
            int ok1 = 0;
            rc = stage1_collect(ctx, len, last_byte, have_last, &n1, &ok1);
            if (rc) return rc;
            if (!ok1) return SJHIP_ERR_STAGE1;  // (first, as in parseMessage)
            if (n1 >= TOKEN_LIMIT) {
                ctx_set_error(ctx, "document of %zu tokens: one context parses fewer than 2^32", n1);
                return SJHIP_ERR_TOOBIG;
            }
            if (n1 > ctx->p_nlay) {  // denser than the layout assumed: nothing of this run counts -- again, the synchronous way
                ctx->p_no_defer = 1;
                if (len <= small_document_bytes()) ctx->p_dense = 1;
                rc = parse_begin(ctx, d_msg, len, flags, last_byte, have_last, tape_len, strings_len);
                ctx->p_no_defer = 0;
                return rc;
            }
            ctx->p_n = n1;
            ctx->p_collected = 1;
        }
        if (hs->err & S2_ERR_SERIAL_STRINGS) {  // pathological surrogate run: measure again with the per-string walks
            if (len > ND_LIMIT) {
                ctx_set_error(ctx, "the per-string path (a surrogate run or a string beyond the byte-parallel path) works on documents of up to 4 GiB");
                return SJHIP_ERR_TOOBIG;
            }
            ctx->p_aux = nullptr;
            HIPCHK(hipMemsetAsync(ctx->d_s2z.p, 0, stage2_zero_bytes(), ctx->stream), "stage2 state reset");
            HIPCHK(stage2_launch_measure(s2_args(ctx, 0, 0, 0)), "stage2 launch (measure, per-string)");
            HIPCHK(hipMemcpyAsync(hs, ctx->d_s2z.p, sizeof(S2State), hipMemcpyDeviceToHost, ctx->stream), "D2H stage2 sizes");
            HIPCHK(hipStreamSynchronize(ctx->stream), "stage2 sizes sync");
        }
        if (tape_len) *tape_len = (size_t)hs->tape_len;
        if (strings_len) *strings_len = ctx->p_aux ? (size_t)hs->strings_len_masks : (size_t)hs->strings_len;
    }
    return SJHIP_OK;
}

static int parse_finish(sjhip_ctx *ctx, uint64_t tape_base, uint64_t strings_base, uint64_t msg_base, size_t *tape_len,
                        size_t *strings_len) {
    if (!ctx->pending) {
        ctx_set_error(ctx, "no parse in progress");
        return SJHIP_ERR_ARG;
    }
    ctx->pending = 0;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    S2State *hs = (S2State *)(ctx->h_scratch + 256);
    // a small document that came from a host buffer: the state, the tape and Strings.B go to pinned memory with the last
    // launch of the chain (sj_ctx.h h_pack); sjhip_fetch then copies from there
    const bool pack = ctx->want_pack && ctx->p_deferred && tape_base == 0 && strings_base == 0 && msg_base == 0;
    ctx->want_pack = 0;
    ctx->pack_valid = 0;
    if (pack && !ctx->h_pack && sj::pinned_alloc((void **)&ctx->h_pack, PACK_BYTES) != hipSuccess) {
        (void)hipGetLastError();
        ctx->h_pack = nullptr;
    }
    auto run = [&]() -> int {
        const S2Args a = s2_args(ctx, tape_base, strings_base, msg_base);
        HIPCHK(stage2_launch_emit(a), "stage2 launch (emit)");
        if (pack && ctx->h_pack) {
            HIPCHK(stage2_launch_pack(a, ctx->h_pack, PACK_BYTES), "stage2 launch (pack)");
            HIPCHK(hipStreamSynchronize(ctx->stream), "stage2 sync");
            memcpy(hs, ctx->h_pack, sizeof(S2State));
            ctx->pack_valid = *(const unsigned long long *)(ctx->h_pack + 64) != 0;
        } else {
            HIPCHK(stage2_launch_state_out(a, hs), "stage2 state");
            HIPCHK(hipStreamSynchronize(ctx->stream), "stage2 sync");
        }
        if (hs->bignum_count && !(hs->err & S2_ERR_SERIAL_STRINGS)) {  // rare: >19-digit mantissas that need the exact tie-break
            ctx->pack_valid = 0;  // (k_pack did not copy: the tape is not final)
            HIPCHK(stage2_launch_bignum(a), "stage2 launch (bignum)");
            HIPCHK(hipMemcpyAsync(hs, ctx->d_s2z.p, sizeof(S2State), hipMemcpyDeviceToHost, ctx->stream), "D2H stage2 state");
            HIPCHK(hipStreamSynchronize(ctx->stream), "stage2 sync");
        }
        return SJHIP_OK;
    };
    int rc = run();
    if (rc) return rc;
    {  // debug build: an out-of-bounds access anywhere in the stage-2 kernels fails the parse, whatever its verdict
        unsigned hits = 0, id = 0;
        unsigned long long index = 0, size = 0;
        if (stage2_debug_bounds(&hits, &id, &index, &size) && hits) {
            ctx_set_error(ctx, "bounds check: %u out-of-bounds accesses, the first to array %u (sj_bounds.h ArrId) at element %llu of %llu",
                          hits, id, index, size);
            return SJHIP_ERR_HIP;
        }
    }
    if (ctx->p_deferred && !ctx->p_collected) {  // stage 1's verdict first, as in parseMessage (parse_json_amd64.go:97-105,123-126)
        size_t n = 0;
        int ok = 0;
        rc = stage1_collect(ctx, ctx->p_len, ctx->p_last, ctx->p_have_last, &n, &ok);
        if (rc) return rc;
        if (!ok) return SJHIP_ERR_STAGE1;
        if (n >= TOKEN_LIMIT) {
            ctx_set_error(ctx, "document of %zu tokens: one context parses fewer than 2^32", n);
            return SJHIP_ERR_TOOBIG;
        }
        if (n > ctx->p_nlay) return PARSE_AGAIN_SYNCHRONOUS;  // denser than the layout assumed: nothing of this run counts
        ctx->p_n = n;  // (the arrays stay laid out for p_nlay)
    }
    if ((hs->err & S2_ERR_SERIAL_STRINGS) && ctx->p_aux) {
        // a run of > SURROGATE_WALK_CAP adjacent high-surrogate escapes (sj_strings.h): the byte-parallel string path
        // gave up, nothing of this run is a verdict.  Stage 2 again with the per-string walks (linear in the run).
        if (ctx->p_len > ND_LIMIT) {
            ctx_set_error(ctx, "the per-string path (a surrogate run or a string beyond the byte-parallel path) works on documents of up to 4 GiB");
            return SJHIP_ERR_TOOBIG;
        }
        ctx->p_aux = nullptr;
        HIPCHK(hipMemsetAsync(ctx->d_s2z.p, 0, stage2_zero_bytes(), ctx->stream), "stage2 state reset");
        HIPCHK(stage2_launch_measure(s2_args(ctx, tape_base, strings_base, msg_base)), "stage2 launch (measure, per-string)");
        rc = run();
        if (rc) return rc;
    }
    if (hs->err & 8u) {  // a bounded spin loop of a scan kernel ran out: internal error, never a verdict
        ctx_set_error(ctx, "stage-2 scan aborted (internal synchronisation timeout)");
        return SJHIP_ERR_HIP;
    }
    if (hs->err & 4u) {
        ctx_set_error(ctx, "tape longer than 2^32 words, Strings.B longer than 4 GiB, or a 4096-token tile that spans 4 GiB of the message");
        return SJHIP_ERR_TOOBIG;
    }
    if (hs->err) return SJHIP_ERR_STAGE2;
    // the token density of this parse (tokens per KiB, rounded up): what the next large parse of the context is laid out for
    // (only LARGE parses leave one -- a small sparse document in between must not send the next large dense one through a
    // layout that fails and a second run of both stages -- and a sparser large document lowers it only half way: the price of
    // too large a layout is a few empty blocks per grid, the price of too small a one is the whole parse twice)
    if (ctx->p_len > small_document_bytes()) {
        const uint32_t dq = (uint32_t)(((unsigned __int128)ctx->p_n * 1024 + ctx->p_len - 1) / ctx->p_len) + 1;
        ctx->p_density_q = dq >= ctx->p_density_q ? dq : (uint32_t)(((uint64_t)dq + ctx->p_density_q + 1) / 2);
    }
    ctx->tape_len = (size_t)hs->tape_len;
    ctx->strings_len = ctx->p_aux ? (size_t)hs->strings_len_masks : (size_t)hs->strings_len;
    ctx->q_records = hs->records;
    ctx->q_valid = tape_base == 0 && strings_base == 0 && msg_base == 0;  // filter / serializer / MarshalJSON work on unsharded results
    ctx->r_valid = 1;  // the path / count queries also on a shard (in the merged index space)
    ctx->r_tape_base = tape_base;
    ctx->r_strings_base = strings_base;
    ctx->r_msg_base = msg_base;
    ctx->kf_valid = (ctx->p_flags & SJHIP_FLAG_KEY_FLAGS) && ctx->d_keyflag.p;  // (indexed by the context's own tape offsets: a shard's as well)
    if (tape_len) *tape_len = ctx->tape_len;
    if (strings_len) *strings_len = ctx->strings_len;
    return SJHIP_OK;
}

static int parse_on_device(sjhip_ctx *ctx, const void *d_msg, size_t len, uint32_t flags, uint8_t last_byte,
                           int have_last, size_t *tape_len, size_t *strings_len) {
    const int want_pack = ctx->want_pack;
    int rc = parse_begin(ctx, d_msg, len, flags, last_byte, have_last, nullptr, nullptr);
    if (rc) return rc;
    rc = parse_finish(ctx, 0, 0, 0, tape_len, strings_len);
    if (rc == PARSE_AGAIN_SYNCHRONOUS) {  // a document denser than its layout: a small one with more than one token per four bytes,
                                          // a large one denser than the context's last parse (this parse leaves the new density)
        ctx->p_no_defer = 1;
        if (len <= small_document_bytes()) ctx->p_dense = 1;  // (minified numeric arrays come in series: the next ones are laid out for one token per byte)
        ctx->want_pack = want_pack;
        rc = parse_begin(ctx, d_msg, len, flags, last_byte, have_last, nullptr, nullptr);
        ctx->p_no_defer = 0;
        if (rc) return rc;
        rc = parse_finish(ctx, 0, 0, 0, tape_len, strings_len);
    }
    return rc;
}

// batch_api.hip: the whole parse of the message it packed into the context's message arena
int sj::parse_packed(sjhip_ctx *ctx, size_t len, uint32_t flags, uint8_t last_byte, int have_last, size_t *tape_len,
                     size_t *strings_len) {
    return parse_on_device(ctx, ctx->d_msg.p, len, flags, last_byte, have_last, tape_len, strings_len);
}

int sjhip_parse_shard_begin(sjhip_ctx *ctx, const void *d_msg, size_t len, uint32_t flags, size_t *tape_len,
                            size_t *strings_len) {
    if (!ctx || !tape_len || !strings_len) return SJHIP_ERR_ARG;
    *tape_len = *strings_len = 0;
    return parse_begin(ctx, d_msg, len, flags, 0, 0, tape_len, strings_len);
}

int sjhip_parse_shard_finish(sjhip_ctx *ctx, uint64_t tape_base, uint64_t strings_base, uint64_t msg_base) {
    if (!ctx) return SJHIP_ERR_ARG;
    return parse_finish(ctx, tape_base, strings_base, msg_base, nullptr, nullptr);
}

int sjhip_parse_device(sjhip_ctx *ctx, const void *d_msg, size_t len, uint32_t flags, size_t *tape_len,
                       size_t *strings_len) {
    if (!ctx) return SJHIP_ERR_ARG;
    if (nd_too_big(len, flags)) {  // shards on this device, each parsing its window of the message in place
        ctx->tape_len = ctx->strings_len = 0;
        ctx->q_valid = ctx->r_valid = ctx->ser_valid = ctx->ms_valid = ctx->f_valid = ctx->pack_valid = ctx->pending = 0;
        HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
        HIPCHK(hipStreamSynchronize(ctx->stream), "stream sync");  // (the shards run on streams of their own)
        return parse_nd_big(ctx, (const uint8_t *)d_msg, len, flags, true, nd_shard_bytes(), tape_len, strings_len, nullptr, nullptr);
    }
    return parse_on_device(ctx, d_msg, len, flags, 0, 0, tape_len, strings_len);
}

int sjhip_parse(sjhip_ctx *ctx, const uint8_t *msg, size_t len, uint32_t flags, size_t *tape_len, size_t *strings_len,
                size_t *msg_off, size_t *msg_len) {
    if (!ctx) return SJHIP_ERR_ARG;
    size_t off = 0, mlen = 0;
    if (len) trim_space(msg, len, &off, &mlen);  // pj.Message = bytes.TrimSpace(msg), parse_json_amd64.go:55
    if (msg_off) *msg_off = off;
    if (msg_len) *msg_len = mlen;
    if (tape_len) *tape_len = 0;
    if (strings_len) *strings_len = 0;
    ctx->tape_len = ctx->strings_len = 0;
    ctx->big_valid = 0;
    if (mlen == 0) return SJHIP_ERR_STAGE1;
    if (nd_too_big(mlen, flags)) {  // shards of the host message, H2D straight from the caller's buffer
        ctx->q_valid = ctx->r_valid = ctx->ser_valid = ctx->ms_valid = ctx->f_valid = ctx->pack_valid = ctx->pending = 0;
        return parse_nd_big(ctx, msg, len, flags, false, nd_shard_bytes(), tape_len, strings_len, nullptr, nullptr);
    }
    if (mlen > SINGLE_LIMIT) {  // before anything is copied to the device
        ctx_set_error(ctx, "document of %zu bytes: one context parses up to %zu", mlen, SINGLE_LIMIT);
        return SJHIP_ERR_TOOBIG;
    }
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    int rc = arena_reserve(ctx, ctx->d_msg, mlen + 128);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(ctx->d_msg.p, msg + off, mlen, hipMemcpyHostToDevice, ctx->stream), "H2D message");
    ctx->want_pack = 1;  // the caller holds host buffers: its sjhip_fetch will want the result there
    rc = parse_on_device(ctx, ctx->d_msg.p, mlen, flags, msg[off + mlen - 1], 1, tape_len, strings_len);
    ctx->want_pack = 0;
    return rc;
}

int sjhip_debug_bounds_selftest(void) { return stage2_debug_bounds_selftest(); }

void sjhip_trim_space(const uint8_t *msg, size_t len, size_t *off, size_t *out_len) {
    size_t o = 0, l = 0;
    if (len) trim_space(msg, len, &o, &l);
    if (off) *off = o;
    if (out_len) *out_len = l;
}

int sjhip_fetch(sjhip_ctx *ctx, uint64_t *tape_dst, uint8_t *strings_dst) {
    if (!ctx) return SJHIP_ERR_ARG;
    if (ctx->big_valid) return fetch_nd_big(ctx, tape_dst, strings_dst);  // every shard straight into its slice
    if (ctx->pack_valid && ctx->h_pack) {  // the result of a small sjhip_parse is already in pinned host memory
        if (ctx->tape_len && tape_dst) memcpy(tape_dst, ctx->h_pack + STAGE2_PACK_HEAD, ctx->tape_len * sizeof(uint64_t));
        if (ctx->strings_len && strings_dst)
            memcpy(strings_dst, ctx->h_pack + STAGE2_PACK_HEAD + ctx->tape_len * sizeof(uint64_t), ctx->strings_len);
        return SJHIP_OK;
    }
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    if (ctx->tape_len && tape_dst)
        HIPCHK(hipMemcpyAsync(tape_dst, ctx->d_tape.p, ctx->tape_len * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream),
               "D2H tape");
    if (ctx->strings_len && strings_dst)
        HIPCHK(hipMemcpyAsync(strings_dst, ctx->d_strings.p, ctx->strings_len, hipMemcpyDeviceToHost, ctx->stream),
               "D2H strings");
    HIPCHK(hipStreamSynchronize(ctx->stream), "fetch sync");
    return SJHIP_OK;
}

static size_t view_limit_bytes() {
    static const size_t v = [] {
        const char *e = getenv("SJHIP_VIEW_LIMIT_BYTES");
        return e ? (size_t)strtoull(e, nullptr, 0) : (size_t)4 << 30;
    }();
    return v;
}

int sjhip_fetch_view(sjhip_ctx *ctx, const uint64_t **tape, const uint8_t **strings) {
    if (!ctx || !tape || !strings) return SJHIP_ERR_ARG;
    *tape = nullptr;
    *strings = nullptr;
    const size_t tb = ctx->tape_len * sizeof(uint64_t), sb = ctx->strings_len;
    if (!ctx->big_valid && ctx->pack_valid && ctx->h_pack) {  // a small sjhip_parse: nothing to move
        if (tb) *tape = (const uint64_t *)(ctx->h_pack + STAGE2_PACK_HEAD);
        if (sb) *strings = ctx->h_pack + STAGE2_PACK_HEAD + tb;
        return SJHIP_OK;
    }
    const size_t need = tb + sb + 64;
    if (need > view_limit_bytes()) {
        ctx_set_error(ctx, "sjhip_fetch_view: a result of %zu bytes is beyond SJHIP_VIEW_LIMIT_BYTES (%zu): use sjhip_fetch", tb + sb,
                      view_limit_bytes());
        return SJHIP_ERR_TOOBIG;
    }
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    if (need > ctx->h_view_cap) {
        if (ctx->h_view) (void)hipHostFree(ctx->h_view);
        ctx->h_view = nullptr;
        ctx->h_view_cap = 0;
        const size_t cap = (need + need / 4 + ((size_t)1 << 20)) & ~(size_t)4095;
        const hipError_t e = sj::pinned_alloc((void **)&ctx->h_view, cap);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            ctx->h_view = nullptr;
            ctx_set_error(ctx, "hipHostMalloc of %zu bytes for the result view: %s", cap, hipGetErrorString(e));
            return SJHIP_ERR_HIP;
        }
        ctx->h_view_cap = cap;
    }
    uint64_t *const t = (uint64_t *)ctx->h_view;
    uint8_t *const st = ctx->h_view + tb;
    const int rc = sjhip_fetch(ctx, tb ? t : nullptr, sb ? st : nullptr);  // (sharded big ND results included)
    if (rc) return rc;
    if (tb) *tape = t;
    if (sb) *strings = st;
    return SJHIP_OK;
}
