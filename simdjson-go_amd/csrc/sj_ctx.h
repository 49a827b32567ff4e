// sj_ctx.h -- host-side context behind the opaque sjhip_ctx of include/sjhip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "sj_device.h"

namespace sj {
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    unsigned gen = 0;  // counts the allocations behind p (arena_reserve, sjhip_ctx_trim): "is this still the memory I prepared?"
};
}  // namespace sj

struct sjhip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;      // stream all work is queued on
    hipStream_t own_stream = nullptr;  // created with the context
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint8_t *h_scratch = nullptr;      // 4 KiB pinned: state read-backs
    // small documents parsed from a host buffer (sjhip_parse): the last kernel of the chain also writes the stage-2 state,
    // the tape and Strings.B into this pinned block (over PCIe, no copy commands), and sjhip_fetch is two memcpy
    uint8_t *h_pack = nullptr;
    int want_pack = 0;                 // set by sjhip_parse for the parse it starts
    int pack_valid = 0;                // h_pack holds the result of the last parse
    uint8_t *h_view = nullptr;         // sjhip_fetch_view of results that did not travel with the last launch (grows)
    size_t h_view_cap = 0;
    uint8_t *h_in = nullptr;           // sjhip_input_block: pinned block the caller reads its input into
    size_t h_in_cap = 0;
    uint8_t *h_stage = nullptr;        // pinned staging of sjhip_parse_batch: runs of small documents travel as one copy
    size_t h_stage_cap = 0;
    sj::DevBuf d_msg, d_pos, d_ws, d_kat, d_tape, d_strings, d_s2, d_s2z, d_aux;
    sj::DevBuf d_keyflag;              // SJHIP_FLAG_KEY_FLAGS: key flags of the string entries, for marshal.hip
    int kf_valid = 0;                  // ... and they belong to the resident result
    sj::DevBuf d_scol, d_stab;         // serializer with de-duplication: the string column, the hash table
    sj::DevBuf d_q, d_qtape, d_qstrings;  // queries over the last result (query.hip): work arrays, filtered tape / Strings.B
    unsigned ws_clean_gen = 0;         // d_ws.gen of the allocation that has been zeroed for stage 1 (0: none; stage1_enqueue)
    sj::S1Ws s1ws;                     // ... and its launch count (sj_device.h: a launch cleans up for the next one)
    unsigned s1_par = 0;               // control slot of the last stage-1 launch (Stage1State::c[]: stage 2 reads has_starter there)
    sj::Stage1State s1;                // last stage-1 state (host copy)
    // last parse (kept on the device until sjhip_fetch)
    size_t tape_len = 0, strings_len = 0;
    int q_valid = 0;              // the device holds the whole result of an unsharded parse: queries are possible
    int r_valid = 0;              // the device holds the result of a parse -- whole, or one shard of a sharded ParseND whose stored
    uint64_t r_tape_base = 0, r_strings_base = 0, r_msg_base = 0;  // indices carry these bases (query.hip works in the merged index space)
    uint32_t q_records = 0;       // record-separating newline runs of that parse (records - 1)
    size_t q_tape_len = 0, q_strings_len = 0;  // last sjhip_filter_where
    int f_valid = 0;              // a filtered result is resident (sjhip_fetch_filtered)
    int ser_valid = 0;            // last sjhip_serialize (serialize.hip): column sizes, framed stream size
    size_t ser_tags = 0, ser_vals = 0, ser_rest = 0, ser_stream = 0, ser_slen = 0;
    int ser_dedup = 0;
    size_t des_msg_len = 0;       // last sjhip_deserialize: length of pj.Message (in d_msg)
    int ms_valid = 0;             // last sjhip_marshal_json (marshal.hip): the text is in d_qtape
    size_t ms_len = 0;
    // a parse between its two phases (sjhip_parse_shard_begin / _finish)
    int pending = 0;
    int p_deferred = 0;   // stage 1's result has not been collected yet (small documents: one synchronisation per parse)
    int p_collected = 0;  // ... but a shard's phase 1 has: its sizes and stage 1's verdict arrive with one synchronisation (parse_begin)
    int p_no_defer = 0;   // the deferred run met more tokens than its layout holds: this parse takes the synchronous path
    uint32_t p_density_q = 0;  // tokens per KiB of the context's last successful parse (+1), 0: none yet -- a large document is
                               // then parsed without the host round trip between the stages, laid out for that density + 1/16 (or what the arenas hold)
    int p_dense = 0;      // sticky: a document of this context was denser than one token per four bytes -- later deferred
                          // parses are laid out for one token per byte instead of paying the second parse again
    uint8_t p_last = 0;   // ... its caller-supplied last byte
    int p_have_last = 0;
    size_t p_nlay = 0;    // the token count the stage-2 arrays were laid out for (>= p_n)
    const void *p_msg = nullptr;
    size_t p_len = 0, p_n = 0;
    uint32_t p_flags = 0;
    void *p_aux = nullptr;
    uint8_t *p_kind = nullptr;  // token kinds of the pending parse (behind the positions in d_pos)
    // an ND message beyond 4 GiB - 64 is parsed shard by shard by a multi handle the context owns (multi_api.hip
    // parse_nd_big); the merged result waits there for sjhip_fetch
    struct sjhip_multi *big = nullptr;
    int big_valid = 0;
    char err[256];
};

namespace sj {
// Pinned host memory of the library: portable (every device context may use it, not only the one that was current when
// it was allocated -- streams and multi handles own contexts on several devices) and mapped (kernels of any of them
// write results straight into it: the stage-1 verdict word, k_pack, the batch end check).
inline hipError_t pinned_alloc(void **p, size_t bytes) { return hipHostMalloc(p, bytes, hipHostMallocPortable | hipHostMallocMapped); }
void ctx_set_error(sjhip_ctx *ctx, const char *fmt, ...);
int ctx_hip_fail(sjhip_ctx *ctx, hipError_t e, const char *what);
int arena_reserve(sjhip_ctx *ctx, DevBuf &b, size_t bytes);
int parse_packed(sjhip_ctx *ctx, size_t len, uint32_t flags, uint8_t last_byte, int have_last, size_t *tape_len,
                 size_t *strings_len);  // parse_api.hip
int parse_nd_big(sjhip_ctx *ctx, const uint8_t *msg, size_t len, uint32_t flags, bool d_resident, size_t shard_bytes,
                 size_t *tape_len, size_t *strings_len, size_t *msg_off, size_t *msg_len);  // multi_api.hip
int fetch_nd_big(sjhip_ctx *ctx, uint64_t *tape_dst, uint8_t *strings_dst);
void release_nd_big(sjhip_ctx *ctx);
size_t nd_big_device_bytes(const sjhip_ctx *ctx);  // arenas of the shard contexts of a sharded ND parse
// the shards of the merged result of parse_nd_big, in document order (empty shards have no context to look at: null)
int nd_big_shards(const sjhip_ctx *ctx);
sjhip_ctx *nd_big_shard(const sjhip_ctx *ctx, int k);
int stage1_enqueue(sjhip_ctx *ctx, const void *d_msg, size_t len, int ndjson, void *d_pos, size_t pos_cap, void *str_aux,
                   uint8_t *d_kind, void *zero2, size_t zero2_bytes, unsigned long long *host_rec = nullptr);
// host_rec: the launch's record in pinned host memory (S1_HOST_WORDS words; null: the context's own at h_scratch)
int stage1_collect(sjhip_ctx *ctx, size_t len, uint8_t last_byte, int have_last, size_t *n, int *ok,
                   const unsigned long long *host_rec = nullptr);
int stage1_run_device(sjhip_ctx *ctx, const void *d_msg, size_t len, int ndjson, void *d_pos, size_t pos_cap,
                      uint8_t last_byte, int have_last, size_t *n, int *ok, void *str_aux = nullptr, uint8_t *d_kind = nullptr,
                      void *zero2 = nullptr, size_t zero2_bytes = 0);
}  // namespace sj
