// multi_api.hip -- ParseND over several GPUs inside the library: sjhip_multi_*.
//
// The reference's ParseND is one call in one process (simdjson_amd64.go:82-94); its records are independent and
// ParseNDStream already parses them block by block (:156-192).  Here the message is cut at record boundaries into one
// shard per device, every shard runs the two-phase shard parse of parse_api.hip on its own context (own device, own
// HIP stream, one host thread each), and the merged ParsedJson is the concatenation of the shard tapes / Strings.B:
//   phase 1 (all shards in parallel)  H2D of the shard, stage 1, stage 2 up to the scans -> (tape_len, strings_len)
//   host                              exclusive prefix sums of the sizes: 16 bytes per shard, no device collective
//   phase 2 (all shards in parallel)  tape words with tape / Strings.B / Message indices rebased by the shard's bases
//   fetch                             every shard copies its piece straight into its slice of the caller's buffers
// Error precedence as in parseMessage (parse_json_amd64.go:97-105,123-126): a stage-1 failure of any shard wins over
// stage-2 failures.  The multi-process form of the same path (one rank per GPU, sizes exchanged by an RCCL all_gather)
// is sjhip/ndshard.py; both produce the tape of the whole document bit for bit.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <thread>
#include <vector>

#include "../../include/sjhip.h"
#include "sj_ctx.h"
#include "sj_host.h"

namespace {
struct Shard {
    sjhip_ctx *ctx = nullptr;
    int device = 0;
    size_t start = 0, len = 0;  // window of the (trimmed) shard inside the caller's message
    size_t tape_len = 0, strings_len = 0;
    size_t tape_base = 0, strings_base = 0;
    int rc = 0;
};
}  // namespace

struct sjhip_multi {
    std::vector<Shard> shards;
    size_t tape_len = 0, strings_len = 0;
    int valid = 0;
    char err[256] = {0};
};

sjhip_multi *sjhip_multi_create(const int *devices, int n) {
    const int have = sjhip_device_count();
    if (have <= 0) return nullptr;
    std::vector<int> devs;
    if (!devices || n <= 0) {
        for (int d = 0; d < have; d++) devs.push_back(d);
    } else {
        for (int k = 0; k < n; k++) {
            if (devices[k] < 0 || devices[k] >= have) return nullptr;
            devs.push_back(devices[k]);
        }
    }
    sjhip_multi *m = new sjhip_multi();
    m->shards.resize(devs.size());
    for (size_t k = 0; k < devs.size(); k++) {
        m->shards[k].device = devs[k];
        m->shards[k].ctx = sjhip_ctx_create(devs[k]);
        if (!m->shards[k].ctx) {
            sjhip_multi_destroy(m);
            return nullptr;
        }
    }
    return m;
}

void sjhip_multi_destroy(sjhip_multi *m) {
    if (!m) return;
    for (Shard &s : m->shards)
        if (s.ctx) sjhip_ctx_destroy(s.ctx);
    delete m;
}

int sjhip_multi_shards(const sjhip_multi *m) { return m ? (int)m->shards.size() : 0; }
const char *sjhip_multi_last_error(const sjhip_multi *m) { return m ? m->err : "no handle"; }

// runs f(shard) for every shard on its own host thread (one shard: on the caller's)
template <typename F>
static void for_shards(sjhip_multi *m, F f) {
    if (m->shards.size() == 1) {
        f(m->shards[0]);
        return;
    }
    std::vector<std::thread> th;
    for (Shard &s : m->shards) th.emplace_back([&s, &f] { f(s); });
    for (std::thread &t : th) t.join();
}

// the verdict all shards share: 0, or the code to return (stage 1 first)
static int agree(sjhip_multi *m, const char *phase) {
    int code = 0;
    for (const Shard &s : m->shards)
        if (s.rc == SJHIP_ERR_STAGE1) code = SJHIP_ERR_STAGE1;
    for (size_t k = 0; k < m->shards.size() && code == 0; k++)
        if (m->shards[k].rc != 0) code = m->shards[k].rc;
    if (code != 0 && code != SJHIP_ERR_STAGE1 && code != SJHIP_ERR_STAGE2)
        for (size_t k = 0; k < m->shards.size(); k++)
            if (m->shards[k].rc == code) {
                snprintf(m->err, sizeof m->err, "shard %zu (%s): %s", k, phase, sjhip_last_error(m->shards[k].ctx));
                break;
            }
    return code;
}

int sjhip_parse_nd_multi(sjhip_multi *m, const uint8_t *msg, size_t len, uint32_t flags, size_t *tape_len, size_t *strings_len,
                         size_t *msg_off, size_t *msg_len) {
    if (!m || m->shards.empty()) return SJHIP_ERR_ARG;
    m->valid = 0;
    m->tape_len = m->strings_len = 0;
    size_t g_off = 0, g_len = 0;
    if (len) sj::trim_space(msg, len, &g_off, &g_len);  // pj.Message = bytes.TrimSpace(msg), parse_json_amd64.go:55
    if (msg_off) *msg_off = g_off;
    if (msg_len) *msg_len = g_len;
    if (tape_len) *tape_len = 0;
    if (strings_len) *strings_len = 0;
    if (g_len == 0) return SJHIP_ERR_STAGE1;
    // Record cuts: right behind the first raw newline at or after k * len / n.  A raw newline never lies inside a string
    // of a valid document (stage-1 error, find_quote_mask_and_bits_amd64.s:67-80) and inside a record it is a stage-2
    // error (stage2_build_tape_amd64.go:196-221), so every raw newline of a valid ND document separates records.
    const size_t n = m->shards.size();
    size_t cut = 0;
    for (size_t k = 0; k < n; k++) {
        Shard &s = m->shards[k];
        size_t end = len;
        if (k + 1 < n) {
            size_t target = len * (k + 1) / n;
            if (target < cut) target = cut;
            const void *nl = target < len ? memchr(msg + target, '\n', len - target) : nullptr;
            end = nl ? (size_t)((const uint8_t *)nl - msg) + 1 : len;
        }
        size_t off = 0, ln = 0;
        if (end > cut) sj::trim_space(msg + cut, end - cut, &off, &ln);
        s.start = cut + off;
        s.len = ln;
        s.tape_len = s.strings_len = 0;
        s.rc = 0;
        cut = end;
    }
    const uint32_t fl = flags | SJHIP_FLAG_NDJSON;
    for_shards(m, [&](Shard &s) {  // phase 1
        if (s.len == 0) return;
        sjhip_ctx *ctx = s.ctx;
        if (hipSetDevice(s.device) != hipSuccess) {
            s.rc = SJHIP_ERR_HIP;
            return;
        }
        s.rc = sj::arena_reserve(ctx, ctx->d_msg, s.len + 128);
        if (s.rc) return;
        if (hipMemcpyAsync(ctx->d_msg.p, msg + s.start, s.len, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
            s.rc = SJHIP_ERR_HIP;
            return;
        }
        s.rc = sjhip_parse_shard_begin(ctx, ctx->d_msg.p, s.len, fl, &s.tape_len, &s.strings_len);
    });
    int code = agree(m, "phase 1");
    if (code) return code;
    size_t t = 0, b = 0;
    for (Shard &s : m->shards) {  // the only exchange of the data path
        s.tape_base = t;
        s.strings_base = b;
        t += s.tape_len;
        b += s.strings_len;
    }
    for_shards(m, [&](Shard &s) {  // phase 2
        if (s.len == 0) return;
        if (hipSetDevice(s.device) != hipSuccess) {
            s.rc = SJHIP_ERR_HIP;
            return;
        }
        s.rc = sjhip_parse_shard_finish(s.ctx, s.tape_base, s.strings_base, s.start - g_off);
    });
    code = agree(m, "phase 2");
    if (code) return code;
    m->tape_len = t;
    m->strings_len = b;
    m->valid = 1;
    if (tape_len) *tape_len = t;
    if (strings_len) *strings_len = b;
    return SJHIP_OK;
}

int sjhip_fetch_multi(sjhip_multi *m, uint64_t *tape_dst, uint8_t *strings_dst) {
    if (!m) return SJHIP_ERR_ARG;
    if (!m->valid) {
        snprintf(m->err, sizeof m->err, "no merged result (sjhip_fetch_multi follows a successful sjhip_parse_nd_multi)");
        return SJHIP_ERR_ARG;
    }
    for_shards(m, [&](Shard &s) {
        s.rc = 0;
        if (s.len == 0) return;
        if (hipSetDevice(s.device) != hipSuccess) {
            s.rc = SJHIP_ERR_HIP;
            return;
        }
        s.rc = sjhip_fetch(s.ctx, tape_dst ? tape_dst + s.tape_base : nullptr, strings_dst ? strings_dst + s.strings_base : nullptr);
    });
    return agree(m, "fetch");
}
