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
int sjhip_multi_shard_device(const sjhip_multi *m, int shard) {
    if (!m || shard < 0 || (size_t)shard >= m->shards.size()) return -1;
    const sjhip_ctx *c = m->shards[(size_t)shard].ctx;
    if (!c || !c->d_tape.p) return -1;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, c->d_tape.p) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return at.device;
}

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

// A message that lies in device memory is only looked at through small windows copied to the host: the record cuts and
// the whitespace at the ends of every shard (the reference trims what it parses, simdjson_amd64.go:87,
// parse_json_amd64.go:55).  JSON whitespace and the other ASCII blanks bytes.TrimSpace removes; a shard of a
// device-resident message that ends in a multi-byte Unicode blank keeps it (and fails like any other stray byte).
namespace {
constexpr size_t WINDOW = 64 << 10;
struct DevPeek {
    const uint8_t *d_msg;
    std::vector<uint8_t> buf = std::vector<uint8_t>(WINDOW);
    bool ok = true;
    const uint8_t *get(size_t at, size_t n) {  // n <= WINDOW
        if (hipMemcpy(buf.data(), d_msg + at, n, hipMemcpyDeviceToHost) != hipSuccess) ok = false;
        return buf.data();
    }
    // first '\n' at or behind `from` (len if there is none)
    size_t find_newline(size_t from, size_t len) {
        for (size_t at = from; at < len && ok; at += WINDOW) {
            const size_t n = len - at < WINDOW ? len - at : WINDOW;
            const void *nl = memchr(get(at, n), '\n', n);
            if (nl) return at + (size_t)((const uint8_t *)nl - buf.data());
        }
        return len;
    }
    static bool blank(uint8_t c) { return c == ' ' || (c >= '\t' && c <= '\r'); }
    void trim(size_t a, size_t b, size_t *off, size_t *ln) {  // [a, b) without the ASCII blanks at its ends
        while (a < b && ok) {
            const size_t n = b - a < WINDOW ? b - a : WINDOW;
            const uint8_t *w = get(a, n);
            size_t k = 0;
            while (k < n && blank(w[k])) k++;
            a += k;
            if (k < n) break;
        }
        while (b > a && ok) {
            const size_t n = b - a < WINDOW ? b - a : WINDOW;
            const uint8_t *w = get(b - n, n);
            size_t k = n;
            while (k > 0 && blank(w[k - 1])) k--;
            b -= n - k;
            if (k > 0) break;
        }
        *off = a;
        *ln = b - a;
    }
};
}  // namespace

// msg: host memory, or (d_resident) device memory of the device every shard of m runs on -- then nothing is copied, a
// shard parses its window of the message in place
static int multi_parse(sjhip_multi *m, const uint8_t *msg, size_t len, uint32_t flags, bool d_resident, size_t *tape_len,
                       size_t *strings_len, size_t *msg_off, size_t *msg_len) {
    if (!m || m->shards.empty()) return SJHIP_ERR_ARG;
    m->valid = 0;
    m->tape_len = m->strings_len = 0;
    DevPeek peek{msg};
    size_t g_off = 0, g_len = 0;
    if (len) {
        if (d_resident) peek.trim(0, len, &g_off, &g_len);
        else sj::trim_space(msg, len, &g_off, &g_len);  // pj.Message = bytes.TrimSpace(msg), parse_json_amd64.go:55
    }
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
            if (d_resident) {
                const size_t nl = peek.find_newline(target, len);
                end = nl < len ? nl + 1 : len;
            } else {
                const void *nl = target < len ? memchr(msg + target, '\n', len - target) : nullptr;
                end = nl ? (size_t)((const uint8_t *)nl - msg) + 1 : len;
            }
        }
        size_t off = 0, ln = 0;
        if (end > cut) {
            if (d_resident) {
                peek.trim(cut, end, &off, &ln);
                off -= cut;
            } else {
                sj::trim_space(msg + cut, end - cut, &off, &ln);
            }
        }
        s.start = cut + off;
        s.len = ln;
        s.tape_len = s.strings_len = 0;
        s.rc = 0;
        cut = end;
    }
    if (!peek.ok) {
        snprintf(m->err, sizeof m->err, "reading the device-resident message failed");
        return SJHIP_ERR_HIP;
    }
    const uint32_t fl = flags | SJHIP_FLAG_NDJSON;
    for_shards(m, [&](Shard &s) {  // phase 1
        if (s.len == 0) return;
        sjhip_ctx *ctx = s.ctx;
        if (hipSetDevice(s.device) != hipSuccess) {
            s.rc = SJHIP_ERR_HIP;
            return;
        }
        if (d_resident) {
            s.rc = sjhip_parse_shard_begin(ctx, msg + s.start, s.len, fl, &s.tape_len, &s.strings_len);
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

int sjhip_parse_nd_multi(sjhip_multi *m, const uint8_t *msg, size_t len, uint32_t flags, size_t *tape_len, size_t *strings_len,
                         size_t *msg_off, size_t *msg_len) {
    return multi_parse(m, msg, len, flags, false, tape_len, strings_len, msg_off, msg_len);
}

// ---- ND messages beyond one context's reach (uint32 positions: 4 GiB - 64) ----------------------------------------------
// The reference parses "arbitrarily large" ND inputs (README.md:567-569: its index stream is deltas,
// flatten_bits_amd64.s:38-41).  Here sjhip_parse / sjhip_parse_device hand such a message to a multi handle the context
// owns: shards of at most `shard_bytes` cut at record boundaries, every shard on its own context with absolute uint32
// positions inside the shard and 64-bit bases added on the device -- the sharded ParseND path, on one device or, for a
// host message, round robin over every visible one.  All shards are resident at once (their tapes wait for sjhip_fetch).
int sj::parse_nd_big(sjhip_ctx *ctx, const uint8_t *msg, size_t len, uint32_t flags, bool d_resident, size_t shard_bytes,
                     size_t *tape_len, size_t *strings_len, size_t *msg_off, size_t *msg_len) {
    const size_t want = (len + shard_bytes - 1) / shard_bytes;
    if (want > 4096) {
        sj::ctx_set_error(ctx, "ND message of %zu bytes: more than 4096 shards", len);
        return SJHIP_ERR_TOOBIG;
    }
    if (ctx->big && (size_t)sjhip_multi_shards(ctx->big) != want) {
        sjhip_multi_destroy(ctx->big);
        ctx->big = nullptr;
    }
    if (!ctx->big) {
        const int have = sjhip_device_count();
        std::vector<int> devs(want);
        // a device-resident message is parsed where it lies; a host message uses every visible device, this one first
        for (size_t k = 0; k < want; k++) devs[k] = d_resident || have < 1 ? ctx->device : (int)((ctx->device + k) % (size_t)have);
        ctx->big = sjhip_multi_create(devs.data(), (int)want);
        if (!ctx->big) {
            sj::ctx_set_error(ctx, "ND message of %zu bytes: no contexts for %zu shards", len, want);
            return SJHIP_ERR_HIP;
        }
    }
    ctx->big_valid = 0;
    const int rc = multi_parse(ctx->big, msg, len, flags, d_resident, tape_len, strings_len, msg_off, msg_len);
    if (rc == SJHIP_OK) {
        ctx->big_valid = 1;
        // the totals of the merged result, for sjhip_fetch_view (which sizes its view block from the context)
        ctx->tape_len = ctx->big->tape_len;
        ctx->strings_len = ctx->big->strings_len;
    } else if (rc != SJHIP_ERR_STAGE1 && rc != SJHIP_ERR_STAGE2) sj::ctx_set_error(ctx, "%s", sjhip_multi_last_error(ctx->big));
    (void)hipSetDevice(ctx->device);
    return rc;
}
int sj::fetch_nd_big(sjhip_ctx *ctx, uint64_t *tape_dst, uint8_t *strings_dst) {
    const int rc = sjhip_fetch_multi(ctx->big, tape_dst, strings_dst);
    if (rc) sj::ctx_set_error(ctx, "%s", sjhip_multi_last_error(ctx->big));
    (void)hipSetDevice(ctx->device);
    return rc;
}
size_t sj::nd_big_device_bytes(const sjhip_ctx *ctx) {
    size_t total = 0;
    if (ctx->big)
        for (const Shard &s : ctx->big->shards) total += sjhip_ctx_device_bytes(s.ctx);
    return total;
}

int sj::nd_big_shards(const sjhip_ctx *ctx) { return ctx->big && ctx->big_valid ? (int)ctx->big->shards.size() : 0; }
sjhip_ctx *sj::nd_big_shard(const sjhip_ctx *ctx, int k) {
    if (!ctx->big || k < 0 || (size_t)k >= ctx->big->shards.size()) return nullptr;
    const Shard &s = ctx->big->shards[(size_t)k];
    return s.len ? s.ctx : nullptr;
}

void sj::release_nd_big(sjhip_ctx *ctx) {
    if (ctx->big) sjhip_multi_destroy(ctx->big);
    ctx->big = nullptr;
    ctx->big_valid = 0;
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
