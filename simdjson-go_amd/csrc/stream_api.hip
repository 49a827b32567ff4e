// stream_api.hip -- ParseNDStream (simdjson_amd64.go:101-216) inside the library: sjhip_stream_*.
//
// The reference reads the input in 10 MiB blocks (tmpSize, :127), extends every block to the end of its last record,
// parses (GOMAXPROCS+1)/2 blocks concurrently -- each an independent NDJSON document with every string copied --
// and delivers the results in input order through a channel of channels; the first error ends the stream; input
// buffers (tmpPool) and result buffers (reuse) are recycled.
//
// Here a stream owns N slots.  A slot = one sjhip_ctx (own HIP stream + device arenas) on one of the stream's
// devices (round robin) + one PINNED input block + PINNED tape / Strings.B result buffers + one worker thread.
//   caller                          worker of the slot                                   caller
//   acquire -> fill -> submit  ==>  H2D (pinned, async) -> kernels -> D2H (pinned)  ==>  next (in order) -> release
// The caller reads its input straight into the pinned block (no staging copy) and copies the result out of pinned
// memory into its own slices; H2D of one block, the kernels of another and the D2H of a third overlap because every
// slot has its own HIP stream.  Nothing the caller passes in is retained after a call returns (cgo rule): blocks and
// results live in memory the library owns.
#include <hip/hip_runtime.h>
#include <string.h>

#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/sjhip.h"
#include "sj_ctx.h"

namespace {

enum SlotState { FREE, FILLING, QUEUED, RUNNING, DONE, DELIVERED };

struct Slot {
    sjhip_ctx *ctx = nullptr;
    int device = 0;
    uint8_t *in = nullptr;  // pinned, in_cap bytes (block_cap unless a long record made it grow)
    size_t in_cap = 0, in_len = 0;
    uint64_t *tape = nullptr;  // pinned result buffers, grown on demand
    size_t tape_cap = 0;
    uint8_t *strings = nullptr;
    size_t strings_cap = 0;
    size_t tape_len = 0, strings_len = 0, msg_off = 0, msg_len = 0;
    uint64_t records = 0;  // filtered streams: matching records of the block
    int rc = 0;
    char err[256] = {0};
    SlotState state = FREE;
    uint64_t seq = 0;  // submission number of the block it holds
    std::thread worker;
};

}  // namespace

struct sjhip_stream {
    std::vector<Slot> slots;
    size_t block_cap = 0;
    uint32_t flags = 0;
    std::mutex mu;
    std::condition_variable cv;
    uint64_t next_submit = 0, next_deliver = 0;  // sequence numbers
    int filling = -1;                            // slot handed out by acquire
    int delivered = -1;                          // slot handed out by next
    bool failed = false, quit = false;
    // optional filter (sjhip_stream_set_filter): every block is parsed, filtered on the device, and only the matching
    // records' (Tape, Strings.B) cross PCIe
    bool filter = false;
    std::vector<uint8_t> fkey, fval;
    uint64_t fail_seq = ~0ull;  // lowest sequence number of a block that failed: nothing behind it needs parsing
    char err[256] = {0};
};

static int grow_pinned(void **p, size_t *cap, size_t want) {
    if (want <= *cap) return 0;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t n = want + want / 4 + 4096;
    if (sj::pinned_alloc(p, n) != hipSuccess) return SJHIP_ERR_HIP;
    *cap = n;
    return 0;
}

static void worker_main(sjhip_stream *s, int k) {
    Slot &sl = s->slots[k];
    std::unique_lock<std::mutex> lk(s->mu);
    for (;;) {
        s->cv.wait(lk, [&] { return s->quit || sl.state == QUEUED; });
        if (s->quit) return;
        sl.state = RUNNING;
        const bool skip = s->failed || sl.seq > s->fail_seq;  // behind the first error nothing is parsed (:207-211)
        lk.unlock();
        int rc = SJHIP_ERR_STREAM_CLOSED;
        if (!skip) {
            // parseMessage on the block: TrimSpace + H2D from the pinned block + stage 1 + stage 2 (parse_api.hip)
            rc = sjhip_parse(sl.ctx, sl.in, sl.in_len, s->flags | SJHIP_FLAG_NDJSON, &sl.tape_len, &sl.strings_len, &sl.msg_off,
                             &sl.msg_len);
            sl.records = 0;
            if (rc == SJHIP_OK && s->filter)  // countWhere / filter on the device-resident tape (query.hip)
                rc = sjhip_filter_where(sl.ctx, s->fkey.data(), s->fkey.size(), s->fval.data(), s->fval.size(), &sl.records,
                                        &sl.tape_len, &sl.strings_len);
            if (rc == SJHIP_OK) {
                size_t tc = sl.tape_cap * sizeof(uint64_t);
                void *tp = sl.tape;
                if (grow_pinned(&tp, &tc, sl.tape_len * sizeof(uint64_t)) ||
                    grow_pinned((void **)&sl.strings, &sl.strings_cap, sl.strings_len + 1))
                    rc = SJHIP_ERR_HIP;
                sl.tape = (uint64_t *)tp;
                sl.tape_cap = tc / sizeof(uint64_t);
                if (rc != SJHIP_OK) snprintf(sl.err, sizeof sl.err, "pinned result buffers (%zu tape words, %zu string bytes): allocation failed",
                                             sl.tape_len, sl.strings_len);
                else if ((rc = s->filter ? sjhip_fetch_filtered(sl.ctx, sl.tape, sl.strings)
                                         : sjhip_fetch(sl.ctx, sl.tape, sl.strings)) != SJHIP_OK)  // D2H into pinned memory
                    snprintf(sl.err, sizeof sl.err, "%s", sjhip_last_error(sl.ctx));
            } else {
                snprintf(sl.err, sizeof sl.err, "%s", sjhip_last_error(sl.ctx));
            }
        }
        lk.lock();
        if (rc != SJHIP_OK && rc != SJHIP_ERR_STREAM_CLOSED && sl.seq < s->fail_seq) s->fail_seq = sl.seq;
        sl.rc = rc;
        sl.state = DONE;
        s->cv.notify_all();
    }
}

sjhip_stream *sjhip_stream_create(int first_device, int n_devices, size_t block_bytes, int slots, uint32_t flags) {
    const int have = sjhip_device_count();
    if (have <= 0 || first_device < 0 || first_device >= have) return nullptr;
    if (n_devices <= 0 || first_device + n_devices > have) n_devices = have - first_device;
    if (block_bytes == 0) block_bytes = (size_t)10 << 20;  // tmpSize
    if (slots <= 0) slots = 3 * n_devices;                 // one block each in H2D, kernels, D2H per device
    sjhip_stream *s = new sjhip_stream();
    s->block_cap = block_bytes;
    s->flags = flags | SJHIP_FLAG_COPY_STRINGS;  // pj.copyStrings = true (:180): the block buffer is recycled
    s->slots.resize((size_t)slots);
    bool ok = true;
    for (int k = 0; k < slots && ok; k++) {
        Slot &sl = s->slots[(size_t)k];
        sl.device = first_device + k % n_devices;
        sl.ctx = sjhip_ctx_create(sl.device);
        ok = sl.ctx != nullptr && hipSetDevice(sl.device) == hipSuccess &&
             sj::pinned_alloc((void **)&sl.in, block_bytes + 64) == hipSuccess;
        sl.in_cap = block_bytes;
    }
    if (!ok) {
        for (Slot &sl : s->slots) {
            if (sl.in) (void)hipHostFree(sl.in);
            if (sl.ctx) sjhip_ctx_destroy(sl.ctx);
        }
        delete s;
        return nullptr;
    }
    for (int k = 0; k < slots; k++) s->slots[(size_t)k].worker = std::thread(worker_main, s, k);
    return s;
}

void sjhip_stream_destroy(sjhip_stream *s) {
    if (!s) return;
    {
        std::unique_lock<std::mutex> lk(s->mu);
        // blocks still queued are dropped; a running one is allowed to finish
        s->cv.wait(lk, [&] {
            for (Slot &sl : s->slots)
                if (sl.state == RUNNING) return false;
            return true;
        });
        s->quit = true;
        s->cv.notify_all();
    }
    for (Slot &sl : s->slots) {
        if (sl.worker.joinable()) sl.worker.join();
        (void)hipSetDevice(sl.device);
        if (sl.in) (void)hipHostFree(sl.in);
        if (sl.tape) (void)hipHostFree(sl.tape);
        if (sl.strings) (void)hipHostFree(sl.strings);
        if (sl.ctx) sjhip_ctx_destroy(sl.ctx);
    }
    delete s;
}

const char *sjhip_stream_last_error(const sjhip_stream *s) { return s ? s->err : "no stream"; }
size_t sjhip_stream_block_capacity(const sjhip_stream *s) { return s ? s->block_cap : 0; }
int sjhip_stream_slots(const sjhip_stream *s) { return s ? (int)s->slots.size() : 0; }

// blocks of the stream that have been submitted and not yet delivered
int sjhip_stream_in_flight(sjhip_stream *s) {
    if (!s) return 0;
    std::lock_guard<std::mutex> lk(s->mu);
    return (int)(s->next_submit - s->next_deliver);
}

int sjhip_stream_acquire(sjhip_stream *s, uint8_t **block, size_t *capacity) {
    if (!s || !block) return SJHIP_ERR_ARG;
    std::unique_lock<std::mutex> lk(s->mu);
    if (s->filling >= 0) {
        snprintf(s->err, sizeof s->err, "a block is already acquired");
        return SJHIP_ERR_ARG;
    }
    if (s->failed) return SJHIP_ERR_STREAM_CLOSED;
    // slots are used in ring order, so submission order = slot order and delivery frees them in the same order
    const int k = (int)(s->next_submit % s->slots.size());
    if (s->slots[(size_t)k].state != FREE) return SJHIP_STREAM_FULL;  // deliver (next + release) a block first
    s->slots[(size_t)k].state = FILLING;
    s->filling = k;
    *block = s->slots[(size_t)k].in;
    if (capacity) *capacity = s->slots[(size_t)k].in_cap;
    return SJHIP_OK;
}

// a record that does not end inside the acquired block: a larger pinned block for this slot, first `keep` bytes kept
int sjhip_stream_grow(sjhip_stream *s, size_t keep, size_t new_capacity, uint8_t **block) {
    if (!s || !block) return SJHIP_ERR_ARG;
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->filling < 0) return SJHIP_ERR_ARG;
    Slot &sl = s->slots[(size_t)s->filling];
    if (new_capacity <= sl.in_cap) {
        *block = sl.in;
        return SJHIP_OK;
    }
    uint8_t *bigger = nullptr;
    (void)hipSetDevice(sl.device);
    if (sj::pinned_alloc((void **)&bigger, new_capacity + 64) != hipSuccess) {
        snprintf(s->err, sizeof s->err, "pinned block of %zu bytes: allocation failed", new_capacity);
        return SJHIP_ERR_HIP;
    }
    if (keep) memcpy(bigger, sl.in, keep < sl.in_cap ? keep : sl.in_cap);
    (void)hipHostFree(sl.in);
    sl.in = bigger;
    sl.in_cap = new_capacity;
    *block = bigger;
    return SJHIP_OK;
}

// hand an acquired block back unused (the reader was exhausted: `if len(tmp) > 0 ... else tmpPool.Put(tmp)`, :178,:205)
int sjhip_stream_cancel(sjhip_stream *s) {
    if (!s) return SJHIP_ERR_ARG;
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->filling < 0) return SJHIP_ERR_ARG;
    s->slots[(size_t)s->filling].state = FREE;
    s->filling = -1;
    return SJHIP_OK;
}

int sjhip_stream_submit(sjhip_stream *s, size_t len) {
    if (!s) return SJHIP_ERR_ARG;
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->filling < 0 || len > s->slots[(size_t)s->filling].in_cap) {
        snprintf(s->err, sizeof s->err, "submit without acquire, or block longer than the capacity");
        return SJHIP_ERR_ARG;
    }
    Slot &sl = s->slots[(size_t)s->filling];
    sl.in_len = len;
    sl.seq = s->next_submit++;
    sl.state = QUEUED;
    s->filling = -1;
    s->cv.notify_all();
    return SJHIP_OK;
}

int sjhip_stream_submit_copy(sjhip_stream *s, const uint8_t *block, size_t len) {
    uint8_t *dst = nullptr;
    size_t cap = 0;
    int rc = sjhip_stream_acquire(s, &dst, &cap);
    if (rc) return rc;
    if (len > cap) {
        rc = sjhip_stream_grow(s, 0, len, &dst);
        if (rc) {
            std::lock_guard<std::mutex> lk(s->mu);
            s->slots[(size_t)s->filling].state = FREE;
            s->filling = -1;
            return rc;
        }
    }
    if (len) memcpy(dst, block, len);
    return sjhip_stream_submit(s, len);
}

int sjhip_stream_next(sjhip_stream *s, sjhip_stream_result *out) {
    if (!s || !out) return SJHIP_ERR_ARG;
    std::unique_lock<std::mutex> lk(s->mu);
    memset(out, 0, sizeof *out);
    if (s->delivered >= 0) {
        snprintf(s->err, sizeof s->err, "the previous result has not been released");
        return SJHIP_ERR_ARG;
    }
    if (s->failed) {  // the stream ended with an error: whatever is still in flight is dropped
        while (s->next_deliver != s->next_submit) {
            Slot &d = s->slots[(size_t)(s->next_deliver % s->slots.size())];
            s->cv.wait(lk, [&] { return d.state == DONE; });
            d.state = FREE;
            s->next_deliver++;
        }
        s->cv.notify_all();
        return SJHIP_ERR_STREAM_CLOSED;
    }
    if (s->next_deliver == s->next_submit) return SJHIP_STREAM_EMPTY;
    const int k = (int)(s->next_deliver % s->slots.size());
    Slot &sl = s->slots[(size_t)k];
    s->cv.wait(lk, [&] { return sl.state == DONE; });
    s->next_deliver++;
    if (sl.rc != SJHIP_OK) {
        // the first error ends the stream: what was submitted behind it is dropped (simdjson_amd64.go:207-211)
        const int rc = sl.rc;
        if (!s->failed && rc != SJHIP_ERR_STREAM_CLOSED) {
            s->failed = true;
            snprintf(s->err, sizeof s->err, "%s", sl.err);
        }
        sl.state = FREE;
        s->cv.notify_all();
        return rc;
    }
    sl.state = DELIVERED;
    s->delivered = k;
    out->tape = sl.tape;
    out->tape_len = sl.tape_len;
    out->strings = sl.strings;
    out->strings_len = sl.strings_len;
    out->message = sl.in + sl.msg_off;
    out->message_len = sl.msg_len;
    out->device = sl.device;
    out->records = sl.records;
    return SJHIP_OK;
}

// Compose the stream with the query of query.hip: from now on every block's result is the (Tape, Strings.B) of the
// records whose root object has `key` with the string value `value` (sjhip_filter_where) -- what ParseND returns for the
// document made of the block's matching lines -- and sjhip_stream_result::records counts them.  Set before the first
// block is submitted (klen = 0 turns the filter off).
int sjhip_stream_set_filter(sjhip_stream *s, const uint8_t *key, size_t klen, const uint8_t *value, size_t vlen) {
    if (!s || (klen && !key) || (vlen && !value)) return SJHIP_ERR_ARG;
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->next_submit != s->next_deliver || s->filling >= 0) {
        snprintf(s->err, sizeof s->err, "sjhip_stream_set_filter: blocks are in flight");
        return SJHIP_ERR_ARG;
    }
    s->filter = klen != 0;
    s->fkey.assign(key, key + klen);
    s->fval.assign(value, value + vlen);
    return SJHIP_OK;
}

// would sjhip_stream_next return without waiting?
int sjhip_stream_ready(sjhip_stream *s) {
    if (!s) return 0;
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->delivered >= 0) return 0;
    if (s->failed) return 1;
    if (s->next_deliver == s->next_submit) return 0;
    return s->slots[(size_t)(s->next_deliver % s->slots.size())].state == DONE ? 1 : 0;
}

int sjhip_stream_release(sjhip_stream *s) {
    if (!s) return SJHIP_ERR_ARG;
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->delivered < 0) return SJHIP_ERR_ARG;
    s->slots[(size_t)s->delivered].state = FREE;
    s->delivered = -1;
    s->cv.notify_all();
    return SJHIP_OK;
}
