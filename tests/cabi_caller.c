/* cabi_caller.c -- a plain C program that drives libsjhip through include/sjhip.h with exactly the call sequences of the
 * Go binding (simdjson-go_amd/go/simdjson_hip.go), for the case that nobody here can compile that file: Parse with a
 * recycled ParsedJson (parseMessageHip), ParseND over several contexts (parseMessageMulti), ParseBatch and the
 * reader / deliverer protocol of ParseNDStream.  Built with gcc as C11 (tests/test_cabi_caller.py): that the header is
 * a C header and every symbol links is the `-m "not gpu"` half; the `-m gpu` half runs the modes below and compares
 * what they dump with the oracle.
 *
 *   cabi_caller parse  <in> <out> <flags> <repeat>      sjhip_parse + sjhip_fetch, buffers recycled across calls
 *   cabi_caller multi  <in> <out> <flags> <shards>      sjhip_multi_create / sjhip_parse_nd_multi / sjhip_fetch_multi
 *   cabi_caller batch  <out> <in1> <in2> ...            sjhip_parse_batch + sjhip_fetch
 *   cabi_caller stream <in> <out> <block> <slots>       acquire / grow / submit / ready / next / release
 * Dump format (little endian u64 words): parse / multi / batch: rc, tape_len, strings_len, msg_off, msg_len, tape,
 * strings (padded to 8 bytes).  stream: per delivered block rc, tape_len, strings_len, message_len, tape, strings
 * (padded), message (padded); the last record has rc != 0 (SJHIP_STREAM_EMPTY = clean end) and no payload. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sjhip.h"

static uint8_t *read_file(const char *path, size_t *len) {
    FILE *f = fopen(path, "rb");
    if (!f) {
        fprintf(stderr, "cannot open %s\n", path);
        exit(2);
    }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *b = (uint8_t *)malloc((size_t)n + 1);
    if (n && fread(b, 1, (size_t)n, f) != (size_t)n) exit(2);
    fclose(f);
    *len = (size_t)n;
    return b;
}
static void put_u64(FILE *f, uint64_t v) { fwrite(&v, 8, 1, f); }
static void put_padded(FILE *f, const void *p, size_t n) {
    static const uint8_t zero[8] = {0};
    if (n) fwrite(p, 1, n, f);
    if (n % 8) fwrite(zero, 1, 8 - n % 8, f);
}

/* a ParsedJson as the Go side holds it: slices with a capacity that survives calls (the `reuse` argument) */
typedef struct {
    uint64_t *tape;
    size_t tape_len, tape_cap;
    uint8_t *strings;
    size_t strings_len, strings_cap;
} parsed_json;
static void pj_resize(parsed_json *pj, size_t tl, size_t sl) { /* `if cap(pj.Tape) < tapeLen { make(...) }` */
    if (pj->tape_cap < tl) {
        free(pj->tape);
        pj->tape = (uint64_t *)malloc(tl * 8 + 8);
        pj->tape_cap = tl;
    }
    if (pj->strings_cap < sl) {
        free(pj->strings);
        pj->strings = (uint8_t *)malloc(sl + 8);
        pj->strings_cap = sl;
    }
    pj->tape_len = tl;
    pj->strings_len = sl;
}
static void dump_result(FILE *f, int rc, const parsed_json *pj, size_t off, size_t len) {
    put_u64(f, (uint64_t)(int64_t)rc);
    put_u64(f, rc ? 0 : pj->tape_len);
    put_u64(f, rc ? 0 : pj->strings_len);
    put_u64(f, off);
    put_u64(f, len);
    if (!rc) {
        put_padded(f, pj->tape, pj->tape_len * 8);
        put_padded(f, pj->strings, pj->strings_len);
    }
}

static int mode_parse(int argc, char **argv) {
    if (argc < 6) return 2;
    size_t len;
    uint8_t *msg = read_file(argv[2], &len);
    const uint32_t flags = (uint32_t)strtoul(argv[4], NULL, 0);
    const int repeat = atoi(argv[5]);
    if (!sjhip_supported()) return 3;
    sjhip_ctx *ctx = sjhip_ctx_create(0);
    if (!ctx) return 3;
    parsed_json pj = {0};
    int rc = 0;
    size_t tl = 0, sl = 0, off = 0, ml = 0;
    for (int it = 0; it < repeat; it++) {
        /* odd iterations parse a prefix-free variant of the input (the first half up to a point where it is invalid or
         * smaller) so that the recycled buffers see results of different sizes, like a reused ParsedJson does */
        const size_t n = (it & 1) && len > 2 ? len / 2 : len;
        rc = sjhip_parse(ctx, n ? msg : NULL, n, flags, &tl, &sl, &off, &ml);
        if (rc == SJHIP_OK) {
            pj_resize(&pj, tl, sl);
            rc = sjhip_fetch(ctx, tl ? pj.tape : NULL, sl ? pj.strings : NULL);
        }
        if (rc != SJHIP_OK && rc != SJHIP_ERR_STAGE1 && rc != SJHIP_ERR_STAGE2) fprintf(stderr, "sjhip: %s\n", sjhip_last_error(ctx));
    }
    if (!(repeat & 1)) { /* the last iteration parsed the half: finish with the whole message */
        rc = sjhip_parse(ctx, len ? msg : NULL, len, flags, &tl, &sl, &off, &ml);
        if (rc == SJHIP_OK) {
            pj_resize(&pj, tl, sl);
            rc = sjhip_fetch(ctx, tl ? pj.tape : NULL, sl ? pj.strings : NULL);
        }
    }
    if (rc == SJHIP_OK) { /* the same result read in place (sjhip_fetch_view) must be what sjhip_fetch copied out */
        const uint64_t *vt = NULL;
        const uint8_t *vs = NULL;
        rc = sjhip_fetch_view(ctx, &vt, &vs);
        if (rc == SJHIP_OK && ((tl && (!vt || memcmp(vt, pj.tape, tl * 8))) || (sl && (!vs || memcmp(vs, pj.strings, sl))))) {
            fprintf(stderr, "sjhip_fetch_view differs from sjhip_fetch\n");
            rc = 99;
        }
    }
    FILE *f = fopen(argv[3], "wb");
    dump_result(f, rc, &pj, off, ml);
    fclose(f);
    sjhip_ctx_destroy(ctx);
    return 0;
}

static int mode_multi(int argc, char **argv) {
    if (argc < 6) return 2;
    size_t len;
    uint8_t *msg = read_file(argv[2], &len);
    const uint32_t flags = (uint32_t)strtoul(argv[4], NULL, 0) | SJHIP_FLAG_NDJSON;
    const int shards = atoi(argv[5]);
    const int ndev = sjhip_device_count();
    if (ndev < 1) return 3;
    int devices[64];
    for (int i = 0; i < shards && i < 64; i++) devices[i] = i % ndev; /* every visible device, round robin */
    sjhip_multi *m = shards > 0 ? sjhip_multi_create(devices, shards) : sjhip_multi_create(NULL, 0);
    if (!m) return 3;
    parsed_json pj = {0};
    size_t tl = 0, sl = 0, off = 0, ml = 0;
    int rc = 0;
    for (int it = 0; it < 2; it++) { /* the handle is pooled by the shim: use it twice */
        rc = sjhip_parse_nd_multi(m, len ? msg : NULL, len, flags, &tl, &sl, &off, &ml);
        if (rc == SJHIP_OK) {
            pj_resize(&pj, tl, sl);
            rc = sjhip_fetch_multi(m, tl ? pj.tape : NULL, sl ? pj.strings : NULL);
        }
    }
    if (rc != SJHIP_OK && rc != SJHIP_ERR_STAGE1 && rc != SJHIP_ERR_STAGE2) fprintf(stderr, "sjhip: %s\n", sjhip_multi_last_error(m));
    FILE *f = fopen(argv[3], "wb");
    dump_result(f, rc, &pj, off, ml);
    put_u64(f, (uint64_t)sjhip_multi_shards(m));
    fclose(f);
    sjhip_multi_destroy(m);
    return 0;
}

static int mode_batch(int argc, char **argv) {
    if (argc < 4) return 2;
    const size_t n = (size_t)(argc - 3);
    const uint8_t **ptrs = (const uint8_t **)calloc(n + 1, sizeof *ptrs);
    size_t *lens = (size_t *)calloc(n + 1, sizeof *lens);
    for (size_t i = 0; i < n; i++) ptrs[i] = read_file(argv[3 + i], &lens[i]);
    sjhip_ctx *ctx = sjhip_ctx_create(0);
    if (!ctx) return 3;
    parsed_json pj = {0};
    size_t tl = 0, sl = 0;
    int rc = sjhip_parse_batch(ctx, ptrs, lens, n, SJHIP_FLAG_COPY_STRINGS, &tl, &sl);
    if (rc == SJHIP_OK) {
        pj_resize(&pj, tl, sl);
        rc = sjhip_fetch(ctx, tl ? pj.tape : NULL, sl ? pj.strings : NULL);
    }
    FILE *f = fopen(argv[2], "wb");
    dump_result(f, rc, &pj, 0, 0);
    fclose(f);
    sjhip_ctx_destroy(ctx);
    return 0;
}

/* ParseNDStream (simdjson_amd64.go:116-216 / the shim's reader + deliverer) on one thread: finished blocks are
 * delivered before every read (sjhip_stream_ready), a full stream delivers one block to make room */
static int deliver(sjhip_stream *st, FILE *f, int *ended) {
    sjhip_stream_result out;
    const int rc = sjhip_stream_next(st, &out);
    if (rc == SJHIP_STREAM_EMPTY) return rc;
    if (rc != SJHIP_OK) { /* the first error is the last value of the stream */
        put_u64(f, (uint64_t)(int64_t)rc);
        *ended = 1;
        return rc;
    }
    put_u64(f, 0);
    put_u64(f, out.tape_len);
    put_u64(f, out.strings_len);
    put_u64(f, out.message_len);
    put_padded(f, out.tape, out.tape_len * 8); /* `copy(pj.Tape, unsafe.Slice(out.tape, tl))` */
    put_padded(f, out.strings, out.strings_len);
    put_padded(f, out.message, out.message_len);
    sjhip_stream_release(st);
    return SJHIP_OK;
}
static int mode_stream(int argc, char **argv) {
    if (argc < 6) return 2;
    size_t len;
    uint8_t *in = read_file(argv[2], &len);
    const size_t block = (size_t)strtoull(argv[4], NULL, 0);
    const int slots = atoi(argv[5]);
    const size_t reserve = block / 8 + 64;
    sjhip_stream *st = sjhip_stream_create(0, 0, block + reserve, slots, 0);
    if (!st) return 3;
    FILE *f = fopen(argv[3], "wb");
    size_t at = 0;
    int ended = 0, eof = 0;
    while (!ended && !eof) {
        while (!ended && sjhip_stream_ready(st)) deliver(st, f, &ended);
        if (ended) break;
        uint8_t *blk = NULL;
        size_t cap = 0;
        int rc = sjhip_stream_acquire(st, &blk, &cap);
        if (rc == SJHIP_STREAM_FULL) {
            deliver(st, f, &ended);
            continue;
        }
        if (rc != SJHIP_OK) break; /* closed after a failed block */
        /* io.ReadFull(rd, buf[:blockSize]), then rd.ReadBytes('\n') to the end of the current record */
        size_t n = len - at < block ? len - at : block;
        memcpy(blk, in + at, n);
        at += n;
        if (n == block) {
            const uint8_t *nl = (const uint8_t *)memchr(in + at, '\n', len - at);
            const size_t rest = nl ? (size_t)(nl - (in + at)) + 1 : len - at;
            if (n + rest > cap) {
                if (sjhip_stream_grow(st, n, n + rest, &blk) != SJHIP_OK) {
                    sjhip_stream_cancel(st);
                    break;
                }
            }
            memcpy(blk + n, in + at, rest);
            n += rest;
            at += rest;
            if (!nl) eof = 1;
        } else {
            eof = 1;
        }
        if (n > 0) sjhip_stream_submit(st, n);
        else sjhip_stream_cancel(st);
    }
    while (!ended) {
        if (deliver(st, f, &ended) == SJHIP_STREAM_EMPTY) {
            put_u64(f, (uint64_t)SJHIP_STREAM_EMPTY); /* io.EOF */
            break;
        }
    }
    fclose(f);
    sjhip_stream_destroy(st);
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    if (!strcmp(argv[1], "parse")) return mode_parse(argc, argv);
    if (!strcmp(argv[1], "multi")) return mode_multi(argc, argv);
    if (!strcmp(argv[1], "batch")) return mode_batch(argc, argv);
    if (!strcmp(argv[1], "stream")) return mode_stream(argc, argv);
    if (!strcmp(argv[1], "symbols")) { /* touches nothing: the link is the test */
        printf("%d\n", SJHIP_OK);
        return 0;
    }
    return 2;
}
