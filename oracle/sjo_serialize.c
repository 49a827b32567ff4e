/*
 * sjo_serialize.c -- ORACLE (test infrastructure only, see sjo.h).
 * Restatement of Serializer.Serialize / Deserialize, format version 3 (parsed_serialize.go:200-431, 466-695) with
 * CompressNone (every block type 0; S2 / zstd are compression of the same columns and stay on the host).
 *
 * One thing cannot be restated bit for bit: the string de-duplication (indexString, :836-857) keys its 16 384-entry
 * table with Go's runtime.memhash, whose seed is random per process -- the reference's own output differs from run
 * to run.  `dedup` = 1 restates the algorithm with FNV-1a in its place (same table size, same replace-on-miss
 * policy); `dedup` = 0 appends every string (the columns a parser-order Strings.B gives directly).  What the reference
 * pins (parsed_serialize_test.go:220-340) is the round trip: Deserialize(Serialize(pj)) marshals to the same JSON.
 * The framing itself is pinned by tests/golden/serialize_v3_vectors.py: five streams written out by hand from the format
 * comment and the encoding loop (:201-236, :283-341, :376-431) for documents whose bytes do not depend on the hash
 * (distinct strings are never merged, an immediately repeated one always is); this file must reproduce them byte for
 * byte (tests/test_oracle_serialize.py).
 */
#include "sjo.h"

#include <stdlib.h>
#include <string.h>

#define STRING_BITS 14
#define STRING_SIZE (1u << STRING_BITS)
#define STRING_MASK (STRING_SIZE - 1u)

typedef struct {
    uint8_t *p;
    size_t len, cap;
} buf_t;

static void put(buf_t *b, const void *src, size_t n) {
    if (b->len + n > b->cap) {
        b->cap = (b->len + n) * 2 + 64;
        b->p = (uint8_t *)realloc(b->p, b->cap);
    }
    if (n) memcpy(b->p + b->len, src, n);
    b->len += n;
}
static void put_u64(buf_t *b, uint64_t v) { put(b, &v, 8); } /* binary.LittleEndian.PutUint64 on a little-endian host */
static void put_byte(buf_t *b, uint8_t v) { put(b, &v, 1); }
static void put_uvarint(buf_t *b, uint64_t v) { /* binary.PutUvarint */
    while (v >= 0x80) {
        put_byte(b, (uint8_t)v | 0x80);
        v >>= 7;
    }
    put_byte(b, (uint8_t)v);
}

/* Serialize (:200-431).  The three columns are also returned separately (tags / values / stringBuf) for the tests.
 * msg: pj.Message (strings that were not copied point into it).  Returns 0, or -1 on an unknown tag. */
int sjo_serialize(const uint64_t *tape, size_t tape_len, const uint8_t *strings, size_t strings_len, const uint8_t *msg,
                  size_t msg_len, int dedup, uint8_t **out, size_t *out_len, uint8_t **tags_out, size_t *tags_len,
                  uint8_t **values_out, size_t *values_len, uint8_t **sbuf_out, size_t *sbuf_len) {
    buf_t tags = {0}, vals = {0}, sbuf = {0}, dst = {0};
    uint32_t *table = (uint32_t *)calloc(STRING_SIZE, sizeof(uint32_t)); /* offsets + 1; 0 = empty (:241-246) */
    (void)strings_len;
    (void)msg_len;
    for (size_t off = 0; off < tape_len; off++) {
        const uint64_t entry = tape[off];
        uint8_t ntype = (uint8_t)(entry >> 56);
        const uint64_t payload = entry & SJO_JSONVALUEMASK;
        switch (ntype) {
        case 'N': /* TagNop: skip counts are rebuilt on the way back */
            break;
        case '"': {
            const uint64_t len = tape[off + 1];
            const uint8_t *sb = (payload & SJO_STRINGBUFBIT) ? strings + (payload & (SJO_STRINGBUFBIT - 1)) : msg + payload;
            uint64_t offset;
            uint32_t h = 2166136261u; /* stand-in for memhash (see the header) */
            for (uint64_t k = 0; k < len; k++) h = (h ^ sb[k]) * 16777619u;
            h &= STRING_MASK;
            long o = (long)table[h] - 1;
            if (dedup && o >= 0 && (size_t)o + len <= sbuf.len && memcmp(sbuf.p + o, sb, len) == 0) {
                offset = (uint64_t)o;
            } else {
                offset = sbuf.len;
                put(&sbuf, sb, len);
                table[h] = (uint32_t)(offset + 1);
            }
            put_u64(&vals, offset);
            put_u64(&vals, len);
            off++;
            break;
        }
        case 'u':
        case 'l':
            put_u64(&vals, tape[off + 1]);
            off++;
            break;
        case 'd':
            if (payload == 0) {
                put_u64(&vals, tape[off + 1]);
            } else { /* tagFloatWithFlag 'e': the whole entry travels (:313-320) */
                ntype = 'e';
                put_u64(&vals, entry);
                put_u64(&vals, tape[off + 1]);
            }
            off++;
            break;
        case 'n':
        case 't':
        case 'f':
            break;
        case '{':
        case '[':
        case 'r': /* (Offset - Current offset); roots rely on wrap-around (:324-328) */
            put_u64(&vals, payload - (uint64_t)off);
            break;
        case '}':
        case ']':
        case 0:
            break;
        default:
            free(table);
            free(tags.p);
            free(vals.p);
            free(sbuf.p);
            return -1;
        }
        put_byte(&tags, ntype);
    }
    free(table);
    /* container (:381-426) */
    buf_t rest = {0};
    put_uvarint(&rest, tape_len);
    put_byte(&rest, 0); /* Strings: uncompressed size 0 */
    put_byte(&rest, 0); /* Strings: block size 0 */
    put_uvarint(&rest, sbuf.len);
    put_uvarint(&rest, sbuf.len + 1); /* block = type byte + data */
    put_byte(&rest, 0);
    put(&rest, sbuf.p, sbuf.len);
    put_uvarint(&rest, tags.len);
    put_uvarint(&rest, tags.len + 1);
    put_byte(&rest, 0);
    put(&rest, tags.p, tags.len);
    put_uvarint(&rest, vals.len);
    put_uvarint(&rest, vals.len + 1);
    put_byte(&rest, 0);
    put(&rest, vals.p, vals.len);
    put_byte(&dst, 3); /* serializedVersion */
    put_uvarint(&dst, rest.len);
    put(&dst, rest.p, rest.len);
    free(rest.p);
    *out = dst.p;
    *out_len = dst.len;
    *tags_out = tags.p;
    *tags_len = tags.len;
    *values_out = vals.p;
    *values_len = vals.len;
    *sbuf_out = sbuf.p;
    *sbuf_len = sbuf.len;
    return 0;
}

/* ---- Deserialize (:466-695), uncompressed blocks only ---- */
typedef struct {
    const uint8_t *p;
    size_t len, pos;
} rd_t;
static int rd_uvarint(rd_t *r, uint64_t *v) {
    uint64_t x = 0;
    unsigned s = 0;
    for (int i = 0; i < 10; i++) {
        if (r->pos >= r->len) return -1;
        const uint8_t b = r->p[r->pos++];
        if (b < 0x80) {
            *v = x | ((uint64_t)b << s);
            return 0;
        }
        x |= (uint64_t)(b & 0x7f) << s;
        s += 7;
    }
    return -1;
}
/* decBlock (:697-757): block size, type byte, data */
static int rd_block(rd_t *r, uint8_t *dst, size_t want) {
    uint64_t size;
    if (rd_uvarint(r, &size)) return -1;
    if (size > r->len - r->pos) return -1;
    if (size == 0 && want == 0) return 0;
    if (size < 1) return -1;
    const uint8_t typ = r->p[r->pos++];
    size--;
    if (typ != 0 || size != want) return -1; /* only blockTypeUncompressed here */
    memcpy(dst, r->p + r->pos, want);
    r->pos += want;
    return 0;
}

/* -> 0 ok; tape / strings / message are malloc'ed (sjo_free).  Negative: malformed input. */
int sjo_deserialize(const uint8_t *src, size_t src_len, uint64_t **tape_out, size_t *tape_len, uint8_t **strings_out,
                    size_t *strings_len, uint8_t **message_out, size_t *message_len) {
    rd_t r = {src, src_len, 0};
    uint64_t c, ts, ss, ms, ntags, nvals;
    if (r.len < 1 || r.p[r.pos++] > 3) return -1;
    if (rd_uvarint(&r, &c) || c > r.len - r.pos) return -2;
    if (rd_uvarint(&r, &ts) || rd_uvarint(&r, &ss)) return -3;
    uint64_t *tape = (uint64_t *)calloc(ts + 1, 8);
    uint8_t *strs = (uint8_t *)malloc(ss + 1);
    if (rd_block(&r, strs, ss)) goto bad;
    if (rd_uvarint(&r, &ms)) goto bad;
    uint8_t *msg = (uint8_t *)malloc(ms + 1);
    uint8_t *tags = NULL, *vals = NULL;
    if (rd_block(&r, msg, ms)) goto bad2;
    if (rd_uvarint(&r, &ntags)) goto bad2;
    tags = (uint8_t *)malloc(ntags + 1);
    if (rd_block(&r, tags, ntags)) goto bad2;
    if (rd_uvarint(&r, &nvals)) goto bad2;
    vals = (uint8_t *)malloc(nvals + 1);
    if (rd_block(&r, vals, nvals)) goto bad2;
    {
        /* reconstruct the tape (:592-686) */
        size_t off = 0, vp = 0;
        uint64_t nskips = 0;
#define NEED(n)                      \
    do {                             \
        if (nvals - vp < (n)) goto bad2; \
    } while (0)
#define VAL(k) (*(const uint64_t *)(vals + vp + 8 * (k)))
        for (uint64_t ti = 0; ti < ntags; ti++) {
            if (off == ts) goto bad2;
            const uint8_t t = tags[ti];
            const uint64_t tag_dst = (uint64_t)t << 56;
            if (nskips > 0 && t != 'N') {
                for (uint64_t i = 0; i < nskips; i++) tape[off++] = ((uint64_t)'N' << 56) | (nskips - i);
                nskips = 0;
            }
            switch (t) {
            case 'N':
                nskips++;
                break;
            case '"':
                NEED(16);
                tape[off] = tag_dst | VAL(0);
                tape[off + 1] = VAL(1);
                vp += 16;
                off += 2;
                break;
            case 'd':
            case 'l':
            case 'u':
                NEED(8);
                tape[off] = tag_dst;
                tape[off + 1] = VAL(0);
                vp += 8;
                off += 2;
                break;
            case 'e':
                NEED(16);
                tape[off] = VAL(0);
                tape[off + 1] = VAL(1);
                vp += 16;
                off += 2;
                break;
            case 'n':
            case 't':
            case 'f':
            case 0:
                tape[off++] = tag_dst;
                break;
            case '{':
            case '[': {
                NEED(8);
                const uint64_t val = VAL(0) + (uint64_t)off;
                vp += 8;
                if (val > ts || val == 0) goto bad2;
                tape[off] = tag_dst | val;
                tape[val - 1] = ((uint64_t)(t == '{' ? '}' : ']') << 56) | (uint64_t)off; /* tagOpenToClose */
                off++;
                break;
            }
            case 'r': {
                NEED(8);
                const uint64_t val = VAL(0) + (uint64_t)off;
                vp += 8;
                if (val > ts) goto bad2;
                tape[off++] = tag_dst | val;
                break;
            }
            case '}':
            case ']':
                if ((tape[off] >> 56) != t) goto bad2; /* written by its opening tag */
                off++;
                break;
            default:
                goto bad2;
            }
        }
        for (uint64_t i = 0; i < nskips; i++) tape[off++] = ((uint64_t)'N' << 56) | (nskips - i);
        if (off != ts || vp != nvals) goto bad2;
    }
    free(tags);
    free(vals);
    *tape_out = tape;
    *tape_len = ts;
    *strings_out = strs;
    *strings_len = ss;
    *message_out = msg;
    *message_len = ms;
    return 0;
bad2:
    free(msg);
    free(tags);
    free(vals);
bad:
    free(tape);
    free(strs);
    return -4;
}
