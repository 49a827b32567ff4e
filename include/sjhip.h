/*
 * sjhip.h -- C ABI of libsjhip: the MI355X (gfx950) engine behind the simdjson-go
 * Parse()/ParseND() hot path.
 *
 * This is the drop-in boundary.  The reference selects its backend with build tags
 * (simdjson_amd64.go:1 vs simdjson_other.go:1); a backend has to provide SupportedCPU,
 * Parse, ParseND and ParseNDStream (simdjson_other.go:29-76), all of which funnel into
 *     (*internalParsedJson).parseMessage(msg []byte, ndjson bool) error      parse_json_amd64.go:52
 * whose only outputs are pj.Message (TrimSpace'd alias of the input), pj.Tape []uint64 and
 * pj.Strings.B []byte.  The Go shim (simdjson-go_amd/go/simdjson_hip.go)
 * binds exactly the functions below through cgo; the Python mirror (sjhip package) binds the
 * same symbols through ctypes.
 *
 * Conventions (they mirror the Go<->asm seam of find_subroutines_amd64.go):
 *   - plain pointers + explicit sizes; the library never retains a caller pointer after the
 *     call returns (cgo rule), hence the two-call parse/fetch protocol;
 *   - one sjhip_ctx per concurrent parse (it owns a HIP stream and device arenas that are
 *     recycled across calls -- the role of the reference's `reuse *ParsedJson`);
 *   - functions return 0 on success, SJHIP_ERR_* otherwise; sjhip_last_error() explains.
 */
#ifndef SJHIP_H
#define SJHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sjhip_ctx sjhip_ctx;

/* flags for sjhip_parse: parse_json_amd64.go:58-62 (ndjson) and options.go:13 (WithCopyStrings) */
#define SJHIP_FLAG_NDJSON 1u
#define SJHIP_FLAG_COPY_STRINGS 2u
/* The caller is going to call sjhip_marshal_json on this result: the parse also leaves, on the device, one byte per
 * string entry of the tape saying whether it is an object key (the parser knows: the token behind it is ':'), and
 * MarshalJSON neither recovers that from the token array (three launches) nor needs its counting pass: it becomes one
 * pass over the tape (marshal.hip).  No effect on the result of the parse. */
#define SJHIP_FLAG_KEY_FLAGS 4u

/* return codes */
#define SJHIP_OK 0
#define SJHIP_ERR_STAGE1 1  /* "Failed to find all structural indices for stage 1" parse_json_amd64.go:93 */
#define SJHIP_ERR_STAGE2 2  /* "Bad parsing while executing stage 2"               parse_json_amd64.go:81 */
#define SJHIP_ERR_NODEVICE 3 /* "Host CPU does not meet target specs" analogue     simdjson_amd64.go:43  */
#define SJHIP_ERR_TOOBIG 4  /* plain stage 1: message longer than 4 GiB - 64 (it hands out uint32 positions); whole parse: more than
                             * 2^32 tokens / tape words / bytes of Strings.B, tokens 4 GiB apart, or a message beyond 256 GiB */
#define SJHIP_ERR_ARG 5
#define SJHIP_STREAM_FULL 6  /* sjhip_stream_acquire: every slot holds a block: take a result first */
#define SJHIP_STREAM_EMPTY 7 /* sjhip_stream_next: nothing submitted is outstanding */
#define SJHIP_ERR_STREAM_CLOSED 8 /* the stream has delivered an error: it accepts and delivers nothing more */
#define SJHIP_ERR_HIP (-1)  /* a HIP runtime call failed */

/* ---- backend presence: replaces SupportedCPU() (simdjson_amd64.go:37) -------------------- */
int sjhip_supported(void);      /* 1 iff a gfx950 device is visible */
int sjhip_device_count(void);

/* ---- context ------------------------------------------------------------------------------ */
sjhip_ctx *sjhip_ctx_create(int device);          /* NULL if the device is unusable */
void sjhip_ctx_destroy(sjhip_ctx *ctx);
const char *sjhip_last_error(const sjhip_ctx *ctx);
/* run the context's work on an existing HIP stream (e.g. torch's current stream); NULL = own stream */
int sjhip_ctx_set_stream(sjhip_ctx *ctx, void *hip_stream);
/* A context's arenas only grow (they are the capacity a recycled `reuse *ParsedJson` carries, simdjson_amd64.go:46-51),
 * sized by the largest message it has parsed.  sjhip_ctx_device_bytes reports what it holds on the device right now;
 * sjhip_ctx_trim gives all of it back (device arenas, the pinned result blocks, the contexts of a sharded ND parse) and
 * drops the resident result -- a pool calls it on a context that has just parsed an unusually large message.  The
 * next parse allocates what it needs. */
size_t sjhip_ctx_device_bytes(const sjhip_ctx *ctx);
int sjhip_ctx_trim(sjhip_ctx *ctx);

/* ---- whole parse: replaces parseMessage (parse_json_amd64.go:52-127) ------------------------
 * msg is a HOST buffer.  The library applies bytes.TrimSpace (parse_json_amd64.go:55) and reports
 * the trimmed window so that the caller can alias pj.Message = msg[msg_off : msg_off+msg_len].
 * On SJHIP_OK the tape/strings stay on the device until sjhip_fetch copies them into
 * caller-owned memory of at least tape_len*8 / strings_len bytes. */
int sjhip_parse(sjhip_ctx *ctx, const uint8_t *msg, size_t len, uint32_t flags, size_t *tape_len,
                size_t *strings_len, size_t *msg_off, size_t *msg_len);
int sjhip_fetch(sjhip_ctx *ctx, uint64_t *tape_dst, uint8_t *strings_dst);
/* The same result WITHOUT the copy into caller memory: *tape / *strings point at tape_len words / strings_len bytes in
 * pinned host memory that the context owns, valid until the next call that parses on this context (or destroys it).
 * This is what `reuse *ParsedJson` means in the reference (simdjson_amd64.go:46-51: the arrays of the recycled
 * ParsedJson are overwritten by the next parse): a binding that keeps one context per recycled ParsedJson hands these
 * pointers out as pj.Tape / pj.Strings.  A small document parsed by sjhip_parse is
 * already there (its last kernel wrote the result over PCIe); anything else is copied device -> pinned block here
 * (the block grows on demand, SJHIP_ERR_TOOBIG beyond SJHIP_VIEW_LIMIT_BYTES, default 4 GiB: use sjhip_fetch).
 * Either pointer is NULL when its length is 0. */
int sjhip_fetch_view(sjhip_ctx *ctx, const uint64_t **tape, const uint8_t **strings);

/* A pinned host block of at least `bytes` bytes owned by the context, for callers that can read their input (a file, a
 * socket) straight into it -- the role of the reference's tmpPool blocks in ParseNDStream (simdjson_amd64.go:127-135) for
 * a single Parse: sjhip_parse(ctx, block, len, ...) then copies host -> device at the pinned rate (twitter.json: 19
 * instead of 28 us).  Valid until the next sjhip_input_block call with a larger size, sjhip_ctx_trim or destroy. */
uint8_t *sjhip_input_block(sjhip_ctx *ctx, size_t bytes);

/* Same parse on a message that is already resident in device memory (already trimmed). Used by
 * bench.py (inputs in HBM before the timed region) and by the multi-GPU shard path. */
int sjhip_parse_device(sjhip_ctx *ctx, const void *d_msg, size_t len, uint32_t flags, size_t *tape_len,
                       size_t *strings_len);

/* ---- one NDJSON shard of a larger document: the multi-GPU ParseND path -------------------------------
 * ParseND's records are independent (simdjson_amd64.go:82, and ParseNDStream parses 10 MiB blocks on their own,
 * :156-192), so a document cut at record boundaries is parsed shard by shard, one shard per GPU.  The merged
 * ParsedJson is the concatenation of the shard tapes / Strings.B, provided every index a shard's tape stores is
 * rebased by where the shard starts in the merged Tape / Strings.B / Message.  Those three offsets are the
 * exclusive prefix sums of the preceding shards' sizes -- the only data the shards exchange (8+8 bytes per rank).
 *   begin : stage 1 + stage 2 up to the scan; returns this shard's tape_len / strings_len (message already on
 *           the device, trimmed, starting at a record boundary)
 *   ...   : all-gather the sizes (RCCL), compute the bases
 *   finish: emits tape and Strings.B with the rebased indices; then sjhip_fetch as usual. */
int sjhip_parse_shard_begin(sjhip_ctx *ctx, const void *d_msg, size_t len, uint32_t flags, size_t *tape_len,
                            size_t *strings_len);
int sjhip_parse_shard_finish(sjhip_ctx *ctx, uint64_t tape_base, uint64_t strings_base, uint64_t msg_base);
/* ---- ParseND over several GPUs in one call (simdjson_amd64.go:82-94 is one call in one process) -------------------
 * A handle owns one context per entry of `devices` (NULL / 0 = every visible device; a device may be listed more than
 * once: several shards on one GPU).  sjhip_parse_nd_multi cuts the HOST message at record boundaries into one shard per
 * entry, runs the two-phase shard parse above on all of them in parallel (one host thread per shard; the sizes meet in
 * a host prefix sum of 16 bytes per shard -- no device collective) and sjhip_fetch_multi copies every shard's piece
 * straight into its slice of the caller's Tape / Strings.B: the result is bit for bit the ParsedJson of ParseND on
 * the whole message.  A stage-1 failure of any shard wins over stage-2 failures (parse_json_amd64.go:97-105,123-126). */
typedef struct sjhip_multi sjhip_multi;
sjhip_multi *sjhip_multi_create(const int *devices, int n);
void sjhip_multi_destroy(sjhip_multi *m);
int sjhip_multi_shards(const sjhip_multi *m);
/* the device that holds the tape of shard `shard` after a parse, as the HIP runtime reports it for that allocation
 * (hipPointerGetAttributes), -1 if the shard has parsed nothing yet: lets a caller (and the tests) see that the shards
 * really sit on the devices they were asked for */
int sjhip_multi_shard_device(const sjhip_multi *m, int shard);
const char *sjhip_multi_last_error(const sjhip_multi *m);
int sjhip_parse_nd_multi(sjhip_multi *m, const uint8_t *msg, size_t len, uint32_t flags, size_t *tape_len,
                         size_t *strings_len, size_t *msg_off, size_t *msg_len);
int sjhip_fetch_multi(sjhip_multi *m, uint64_t *tape_dst, uint8_t *strings_dst);
/* ---- many documents, one launch set (the goroutine-per-Parse shape of benchmarks_test.go:60-75, batched) ------------
 * The documents are packed into one device message -- each trimmed like Parse() trims it (parse_json_amd64.go:55),
 * separated by '\n', a raw '\n' INSIDE a document replaced by '\r' (whitespace either way outside strings, the same
 * stage-1 error inside one; "1\n2" stays the error it is in Parse()) -- and parsed as one ND document.  The result is
 * what ParseND of that message returns: document i is root i of the tape (Iter.Advance walks them), all string words
 * point into one Strings.B; sjhip_fetch and every query / serializer call work on it.  An empty document or any invalid
 * one fails the whole batch with the code Parse() of that document returns (stage 1 before stage 2) -- including the
 * end-of-message rule of stage 1 (the last structural must close a container, stage1_find_marks_amd64.go:115-129), to
 * which every document is held while the batch is packed: a scalar, a truncated or an all-whitespace document is
 * SJHIP_ERR_STAGE1 wherever it stands.  Needs SJHIP_FLAG_COPY_STRINGS.
 * sjhip_parse_batch_device: the documents lie in ONE device buffer at offs[i] (lens[i] bytes, taken untrimmed: JSON
 * whitespace around a document is whitespace of its record). */
int sjhip_parse_batch(sjhip_ctx *ctx, const uint8_t *const *msgs, const size_t *lens, size_t n, uint32_t flags,
                      size_t *tape_len, size_t *strings_len);
int sjhip_parse_batch_device(sjhip_ctx *ctx, const void *d_buf, const size_t *offs, const size_t *lens, size_t n,
                             uint32_t flags, size_t *tape_len, size_t *strings_len);
/* bytes.TrimSpace exactly as parseMessage applies it (parse_json_amd64.go:55); for hosts that are not Go */
void sjhip_trim_space(const uint8_t *msg, size_t len, size_t *off, size_t *out_len);

/* ---- queries on the device-resident result (no reference counterpart in the parser: they replace what callers do with
 * Iter / Object.FindKey on the host, ndjson_test.go:421-471, parsed_object.go:97-138, README.md:226-269) ------------
 * Both work on the result of the last successful sjhip_parse / sjhip_parse_device of `ctx` (still on the device).
 * Results larger than one context (round 6): after an ND message beyond 4 GiB -- parsed shard by shard, every shard resident on
 * its own context -- sjhip_count_where, the path / key-set queries below and sjhip_marshal_json run shard by shard and return
 * what they return on the merged ParsedJson (counts added up, per-record answers in document order holding indexes of the MERGED
 * tape, texts joined with the newline between two records), like the reference's Iter on any ParsedJson (parsed_json.go:96,125,833);
 * sjhip_filter_where and sjhip_serialize need the result of one context and say so (SJHIP_ERR_ARG).
 * A record matches when its root value is an object whose FIRST member with key == `key` (top level only, like
 * Object.FindKey) has a string value == `value` (compared after unescaping) -- the reference's countWhere.
 *   count_where : number of matching records; 8 bytes cross PCIe.
 *   filter_where: compacts the matching records into a new self-contained (Tape, Strings.B) on the device, identical
 *                 to ParseND of the document made of the matching lines (root chain re-linked, container / string
 *                 offsets rebased); sjhip_fetch_filtered copies it to the host.  Needs SJHIP_FLAG_COPY_STRINGS. */
int sjhip_count_where(sjhip_ctx *ctx, const uint8_t *key, size_t klen, const uint8_t *value, size_t vlen, uint64_t *count);
int sjhip_filter_where(sjhip_ctx *ctx, const uint8_t *key, size_t klen, const uint8_t *value, size_t vlen,
                       uint64_t *n_records, size_t *tape_len, size_t *strings_len);
int sjhip_fetch_filtered(sjhip_ctx *ctx, uint64_t *tape_dst, uint8_t *strings_dst);

/* Paths, typed values and key sets on the same device-resident result (round 5).  A path is n_keys keys, concatenated in
 * `keys`, key j being key_lens[j] bytes long (at most 16 keys, 1024 bytes together); every call evaluates it on the root
 * value of EVERY record (one record for a plain document) with the semantics of the reference's host API:
 *   sjhip_find_path        Iter.FindElement(path...) (parsed_json.go:833-865) = Object.FindPath (parsed_object.go:256-313):
 *                          into the root and into objects, not into arrays; at every level the first member with the key
 *                          wins.  index_out[r] = tape index of the element's value (tape[index] is its tag word), or
 *                          SJHIP_PATH_NOT_FOUND (ErrPathNotFound) or SJHIP_PATH_NOT_OBJECT (the root value, or the value
 *                          of a key that is not the last one, is not an object: the reference's type errors).
 *                          *records = number of records; cap = room in index_out (records).
 *   sjhip_count_where_path number of records whose element at `path` exists and satisfies `op`:
 *                          EXISTS; EQ_STRING (value = vlen bytes, compared after unescaping, Iter.StringBytes);
 *                          EQ_INT / EQ_UINT / EQ_FLOAT (value = an int64_t / uint64_t / double, vlen 8; the element is
 *                          converted the way Iter.Int / Uint / Float convert between the three number tags,
 *                          parsed_json.go:560-727 -- with the amd64 results at the two edges the reference lets through:
 *                          a float of exactly 2^63 is MinInt64 for EQ_INT, one of exactly 2^64 is 0 for EQ_UINT); EQ_BOOL (value = one byte); IS_NULL.  8 bytes cross PCIe.
 *   sjhip_project_keys     Object.ForEach(fn, onlyKeys) (parsed_object.go:142-196) on the root object of every record: the
 *                          members whose key is in the set, in document order, at most n_keys of them (the reference stops
 *                          after len(onlyKeys) deliveries).  out[r * n_keys + j] = key number << 56 | tape index of the
 *                          value of the j-th delivered member, ~0 when there is no j-th.  The keys must be distinct. */
#define SJHIP_PATH_NOT_FOUND (~0ull)
#define SJHIP_PATH_NOT_OBJECT (~0ull - 1ull)
enum { SJHIP_OP_EXISTS = 0, SJHIP_OP_EQ_STRING = 1, SJHIP_OP_EQ_INT = 2, SJHIP_OP_EQ_UINT = 3, SJHIP_OP_EQ_FLOAT = 4,
       SJHIP_OP_EQ_BOOL = 5, SJHIP_OP_IS_NULL = 6 };
int sjhip_find_path(sjhip_ctx *ctx, const uint8_t *keys, const uint32_t *key_lens, uint32_t n_keys, uint64_t *index_out,
                    size_t cap, size_t *records);
int sjhip_count_where_path(sjhip_ctx *ctx, const uint8_t *keys, const uint32_t *key_lens, uint32_t n_keys, int op,
                           const void *value, size_t vlen, uint64_t *count);
int sjhip_project_keys(sjhip_ctx *ctx, const uint8_t *keys, const uint32_t *key_lens, uint32_t n_keys, uint64_t *out,
                       size_t cap_records, size_t *records);

/* ---- Serializer.Serialize on the device (parsed_serialize.go:200-431, format version 3) -----------------------------
 * Splits the device-resident tape of the last parse (SJHIP_FLAG_COPY_STRINGS) into the reference's three columns --
 * tags (one byte per tape entry), values (8 / 16 bytes per value-bearing entry), strings (= Strings.B, the reference's
 * string buffer without de-duplication hits) -- and frames them as a CompressNone stream (every block type 0), which
 * the reference's Deserialize reads; S2 / zstd compression of the columns (CompressFast / Default / Best) is host work.
 *   serialize        : builds the columns on the device; sizes of the columns and of the framed stream
 *   fetch_serialized : writes the framed stream into `dst` (>= stream_len bytes): header varints from the host, the
 *                      three columns straight from the device */
int sjhip_serialize(sjhip_ctx *ctx, size_t *tags_len, size_t *values_len, size_t *strings_len, size_t *stream_len);
/* SJHIP_SER_DEDUP: strings are de-duplicated like the reference's indexString (parsed_serialize.go:836-857) does -- a
 * string equal to the first string of the document in its hash slot is stored once (2^20 slots; deterministic, where the
 * reference's table is keyed by a per-process random hash) -- and the string column holds the kept strings only. */
#define SJHIP_SER_DEDUP 1u
int sjhip_serialize_ex(sjhip_ctx *ctx, uint32_t flags, size_t *tags_len, size_t *values_len, size_t *strings_len,
                       size_t *stream_len);
int sjhip_fetch_serialized(sjhip_ctx *ctx, uint8_t *dst, size_t cap, size_t *len);
/* Serializer.Deserialize (parsed_serialize.go:466-695) of a stream whose blocks are uncompressed (what
 * sjhip_fetch_serialized writes; S2 / zstd blocks are decompressed on the host first): the tape is rebuilt on the device
 * from the tag and value columns (two prefix sums and one scatter pass; closing brackets from their openers).  Like the
 * reference's result the strings point into pj.Message (= the string column) and Strings.B is what the stream carried
 * (empty for version 3).  sjhip_fetch copies Tape / Strings.B, sjhip_fetch_message pj.Message (message_len bytes). */
int sjhip_deserialize(sjhip_ctx *ctx, const uint8_t *stream, size_t len, size_t *tape_len, size_t *strings_len,
                      size_t *message_len);
int sjhip_fetch_message(sjhip_ctx *ctx, uint8_t *dst);

/* ---- Iter.MarshalJSON on the device (parsed_json.go:401-556) ---------------------------------------------------------
 * The device-resident result of the last parse as compact JSON text, records separated by '\n' -- byte for byte what
 * pj.Iter().MarshalJSON() returns: escapeBytes for strings, strconv.AppendInt / AppendUint, appendFloat (the reference's
 * copy of Go's Ryu shortest formatting with its ES6-style %f / %e choice).  The text stays on the device until
 * sjhip_fetch_marshaled copies it into `dst` (>= text_len bytes).  SJHIP_ERR_TOOBIG for a document of 4 GiB or more that was
 * parsed without SJHIP_FLAG_COPY_STRINGS (the strings that are not copied then lie at message offsets beyond 32 bits). */
int sjhip_marshal_json(sjhip_ctx *ctx, size_t *text_len);
int sjhip_fetch_marshaled(sjhip_ctx *ctx, uint8_t *dst);

/* ---- ParseNDStream: replaces the block pipeline of simdjson_amd64.go:101-216 --------------------------------------
 * The binding cuts the input into blocks that end at a record boundary (simdjson_amd64.go:155-176; tmpSize = 10 MiB)
 * and feeds them to a stream; every block is parsed as an independent NDJSON document with every string copied
 * (:180) and the results come back in submission order.  A stream owns `slots` blocks in flight, spread round robin
 * over `n_devices` devices starting at `first_device` (0 devices = all that are visible; 0 slots = 3 per device);
 * every slot has its own context, HIP stream, pinned input block and pinned result buffers, so the H2D copy of one
 * block, the kernels of another and the D2H copy of a third overlap.
 *   acquire  : a pinned block of sjhip_stream_block_capacity() bytes to read the input into (the reference's tmpPool);
 *              SJHIP_STREAM_FULL when every slot is busy (take a result first)
 *   grow     : a record that runs past the acquired block: a larger pinned block, the first `keep` bytes kept
 *   submit   : queue the acquired block (its first `len` bytes)            submit_copy = acquire + memcpy + submit
 *   next     : the result of the oldest outstanding block (blocks until it is done).  Tape / Strings.B / Message
 *              point into memory of the stream and stay valid until sjhip_stream_release (copy them into the
 *              caller's slices: the reference's `reuse` recycling is the caller's side of that copy).  A block that
 *              fails returns its error code (SJHIP_ERR_STAGE1 / _STAGE2 / ...), which ends the stream like the
 *              reference's first Stream{Error}: later calls return SJHIP_ERR_STREAM_CLOSED.  SJHIP_STREAM_EMPTY when
 *              nothing is outstanding (the caller reports io.EOF once its reader is exhausted).
 *   ready    : 1 if the oldest outstanding block has finished, i.e. sjhip_stream_next would return at once, else 0
 *              (a single-threaded caller drains finished blocks with it before every blocking read of its input,
 *              so that results are not withheld while the reader waits for more data)
 * One thread may submit while another takes results. */
typedef struct sjhip_stream sjhip_stream;
typedef struct sjhip_stream_result {
    const uint64_t *tape;
    size_t tape_len;
    const uint8_t *strings;
    size_t strings_len;
    const uint8_t *message; /* TrimSpace'd block (inside the pinned input block) */
    size_t message_len;
    int device;
    uint64_t records;       /* filtered streams (sjhip_stream_set_filter): matching records of the block, else 0 */
} sjhip_stream_result;
sjhip_stream *sjhip_stream_create(int first_device, int n_devices, size_t block_bytes, int slots, uint32_t flags);
void sjhip_stream_destroy(sjhip_stream *s);
size_t sjhip_stream_block_capacity(const sjhip_stream *s);
int sjhip_stream_slots(const sjhip_stream *s);
int sjhip_stream_in_flight(sjhip_stream *s);
const char *sjhip_stream_last_error(const sjhip_stream *s);
int sjhip_stream_acquire(sjhip_stream *s, uint8_t **block, size_t *capacity);
int sjhip_stream_grow(sjhip_stream *s, size_t keep, size_t new_capacity, uint8_t **block);
int sjhip_stream_submit(sjhip_stream *s, size_t len);
int sjhip_stream_cancel(sjhip_stream *s); /* hand the acquired block back unused */
int sjhip_stream_submit_copy(sjhip_stream *s, const uint8_t *block, size_t len);
int sjhip_stream_next(sjhip_stream *s, sjhip_stream_result *out);
int sjhip_stream_ready(sjhip_stream *s);
/* Compose the stream with sjhip_filter_where: every block is parsed and filtered on the device and only the matching
 * records' (Tape, Strings.B) -- identical to ParseND of the block's matching lines -- cross PCIe; result.records counts
 * them (a block without matches delivers the empty result: tape_len 0).  Set before the first block is submitted;
 * klen = 0 turns the filter off. */
int sjhip_stream_set_filter(sjhip_stream *s, const uint8_t *key, size_t klen, const uint8_t *value, size_t vlen);
int sjhip_stream_release(sjhip_stream *s);

/* ---- stage 1 only: replaces findStructuralIndices (stage1_find_marks_amd64.go:41-148) --------
 * pos_out receives ABSOLUTE uint32 byte positions (running sum of the reference's deltas).
 * *ok = the reference's return value (error_mask == 0 && indexTotal > 0 && end-of-doc checks). */
int sjhip_stage1(sjhip_ctx *ctx, const uint8_t *msg, size_t len, int ndjson, uint32_t *pos_out,
                 size_t pos_cap, size_t *n, int *ok);
/* device-resident variant: d_pos must hold pos_cap uint32; nothing is copied back but the counts */
int sjhip_stage1_device(sjhip_ctx *ctx, const void *d_msg, size_t len, int ndjson, void *d_pos,
                        size_t pos_cap, size_t *n, int *ok);
/* Queued form of sjhip_stage1_device for a caller that keeps several messages (or blocks of a stream) in flight, as
 * ParseNDStream's reader does with its 10 MB blocks (simdjson_amd64.go:127-215: the next block is read and indexed while the
 * previous one is parsed): _queue launches behind what is already on the context's stream and returns at once; the launch
 * leaves count, end state and error bits in record `slot` (0 .. SJHIP_STAGE1_QUEUE_SLOTS-1) of the context's pinned host
 * memory.  _wait synchronises with the stream; _result turns a record into (*n, *ok) exactly like sjhip_stage1_device and must
 * only be called for a slot whose launch _wait has covered (else SJHIP_ERR_HIP "left no result").  A slot is free again once
 * its result has been taken.  d_msg / d_pos of a queued launch must stay untouched until then. */
#define SJHIP_STAGE1_QUEUE_SLOTS 64
int sjhip_stage1_device_queue(sjhip_ctx *ctx, const void *d_msg, size_t len, int ndjson, void *d_pos,
                              size_t pos_cap, int slot);
int sjhip_stage1_device_wait(sjhip_ctx *ctx);
int sjhip_stage1_device_result(sjhip_ctx *ctx, int slot, size_t len, size_t *n, int *ok);
/* launches the stage-1 kernel `iters` times back to back on the context's stream and returns the
 * average kernel duration in milliseconds measured with hipEvents on that stream */
int sjhip_stage1_time(sjhip_ctx *ctx, const void *d_msg, size_t len, int ndjson, void *d_pos,
                      size_t pos_cap, int iters, float *ms_per_launch);

/* ---- profiling aids (not needed by a binding) -------------------------------------------------------------------
 * sjhip_stage1_set_variant: kernel variant used by this process for stage 1 (A/B runs on hardware): 0 512-thread
 *   blocks with barriers, 1 1024 with barriers (default), 2 768 with barriers, 3 1024 barrier-free with 2 tiles in
 *   flight per block, 4 the same with 3;
 *   -1 = the SJHIP_S1_VARIANT environment variable or the default.  Returns the variant in effect.
 * sjhip_stage1_trace: one stage-1 launch of a profiling build of the current variant that stamps s_memtime at the
 *   phase boundaries of every (tile, wave): trace_out[(tile * waves + wave) * words + k], k = 0 phase A begins,
 *   1 phase A done, 2 serial section done (only the wave that ran it), 3 state of the tile known, 4 flatten done,
 *   5 HW_ID | XCC_ID << 32.  Plain (non-ND) stage 1 of a device-resident message. */
int sjhip_stage1_set_variant(int variant);
/* The library can be built with -DSJ_DEBUG_BOUNDS (csrc/sj_bounds.h; __graft_entry__.build_lib(debug_bounds=True) ->
 * libsjhip_dbg.so): every array of the parse path is then reached through a bounds-checked view, and a parse during
 * which a kernel touched an element outside its array fails with SJHIP_ERR_HIP ("bounds check: ...").
 * sjhip_debug_bounds_selftest: -1 in the product build; in the debug build it runs a kernel with two deliberate
 * violations and returns how many were recorded (2). */
int sjhip_debug_bounds_selftest(void);
int sjhip_stage1_trace(sjhip_ctx *ctx, const void *d_msg, size_t len, void *d_pos, size_t pos_cap, uint64_t *trace_out,
                       size_t trace_cap_words, unsigned *tiles, int *waves, int *words);

/* ---- per-routine known-answer entry points (one 64-byte chunk, executed on the GPU) ----------
 * Same signatures as the Go wrappers in find_subroutines_amd64.go so that the reference's
 * per-routine tests (find_subroutines_amd64_test.go) can be replayed against the device code. */
int sjhip_find_odd_backslash_sequences(sjhip_ctx *ctx, const uint8_t in[64],
                                       uint64_t *prev_iter_ends_odd_backslash, uint64_t *odd_ends);      /* :86  */
int sjhip_find_quote_mask_and_bits(sjhip_ctx *ctx, const uint8_t in[64], uint64_t odd_ends,
                                   uint64_t *prev_iter_inside_quote, uint64_t *quote_bits,
                                   uint64_t *error_mask, uint64_t *quote_mask);                          /* :60  */
int sjhip_find_whitespace_and_structurals(sjhip_ctx *ctx, const uint8_t in[64], uint64_t *whitespace,
                                          uint64_t *structurals);                                        /* :206 */
int sjhip_finalize_structurals(sjhip_ctx *ctx, uint64_t structurals, uint64_t whitespace,
                               uint64_t quote_mask, uint64_t quote_bits,
                               uint64_t *prev_iter_ends_pseudo_pred, uint64_t *out);                     /* :35  */
int sjhip_find_newline_delimiters(sjhip_ctx *ctx, const uint8_t in[64], uint64_t quote_mask,
                                  uint64_t *mask);                                                       /* :40  */
int sjhip_flatten_bits_incremental(sjhip_ctx *ctx, uint32_t *base, int *base_index, uint64_t mask,
                                   uint64_t *carried, uint64_t *position);                               /* :229 */

#ifdef __cplusplus
}
#endif
#endif
