/*
 * sjo.h -- CPU ORACLE for the simdjson-go Parse()/ParseND() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C, scalar restatement of the
 * reference's algorithm (minio/simdjson-go, Go + AVX2 Plan-9 assembly).  It is
 * used solely as the checker by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg.  Nothing in the product path (simdjson-go_amd/) may link,
 * import or call it.
 *
 * Parity pinning: every function below is checked against the reference's own
 * golden vectors (the JSON files under tests/golden, extracted by tools/extract_goldens.py
 * from the reference *_test.go tables).  Number conversion in the reference
 * lives in the Go standard library (strconv.ParseInt/ParseUint/ParseFloat,
 * module std, go 1.22-1.24 per reference go.mod:3); it is restated here with
 * glibc strtod (correctly rounded, round-half-even == strconv.ParseFloat) behind
 * a re-statement of Go's decimal float grammar, and pinned by the reference's
 * number tables (parse_json_amd64_test.go:223-249,287-318,349-502).
 * One behaviour is NOT pinned by any reference vector ("parity unpinned"):
 * hex digits < 0x30 inside \uXXXX (see sjo_parse_string.c, quirk Q3).
 *
 * All file:line citations are relative to the reference repository root.
 */
#ifndef SJO_H
#define SJO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- tape constants: parsed_json.go:26-30, 1076-1094 ---- */
#define SJO_JSONVALUEMASK 0x00ffffffffffffffULL
#define SJO_JSONTAGOFFSET 56
#define SJO_STRINGBUFBIT 0x0080000000000000ULL
#define SJO_INDEX_SIZE 1536                    /* parsed_json.go:74 */
#define SJO_INDEX_SIZE_SAFE (1536 - 128)       /* parsed_json.go:75 */

/* flags for sjo_parse */
#define SJO_FLAG_NDJSON 1u
#define SJO_FLAG_COPY_STRINGS 2u

/* return codes of sjo_parse (parse_json_amd64.go:81,93) */
#define SJO_OK 0
#define SJO_ERR_STAGE1 1 /* "Failed to find all structural indices for stage 1" */
#define SJO_ERR_STAGE2 2 /* "Bad parsing while executing stage 2" */

/* ---- stage-1 per-64-byte routines (find_subroutines_amd64.go wrappers) ---- */
uint64_t sjo_find_odd_backslash_sequences(const uint8_t *in64, uint64_t *prev_iter_ends_odd_backslash);
uint64_t sjo_find_quote_mask_and_bits(const uint8_t *in64, uint64_t odd_ends,
                                      uint64_t *prev_iter_inside_quote, uint64_t *quote_bits,
                                      uint64_t *error_mask);
void sjo_find_whitespace_and_structurals(const uint8_t *in64, uint64_t *whitespace, uint64_t *structurals);
uint64_t sjo_finalize_structurals(uint64_t structurals, uint64_t whitespace, uint64_t quote_mask,
                                  uint64_t quote_bits, uint64_t *prev_iter_ends_pseudo_pred);
uint64_t sjo_find_newline_delimiters(const uint8_t *in64, uint64_t quote_mask);
void sjo_flatten_bits_incremental(uint32_t *base, int *base_index, uint64_t mask, uint64_t *carried,
                                  uint64_t *position);
uint64_t sjo_find_structural_bits(const uint8_t *in64, uint64_t *prev_iter_ends_odd_backslash,
                                  uint64_t *prev_iter_inside_quote, uint64_t *error_mask,
                                  uint64_t *prev_iter_ends_pseudo_pred);
uint64_t sjo_find_structural_bits_in_slice(const uint8_t *buf, uint64_t len,
                                           uint64_t *prev_iter_ends_odd_backslash,
                                           uint64_t *prev_iter_inside_quote, uint64_t *error_mask,
                                           uint64_t *prev_iter_ends_pseudo_pred, uint32_t *indexes,
                                           int *index, uint64_t *carried, uint64_t *position,
                                           uint64_t ndjson);

/* ---- stage-1 driver (stage1_find_marks_amd64.go:41-148) ----
 * Emits ABSOLUTE byte positions (the running sum of the reference's uint32
 * deltas) of every structural index that the reference hands to stage 2.
 * Returns 1 if the reference's findStructuralIndices() returns true. */
int sjo_find_structural_indices(const uint8_t *msg, size_t len, int ndjson, uint32_t *pos_out,
                                size_t pos_cap, size_t *n_out);

/* ---- strings (parse_string_amd64.s) ---- src points at the byte AFTER the opening quote */
int sjo_parse_string_validate_only(const uint8_t *src, size_t avail, uint64_t *str_length,
                                   uint64_t *dst_length);
int sjo_parse_string(const uint8_t *src, size_t avail, uint8_t *dst, uint64_t *dst_length);

/* ---- numbers (parse_number.go:65-135) ---- returns tag word (0 on failure) */
uint64_t sjo_parse_number(const uint8_t *buf, size_t len, uint64_t *val);

/* ---- atoms (stage2_build_tape_amd64.go:124-158) ---- */
int sjo_is_valid_true_atom(const uint8_t *buf, size_t len);
int sjo_is_valid_false_atom(const uint8_t *buf, size_t len);
int sjo_is_valid_null_atom(const uint8_t *buf, size_t len);

/* ---- bytes.TrimSpace (Go std) as used at parse_json_amd64.go:55 ---- */
void sjo_trim_space(const uint8_t *msg, size_t len, size_t *off, size_t *out_len);

/* ---- whole parse: parseMessage (parse_json_amd64.go:52-127) ----
 * Allocates *tape / *strings with malloc (caller frees with sjo_free).  On error
 * the outputs are NULL/0.  msg_off/msg_len describe pj.Message (TrimSpace'd) inside msg. */
int sjo_parse(const uint8_t *msg, size_t len, uint32_t flags, uint64_t **tape, size_t *tape_len,
              uint8_t **strings, size_t *strings_len, size_t *msg_off, size_t *msg_len);
void sjo_free(void *p);

/* ---- sjo_fast.c: the same routines with the reference's AVX2 / PCLMULQDQ instruction shapes, for the CPU baseline
 * of bench.py (shapes B1 / B2 / B3).  Bit-identical to the scalar functions above. ---- */
typedef struct sjo_fast sjo_fast;
int sjo_avx2_available(void);
int sjo_find_structural_indices_avx2(const uint8_t *msg, size_t len, int ndjson, uint32_t *pos_out, size_t pos_cap,
                                     size_t *n_out);
sjo_fast *sjo_fast_create(void);
void sjo_fast_destroy(sjo_fast *w);
int sjo_fast_parse(sjo_fast *w, const uint8_t *msg, size_t len, uint32_t flags, int threads, const uint64_t **tape,
                   size_t *tape_len, const uint8_t **strings, size_t *strings_len);
double sjo_bench_stage1(const uint8_t *msg, size_t len, int ndjson, int iters, int avx2, size_t *n_out);
double sjo_bench_parse(const uint8_t *msg, size_t len, uint32_t flags, int threads, int iters, int *rc_out,
                       size_t *tape_len_out);
double sjo_bench_nd_blocks(const uint8_t *msg, size_t len, int threads, size_t block_bytes, int iters, int *failed_out);

/* ---- sjo_marshal.c: Iter.MarshalJSONBuffer on a whole ParsedJson (parsed_json.go:401-556) ---- */
int sjo_format_float(uint64_t bits, char *out40);
int sjo_marshal_json(const uint64_t *tape, size_t n, const uint8_t *strings, const uint8_t *msg, uint8_t **out,
                     size_t *out_len);

/* ---- sjo_serialize.c: Serializer.Serialize / Deserialize, format v3, CompressNone (parsed_serialize.go) ---- */
int sjo_serialize(const uint64_t *tape, size_t tape_len, const uint8_t *strings, size_t strings_len, const uint8_t *msg,
                  size_t msg_len, int dedup, uint8_t **out, size_t *out_len, uint8_t **tags_out, size_t *tags_len,
                  uint8_t **values_out, size_t *values_len, uint8_t **sbuf_out, size_t *sbuf_len);
int sjo_deserialize(const uint8_t *src, size_t src_len, uint64_t **tape_out, size_t *tape_len, uint8_t **strings_out,
                    size_t *strings_len, uint8_t **message_out, size_t *message_len);

#ifdef __cplusplus
}
#endif
#endif
