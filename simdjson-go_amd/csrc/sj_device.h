// sj_device.h -- device-side state shared between the kernels and the host launcher.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace sj {

// Lives at the start of the stage-1 workspace; two sets of tile descriptors behind it.  Round 6: no preparation kernel in front
// of a launch any more.  The launches on a workspace are counted (their owner keeps the epoch); launch k works on control
// slot k & 1 and on descriptor set k & 1 -- both zero when it begins -- and zeroes slot and set (k + 1) & 1, which launch
// k - 1 used and nobody looks at any more: cleaning needs no ordering inside the launch, every block does a slice when it
// starts.  The result words are overwritten by every launch.  A fresh workspace is zeroed once (epoch 0).
struct Stage1Ctrl {
    uint32_t tile_counter;   // dynamic tile id dispenser
    uint32_t error;          // OR of (control char inside string)   -> reference error_mask != 0; bit 31: a bounded spin ran out
    uint32_t has_starter;    // whole parse: some unit holds a backslash that starts an escape (stage 2 reads it on the device:
                             // WithCopyStrings(false) of a message without one copies nothing and measures nothing)
    uint32_t pad;
};
struct Stage1State {
    Stage1Ctrl c[2];
    uint64_t total;          // number of structural indexes
    uint32_t ends_in_quote;  // reference prev_iter_inside_quote != 0 at the end
    uint32_t error;          // host copy only: c[].error as the host records deliver it
    uint32_t last_byte;      // host copy only: msg[len - 1] (the end-of-document verdict needs it)
    uint32_t pad[3];
};
static_assert(sizeof(Stage1State) == 64, "Stage1State must stay one 64-byte line");
// The host record of stage 1: three 8-byte words of pinned host memory, zeroed by the caller.  Word 0 is stored by the block that
// flattens the last tile (count, state at the end, the last message byte); word 1 / word 2 are set to 1 by any block that met a
// control character inside a string / whose bounded spin ran out.  The host reads them after it has synchronised with the stream.
static constexpr uint64_t S1_HOST_VALID = 1ull << 63, S1_HOST_IN_QUOTE = 1ull << 60, S1_HOST_TOTAL_MASK = (1ull << 40) - 1;
static constexpr int S1_HOST_LAST_SHIFT = 48;  // 8 bits: msg[len - 1]
static constexpr int S1_HOST_WORDS = 3;

// stage-2 totals and flags (zeroed before every launch)
struct S2State {
    uint32_t err;            // bit0: stage-2 failure, bit2: tape would exceed 2^32 words, bit3: internal scan timeout,
                             // bit4: S2_ERR_SERIAL_STRINGS
    uint32_t bignum_count;   // numbers queued for the big-integer tie-break
    int32_t final_depth;
    uint32_t records;        // record-separating newline runs (ND)
    unsigned long long tape_len;
    unsigned long long strings_len;
    uint32_t n_br;           // number of bracket tokens (size of the compact bracket view)
    uint32_t tail_mask;      // allowed contexts of the gap behind the last bracket (sj_stage2.h)
    unsigned long long strings_len_masks;  // Strings.B length according to the emit masks (both copy modes; without masks: strings_len)
    uint32_t num_count;      // number tokens queued for k_numbers
    uint32_t str_count;      // (unused since round 5)
    uint32_t n_strings;      // every string copied: opening quotes of the message (the unit scan's second total)
    uint32_t pad[1];
};
static_assert(sizeof(S2State) == 64, "S2State must stay one 64-byte line");
// a run of more than SURROGATE_WALK_CAP adjacent high-surrogate escapes (sj_strings.h): the byte-parallel string
// path gives up and the host repeats stage 2 with the per-string walks (no verdict is taken from the first run)
static constexpr uint32_t S2_ERR_SERIAL_STRINGS = 16u;

// Everything a stage-2 launch needs.
struct S2Args {
    const void *d_msg;
    size_t len;
    const uint32_t *d_pos;
    const uint8_t *d_kind;  // the token kinds stage 1 wrote next to the positions
    size_t n;               // tokens; with n_dev an upper bound (the arrays and grids are sized for it)
    const unsigned long long *n_dev;  // null, or Stage1State::total on the device: the host did not wait for stage 1
    const uint32_t *s1_has_starter;   // Stage1State::has_starter on the device (null: unknown, assume there are escapes)
    uint32_t flags;
    void *ws_zero;          // stage2_zero_bytes(): zero before the measure phase (stage 1's preparation kernel does it)
    void *ws;               // stage2_workspace_bytes(n)
    uint64_t *d_tape;
    size_t tape_cap;
    uint8_t *d_strings;
    size_t strings_cap;
    uint64_t tape_base, strings_base, msg_base;
    void *str_aux;          // string masks of stage 1 (str_aux_layout) or null: per-string walks
    uint8_t *d_keyflag;     // null, or tape_cap / 2 + 8 bytes: [tape index of a string entry >> 1] = 1 iff it is an object key
                            // (SJHIP_FLAG_KEY_FLAGS; indices inside this parse's tape, without tape_base)
    hipStream_t stream;
};
size_t stage2_zero_bytes();
size_t stage2_workspace_bytes(size_t n_tokens);
hipError_t stage2_launch_measure(const S2Args &a);
hipError_t stage2_launch_emit(const S2Args &a);
hipError_t stage2_launch_bignum(const S2Args &a);
// Behind the emit phase: the 64-byte state to h_dst, and -- if the parse succeeded, needs no bignum pass and
// 8 * tape_len + strings_len fits `cap` -- the tape to h_dst + STAGE2_PACK_HEAD and Strings.B behind it; h_dst[64] (u64)
// says whether the payload is there.  h_dst is pinned host memory mapped into the device (one launch, no copy commands).
constexpr size_t STAGE2_PACK_HEAD = 128;
hipError_t stage2_launch_pack(const S2Args &a, void *h_dst, size_t cap);
// the 64-byte state alone to h_dst (pinned, device-mapped): a one-wave kernel instead of a device-to-host copy command
hipError_t stage2_launch_state_out(const S2Args &a, void *h_dst);

// debug build (-DSJ_DEBUG_BOUNDS, sj_bounds.h): 1 and the record of the out-of-bounds accesses since the last call (cleared);
// 0 in the product build.  _selftest: -1 in the product build, else the violations recorded for two deliberate ones
int stage2_debug_bounds(unsigned *hits, unsigned *id, unsigned long long *index, unsigned long long *size);
int stage1_debug_bounds(unsigned *hits, unsigned *id, unsigned long long *index, unsigned long long *size);  // stage1.hip
int stage2_debug_bounds_selftest();

size_t stage1_workspace_bytes(size_t len);
// A stage-1 workspace and what its owner keeps about it: the launches on it are counted (Stage1State above).
struct S1Ws {
    void *p = nullptr;
    size_t bytes = 0;
    unsigned epoch = 0;       // launches since the workspace was zeroed
    unsigned prev_tiles = 0;  // descriptors the last launch used (the next one zeroes them)
};
// Zeroes a workspace whose contents are unknown and resets its epoch (a parse does not need it: every launch leaves the
// workspace ready for the next one).  zero2 / zero2_bytes: a second region to zero, or null
hipError_t stage1_prepare(S1Ws &ws, hipStream_t stream, void *zero2 = nullptr, size_t zero2_bytes = 0);
// String-mask workspace of the whole parse (copy_strings): stage 1 fills qm / st / unit_h, the string
// kernels of stage 2 add one 16-byte record per chunk (sj_strings.h ChunkRec) and unit_cnt.  `span` = lead + len (bytes from the 64-byte aligned
// base of the message); everything is sized in whole 4 KiB units.
struct StrAux {
    size_t units, chunks, bytes;
    uint64_t *qm, *st;
    void *rec;  // ChunkRec[chunks]
    uint32_t *unit_cnt;
    uint32_t *tile_unit;  // [units + 1] the unit that holds token 4096 T (stage1.hip; positions wrap beyond 4 GiB)
    uint32_t *unit_str;   // per unit: strings that begin in it, then (k_scans) their exclusive prefix (every string copied)
    uint8_t *unit_h;
    uint64_t *unit_slow;  // per unit: chunks that hold an escaped character other than " \\ / b f n r t (a \u or an invalid escape)
    uint32_t *unit_tq;    // per unit, WithCopyStrings(false): aligned offset of the closing quote of the string that is open at the unit's end
    uint8_t *unit_copy;   // per unit, WithCopyStrings(false): the states at the unit's ends of the byte-parallel selective copy (stage2.hip USEL_*)
};
inline StrAux str_aux_layout(void *buf, size_t span) {
    StrAux a;
    a.units = (span + 4095) / 4096 + 1;
    a.chunks = a.units * 64;
    char *w = reinterpret_cast<char *>(buf);
    auto carve = [&](size_t n) {
        char *r = w;
        w += (n + 255) / 256 * 256;
        return r;
    };
    a.qm = reinterpret_cast<uint64_t *>(carve(a.chunks * 8));
    a.st = reinterpret_cast<uint64_t *>(carve(a.chunks * 8));
    a.rec = carve(a.chunks * 16);
    a.unit_cnt = reinterpret_cast<uint32_t *>(carve(a.units * 4));
    a.unit_str = reinterpret_cast<uint32_t *>(carve(a.units * 4));
    a.tile_unit = reinterpret_cast<uint32_t *>(carve((a.units + 1) * 4));
    a.unit_h = reinterpret_cast<uint8_t *>(carve(a.units));
    a.unit_slow = reinterpret_cast<uint64_t *>(carve(a.units * 8));
    a.unit_copy = reinterpret_cast<uint8_t *>(carve(a.units));
    a.unit_tq = reinterpret_cast<uint32_t *>(carve(a.units * 4));
    a.bytes = (size_t)(w - reinterpret_cast<char *>(buf));
    return a;
}
inline size_t str_aux_bytes(size_t span) { return str_aux_layout(nullptr, span).bytes + 256; }
// `ndjson` of the launch functions: bit 0 NDJSON | S1_WANT_STARTER_FLAG (whole parse with WithCopyStrings(false): the kernel also
// leaves Stage1State::has_starter -- one look and at most one store per block; nobody else pays for it)
static constexpr int S1_WANT_STARTER_FLAG = 0x100;
// aux_buf: string masks for the whole parse (str_aux_layout); d_kind: [pos_cap] token kinds next to the positions.
// h_state: S1_HOST_WORDS 8-byte words of pinned host memory (device-visible, zeroed by the caller): the host record -- the
// host needs a stream synchronisation but no copy.  zero2: a region the launch zeroes for the kernels behind it (the stage-2
// state of this parse).  d_trace: profiling builds of the kernel (stage1_trace_words() zeroed u64)
hipError_t stage1_launch(const void *d_msg, size_t len, int ndjson, uint32_t *d_pos, size_t pos_cap, S1Ws &ws,
                         hipStream_t stream, void *aux_buf = nullptr, uint8_t *d_kind = nullptr,
                         unsigned long long *h_state = nullptr, void *zero2 = nullptr, size_t zero2_bytes = 0,
                         unsigned long long *d_trace = nullptr);
// kernel variant for A/B runs (-1: SJHIP_S1_VARIANT or the default); per-phase trace size of one launch
int stage1_set_variant(int v);
size_t stage1_trace_words(size_t len, size_t lead, unsigned *tiles_out, int *waves_out);

}  // namespace sj
