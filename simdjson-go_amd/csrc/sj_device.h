// sj_device.h -- device-side state shared between the kernels and the host launcher.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace sj {

// Lives at the start of the stage-1 workspace; zeroed (with the tile descriptors that
// follow it) by a hipMemsetAsync node before every launch.
struct Stage1State {
    uint32_t tile_counter;   // dynamic tile id dispenser
    uint32_t error;          // OR of (control char inside string)   -> reference error_mask != 0
    uint64_t total;          // number of structural indexes
    uint32_t ends_in_quote;  // reference prev_iter_inside_quote != 0 at the end
    uint32_t pad[11];
};
static_assert(sizeof(Stage1State) == 64, "Stage1State must stay one 64-byte line");

// stage-2 totals and flags (zeroed before every launch)
struct S2State {
    uint32_t err;            // bit0: stage-2 failure, bit2: tape would exceed 2^32 words
    uint32_t bignum_count;   // numbers queued for the big-integer tie-break
    int32_t final_depth;
    uint32_t records;        // record-separating newline runs (ND)
    unsigned long long tape_len;
    unsigned long long strings_len;
    uint32_t n_br;           // number of bracket tokens (size of the compact bracket view)
    uint32_t pad[7];
};
static_assert(sizeof(S2State) == 64, "S2State must stay one 64-byte line");

size_t stage2_workspace_bytes(size_t n_tokens);
hipError_t stage2_launch(const void *d_msg, size_t len, const uint32_t *d_pos, size_t n, uint32_t flags, void *ws,
                         uint64_t *d_tape, size_t tape_cap, uint8_t *d_strings, size_t strings_cap,
                         hipStream_t stream);

hipError_t stage2_launch_measure(const void *d_msg, size_t len, const uint32_t *d_pos, size_t n, uint32_t flags, void *ws,
                                 hipStream_t stream);
hipError_t stage2_launch_emit(const void *d_msg, size_t len, const uint32_t *d_pos, size_t n, uint32_t flags, void *ws,
                              uint64_t *d_tape, size_t tape_cap, uint8_t *d_strings, size_t strings_cap,
                              uint64_t tape_base, uint64_t strings_base, uint64_t msg_base, hipStream_t stream);

size_t stage1_workspace_bytes(size_t len);
hipError_t stage1_prepare(size_t len, size_t lead, void *ws, hipStream_t stream);
hipError_t stage1_launch_prepared(const void *d_msg, size_t len, int ndjson, uint32_t *d_pos, size_t pos_cap,
                                  void *ws, hipStream_t stream);
hipError_t stage1_launch(const void *d_msg, size_t len, int ndjson, uint32_t *d_pos, size_t pos_cap, void *ws,
                         hipStream_t stream);

}  // namespace sj
