/*
 * sjo_internal.h -- ORACLE internals shared between sjo_stage2.c and sjo_fast.c (test infrastructure only, see sjo.h).
 */
#ifndef SJO_INTERNAL_H
#define SJO_INTERNAL_H
#include <stddef.h>
#include <stdint.h>

typedef int (*sjo_validate_fn)(const uint8_t *src, size_t avail, uint64_t *str_length, uint64_t *dst_length);
typedef int (*sjo_copy_fn)(const uint8_t *src, size_t avail, uint8_t *dst, uint64_t *dst_length);

typedef struct {
    const uint8_t *msg;
    size_t len;
    int copy_strings;
    uint64_t *tape;
    size_t tape_len, tape_cap;
    uint8_t *strs;
    size_t strs_len, strs_cap;
    uint64_t *scope; /* containingScopeOffset */
    size_t scope_len, scope_cap;
    const uint32_t *pos;
    size_t npos, ipos;
    /* sjo_fast.c: string routines with AVX2 windows (NULL: the scalar ones), and the live feed of the 2-thread shape */
    sjo_validate_fn validate_string;
    sjo_copy_fn copy_string;
    const size_t *live_npos;
    const int *live_done;
} pj_t;

int sjo_unified_machine(pj_t *pj);
/* the string walk of sjo_parse_string.c resumed at (pos, out) */
int sjo_string_walk_from(const uint8_t *src, size_t avail, uint8_t *dst, size_t pos, size_t out, uint64_t *str_length,
                         uint64_t *dst_length);
#endif
