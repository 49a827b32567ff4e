// sj_bounds.h -- bounds-checked views of the device arenas for the debug build (-DSJ_DEBUG_BOUNDS).
//
// The stage-1 kernels write positions, kinds and the string masks through such views as well (stage1.hip).
// The kernels of stage2.hip reach every array of the parse -- message, positions, kinds, the stage-2 work arrays, the
// string masks and records, tape and Strings.B -- through the fields of S2Dev / StrView.  Those fields are declared as
// Arr<T>: in the product build that is a plain T* (no code changes, no cost); in the debug build it is a pointer with
// its element count, every a[i] is checked, and every place that forms a pointer for a wider access (a 16-byte load
// of four positions, a window of the message handed to the number parser) states how many elements it is going to
// touch: arr_at(a, first, count).  A violation is recorded (array id, index, size; the first one wins) and the access
// is redirected to an element inside the array, so the kernel finishes, nothing outside the arena is touched, and the
// host turns the record into an error of the parse (parse_api.hip) -- a test suite run under this build
// (tools/gpu_debug_bounds.sh) fails on the first out-of-bounds access anywhere in the parse path instead of relying on
// the access landing on an unmapped page.  SURVEY.md section 5 (sanitizer / race-detection row).
#pragma once
#include <stdint.h>

#include <type_traits>

#include "sj_chunk.h"  // SJ_HD

namespace sj {

// array ids (reported with a hit)
enum ArrId : uint32_t {
    A_NONE = 0, A_MSG, A_POS, A_KIND, A_DLEN, A_STR_OFF, A_NL_OFF, A_NUMQ, A_BIGQ, A_STRQ, A_BR_DEPTH, A_BR_OFF, A_BR_INFO, A_AGG,
    A_LEV, A_TAPE, A_STRINGS, A_STR_OUT, A_REC, A_UNIT_CNT, A_SV_BASE, A_SV_QM, A_SV_Q, A_SV_ST, A_SV_UNIT_H, A_SV_UNIT_SLOW,
    A_SELFTEST, A_KEYFLAG,
    A_UNIT_COPY, A_S1_POS, A_S1_KIND, A_S1_QM, A_S1_Q, A_S1_ST, A_S1_UNIT_H, A_S1_UNIT_SLOW,  // stage 1's outputs (stage1.hip)
    A_S1_REC, A_S1_UNIT_CNT, A_S1_UNIT_COPY, A_S1_UNIT_STR, A_UNIT_STR, A_SOFF,
    A_S1_TILE_UNIT, A_TILE_UNIT, A_UNIT_TQ
};

#if defined(SJ_DEBUG_BOUNDS)

struct BoundsHit {
    unsigned int hits, id;
    unsigned long long index, size;
};
#if defined(__HIPCC__)
static __device__ BoundsHit g_bounds_hit;                 // one per translation unit; stage2.hip reads and clears its own
static __device__ unsigned long long g_bounds_sink[8];    // target of accesses to an array without a single element
__device__ __forceinline__ void bounds_report(uint32_t id, unsigned long long index, unsigned long long size) {
    if (atomicAdd(&g_bounds_hit.hits, 1u) == 0u) {
        g_bounds_hit.id = id;
        g_bounds_hit.index = index;
        g_bounds_hit.size = size;
    }
}
#else
inline void bounds_report(uint32_t, unsigned long long, unsigned long long) {}
static unsigned long long g_bounds_sink[8];
#endif

template <typename T>
struct Arr {
    T *p = nullptr;
    unsigned long long n = 0;  // elements
    uint32_t id = A_NONE;
    Arr() = default;
    SJ_HD Arr(T *p_, unsigned long long n_, uint32_t id_) : p(p_), n(n_), id(id_) {}
    SJ_HD Arr(decltype(nullptr)) {}
    template <typename U>
    SJ_HD Arr(const Arr<U> &o) : p(o.p), n(o.n), id(o.id) {}  // Arr<T> -> Arr<const T>
    SJ_HD explicit operator bool() const { return p != nullptr; }
    SJ_HD bool operator!() const { return p == nullptr; }
    SJ_HD T &operator[](unsigned long long i) const {
        if (i >= n) {
            bounds_report(id, i, n);
            if (n == 0) return *reinterpret_cast<T *>(g_bounds_sink);
            i = n - 1;
        }
        return p[i];
    }
    // pointer to elements [first, first + count)
    SJ_HD T *at(unsigned long long first, unsigned long long count) const {
        if (first > n || count > n - first) {
            bounds_report(id, first + count, n);
            if (count > n) return reinterpret_cast<T *>(g_bounds_sink);
            first = n - count;
        }
        return p + first;
    }
};
template <typename T>
SJ_HD T *arr_at(const Arr<T> &a, unsigned long long first, unsigned long long count) { return a.at(first, count); }
template <typename T>
SJ_HD T *arr_raw(const Arr<T> &a) { return a.p; }  // (launchers: memsets and copies of whole arrays)
#define SJ_ARR(ptr, count, id) ::sj::Arr<typename std::remove_pointer<decltype(ptr)>::type>((ptr), (unsigned long long)(count), (id))
#define SJ_ARR_PARAM(T) ::sj::Arr<T>  // a function parameter that is `T *__restrict__` in the product build

#else  // product build: plain pointers

template <typename T>
using Arr = T *;
template <typename T>
SJ_HD T *arr_at(T *a, unsigned long long first, unsigned long long) { return a + first; }
template <typename T>
SJ_HD T *arr_raw(T *a) { return a; }
#define SJ_ARR(ptr, count, id) (ptr)
#define SJ_ARR_PARAM(T) T *__restrict__

#endif

}  // namespace sj

// ---- streaming stores ------------------------------------------------------------------------------------------------
// Output that nobody reads again soon (stage 1's positions, the tape, Strings.B) leaves with the nt bit: it does not displace the
// lines the following kernels -- or the next pass over the same message -- find in the L2 / Infinity Cache (round 6, measured).
#if defined(__HIPCC__)
namespace sj {
template <typename T>
__device__ __forceinline__ void nt_store(T *p, T v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void nt_store4(void *p, uint4 v) {  // (merged into one global_store_dwordx4 ... nt)
    unsigned int *d = reinterpret_cast<unsigned int *>(p);
    __builtin_nontemporal_store(v.x, d);
    __builtin_nontemporal_store(v.y, d + 1);
    __builtin_nontemporal_store(v.z, d + 2);
    __builtin_nontemporal_store(v.w, d + 3);
}
}  // namespace sj
#endif

