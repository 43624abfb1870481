// Shared host/device helpers for the gfx950 Foldclass library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/merizo_search_amd.h"

#define MS_WAVE 64

// ---------------------------------------------------------------- host-side errors ----
extern thread_local char ms_err_buf[512];

#define MS_FAIL(code, ...)                                       \
    do {                                                         \
        snprintf(ms_err_buf, sizeof(ms_err_buf), __VA_ARGS__);   \
        return (code);                                           \
    } while (0)

#define MS_HIP_CHECK(expr)                                                                       \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) MS_FAIL(MS_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

#define MS_LAUNCH_CHECK(name)                                                                        \
    do {                                                                                             \
        hipError_t _e = hipGetLastError();                                                           \
        if (_e != hipSuccess) MS_FAIL(MS_ERR_HIP, "launch of %s failed: %s", name, hipGetErrorString(_e)); \
    } while (0)

static inline size_t ms_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---------------------------------------------------------------- device helpers ------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MS_IDX_NONE 0xFFFFFFFFu   // sentinel row of an empty top-k slot

// Total order of search results: score descending, then row ascending (-0.0 == +0.0).
__device__ __forceinline__ bool ms_better(float sa, uint32_t ia, float sb, uint32_t ib) {
    return (sa > sb) || (sa == sb && ia < ib);
}

__device__ __forceinline__ float ms_readlane_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ uint32_t ms_readlane_u(uint32_t v, int lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}

// Wave-cooperative insertion into a best-first sorted list of k <= 64 entries held one per
// lane in LDS (entry = {score bits, row}).  (cs, ci) is wave-uniform; all 64 lanes must be
// active.  Returns the score of the list's last entry afterwards (the pruning threshold).
__device__ __forceinline__ float ms_wave_insert(uint2 *list, int k, float cs, uint32_t ci, int lane) {
    const bool in = lane < k;
    uint2 e = in ? list[lane] : make_uint2(0u, 0u);
    const float es = __uint_as_float(e.x);
    const bool e_better = in && ms_better(es, e.y, cs, ci);
    const int pos = __popcll(__ballot(e_better));
    float ps = __shfl_up(es, 1);
    uint32_t pi = __shfl_up(e.y, 1);
    float ns = es;
    uint32_t ni = e.y;
    if (lane == pos) { ns = cs; ni = ci; }
    if (lane > pos) { ns = ps; ni = pi; }
    if (in && lane >= pos) list[lane] = make_uint2(__float_as_uint(ns), ni);
    return ms_readlane_f(ns, k - 1);
}
