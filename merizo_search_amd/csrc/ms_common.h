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

// How many entries of a sorted list (best first, empty slots last; entry i at list[i * stride], len <= 64) are better
// than (es, ei): branch-free binary search, step0 = the largest power of two <= len.
__device__ __forceinline__ int ms_count_better(const uint2 *list, int stride, int len, int step0, float es, uint32_t ei) {
    int lo = 0;
    for (int step = step0; step >= 1; step >>= 1) {
        const int mid = lo + step - 1;
        if (mid < len) {
            const uint2 o = list[(size_t)mid * stride];
            if (ms_better(__uint_as_float(o.x), o.y, es, ei)) lo = mid + 1;
        }
    }
    return lo;
}
__device__ __forceinline__ int ms_pow2_floor(int x) { return 1 << (31 - __clz(x)); }      // x >= 1

// The total order as ONE unsigned 64-bit compare: ms_better(a, b) <=> ms_order_key(a) > ms_order_key(b) (scores are never
// NaN here; -0.0 ranks as +0.0; an empty slot (-inf, MS_IDX_NONE) has the smallest key any entry can have).
__device__ __forceinline__ unsigned long long ms_order_key(uint2 e) {
    uint32_t b = __float_as_uint(__uint_as_float(e.x) + 0.0f);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((unsigned long long)b << 32) | (unsigned long long)(~e.y);
}

// Merge of P <= 256 sorted lists of k <= 64 entries (rank-major in LDS: entry d of list l at ent[d * P + l]) by the 256
// threads of a workgroup, without serial rounds: the k-th best list HEAD is a threshold no answer can be below (k distinct
// rows are at or above it), and hardly anything but the answer is above it.
//   1. thread l ranks its head among the 64 heads of its wave (64 broadcast reads, one 64-bit compare each); the k best
//      heads of each wave are the threshold's candidates (the k best heads overall are among them);
//   2. the k-th best of the <= 4 k candidates is the threshold T;
//   3. thread l walks its list while the entries are >= T and appends them to the survivors (typically 1-3 k entries);
//   4. every survivor counts the survivors that beat it: that is its place in the answer.
// Returns the answer (k entries in LDS, best first, empty slots (-inf, MS_IDX_NONE) last), or nullptr when the shape does
// not fit -- fewer than k lists, no k-th head (empty lists), more than 256 survivors (ties) -- and the caller falls back
// to the head-advance merge.  Every thread of the workgroup must call it; scratch: MS_BLOCK_MERGE_SCRATCH bytes of LDS.
constexpr int MS_BLOCK_MERGE_SCRATCH = 8192;
__device__ __forceinline__ const uint2 *ms_block_merge(const uint2 *ent, char *scratch, int P, int k, int tid) {
    unsigned long long *hk = reinterpret_cast<unsigned long long *>(scratch);      // [256] head keys (0 past P)
    unsigned long long *cand = hk + 256;                                           // [4 k <= 256] threshold candidates
    unsigned long long *skey = cand + 256;                                         // [256] survivors
    uint32_t *sloc = reinterpret_cast<uint32_t *>(skey + 256);                     // [256] their place in ent
    uint32_t *ctl = sloc + 256;                                                    // [0] survivor count, [2..3] threshold
    uint2 *fin = reinterpret_cast<uint2 *>(ctl + 4);                               // [64]
    if (P < k || P > 256) return nullptr;
    const int wave = tid >> 6;
    const unsigned long long my = tid < P ? ms_order_key(ent[tid]) : 0ull;
    hk[tid] = my;
    cand[tid] = 0ull;
    if (tid < 64) fin[tid] = make_uint2(__float_as_uint(-INFINITY), MS_IDX_NONE);
    if (tid == 0) { ctl[0] = 0u; ctl[2] = 0u; ctl[3] = 0u; }
    __syncthreads();
    {
        int rank = 0, rank2 = 0;
        const unsigned long long *wh = hk + 64 * wave;
#pragma unroll 8
        for (int j = 0; j < 64; j += 2) { rank += wh[j] > my ? 1 : 0; rank2 += wh[j + 1] > my ? 1 : 0; }
        rank += rank2;
        // (heads of non-empty lists have distinct keys; empty lists tie with each other: the same value into the same slot)
        if (my != 0ull && rank < k) cand[wave * k + rank] = my;
    }
    __syncthreads();
    {
        const int nc = 4 * k;
        const unsigned long long c = tid < nc ? cand[tid] : 0ull;
        int gt = 0, ge = 0;
        for (int j = 0; j < nc; ++j) { const unsigned long long o = cand[j]; gt += o > c ? 1 : 0; ge += o >= c ? 1 : 0; }
        if (c != 0ull && gt < k && ge >= k) *reinterpret_cast<unsigned long long *>(ctl + 2) = c;      // the k-th best head
    }
    __syncthreads();
    const unsigned long long T = *reinterpret_cast<const unsigned long long *>(ctl + 2);
    if (T == 0ull) return nullptr;
    if (tid < P) {
        for (int d = 0; d < k; ++d) {
            const unsigned long long key = d == 0 ? my : ms_order_key(ent[d * P + tid]);
            if (key < T) break;
            const uint32_t pos = atomicAdd(ctl, 1u);
            if (pos < 256u) { skey[pos] = key; sloc[pos] = (uint32_t)(d * P + tid); }
        }
    }
    __syncthreads();
    const int S = (int)ctl[0];
    if (S > 256) return nullptr;
    if (tid < S) {
        const unsigned long long c = skey[tid];
        int rank = 0;
        for (int j = 0; j < S; ++j) rank += skey[j] > c ? 1 : 0;
        if (rank < k) fin[rank] = ent[sloc[tid]];
    }
    __syncthreads();
    return fin;
}

// Gate of the exact pipeline queued behind a prefiltered search (ms_search.hip: ms_rescore_body): gate[0] == epoch -> some query
// needs it, gate[1] == epoch -> the re-scoring launch of THIS call published its verdict.  A gated launch that finds another epoch in
// gate[1] has been handed a workspace whose re-scoring never finished its count (a poisoned or shared ticket: include/merizo_search_amd.h
// "a workspace serves one stream"): it TRAPS -- the flagged queries would otherwise keep an unproven answer with a zero return code.
// Uniform scalar loads from one line; nothing on the good path but one more compare.
__device__ __forceinline__ bool ms_gate_closed(const uint32_t *gate, uint32_t epoch) {
    if (gate == nullptr) return false;
    if (gate[1] != epoch) __builtin_trap();
    return gate[0] != epoch;
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
