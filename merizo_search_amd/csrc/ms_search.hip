// Brute-force cosine / inner-product top-k over the 128-d domain-embedding database on gfx950.
//
// Reference arithmetic replaced (programs/Foldclass/dbsearch.py):
//   :75-81   search_query_against_db   cosine_similarity(db, q) * mask -> topk   (`.pt` DB)
//   :213-248 knn_exact_faiss           IndexFlat(IP).search per block + ResultHeap merge
//   :303-304 F.normalize(query_embeddings)
//
// This file: normalisation kernels, the launch plan (row streams x query tiles, workspace carve), the two merges of
// per-stream lists and the C entry points.  The scan kernels themselves live in ms_scan.h (instantiated per list
// length in ms_scan_kl*.hip):
//   ms_scan_loader_kernel<KL,AUXM,SAMPLE>  >= 3 query tiles (MFMA-bound), k <= 64, all modes: 4 compute waves share the rows
//                                          a fifth wave streams into an LDS ring by LDS-DMA; SAMPLE = its sample pass
//   ms_scan_kernel<KL,AUX,UB>              1-2 query tiles (HBM-bound), and passes with an upper bound (k > 64)
//   ms_scan_sample_kernel<KL,AUX>          sample pass of the latter: best scores of the first tiles of every stream
// Common to all of them: one wave = one (32-query tile, row stream) pair; the query tile is the MFMA B
// operand in 64 VGPRs for the whole kernel; 32-row tiles arrive by LDS-DMA in an image whose lane = row ds_read_b128
// A-fragment reads are conflict free; k-step s of the v_mfma_f32_32x32x2_f32 chain multiplies
// elements k = s (lanes 0-31) and k = 64 + s (lanes 32-63), i.e. the accumulation order is s = 0..63: (k = s, k = 64 + s)
// -- restated by oracle/oracle.c:dot_ordered(order = 1), which reproduces the scores bit for bit; the running top-k of a
// query lives in the registers of its two lanes; rows are visited in ascending order, so "strictly greater than the k-th
// best" is the exact (score desc, row asc) order.
//
// Kernels here
//   ms_normalize_rows_kernel / ms_row_inv_norms_kernel / ms_prepare_queries_kernel   one wave per 512-byte row
//   ms_block_merge_kernel     per query, one workgroup: threshold = k-th best list head, the few entries at or above it rank
//                             themselves (ms_common.h: ms_block_merge); the usual final merge
//   ms_sample_bound_kernel    per query: k-th largest of the sample pass's maxima = lower bound of the full pass
//   ms_head_merge_kernel      per query: k rounds of "best list head wins" over the per-stream lists staged in LDS (shapes the
//                             block merge declines, and more than 256 lists)
//   ms_partial_merge_kernel   the same merge for large k * P (does not fit LDS): threshold from the list heads + pool
//   ms_kway_merge_kernel      public merge of S sorted lists (shards after the all-gather, blocks when streaming);
//                             ms_kway_merge_any_kernel beyond 64 lists
#include "ms_common.h"

#include <math.h>
#include <stdlib.h>

#include <atomic>
#include <mutex>

thread_local char ms_err_buf[512] = "";

// ------------------------------------------------------------------ normalisation ------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ __launch_bounds__(256) void ms_normalize_rows_kernel(const float *x, float *y, int64_t n, float eps) {   // y may be x
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave0; r < n; r += nwaves) {
        float2 v = *(reinterpret_cast<const float2 *>(x + r * MS_DIM) + lane);
        const float ss = wave_sum(v.x * v.x + v.y * v.y);
        const float nrm = fmaxf(sqrtf(ss), eps);
        v.x = v.x / nrm;
        v.y = v.y / nrm;
        *(reinterpret_cast<float2 *>(y + r * MS_DIM) + lane) = v;
    }
}

__global__ __launch_bounds__(256) void ms_row_inv_norms_kernel(const float *x, int64_t n, float eps, float *inv) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave0; r < n; r += nwaves) {
        const float2 v = *(reinterpret_cast<const float2 *>(x + r * MS_DIM) + lane);
        const float ss = wave_sum(v.x * v.x + v.y * v.y);
        if (lane == 0) inv[r] = 1.0f / fmaxf(sqrtf(ss), eps);
    }
}

// Queries -> zero-padded [nq_pad,128] copy, optionally L2-normalised (cosine mode, eps 1e-8).
__global__ __launch_bounds__(256) void ms_prepare_queries_kernel(const float *q, int nq, int nq_pad, int normalize,
                                                                 float eps, float *qn) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nq_pad) return;
    float2 v = make_float2(0.0f, 0.0f);
    if (row < nq) {
        v = *(reinterpret_cast<const float2 *>(q + (size_t)row * MS_DIM) + lane);
        if (normalize) {
            const float ss = wave_sum(v.x * v.x + v.y * v.y);
            const float nrm = fmaxf(sqrtf(ss), eps);
            v.x = v.x / nrm;
            v.y = v.y / nrm;
        }
    }
    *(reinterpret_cast<float2 *>(qn + (size_t)row * MS_DIM) + lane) = v;
}

#include "ms_scan.h"
#include "ms_scan_pf.h"
#include "ms_scan_pf16.h"

// ------------------------------------------------------------------ partial merge ------
// One workgroup per query merges its P partial lists (each sorted best-first, rank-major
// layout [k][P]) into the final top-k:
//   1. the k-th best of the P list HEADS (k rounds of a block-wide arg-max) is a lower bound
//      of the answer in the total order (score desc, row asc);
//   2. only entries at least that good can matter -- a sorted prefix of at most k lists, at
//      most k*k entries in all -- they are appended to an LDS pool;
//   3. one wave inserts the pool into a sorted k-list (ms_wave_insert) and writes float32 /
//      int64 results (+ the exclusive bound of the next pass when k > 64).
constexpr int MERGE_MAX_P = 1024;
__global__ __launch_bounds__(256) void ms_partial_merge_kernel(const float *part_s, const uint32_t *part_i, int P,
                                                              int k, int64_t row_offset, float *out_s, int64_t *out_i,
                                                              int out_stride, int out_col0, float *ub_s, uint32_t *ub_i, const uint32_t *gate, uint32_t gate_epoch) {
    if (ms_gate_closed(gate, gate_epoch)) return;      // (the exact pipeline behind a prefiltered search: only when it is needed)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint2 *pool = reinterpret_cast<uint2 *>(smem);              // [k*k]
    uint2 *list = pool + (size_t)k * k;                         // [k]
    __shared__ float red_s[4];
    __shared__ uint32_t red_i[4];
    __shared__ int red_slot[4];
    __shared__ int pool_count;
    const int q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *ps = part_s + (size_t)q * k * P;
    const uint32_t *pi = part_i + (size_t)q * k * P;
    constexpr int PER = MERGE_MAX_P / 256;
    float hs[PER];
    uint32_t hi[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int pp = tid + u * 256;
        hs[u] = -INFINITY; hi[u] = MS_IDX_NONE;
        if (pp < P) { hs[u] = ps[pp]; hi[u] = pi[pp]; }
    }
    if (tid == 0) pool_count = 0;
    // 1. k-th best head
    float th_s = -INFINITY;
    uint32_t th_i = MS_IDX_NONE;
    for (int round = 0; round < k; ++round) {
        float bs = -INFINITY; uint32_t bi = MS_IDX_NONE; int bslot = -1;
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if (hi[u] != MS_IDX_NONE && (bslot < 0 || ms_better(hs[u], hi[u], bs, bi))) { bs = hs[u]; bi = hi[u]; bslot = tid + u * 256; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float os = __shfl_xor(bs, off);
            const uint32_t oi = __shfl_xor(bi, off);
            const int oslot = __shfl_xor(bslot, off);
            if (oslot >= 0 && (bslot < 0 || ms_better(os, oi, bs, bi))) { bs = os; bi = oi; bslot = oslot; }
        }
        if (lane == 0) { red_s[wave] = bs; red_i[wave] = bi; red_slot[wave] = bslot; }
        __syncthreads();
        bs = red_s[0]; bi = red_i[0]; bslot = red_slot[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (red_slot[w] >= 0 && (bslot < 0 || ms_better(red_s[w], red_i[w], bs, bi))) { bs = red_s[w]; bi = red_i[w]; bslot = red_slot[w]; }
        __syncthreads();
        if (bslot < 0) { th_s = -INFINITY; th_i = MS_IDX_NONE; break; }   // fewer than k non-empty lists: keep everything
        th_s = bs; th_i = bi;
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if (bslot == tid + u * 256) hi[u] = MS_IDX_NONE;              // remove the winner
    }
    // 2. pool of entries not worse than the threshold
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int pp = tid + u * 256;
        if (pp >= P) continue;
        for (int j = 0; j < k; ++j) {
            const float s = ps[(size_t)j * P + pp];
            const uint32_t i = pi[(size_t)j * P + pp];
            if (i == MS_IDX_NONE) break;
            if (th_i != MS_IDX_NONE && ms_better(th_s, th_i, s, i)) break;     // strictly worse than the threshold
            const int slot = atomicAdd(&pool_count, 1);
            if (slot < k * k) pool[slot] = make_uint2(__float_as_uint(s), i);
        }
    }
    if (tid < k) list[tid] = make_uint2(__float_as_uint(-INFINITY), MS_IDX_NONE);
    __syncthreads();
    // 3. final selection by one wave
    if (wave == 0) {
        const int count = pool_count < k * k ? pool_count : k * k;
        for (int c0 = 0; c0 < count; c0 += 64) {
            const int idx = c0 + lane;
            const uint2 e = idx < count ? pool[idx] : make_uint2(0u, MS_IDX_NONE);
            const int nn = (count - c0) < 64 ? (count - c0) : 64;
            for (int i = 0; i < nn; ++i)
                ms_wave_insert(list, k, ms_readlane_f(__uint_as_float(e.x), i), ms_readlane_u(e.y, i), lane);
        }
        if (lane < k) {
            const uint2 e = list[lane];
            const size_t o = (size_t)q * out_stride + out_col0 + lane;
            out_s[o] = __uint_as_float(e.x);
            out_i[o] = (e.y == MS_IDX_NONE) ? (int64_t)-1 : row_offset + (int64_t)e.y;
            if (lane == k - 1 && ub_s != nullptr) { ub_s[q] = __uint_as_float(e.x); ub_i[q] = e.y; }
        }
    }
}

// Head-advance form of the same merge for small k * P (the usual case: k = 10, P <= 128 lists):
// one wave per query stages the [k][P] block in LDS, then runs k rounds of "best list head
// wins and its list advances" -- a wave-level arg-max per round, no block barrier, no pool.
template <int PER>      // lists per lane: P <= 64 * PER
__global__ __launch_bounds__(64) void ms_head_merge_kernel(const float *part_s, const uint32_t *part_i, int P, int k,
                                                          int64_t row_offset, float *out_s, int64_t *out_i,
                                                          int out_stride, int out_col0, float *ub_s, uint32_t *ub_i, const uint32_t *gate, uint32_t gate_epoch) {
    if (ms_gate_closed(gate, gate_epoch)) return;      // (the exact pipeline behind a prefiltered search: only when it is needed)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ms_head_merge_wave<PER>(reinterpret_cast<uint2 *>(smem), part_s, part_i, P, k, row_offset, out_s, out_i, out_stride, out_col0,
                            ub_s, ub_i, blockIdx.x, threadIdx.x);
}

// The same merge by a whole workgroup per query (ms_block_merge: threshold = the k-th best list head, then the few entries at
// or above it rank themselves; no serial rounds).  The form launch_merge uses for P <= 256 lists when the [k][P] block
// fits in LDS; shapes ms_block_merge declines (fewer lists than k, ties by the hundred) go through the head-advance merge of
// the workgroup's first wave.
// dp / qmap (the exact pass over the flagged queries of a prefiltered search): the number of queries and of lists comes from the
// device plan, and query q of the compacted batch is row qmap[q] of the outputs.
__device__ __forceinline__ void ms_block_merge_body(char *smem, const float *part_s, const uint32_t *part_i, int P, int k,
                                                    int64_t row_offset, float *out_s, int64_t *out_i,
                                                    int out_stride, int out_col0, float *ub_s, uint32_t *ub_i,
                                                    const ScanDevPlan *dp, const int *qmap, int sparse, size_t sm_stride,
                                                    uint2 *keep = nullptr) {
    // keep (round 6; the fused merge + re-scoring launch): the k merged entries are ALSO left in LDS -- keep[rank] = {score bits, row or
    // MS_IDX_NONE}, keep[64].x = 1 when they are all there -- so that the re-scoring behind the merge does not read back through L2 what
    // this workgroup has just written (one round trip of ~1.5 us per call)
    int q = blockIdx.x;
    if (dp != nullptr) {
        if (q >= dp->nq) return;
        P = dp->P;
    }
    const int q_out = qmap != nullptr ? qmap[q] : q;
    const int tid = threadIdx.x, kP = k * P;
    uint2 *ent = reinterpret_cast<uint2 *>(smem + MS_BLOCK_MERGE_SCRATCH);
    const size_t base = (size_t)q * kP;
    // entry (rank r, list l) of this query: rank-major [k][P] per query, or (sm_stride > 0) stream-major [list][query][rank]
    auto at = [&](int r, int l) -> size_t { return sm_stride != 0 ? (size_t)l * sm_stride + (size_t)q * k + r : base + (size_t)r * P + l; };
    if (sparse) {
        // SPARSE lists (the prefilter's: 256 streams, 20 slots each, ~100 entries in all -- a bound filtered the rest): every thread
        // walks its own list (four ranks in one round trip, further ones only where the fourth is occupied) and appends what it finds
        // to a pool; the pool ranks itself (rows are distinct: an entry's place = how many beat it).  40 KB of mostly empty slots
        // are not staged at all: 28 -> ~5 us at C2.  More than 512 entries (clustered data): the general merge below.
        __shared__ int pool_n;
        unsigned long long *pk = reinterpret_cast<unsigned long long *>(smem);            // [512]  (the scratch area: 8 KiB)
        uint2 *pe = reinterpret_cast<uint2 *>(smem + 4096);                                // [512]
        if (tid == 0) pool_n = 0;
        __syncthreads();
        for (int l = tid; l < P; l += 256) {
            float s4[4];
            uint32_t i4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                i4[r] = MS_IDX_NONE; s4[r] = -INFINITY;
                if (r < k) { s4[r] = part_s[at(r, l)]; i4[r] = part_i[at(r, l)]; }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (i4[r] == MS_IDX_NONE) continue;
                const int pos = atomicAdd(&pool_n, 1);
                if (pos < 512) { const uint2 e = make_uint2(__float_as_uint(s4[r]), i4[r]); pk[pos] = ms_order_key(e); pe[pos] = e; }
            }
            if (k > 4 && i4[3] != MS_IDX_NONE) {
                for (int r = 4; r < k; ++r) {
                    const uint32_t ii = part_i[at(r, l)];
                    if (ii == MS_IDX_NONE) break;
                    const int pos = atomicAdd(&pool_n, 1);
                    if (pos < 512) { const uint2 e = make_uint2(__float_as_uint(part_s[at(r, l)]), ii); pk[pos] = ms_order_key(e); pe[pos] = e; }
                }
            }
        }
        __syncthreads();
        const int n = pool_n;
        if (n <= 512) {
            for (int e = tid; e < n; e += 256) {
                const unsigned long long key = pk[e];
                int rank = 0;
                for (int j = 0; j < n; ++j) rank += pk[j] > key ? 1 : 0;
                if (rank < k) {
                    const uint2 v = pe[e];
                    const size_t o = (size_t)q_out * out_stride + out_col0 + rank;
                    out_s[o] = __uint_as_float(v.x);
                    out_i[o] = row_offset + (int64_t)v.y;
                    if (keep != nullptr) keep[rank] = v;
                }
            }
            if (tid < k && tid >= n) {
                const size_t o = (size_t)q_out * out_stride + out_col0 + tid;
                out_s[o] = -INFINITY;
                out_i[o] = -1;
                if (keep != nullptr) keep[tid] = make_uint2(__float_as_uint(-INFINITY), MS_IDX_NONE);
            }
            if (keep != nullptr && tid == 0) keep[64].x = 1u;
            return;
        }
        __syncthreads();            // (the pool is abandoned: the scratch area is the general merge's from here on)
    }
    if (sm_stride != 0) {           // stream-major: thread l copies its list (k contiguous entries) into the [k][P] staging order
        for (int l = tid; l < P; l += 256)
            for (int r = 0; r < k; ++r) ent[(size_t)r * P + l] = make_uint2(__float_as_uint(part_s[at(r, l)]), part_i[at(r, l)]);
    } else if ((kP & 3) == 0) {
        const float4 *ps4 = reinterpret_cast<const float4 *>(part_s + base);
        const uint4 *pi4 = reinterpret_cast<const uint4 *>(part_i + base);
#pragma unroll 4
        for (int e = tid; e < (kP >> 2); e += 256) {
            const float4 sv = ps4[e];
            const uint4 iv = pi4[e];
            uint2 *dst = ent + 4 * e;
            dst[0] = make_uint2(__float_as_uint(sv.x), iv.x); dst[1] = make_uint2(__float_as_uint(sv.y), iv.y);
            dst[2] = make_uint2(__float_as_uint(sv.z), iv.z); dst[3] = make_uint2(__float_as_uint(sv.w), iv.w);
        }
    } else {
#pragma unroll 4
        for (int e = tid; e < kP; e += 256) ent[e] = make_uint2(__float_as_uint(part_s[base + e]), part_i[base + e]);
    }
    __syncthreads();
    const uint2 *fin = ms_block_merge(ent, smem, P, k, tid);
    if (fin == nullptr) {        // (uniform across the workgroup)
        if (tid < 64)       // (`ent` is staged already: the wave reads nothing at part_s / part_i, and q only names the output row)
            ms_head_merge_wave<4, true>(ent, part_s, part_i, P, k, row_offset, out_s, out_i, out_stride, out_col0, ub_s, ub_i, q_out, tid);
        if (keep != nullptr && tid == 0) keep[64].x = 0u;      // (this rare form writes the global lists only: the re-scoring reads them back)
        return;
    }
    if (keep != nullptr) {
        if (tid < k) keep[tid] = fin[tid];
        if (tid == 0) keep[64].x = 1u;
    }
    if (tid < k) {
        const uint2 v = fin[tid];
        const size_t o = (size_t)q_out * out_stride + out_col0 + tid;
        out_s[o] = __uint_as_float(v.x);
        out_i[o] = v.y == MS_IDX_NONE ? (int64_t)-1 : row_offset + (int64_t)v.y;
        if (tid == k - 1 && ub_s != nullptr) { ub_s[q_out] = __uint_as_float(v.x); ub_i[q_out] = v.y; }
    }
}

__global__ __launch_bounds__(256) void ms_block_merge_kernel(const float *part_s, const uint32_t *part_i, int P, int k,
                                                             int64_t row_offset, float *out_s, int64_t *out_i,
                                                             int out_stride, int out_col0, float *ub_s, uint32_t *ub_i, const uint32_t *gate, uint32_t gate_epoch,
                                                             const ScanDevPlan *dp, const int *qmap, int sparse, size_t sm_stride) {
    if (ms_gate_closed(gate, gate_epoch)) return;      // (the exact pipeline behind a prefiltered search: only when it is needed)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ms_block_merge_body(smem, part_s, part_i, P, k, row_offset, out_s, out_i, out_stride, out_col0, ub_s, ub_i, dp, qmap, sparse, sm_stride);
}

// ------------------------------------------------------------------ sample bound -------
// Lower bound of the full pass from the sample pass's lists: the k-th largest (with multiplicity) of the first
// `ranks` entries of all P lists of a query -- scores of distinct rows, so at least k rows score >= the result.
// Values only (no rows, no sorted output): one wave per query, the values in registers, k rounds of "largest value
// below the previous one + how many lanes hold it".  -inf when there are fewer than k sampled rows.
template <int VPL>      // values per lane: ranks * P <= 64 * VPL
// sm_stride > 0: the lists are STREAM-MAJOR ([stream][query][rank], the image scans of the prefilter: ms_scan_pf.h) -- entry (rank r,
// list l) of query q is at l * sm_stride + q * k + r, sm_stride = nq_pad * k; 0: rank-major per query ([query][rank][stream]).
__global__ __launch_bounds__(64) void ms_sample_bound_kernel(const float *part_s, int P, int k, int ranks, float *lb, uint32_t *hist,
                                                            float *hstep, const uint32_t *gate, uint32_t gate_epoch, size_t sm_stride) {
    if (ms_gate_closed(gate, gate_epoch)) return;      // (the exact pipeline behind a prefiltered search: only when it is needed)
    const int q = blockIdx.x, lane = threadIdx.x;
    const float *ps = part_s + (size_t)q * k * P;              // rank-major [k][P]: the first ranks * P floats
    const int count = ranks * P;
    float v[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int idx = lane + 64 * i;
        if (sm_stride == 0) v[i] = idx < count ? ps[idx] : -INFINITY;
        else v[i] = idx < count ? part_s[(size_t)(idx % P) * sm_stride + (size_t)q * k + idx / P] : -INFINITY;
    }
    float kth = -INFINITY, top = -INFINITY, second = -INFINITY, third = -INFINITY;        // (second, third: with multiplicity)
    // wave maximum: 4 DPP steps inside each row of 16 lanes, then the 4 row results through SGPRs
#define MS_DPP_FMAX(CTRL) m = fmaxf(m, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(m), CTRL, 0xF, 0xF, false)));
    if constexpr (VPL <= 8) {
        // Round 6: every lane sorts its own <= 8 values once (a network of compare-exchanges), then k rounds of "the largest lane HEAD, and
        // the lowest lane that holds it pops" -- ~30 instructions per round instead of ~60 (the form below scans all VPL values and takes
        // VPL ballots per round): the k-th largest with multiplicity, the same value.  9.6 -> ~7 us for the image scan's 20 rounds.
#define MS_CE(A, B) { const float hi_ = fmaxf(v[A], v[B]), lo_ = fminf(v[A], v[B]); v[A] = hi_; v[B] = lo_; }
        if constexpr (VPL == 4) { MS_CE(0, 1) MS_CE(2, 3) MS_CE(0, 2) MS_CE(1, 3) MS_CE(1, 2) }
        if constexpr (VPL == 8) {
            MS_CE(0, 1) MS_CE(2, 3) MS_CE(4, 5) MS_CE(6, 7) MS_CE(0, 2) MS_CE(1, 3) MS_CE(4, 6) MS_CE(5, 7) MS_CE(1, 2) MS_CE(5, 6)
            MS_CE(0, 4) MS_CE(1, 5) MS_CE(2, 6) MS_CE(3, 7) MS_CE(2, 4) MS_CE(3, 5) MS_CE(1, 2) MS_CE(3, 4) MS_CE(5, 6)
        }
#undef MS_CE
        for (int round = 0; round < k; ++round) {
            float m = v[0];
            MS_DPP_FMAX(0xB1) MS_DPP_FMAX(0x4E) MS_DPP_FMAX(0x141) MS_DPP_FMAX(0x140)
            m = fmaxf(fmaxf(ms_readlane_f(m, 0), ms_readlane_f(m, 16)), fmaxf(ms_readlane_f(m, 32), ms_readlane_f(m, 48)));
            if (!(m > -INFINITY)) { kth = -INFINITY; break; }             // fewer than k values: no bound
            if (round == 0) top = m;
            if (round == 1) second = m;
            if (round == 2) third = m;
            kth = m;
            const int win = __builtin_ctzll(__ballot(v[0] == m));        // (some lane's head IS the maximum)
            if (lane == win) {
#pragma unroll
                for (int i = 0; i + 1 < VPL; ++i) v[i] = v[i + 1];
                v[VPL - 1] = -INFINITY;
            }
        }
    } else {
    float cur = INFINITY;
    int remaining = k;
    for (int round = 0; round < k; ++round) {
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < VPL; ++i) m = (v[i] < cur) ? fmaxf(m, v[i]) : m;
        MS_DPP_FMAX(0xB1) MS_DPP_FMAX(0x4E) MS_DPP_FMAX(0x141) MS_DPP_FMAX(0x140)
        m = fmaxf(fmaxf(ms_readlane_f(m, 0), ms_readlane_f(m, 16)), fmaxf(ms_readlane_f(m, 32), ms_readlane_f(m, 48)));
        if (!(m > -INFINITY)) break;                              // fewer than k values: no bound
        if (round == 0) top = m;
        int c = 0;
#pragma unroll
        for (int i = 0; i < VPL; ++i) c += __popcll(__ballot(v[i] == m));
        const int seen = k - remaining;                               // values above m so far
        if (seen < 2 && seen + c >= 2) second = m;
        if (seen < 3 && seen + c >= 3) third = m;
        if (c >= remaining) { kth = m; break; }
        remaining -= c;
        cur = m;
    }
    }
#undef MS_DPP_FMAX
    if (lane == 0) lb[q] = kth;
    // Score buckets of the full pass's shared bound (ScanHist): the k-th best of the whole shard sits near the sample's BEST
    // score (the sample is a few percent of the rows), so 16 buckets of (best - k-th) / 12 from the k-th up cover the range
    // the bound moves through; the last bucket is open-ended.  No usable spread: step 0 = no histogram for this query.
    // Round 6: "best - k-th" is taken from the SLOPE of the sample's tail, not from its single best score.  Scores of rank r in a tail fall
    // like kth + b ln(k / r); b from rank 1, 2 and 3 each, the MEDIAN of the three.  On a smooth tail the three agree and the buckets are
    // the ones above; a sampled row that is no part of the tail -- a near-duplicate of the query, a close homologue: cosine 0.97 over a
    // tail that ends at 0.35 -- used to stretch the 16 buckets over the whole gap, the bound of that query then never rose, and its ~1,200
    // rows above the sample bound (not ~60) all went through the rare path, the merge and the re-scoring: the bench's database plants three
    // such rows per query, ~6 of 256 queries caught one in the sample, and the C2 step cost 141 us instead of 130
    // (profiles/r06_hist_step_ab.log).
    if (hist != nullptr) {
        if (lane < 16) hist[(size_t)q * 16 + lane] = 0u;
        if (lane == 0) {
            float spread = (kth > -INFINITY && top > kth) ? top - kth : 0.0f;
            if (spread > 0.0f && k >= 4) {
                const float lk = __logf((float)k);
                const float b1 = (top - kth) / lk, b2 = (second - kth) / (lk - 0.6931472f), b3 = (third - kth) / (lk - 1.0986123f);
                const float bm = fmaxf(fminf(b1, b2), fminf(fmaxf(b1, b2), b3));                 // median
                if (bm > 0.0f && bm < INFINITY) spread = bm * lk;
            }
            float step = spread * (1.0f / 12.0f);
            if (!(step > 0.0f) || !(step < INFINITY) || !(1.0f / step < INFINITY)) step = 0.0f;
            hstep[q] = step;
        }
    }
}

// ------------------------------------------------------------------ prefilter: exact re-scoring ---
// ms_ip_topk_prefiltered: the scan ran on APPROXIMATE scores a(row) (split-bf16 matrix instructions, |a - s| <= E =
// MS_PF_ERR |row| |q|) and kept the kp > k best rows per query by a.  One wave per query re-scores those rows with the exact
// fp32 chain (the k order of the fp32 scan: s = 0..63: (k = s, k = 64 + s), one fmaf each -- the same bits), takes the
// best k under the total order, and proves the answer: every row that is NOT a candidate has a <= a_last (the kp-th
// approximate score), hence s <= a_last + E; if the k-th best exact score is above that, no other row can be in the
// answer or tie with it (the candidate list must be full for that argument: a list with empty slots raises the gate too).  Otherwise (near-ties by the dozen around the k-th best) the query raises the gate and the exact
// pipeline, queued behind this kernel, runs after all.
// The flagged queries are gathered for the exact pass right here (round 4b; a separate one-workgroup kernel before): a flagged
// query takes the next slot of the compacted batch (an atomic counter in library-owned memory: the order of the batch varies from
// run to run, the results do not -- every query's exact pass is its own) and copies its vector, bound and length there; the LAST
// workgroup to finish (a ticket) writes the launch plan of the exact pass for that many queries, raises the gate and resets both
// counters.
struct PfCompact {
    float *qn_c, *lb_c, *qlen_c;
    int *qmap;
    ScanDevPlan *dp;
    uint32_t *gate;          // [0] gate, [1] epoch, [2] flagged count (diagnostics), [4] slot counter, [5] ticket (never reset: counts on)
    uint32_t epoch;
    uint32_t ticket_base;    // value of the ticket counter when this call's first workgroup arrives (the host keeps the running total)
    int64_t n;
    int cus, nq;
};
// The two counters of the compaction (slot counter, ticket) are plain atomic adds -- 256 workgroups arrive within a few
// microseconds, and a compare-and-swap loop per arrival (tried: counters tagged with the call's epoch) serialises them into
// O(n^2) retries: +250 us at C2.  Round 6: the TICKET is never reset -- it counts on from call to call, the host keeps the running
// total per workspace (sync_take_tickets) and hands every call the value its first arrival will see (ticket_base): the last
// workgroup is the one that draws ticket_base + nq - 1.  A counter that is not where the host expects it (a launch that was
// aborted half way, a second stream sharing the workspace against the header's rule, memory poisoned through
// ms_debug_prefilter_poison) never produces that ticket, gate[1] keeps the previous call's epoch, and the gated launches queued
// behind (ms_gate_closed) TRAP instead of leaving the flagged queries with an unproven answer.  The slot counter is still zeroed by
// the last workgroup (its final value is data-dependent).
struct PfRescore {
    const float *db, *qn, *as;
    const int64_t *ai;
    const float *lengths, *qlen;
    float *out_s;
    int64_t *out_i;
    uint32_t *flag;
    int64_t row_offset;
    float err_coef, mincov;
    float q_eps;             // > 0: qn is the caller's RAW query array -- the workgroup normalises its query itself, exactly (round 6)
    int k, kp, fp16_range;
    PfCompact cp;
};
// Every thread of the workgroup calls it (barriers inside); the first wave does the work.  `coherent`: the candidate lists were
// written by THIS workgroup a moment ago (the fused merge + re-scoring launch): read them past the L1.
// The query of a re-scoring workgroup, requested early (round 6: the fused launch asks for it BEFORE it merges the candidate lists, so that its
// round trip runs under the merge's own loads): this lane's float2 of the raw query, or its two floats of the prepared one
struct PfQueryPre { float x, y; };
__device__ __forceinline__ PfQueryPre ms_rescore_query_load(const PfRescore &a, int q, int tid) {
    PfQueryPre r{0.0f, 0.0f};
    if (tid < 64) {
        if (a.q_eps > 0.0f) { const float2 v = *(reinterpret_cast<const float2 *>(a.qn + (size_t)q * MS_DIM) + tid); r.x = v.x; r.y = v.y; }
        else { r.x = a.qn[(size_t)q * MS_DIM + tid]; r.y = a.qn[(size_t)q * MS_DIM + 64 + tid]; }
    }
    return r;
}
__device__ __forceinline__ void ms_rescore_body(const PfRescore &a, int q, int tid, bool coherent, const uint2 *keep = nullptr,
                                                const PfQueryPre *pre = nullptr) {
    __shared__ float qs[128];
    __shared__ float cs[64];
    __shared__ uint32_t ci[64];
    __shared__ float kth;
    __shared__ int slot_s;
    const bool act = tid < 64;
    const int lane = tid & 63;
    const int k = a.k, kp = a.kp;
    const PfCompact &cp = a.cp;
    if (act) {
        if (a.q_eps > 0.0f) {
            // raw query: F.normalize (dbsearch.py:303-304, eps 1e-12) / cosine_similarity's own normalisation (dbsearch.py:78, eps 1e-8) in the
            // arithmetic of ms_normalize_rows_kernel -- float2 per lane, the same butterfly sum, sqrt, max, divide: the same bits -- which
            // is what the exact scores below (and the exact pass's copy of the query) are made of; the image scan in front of this launch
            // normalised the same query approximately, which is all its error bound asks for (ms_scan_pf16.h)
            const PfQueryPre ld = pre != nullptr ? *pre : ms_rescore_query_load(a, q, tid);
            float2 v = make_float2(ld.x, ld.y);
            const float ss = wave_sum(v.x * v.x + v.y * v.y);
            const float nrm = fmaxf(sqrtf(ss), a.q_eps);
            v.x = v.x / nrm;
            v.y = v.y / nrm;
            qs[2 * lane] = v.x;
            qs[2 * lane + 1] = v.y;
        } else {
            const PfQueryPre ld = pre != nullptr ? *pre : ms_rescore_query_load(a, q, tid);
            qs[lane] = ld.x;
            qs[64 + lane] = ld.y;
        }
        if (lane == 0) kth = -INFINITY;
    }
    __syncthreads();
    float qnorm = 0.0f, a_last = 0.0f, s = -INFINITY;
    int64_t row = -1;
    bool full = false;
    if (act) {
        float qq = qs[lane] * qs[lane] + qs[64 + lane] * qs[64 + lane];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) qq += __shfl_xor(qq, off);
        qnorm = sqrtf(qq) * 1.001f;
        const int64_t *aq = a.ai + (size_t)q * kp;
        const float *sq = a.as + (size_t)q * kp;
        if (keep != nullptr && keep[64].x != 0u) {          // (uniform) the candidates are still in LDS
            const uint2 mine = keep[lane < kp ? lane : 0], last = keep[kp - 1];
            row = (lane < kp && mine.y != MS_IDX_NONE) ? (int64_t)mine.y : -1;
            full = last.y != MS_IDX_NONE;
            a_last = __uint_as_float(last.x);
        } else if (coherent) {
            row = lane < kp ? __hip_atomic_load(aq + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
            full = __hip_atomic_load(aq + kp - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 0;
            a_last = __hip_atomic_load(sq + kp - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            row = lane < kp ? aq[lane] : -1;
            full = aq[kp - 1] >= 0;
            a_last = sq[kp - 1];
        }
        if (row >= 0) {
            const float4 *x = reinterpret_cast<const float4 *>(a.db + (size_t)row * MS_DIM);
            float acc = 0.0f;
            // (the whole row requested before the chain starts: four batches of eight loads, each a round trip to a row nobody has
            //  touched, were most of this kernel's 14 us)
            float4 xl[16], xh[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) { xl[t] = x[t]; xh[t] = x[16 + t]; }
            __builtin_amdgcn_sched_barrier(0);          // (or hipcc pulls the loads back next to their uses, eight at a time)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float4 lo = xl[t], hi = xh[t];
                acc = fmaf(lo.x, qs[4 * t + 0], acc); acc = fmaf(hi.x, qs[64 + 4 * t + 0], acc);
                acc = fmaf(lo.y, qs[4 * t + 1], acc); acc = fmaf(hi.y, qs[64 + 4 * t + 1], acc);
                acc = fmaf(lo.z, qs[4 * t + 2], acc); acc = fmaf(hi.z, qs[64 + 4 * t + 2], acc);
                acc = fmaf(lo.w, qs[4 * t + 3], acc); acc = fmaf(hi.w, qs[64 + 4 * t + 3], acc);
            }
            s = acc;
            if (a.lengths != nullptr) {       // MS_MODE_COSINE_UNIT: the fp32 scan's own two multiplications (unit rows: scale 1; dbsearch.py:76,78)
                float sv = s * 1.0f;
                const float mk = (a.qlen[q] >= a.lengths[row] * a.mincov) ? 1.0f : 0.0f;
                sv = sv * mk;
                s = sv;
            }
        }
        cs[lane] = s;
        ci[lane] = row >= 0 ? (uint32_t)row : MS_IDX_NONE;
    }
    __syncthreads();
    if (act) {
        const int nvalid = __popcll(__ballot(row >= 0));
        int rank = 0;
        for (int j = 0; j < kp; ++j) rank += (ci[j] != MS_IDX_NONE && ms_better(cs[j], ci[j], s, ci[lane])) ? 1 : 0;
        const size_t o0 = (size_t)q * k;
        if (row >= 0 && rank < k) { a.out_s[o0 + rank] = s; a.out_i[o0 + rank] = a.row_offset + row; }
        if (lane < k && lane >= nvalid) { a.out_s[o0 + lane] = -INFINITY; a.out_i[o0 + lane] = -1; }
        if (row >= 0 && rank == k - 1) kth = s;
    }
    __syncthreads();
    if (tid == 0) {
        // (kth: k rows score at least this -- the bound the exact scan starts from, should it have to run (-inf: none): lb_c below)
        // (a list that is not full cannot happen on a database of >= 65,536 rows unless rows were lost to a bound: no proof then either)
        // (fp16 formats: the scan scales every query into the fp16 range by a power of two it clamps at 2^+-60 -- a query whose norm is
        //  outside [2^-40, 2^40] is not covered by the error bound: no proof)
        const bool in_range = !a.fp16_range || (qnorm >= 9.094947e-13f && qnorm <= 1.0995116e12f);
        const bool flagged = !full || !in_range || !(kth > a_last + a.err_coef * qnorm);
        a.flag[q] = flagged ? 1u : 0u;
        int slot = -1;
        if (flagged) {
            slot = (int)atomicAdd(cp.gate + 4, 1u);
            cp.qmap[slot] = q;
            cp.lb_c[slot] = kth;
            cp.qlen_c[slot] = a.qlen != nullptr ? a.qlen[q] : 0.0f;
        }
        slot_s = slot;
    }
    __syncthreads();
    const int slot = slot_s;
    if (act && slot >= 0) {
        cp.qn_c[(size_t)slot * MS_DIM + lane] = qs[lane];
        cp.qn_c[(size_t)slot * MS_DIM + 64 + lane] = qs[64 + lane];
    }
    if (tid == 0) {
        __threadfence();                                         // (this workgroup's slot is taken before its ticket)
        const uint32_t done = atomicAdd(cp.gate + 5, 1u);
        // this call's nq tickets are ticket_base .. ticket_base + nq - 1: any other value means the counter was not where the host's
        // running total says (behind: the first arrival sees it; ahead by less than nq: the "last" ticket would be drawn EARLY, by a
        // workgroup that is not the last, and the late arrivals see it here) -- trap, never an unproven answer
        if (done - cp.ticket_base >= (uint32_t)cp.nq) __builtin_trap();
        if (done == cp.ticket_base + (uint32_t)cp.nq - 1u) {    // the last workgroup of THIS call: every flagged query has its slot
            __threadfence();
            const int cnt = (int)atomicAdd(cp.gate + 4, 0u);
            ScanDevPlan d;
            ms_plan_core(cp.n, cnt > 0 ? cnt : 1, cp.cus, &d);
            if (cnt == 0) { d.nq = 0; d.grid = 0; }
            *cp.dp = d;
            cp.gate[0] = cnt > 0 ? cp.epoch : 0u;
            cp.gate[1] = cp.epoch;
            cp.gate[2] = (uint32_t)cnt;
            cp.gate[4] = 0u;                                     // (for the next call on this workspace: stream order; the ticket counts on)
        }
    }
}
// ONE launch for the prefilter's candidate merge and the exact re-scoring behind it (round 5; two launches and a clearing memset
// before): a workgroup merges the per-stream lists of its query into the kp candidates (ms_block_merge_body: the sparse pool, or
// the general merge), then its first wave re-scores them, ranks them, proves the answer or takes a slot of the exact pass.
__global__ __launch_bounds__(256) void ms_merge_rescore_kernel(const float *part_s, const uint32_t *part_i, int P, float *as, int64_t *ai,
                                                               size_t sm_stride, const PfRescore a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint2 keep[65];          // the kp <= 64 merged candidates + [64].x: are they all here?
    const PfQueryPre pre = ms_rescore_query_load(a, (int)blockIdx.x, (int)threadIdx.x);
    ms_block_merge_body(smem, part_s, part_i, P, a.kp, 0, as, ai, a.kp, 0, nullptr, nullptr, nullptr, nullptr, 1, sm_stride, keep);
    __threadfence_block();
    __syncthreads();
    ms_rescore_body(a, (int)blockIdx.x, (int)threadIdx.x, true, keep, &pre);
}

// ------------------------------------------------------------------ public k-way merge -
// One thread per query: classic k-way merge of S lists that are each sorted best-first.
__global__ __launch_bounds__(64) void ms_kway_merge_kernel(const float *scores0, const int64_t *idx0, int64_t score_stride,
                                                           int64_t idx_stride, int S, int nq, int k, float *out_s,
                                                           int64_t *out_i) {
    // list s lives at byte offset s * stride of each array (dense [S,nq,k] arrays: stride = nq*k elements)
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    constexpr int MAXS = 64;
    int head[MAXS];
    for (int s = 0; s < S; ++s) head[s] = 0;
    for (int j = 0; j < k; ++j) {
        int best = -1;
        float bs = 0.0f;
        int64_t bi = 0;
        for (int s = 0; s < S; ++s) {
            if (head[s] >= k) continue;
            const size_t o = (size_t)q * k + head[s];
            const int64_t ci = reinterpret_cast<const int64_t *>(reinterpret_cast<const char *>(idx0) + (size_t)s * idx_stride)[o];
            if (ci < 0) { head[s] = k; continue; }   // padding: list exhausted
            const float cs = reinterpret_cast<const float *>(reinterpret_cast<const char *>(scores0) + (size_t)s * score_stride)[o];
            if (best < 0 || cs > bs || (cs == bs && ci < bi)) { best = s; bs = cs; bi = ci; }
        }
        const size_t oo = (size_t)q * k + j;
        if (best < 0) { out_s[oo] = -INFINITY; out_i[oo] = -1; }
        else { out_s[oo] = bs; out_i[oo] = bi; ++head[best]; }
    }
}

// The same merge for any number of lists (S > 64: no per-thread head array): output j is the best entry that comes
// strictly after output j-1 in the total order (score desc, index asc; indices are unique across the lists), found
// by a linear walk of every sorted list.  O(k * S * k) per query: only used beyond 64 lists.
__global__ __launch_bounds__(64) void ms_kway_merge_any_kernel(const float *scores0, const int64_t *idx0, int64_t score_stride,
                                                               int64_t idx_stride, int S, int nq, int k, float *out_s,
                                                               int64_t *out_i) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    bool have_prev = false;
    float ps = 0.0f;
    int64_t pi = 0;
    for (int j = 0; j < k; ++j) {
        bool found = false;
        float bs = 0.0f;
        int64_t bi = 0;
        for (int s = 0; s < S; ++s) {
            const float *ls = reinterpret_cast<const float *>(reinterpret_cast<const char *>(scores0) + (size_t)s * score_stride) + (size_t)q * k;
            const int64_t *li = reinterpret_cast<const int64_t *>(reinterpret_cast<const char *>(idx0) + (size_t)s * idx_stride) + (size_t)q * k;
            for (int e = 0; e < k; ++e) {
                const int64_t ci = li[e];
                if (ci < 0) break;                                   // padding: list exhausted
                const float cs = ls[e];
                if (have_prev && !(cs < ps || (cs == ps && ci > pi))) continue;   // not after the previous output
                if (!found || cs > bs || (cs == bs && ci < bi)) { found = true; bs = cs; bi = ci; }
                break;                                               // sorted list: its first entry after prev is its best one
            }
        }
        const size_t oo = (size_t)q * k + j;
        if (!found) { for (int r = j; r < k; ++r) { out_s[(size_t)q * k + r] = -INFINITY; out_i[(size_t)q * k + r] = -1; } return; }
        out_s[oo] = bs; out_i[oo] = bi;
        have_prev = true; ps = bs; pi = bi;
    }
}

// ------------------------------------------------------------------ host side ----------
namespace {


// CU count of HIP's current device (the device the caller's tensors live on: the Python front end
// makes it current for every call), cached per device ordinal.
int cu_count_cached() {
    constexpr int MAX_DEV = 64;
    static int cus[MAX_DEV] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return 256;
    if (cus[dev] == 0) {
        int c = 0;
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
        cus[dev] = c;
    }
    return cus[dev];
}


int head_merge_setting() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("MS_HEAD_MERGE"); v = e ? atoi(e) : 1; }
    return v;
}

int block_merge_setting() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("MS_BLOCK_MERGE"); v = e ? atoi(e) : 1; }      // diagnostics: 0 = the head-advance merge
    return v;
}

int64_t pf_few_min_rows(int nq) {   // <= 64 queries take the fp16-image scan from this many rows: 1..32 queries (one query tile, HBM-bound) / 33..64 (two tiles)
    static int64_t v1 = -1, v2 = -1;
    if (v1 < 0) { const char *e = getenv("MS_PF_FEW_MIN_ROWS"); v1 = e ? atoll(e) : (int64_t)MS_PF_FEW_MIN_ROWS; }
    if (v2 < 0) { const char *e = getenv("MS_PF_FEW2_MIN_ROWS"); v2 = e ? atoll(e) : (int64_t)MS_PF_FEW2_MIN_ROWS; }
    return nq > 32 ? (v2 < v1 ? v2 : v1) : v1;
}

#ifndef MS_PF_SAMPLE_COEF_DEFAULT
#define MS_PF_SAMPLE_COEF_DEFAULT 1.2        // the constant of the sample-size rule for the image scans with more than 64 queries (MS_PF_SAMPLE_COEF
                                             // overrides): twice the sample of the fp32 rule's 0.3 -- a visit of the rare path costs these kernels ~900 cycles per
                                             // half tile and a sample tile next to nothing (the sample launch is mostly fixed cost): C2 0.145 -> 0.137 ms per
                                             // call, every other shape within 1 % (profiles/r05_pf_sample_coef_sweep.log); few-query plans keep 0.3
#endif
int sample_min_queries_setting() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("MS_SAMPLE_MIN_NQ"); v = e ? atoi(e) : 8; }
    return v;
}

// Histogram counters of the shared bound (ScanHist) must be zero when a scan starts.  ms_sample_bound_kernel zeroes them
// (every ms_ip_topk / ms_ip_topk_prepare); a staged ms_ip_topk_scan that does not directly follow a prepare on the same
// workspace and shape (the same workspace scanned twice in a row) gets a hipMemsetAsync in front of it.  The record below is
// only that optimisation's bookkeeping: per workspace pointer, did the last staged call leave the counters clean?
struct HistClean { const void *ws; int64_t n; int nq, k; bool clean; };
HistClean g_hist_clean[16];
int g_hist_clean_next = 0;
std::mutex g_hist_clean_mutex;
bool hist_take_clean(const void *ws, int64_t n, int nq, int k) {         // -> were they clean?  (they are dirty afterwards)
    std::lock_guard<std::mutex> lock(g_hist_clean_mutex);
    for (HistClean &e : g_hist_clean)
        if (e.ws == ws && e.n == n && e.nq == nq && e.k == k) { const bool c = e.clean; e.clean = false; return c; }
    return false;
}
void hist_invalidate(const void *ws) {          // a scan ran on this workspace outside the staged bookkeeping: its counters are dirty
    std::lock_guard<std::mutex> lock(g_hist_clean_mutex);
    for (HistClean &e : g_hist_clean)
        if (e.ws == ws) e.clean = false;
}
void hist_mark_clean(const void *ws, int64_t n, int nq, int k) {
    std::lock_guard<std::mutex> lock(g_hist_clean_mutex);
    for (HistClean &e : g_hist_clean)
        if (e.ws == ws) { e.n = n; e.nq = nq; e.k = k; e.clean = true; return; }
    g_hist_clean[g_hist_clean_next] = HistClean{ws, n, nq, k, true};
    g_hist_clean_next = (g_hist_clean_next + 1) % 16;
}

// The arrival counters of the in-launch merge must be zero when a launch starts, and the launch leaves them that way (the
// last arriver resets its counter).  They do NOT live in the caller's workspace -- memory the library cannot vouch for
// between calls (a freed workspace's address may come back holding anything) -- but in a small block the library allocates
// and zeroes itself, one per (workspace pointer, device): calls that share a workspace are serialised by the caller anyway
// (they share the partial lists), so they may share the block.  64 blocks; a 65th workspace takes over the least recently used
// one (after a device synchronisation: a rare event).
struct SyncBlock { const void *ws; int dev; char *mem; uint64_t last_use; uint32_t pf_tickets; };      // pf_tickets: tickets handed out so far = the value of the device's ticket word
constexpr int SYNC_BLOCKS = 64;
constexpr size_t SYNC_BYTES = 512;         // [0,256) 64 arrival counters of the in-launch merge; [256] the prefilter's gate word
SyncBlock g_sync[SYNC_BLOCKS];
int g_sync_used = 0;
uint64_t g_sync_clock = 0;
char *sync_block_for(const void *ws) {
    std::lock_guard<std::mutex> lock(g_hist_clean_mutex);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    for (int i = 0; i < g_sync_used; ++i)
        if (g_sync[i].ws == ws && g_sync[i].dev == dev) { g_sync[i].last_use = ++g_sync_clock; return g_sync[i].mem; }
    if (g_sync_used == SYNC_BLOCKS) {
        // every block is taken: the least recently used one of this device changes hands -- after the device has drained
        // (launches of its old workspace may still be in flight on another stream) and with its counters zeroed again
        int lru = -1;
        for (int i = 0; i < g_sync_used; ++i)
            if (g_sync[i].dev == dev && (lru < 0 || g_sync[i].last_use < g_sync[lru].last_use)) lru = i;
        if (lru < 0 || hipDeviceSynchronize() != hipSuccess || hipMemset(g_sync[lru].mem, 0, SYNC_BYTES) != hipSuccess) return nullptr;
        g_sync[lru].ws = ws;
        g_sync[lru].last_use = ++g_sync_clock;
        g_sync[lru].pf_tickets = 0u;
        return g_sync[lru].mem;
    }
    char *mem = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&mem), SYNC_BYTES) != hipSuccess) return nullptr;
    if (hipMemset(mem, 0, SYNC_BYTES) != hipSuccess) { (void)hipFree(mem); return nullptr; }
    g_sync[g_sync_used++] = SyncBlock{ws, dev, mem, ++g_sync_clock, 0u};
    return mem;
}
// The re-scoring launch of a prefiltered search draws `count` tickets from the block's ticket word (one per query); -> the value the
// first of them will see.  Wraps modulo 2^32 like the device counter.
uint32_t sync_take_tickets(const char *mem, uint32_t count) {
    std::lock_guard<std::mutex> lock(g_hist_clean_mutex);
    for (int i = 0; i < g_sync_used; ++i)
        if (g_sync[i].mem == mem) { const uint32_t base = g_sync[i].pf_tickets; g_sync[i].pf_tickets = base + count; return base; }
    return 0u;
}
int inkernel_norm_setting() {      // MS_MODE_IP_NORMQ: up to this many queries are normalised by the scan's own waves (one batch of row loads)
    static int v = -1;
    if (v < 0) { const char *e = getenv("MS_INKERNEL_NORM_MAX_NQ"); v = e ? atoi(e) : 4; }
    return v;
}
int fused_merge_setting() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("MS_FUSED_MERGE_MAX_NQ"); v = e ? atoi(e) : 2; }      // diagnostics: 0 = always a merge launch
    return v;
}

int hist_setting() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("MS_SHARED_BOUND"); v = e ? atoi(e) : 1; }      // diagnostics: 0 = no shared bound
    return v;
}

double sample_coef_setting() {      // diagnostics: the constant of the sample-size rule when the shared bound is on
    static double v = -1.0;
    if (v < 0.0) { const char *e = getenv("MS_SAMPLE_COEF"); v = e ? atof(e) : 0.05; }
    return v;
}

int prepass_tiles_setting() {
    static int v = -2;
    if (v == -2) {
        const char *e = getenv("MS_PREPASS_TILES");     // diagnostics: force a sample size (0 = no sample pass)
        v = e ? atoi(e) : -1;
        if (v < -1) v = -1;
    }
    return v;
}

// list entries per lane and pass: 5 for k <= 10, 10 for k <= 20, 16 for k <= 32, else 32 (k <= 64)
int pick_kl(int k_pass) {
    const int opts[4] = {5, 10, 16, 32};
    for (int i = 0; i < 4; ++i)
        if (2 * opts[i] >= k_pass) return opts[i];
    return 32;
}

// qpw > 0: the plan of the split-image prefilter scan (ms_scan_pf.h): 4 waves x qpw query tiles per workgroup, one workgroup per CU
// tile_rows: rows per tile of that kernel's image (32: split-bf16, 64: fp16); streams are whole tiles
ScanPlan make_plan(int64_t n, int nq, int k, int cus, int qpw = 0, int tile_rows = 32) {
    ScanPlan pl;
    pl.nq_real = nq;
    pl.qpw = qpw;
    pl.k_pass = k < 64 ? k : 64;
    pl.kl = pick_kl(pl.k_pass);
    int64_t tiles_per_stream;
    if (qpw == 0) {
        ScanDevPlan d;
        ms_plan_core(n, nq, cus, &d);      // (the same arithmetic the device runs for the exact pass behind a prefiltered search)
        pl.n_qtiles = d.n_qtiles; pl.qwb = d.qwb; pl.n_qgroups = d.n_qgroups; pl.nq_pad = d.nq_pad; pl.rows_per_stream = d.rows_per_stream;
        pl.n_streams = d.n_streams; pl.n_sgroups = d.n_sgroups; pl.P = d.P; pl.grid = d.grid;
        tiles_per_stream = d.rows_per_stream / 32;
    } else {
        pl.n_qtiles = (nq + 31) / 32;
        pl.qwb = 4;                        // (one list per (stream, query), as in the loader-wave form)
        const int group_tiles = 4 * qpw;   // query tiles per workgroup
        pl.n_qgroups = (pl.n_qtiles + group_tiles - 1) / group_tiles;
        pl.nq_pad = pl.n_qgroups * group_tiles * 32;
        const int64_t big_tiles = (n + tile_rows - 1) / tile_rows;
        int64_t want = (int64_t)cus / pl.n_qgroups;
        if (want < 1) want = 1;
        if (want > big_tiles) want = big_tiles > 0 ? big_tiles : 1;
        const int64_t big_per_stream = (big_tiles + want - 1) / want;
        tiles_per_stream = big_per_stream * (tile_rows / 32);           // (in 32-row units: the sample-size rule below)
        pl.rows_per_stream = (int)((big_per_stream > 0 ? big_per_stream : 1) * tile_rows);
        pl.n_streams = (int)((n + pl.rows_per_stream - 1) / pl.rows_per_stream);
        if (pl.n_streams < 1) pl.n_streams = 1;
        pl.n_sgroups = pl.n_streams;
        pl.P = pl.n_streams;
        pl.grid = ((pl.n_sgroups + 7) / 8) * 8 * pl.n_qgroups;
    }
    pl.lds_bytes = 4 * 32768 + 4 * 1024;        // tile slots, cosine side data, in-launch bound
    // stream-major lists: the image scans always; the loader-wave kernel (>= 3 query tiles, one pass) when the workgroup-per-query merge
    // reads them (<= 256 lists whose [k][P] staging fits the LDS) -- MS_LIST_SM=0: rank-major as in rounds 1-4
    {
        static const int sm_setting = [] { const char *e = getenv("MS_LIST_SM"); return e ? atoi(e) : 1; }();
        const size_t block_lds = (size_t)MS_BLOCK_MERGE_SCRATCH + (((size_t)pl.k_pass * pl.P + 3) & ~(size_t)3) * sizeof(uint2);
        pl.list_sm = (qpw > 0 || (sm_setting && pl.qwb == 4 && loader_wave_setting() && k <= 64 && pl.P <= 256 && block_lds <= 156 * 1024)) ? 1 : 0;
    }
    // sample pass: the k-th best score of the first few tiles of every stream bounds the answer
    // from below and prunes almost every insertion of the full pass; worth it for long streams
    // Size of the sample: T0 tiles per stream cost T0 tile times; the insertion steps they save in the
    // full pass fall as 1/T0 (candidates per tile = 1024 k / (streams * 32 * T0) while the sample's bound is
    // tighter than a stream's own list).  Minimum at T0 = sqrt(c * tiles_per_stream * k / streams), c from
    // the measured cost of a tile (2.2 us) and of a candidate (0.27 us): 3 tiles at 31 tiles per stream,
    // 9 at 244, 17 at 977 for k = 10 and 128 streams (sweeps at 125k-16M rows x 256 queries agree).
    // (... without the shared bound.  With it -- the loader-wave form of the fp32 scan -- the threshold follows the scan and
    //  the sample only has to start it: the optimum moves to ~0.4 of that, 3-4 tiles at C2 instead of 9 (0.518 against 0.526 ms per
    //  step) and 9 instead of 22 at k = 64 (0.665 against 0.719); profiles/r04_sample_size_sweep.log.  Below k = 5 the sample's best and
    //  k-th best scores are too close for the histogram to have buckets: the old rule)
    pl.prepass_tiles = prepass_tiles_setting();
    if (pl.prepass_tiles < 0) {
        static const double pf_coef = [] { const char *e = getenv("MS_PF_SAMPLE_COEF"); return e ? atof(e) : MS_PF_SAMPLE_COEF_DEFAULT; }();
        const double c = (qpw == 0 && pl.qwb == 4 && pl.k_pass >= 5 && loader_wave_setting() && hist_setting()) ? sample_coef_setting() : (qpw > 0 && nq > 64 ? pf_coef : 0.3);      // (the split-image
                                     // scan only appends between flushes: its thresholds move with the shared bound alone, and it wants the larger sample)
        const double t0 = sqrt(c * (double)tiles_per_stream * ((double)pl.k_pass / 10.0) * (128.0 / (double)pl.n_streams));
        pl.prepass_tiles = t0 < 1.0 ? 1 : (t0 > 32.0 ? 32 : (int)(t0 + 0.5));
    }
    if (tiles_per_stream < 8 * (int64_t)pl.prepass_tiles) pl.prepass_tiles = (int)(tiles_per_stream / 8);
    // short streams / few queries: few insertions anyway -- except in the image scans of the prefilter, whose lists only take candidates at
    // a flush and whose thresholds come from the sample and the shared bound alone: without a sample every tile visits the rare path
    // until the first flush (one query over 1M rows: 158 us against 62 with a sample)
    if (tiles_per_stream < 12 || k > 64 || (qpw == 0 && nq < sample_min_queries_setting())) pl.prepass_tiles = 0;
    if (qpw > 0 && tile_rows == 64) pl.prepass_tiles = (pl.prepass_tiles + 1) / 2;                     // (counted in the kernel's own tiles)
    size_t off = 0;
    pl.off_qn = off;      off += ms_align_up((size_t)pl.nq_pad * MS_DIM * sizeof(float), 256);
    pl.off_inv = off;     off += ms_align_up((size_t)(n > 0 ? n : 1) * sizeof(float), 256);
    pl.off_part_s = off;  off += ms_align_up((size_t)pl.P * pl.nq_pad * pl.k_pass * sizeof(float), 256);
    pl.off_part_i = off;  off += ms_align_up((size_t)pl.P * pl.nq_pad * pl.k_pass * sizeof(uint32_t), 256);
    pl.off_ub_s = off;    off += ms_align_up((size_t)pl.nq_pad * sizeof(float), 256);
    pl.off_ub_i = off;    off += ms_align_up((size_t)pl.nq_pad * sizeof(uint32_t), 256);
    pl.off_lb_s = off;    off += ms_align_up((size_t)pl.nq_pad * sizeof(float), 256);
    pl.off_lb_i = off;    off += ms_align_up((size_t)pl.nq_pad * sizeof(uint32_t), 256);
    pl.off_scr_s = off;   off += ms_align_up((size_t)pl.nq_pad * pl.k_pass * sizeof(float), 256);
    pl.off_scr_i = off;   off += ms_align_up((size_t)pl.nq_pad * pl.k_pass * sizeof(int64_t), 256);
    pl.off_hist = off;    off += ms_align_up((size_t)pl.nq_pad * 16 * sizeof(uint32_t), 256);
    pl.off_hstep = off;   off += ms_align_up((size_t)pl.nq_pad * sizeof(float), 256);
    pl.off_prog = off;    off += qpw > 0 ? ms_align_up((size_t)pl.n_streams * 64, 256) : 0;      // progress words of the image scan's workgroups
    pl.total = off;
    return pl;
}

int check_search_args(const float *db, int64_t n, const float *q, int nq, int k, int mode, const float *inv_norm,
                      const float *lengths, const float *qlen) {
    if (n < 0 || nq < 1 || k < 1) MS_FAIL(MS_ERR_ARG, "ms_ip_topk: need n >= 0, nq >= 1, k >= 1 (n=%lld nq=%d k=%d)",
                                          (long long)n, nq, k);
    if (n >= (int64_t)0x7FFFFFFF) MS_FAIL(MS_ERR_RANGE, "ms_ip_topk: n=%lld rows per call must be < 2^31; shard the database",
                                          (long long)n);
    if ((n > 0 && db == nullptr) || q == nullptr) MS_FAIL(MS_ERR_ARG, "ms_ip_topk: NULL db / q");
    if (mode != MS_MODE_IP_PRENORM && mode != MS_MODE_COSINE_RAW && mode != MS_MODE_COSINE_UNIT && mode != MS_MODE_IP_NORMQ)
        MS_FAIL(MS_ERR_ARG, "ms_ip_topk: unknown mode %d", mode);
    if ((mode == MS_MODE_IP_PRENORM || mode == MS_MODE_IP_NORMQ) && (inv_norm || lengths || qlen))
        MS_FAIL(MS_ERR_ARG, "ms_ip_topk: inv_norm / lengths / qlen are only valid in the cosine modes");
    if (mode == MS_MODE_COSINE_UNIT && inv_norm)
        MS_FAIL(MS_ERR_ARG, "ms_ip_topk: MS_MODE_COSINE_UNIT takes rows that are normalised already, not an inv_norm array");
    // (an empty shard -- a rank of a sharded search with no rows -- has NULL row arrays)
    if (n > 0 && (lengths == nullptr) != (qlen == nullptr)) MS_FAIL(MS_ERR_ARG, "ms_ip_topk: lengths and qlen go together");
    return MS_OK;
}

int launch_scan(const ScanPlan &pl, const ScanParams &sp, hipStream_t st) {
    if (sp.prefilter && sp.pf_image != nullptr && sp.pf_format != MS_PF_BF16X3) {      // the prefilter's scan over the fp16 image
        if (pick_kl(sp.k) == 5) return ms_launch_scan_pf16_kl5(pl, sp, st);
        if (pick_kl(sp.k) == 10) return ms_launch_scan_pf16_kl10(pl, sp, st);
        if (pick_kl(sp.k) == 16) return ms_launch_scan_pf16_kl16(pl, sp, st);
        return ms_launch_scan_pf16_kl32(pl, sp, st);
    }
    if (sp.prefilter && sp.pf_image != nullptr) {                  // the prefilter's scan over the split-bf16 image
        if (pick_kl(sp.k) == 5) return ms_launch_scan_pf2_kl5(pl, sp, st);
        if (pick_kl(sp.k) == 10) return ms_launch_scan_pf2_kl10(pl, sp, st);
        if (pick_kl(sp.k) == 16) return ms_launch_scan_pf2_kl16(pl, sp, st);
        return ms_launch_scan_pf2_kl32(pl, sp, st);
    }
    if (sp.ub_s != nullptr) return ms_launch_scan_kl32ub(pl, sp, st);
    if (pick_kl(sp.k) == 5) return ms_launch_scan_kl5(pl, sp, st);
    if (pick_kl(sp.k) == 10) return ms_launch_scan_kl10(pl, sp, st);
    if (pick_kl(sp.k) == 16) return ms_launch_scan_kl16(pl, sp, st);
    return ms_launch_scan_kl32(pl, sp, st);
}

// dp / qmap: the merge behind the exact pass over a prefiltered search's flagged queries -- the list count comes from the device
// plan (pl.P is then its upper bound), query q of the compacted batch is output row qmap[q]
int launch_merge(const ScanPlan &pl, const ScanParams &sp, int nq, int kp, int64_t row_offset, float *out_s,
                 int64_t *out_i, int out_stride, int col0, float *ub_s, uint32_t *ub_i, hipStream_t st,
                 const ScanDevPlan *dp = nullptr, const int *qmap = nullptr, int sparse = 0, size_t sm_stride = 0) {
    const uint32_t *gate = sp.gate;
    const uint32_t gate_epoch = sp.gate_epoch;
    if (pl.P > MERGE_MAX_P) MS_FAIL(MS_ERR_RANGE, "internal: %d partial lists exceed the merge limit", pl.P);
    const size_t head_lds = (size_t)kp * pl.P * sizeof(uint2);
    const size_t block_lds = (size_t)MS_BLOCK_MERGE_SCRATCH + (((size_t)kp * pl.P + 3) & ~(size_t)3) * sizeof(uint2);
    if (block_lds <= 156 * 1024 && pl.P <= 256 && (block_merge_setting() || dp != nullptr || sm_stride != 0)) {     // the usual case: a workgroup per query
        if (block_lds > 48 * 1024)
            MS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ms_block_merge_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)block_lds));
        hipLaunchKernelGGL(ms_block_merge_kernel, dim3(nq), dim3(256), block_lds, st, sp.part_s, sp.part_i, pl.P, kp, row_offset, out_s,
                           out_i, out_stride, col0, ub_s, ub_i, gate, gate_epoch, dp, qmap, (ub_s == nullptr) ? sparse : 0, sm_stride);
        MS_LAUNCH_CHECK("ms_block_merge_kernel");
        return MS_OK;
    }
    if (dp != nullptr) MS_FAIL(MS_ERR_RANGE, "internal: the exact pass behind the prefilter needs the block merge (P = %d, k = %d)", pl.P, kp);
    if (sm_stride != 0) MS_FAIL(MS_ERR_RANGE, "internal: stream-major lists need the block merge (P = %d, k = %d)", pl.P, kp);
    if (head_lds <= 128 * 1024 && head_merge_setting()) {       // k * P entries fit in LDS: one wave per query, k arg-max rounds
        const int per = (pl.P + 63) / 64;
#define MS_HEAD_MERGE(PER)                                                                                             \
    if (head_lds > 48 * 1024)                                                                                          \
        MS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ms_head_merge_kernel<PER>),                    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)head_lds));                  \
    hipLaunchKernelGGL(ms_head_merge_kernel<PER>, dim3(nq), dim3(64), head_lds, st, sp.part_s, sp.part_i, pl.P, kp,    \
                       row_offset, out_s, out_i, out_stride, col0, ub_s, ub_i, gate, gate_epoch)
        if (per <= 1) { MS_HEAD_MERGE(1); }
        else if (per <= 2) { MS_HEAD_MERGE(2); }
        else if (per <= 4) { MS_HEAD_MERGE(4); }
        else if (per <= 8) { MS_HEAD_MERGE(8); }
        else { MS_HEAD_MERGE(16); }
#undef MS_HEAD_MERGE
        MS_LAUNCH_CHECK("ms_head_merge_kernel");
        return MS_OK;
    }
    const size_t lds = ((size_t)kp * kp + kp) * sizeof(uint2);
    hipLaunchKernelGGL(ms_partial_merge_kernel, dim3(nq), dim3(256), lds, st, sp.part_s, sp.part_i, pl.P, kp, row_offset,
                       out_s, out_i, out_stride, col0, ub_s, ub_i, gate, gate_epoch);
    MS_LAUNCH_CHECK("ms_partial_merge_kernel");
    return MS_OK;
}

// ScanParams of the full pass from the workspace layout (queries prepared, inverse norms in the
// workspace when the caller gave none)
#ifdef MS_STAMP
constexpr size_t MS_STAMP_WORDS = 4 * 4 * 65536;
unsigned long long *ms_stamp_buffer() {
    static unsigned long long *buf = nullptr;
    if (buf == nullptr) {
        if (hipMalloc(reinterpret_cast<void **>(&buf), MS_STAMP_WORDS * 8) != hipSuccess) return nullptr;
        (void)hipMemset(buf, 0, MS_STAMP_WORDS * 8);
    }
    return buf;
}
extern "C" int ms_debug_stamps(unsigned long long *host, int words) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpy(host, ms_stamp_buffer(), (size_t)words * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

// Inner-product mode uses the queries as given: the scan kernels read the caller's array directly
// (no padded copy, one launch less per batch) when it is 16-byte aligned.
// ... and so does MS_MODE_IP_NORMQ when the batch runs in the kernels for 1-2 query tiles, whose waves normalise their
// own query tile (ScanParams::qnorm_eps); larger batches get the prepared copy (ms_prepare_queries_kernel, eps 1e-12).
bool queries_used_in_place(const float *q, int mode, const ScanPlan &pl) {
    if (mode == MS_MODE_IP_PRENORM) return ((uintptr_t)q & 15) == 0;
    if (mode == MS_MODE_IP_NORMQ) return pl.qwb < 4 && pl.nq_real <= inkernel_norm_setting() && ((uintptr_t)q & 7) == 0;
    return false;
}

void fill_scan_params(const ScanPlan &pl, const float *db, int64_t n, const float *q, int nq, const float *inv_norm, const float *lengths,
                      const float *qlen, float mincov, char *ws, int mode, ScanParams *sp) {
    const float *inv = inv_norm;
    if (mode == MS_MODE_COSINE_RAW && inv == nullptr && n > 0) inv = reinterpret_cast<const float *>(ws + pl.off_inv);
    sp->db = db; sp->n = n; sp->nq = nq; sp->nq_pad = pl.nq_pad;
    sp->qn = queries_used_in_place(q, mode, pl) ? q : reinterpret_cast<const float *>(ws + pl.off_qn);
    sp->qnorm_eps = (mode == MS_MODE_IP_NORMQ && queries_used_in_place(q, mode, pl)) ? 1e-12f : 0.0f;
    sp->k = pl.k_pass;
    sp->inv_norm = inv; sp->lengths = lengths; sp->qlen = qlen; sp->mincov = mincov;
    sp->unit_rows = mode == MS_MODE_COSINE_UNIT ? 1 : 0;
    sp->ub_s = nullptr; sp->ub_i = nullptr; sp->lb_s = nullptr; sp->max_tiles = 0;
    sp->hist = nullptr; sp->hstep = nullptr;
    sp->fin_s = nullptr; sp->fin_i = nullptr; sp->fin_row_offset = 0; sp->fin_stride = 0; sp->ticket = nullptr; sp->fin_qmap = nullptr;
    sp->prefilter = 0; sp->gate = nullptr; sp->gate_epoch = 0; sp->pf_image = nullptr; sp->pf_format = 0; sp->qpw = pl.qpw;
    sp->list_sm = pl.list_sm; sp->prog = nullptr; sp->prog_epoch = 0;
    sp->part_s = reinterpret_cast<float *>(ws + pl.off_part_s);
    sp->part_i = reinterpret_cast<uint32_t *>(ws + pl.off_part_i);
    sp->rows_per_stream = pl.rows_per_stream; sp->n_streams = pl.n_streams; sp->n_qtiles = pl.n_qtiles;
    sp->qwb = pl.qwb; sp->n_qgroups = pl.n_qgroups; sp->n_sgroups = pl.n_sgroups; sp->P = pl.P;
#ifdef MS_STAMP
    sp->stamps = ms_stamp_buffer();
#endif
}

// Sample pass: scan the first prepass_tiles tiles of every stream, merge, and leave the k-th
// best score per query in the workspace (off_ub_s doubles as the buffer) as the lower bound
// of the full pass.  Outputs of the merge go to scratch inside the partial-list area's tail.
int run_prepass(const ScanPlan &pl, ScanParams *sp, int nq, char *ws, hipStream_t st) {
    if (pl.prepass_tiles <= 0) return MS_OK;
    ScanParams s0 = *sp;
    s0.max_tiles = pl.prepass_tiles;
    s0.lb_s = nullptr;
    int rc = (s0.prefilter && s0.pf_image != nullptr) ? (s0.pf_format != MS_PF_BF16X3 ? ms_launch_sample_pf16(pl, s0, st) : ms_launch_sample_pf2(pl, s0, st))
             : ((pl.qwb == 4 && loader_wave_setting()) ? ms_launch_sample_loader(pl, s0, st) : launch_scan(pl, s0, st));
    if (rc) return rc;
    float *lb = reinterpret_cast<float *>(ws + pl.off_lb_s);
    // the sample kernel leaves at most 2 entries per (query, stream) and merges the 4 / qwb streams of a workgroup: only the
    // first `ranks` entries of a list can be valid.  Their k-th largest value is the bound: no rows, no sorted list needed.
    int ranks = (2 * (4 / pl.qwb) < s0.k) ? 2 * (4 / pl.qwb) : s0.k;
    // (few query tiles: a workgroup's list holds the maxima of 4 / qwb streams; the best two or so per list bound nearly as
    //  well as all eight, and the selection below reads a quarter of the values: 12 -> 5 us)
    {
        const int enough = (s0.k + pl.P - 1) / pl.P;                 // ranks * P >= k values are needed for a bound at all
        const int cap = 512 / pl.P > enough ? 512 / pl.P : enough;
        if (ranks > cap) ranks = cap > 1 ? cap : 1;
    }
    {   // diagnostics / tuning: MS_BOUND_RANKS caps how many entries per sample list the bound selection reads (fewer = a cheaper, weaker bound)
        static const int cap_env = [] { const char *e = getenv("MS_BOUND_RANKS"); return e ? atoi(e) : 0; }();
        if (cap_env > 0 && cap_env < ranks && (int64_t)cap_env * pl.P >= 4 * (int64_t)s0.k) ranks = cap_env;
    }
    const int vpl = (ranks * pl.P + 63) / 64;
    const size_t sm_stride = s0.list_sm ? (size_t)pl.nq_pad * s0.k : 0;       // (the image scans, and the loader-wave kernel where its merge allows, write stream-major lists)
    if (vpl > 32 && sm_stride != 0) MS_FAIL(MS_ERR_RANGE, "internal: %d sample lists of the image scan exceed the bound selection", pl.P);
    if (vpl <= 32) {
        const bool hist_on = pl.qwb == 4 && loader_wave_setting() && hist_setting();
        uint32_t *hist = hist_on ? reinterpret_cast<uint32_t *>(ws + pl.off_hist) : nullptr;
        float *hstep = reinterpret_cast<float *>(ws + pl.off_hstep);
#define MS_BOUND(V) hipLaunchKernelGGL(ms_sample_bound_kernel<V>, dim3(nq), dim3(64), 0, st, s0.part_s, pl.P, s0.k, ranks, lb, hist, hstep, s0.gate, s0.gate_epoch, sm_stride)
        if (vpl <= 4) { MS_BOUND(4); }
        else if (vpl <= 8) { MS_BOUND(8); }
        else if (vpl <= 16) { MS_BOUND(16); }
        else { MS_BOUND(32); }
#undef MS_BOUND
        MS_LAUNCH_CHECK("ms_sample_bound_kernel");
        if (hist_on) { sp->hist = hist; sp->hstep = hstep; }
    } else {
        float *scratch_s = reinterpret_cast<float *>(ws + pl.off_scr_s);
        int64_t *scratch_i = reinterpret_cast<int64_t *>(ws + pl.off_scr_i);
        uint32_t *lb_i = reinterpret_cast<uint32_t *>(ws + pl.off_lb_i);
        rc = launch_merge(pl, s0, nq, s0.k, 0, scratch_s, scratch_i, s0.k, 0, lb, lb_i, st);
        if (rc) return rc;
    }
    sp->lb_s = lb;
    return MS_OK;
}

// Prepare queries (+ inverse norms if absent) and fill ScanParams for the first pass.
int prepare_scan(const ScanPlan &pl, const float *db, int64_t n, const float *q, int nq, int mode,
                 const float *inv_norm, const float *lengths, const float *qlen, float mincov, char *ws,
                 hipStream_t st, ScanParams *sp) {
    if (!queries_used_in_place(q, mode, pl)) {
        float *qn = reinterpret_cast<float *>(ws + pl.off_qn);
        hipLaunchKernelGGL(ms_prepare_queries_kernel, dim3((pl.nq_pad + 3) / 4), dim3(256), 0, st, q, nq, pl.nq_pad,
                           mode != MS_MODE_IP_PRENORM ? 1 : 0, mode == MS_MODE_IP_NORMQ ? 1e-12f : 1e-8f, qn);
        MS_LAUNCH_CHECK("ms_prepare_queries_kernel");
    }
    if (mode == MS_MODE_COSINE_RAW && inv_norm == nullptr && n > 0) {
        float *inv_ws = reinterpret_cast<float *>(ws + pl.off_inv);
        const int64_t blocks = (n + 3) / 4;
        hipLaunchKernelGGL(ms_row_inv_norms_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, st,
                           db, n, 1e-8f, inv_ws);
        MS_LAUNCH_CHECK("ms_row_inv_norms_kernel");
    }
    fill_scan_params(pl, db, n, q, nq, inv_norm, lengths, qlen, mincov, ws, mode, sp);
    return MS_OK;
}

}  // namespace

extern "C" {

int ms_version(void) { return 210; }      // 200: pf_format in the prefilter entry points (round 5); 210: ms_device_pci_bus_id, ms_debug_prefilter_poison (round 6)
const char *ms_last_error(void) { return ms_err_buf; }

int ms_device_count(void) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

int ms_device_cu_count(void) { return cu_count_cached(); }

int ms_device_pci_bus_id(char *buf, int len) {
    if (buf == nullptr || len < 16) MS_FAIL(MS_ERR_ARG, "ms_device_pci_bus_id: need a buffer of >= 16 bytes");
    int dev = 0;
    MS_HIP_CHECK(hipGetDevice(&dev));
    MS_HIP_CHECK(hipDeviceGetPCIBusId(buf, len, dev));
    return MS_OK;
}

void ms_small_batch_thresholds(int *fused_merge_max_nq, int *inkernel_norm_max_nq) {
    if (fused_merge_max_nq != nullptr) *fused_merge_max_nq = fused_merge_setting();
    if (inkernel_norm_max_nq != nullptr) *inkernel_norm_max_nq = inkernel_norm_setting();
}
int ms_prefilter_max_k(void) { return MS_PREFILTER_MAX_K; }
int64_t ms_pf_few_min_rows(int nq) { return pf_few_min_rows(nq); }

int ms_l2_normalize_rows(float *x, int64_t n, int d, float eps, ms_stream_t stream) {
    if (d != MS_DIM) MS_FAIL(MS_ERR_ARG, "ms_l2_normalize_rows: d must be %d (got %d)", MS_DIM, d);
    if (n < 0 || (n > 0 && x == nullptr)) MS_FAIL(MS_ERR_ARG, "ms_l2_normalize_rows: bad arguments");
    if (n == 0) return MS_OK;
    const int64_t blocks = (n + 3) / 4;
    hipLaunchKernelGGL(ms_normalize_rows_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0,
                       (hipStream_t)stream, x, x, n, eps);
    MS_LAUNCH_CHECK("ms_normalize_rows_kernel");
    return MS_OK;
}

int ms_l2_normalize_rows_to(const float *x, float *y, int64_t n, int d, float eps, ms_stream_t stream) {
    if (d != MS_DIM) MS_FAIL(MS_ERR_ARG, "ms_l2_normalize_rows_to: d must be %d (got %d)", MS_DIM, d);
    if (n < 0 || (n > 0 && (x == nullptr || y == nullptr))) MS_FAIL(MS_ERR_ARG, "ms_l2_normalize_rows_to: bad arguments");
    if (n == 0) return MS_OK;
    const int64_t blocks = (n + 3) / 4;
    hipLaunchKernelGGL(ms_normalize_rows_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0,
                       (hipStream_t)stream, x, y, n, eps);
    MS_LAUNCH_CHECK("ms_normalize_rows_kernel");
    return MS_OK;
}

int ms_row_inv_norms(const float *x, int64_t n, int d, float eps, float *inv_norm, ms_stream_t stream) {
    if (d != MS_DIM) MS_FAIL(MS_ERR_ARG, "ms_row_inv_norms: d must be %d (got %d)", MS_DIM, d);
    if (n < 0 || (n > 0 && (x == nullptr || inv_norm == nullptr))) MS_FAIL(MS_ERR_ARG, "ms_row_inv_norms: bad arguments");
    if (n == 0) return MS_OK;
    const int64_t blocks = (n + 3) / 4;
    hipLaunchKernelGGL(ms_row_inv_norms_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0,
                       (hipStream_t)stream, x, n, eps, inv_norm);
    MS_LAUNCH_CHECK("ms_row_inv_norms_kernel");
    return MS_OK;
}

size_t ms_ip_topk_workspace_bytes(int64_t n, int nq, int k) {
    if (n < 0 || nq < 1 || k < 1) return 0;
    return make_plan(n, nq, k, cu_count_cached()).total;
}

int ms_ip_topk_prepare(const float *db, int64_t n, const float *q, int nq, int k, int mode, const float *inv_norm,
                       const float *lengths, const float *qlen, float mincov, void *workspace, size_t workspace_bytes,
                       ms_stream_t stream) {
    int rc = check_search_args(db, n, q, nq, k, mode, inv_norm, lengths, qlen);
    if (rc) return rc;
    if (k > 64) MS_FAIL(MS_ERR_ARG, "ms_ip_topk_prepare: k <= 64 only (use ms_ip_topk)");
    const ScanPlan pl = make_plan(n, nq, k, cu_count_cached());
    if (workspace == nullptr || workspace_bytes < pl.total)
        MS_FAIL(MS_ERR_WORKSPACE, "ms_ip_topk_prepare: workspace %zu < %zu bytes", workspace_bytes, pl.total);
    ScanParams sp;
    rc = prepare_scan(pl, db, n, q, nq, mode, inv_norm, lengths, qlen, mincov, (char *)workspace, (hipStream_t)stream, &sp);
    if (rc) return rc;
    rc = run_prepass(pl, &sp, nq, (char *)workspace, (hipStream_t)stream);
    if (rc == MS_OK && sp.hist != nullptr) hist_mark_clean(workspace, n, nq, k);      // (ms_sample_bound_kernel zeroed the counters)
    return rc;
}

int ms_ip_topk_scan(const float *db, int64_t n, const float *q, int nq, int k, int mode, const float *inv_norm,
                    const float *lengths, const float *qlen, float mincov, void *workspace, size_t workspace_bytes,
                    ms_stream_t stream) {
    int rc = check_search_args(db, n, q, nq, k, mode, inv_norm, lengths, qlen);
    if (rc) return rc;
    if (k > 64) MS_FAIL(MS_ERR_ARG, "ms_ip_topk_scan: k <= 64 only (use ms_ip_topk)");
    const ScanPlan pl = make_plan(n, nq, k, cu_count_cached());
    if (workspace == nullptr || workspace_bytes < pl.total)
        MS_FAIL(MS_ERR_WORKSPACE, "ms_ip_topk_scan: workspace %zu < %zu bytes", workspace_bytes, pl.total);
    char *ws = (char *)workspace;
    ScanParams sp;
    // same parameters as ms_ip_topk_prepare left in the workspace (queries, inverse norms, lower bound)
    fill_scan_params(pl, db, n, q, nq, inv_norm, lengths, qlen, mincov, ws, mode, &sp);
    if (pl.prepass_tiles > 0) {
        sp.lb_s = reinterpret_cast<const float *>(ws + pl.off_lb_s);
        if (pl.qwb == 4 && loader_wave_setting() && hist_setting()) {      // as run_prepass left them
            sp.hist = reinterpret_cast<uint32_t *>(ws + pl.off_hist);
            sp.hstep = reinterpret_cast<const float *>(ws + pl.off_hstep);
            // the counters must start at zero for EVERY scan (a second scan after one prepare would otherwise count rows twice)
            if (!hist_take_clean(workspace, n, nq, k))
                MS_HIP_CHECK(hipMemsetAsync(sp.hist, 0, (size_t)pl.nq_pad * 16 * sizeof(uint32_t), (hipStream_t)stream));
        }
    }
    return launch_scan(pl, sp, (hipStream_t)stream);
}

int ms_ip_topk_finish(int64_t n, int64_t row_offset, int nq, int k, float *out_scores, int64_t *out_idx,
                      void *workspace, size_t workspace_bytes, ms_stream_t stream) {
    if (k < 1 || k > 64 || nq < 1 || out_scores == nullptr || out_idx == nullptr)
        MS_FAIL(MS_ERR_ARG, "ms_ip_topk_finish: bad arguments");
    const ScanPlan pl = make_plan(n, nq, k, cu_count_cached());
    if (workspace == nullptr || workspace_bytes < pl.total)
        MS_FAIL(MS_ERR_WORKSPACE, "ms_ip_topk_finish: workspace %zu < %zu bytes", workspace_bytes, pl.total);
    char *ws = (char *)workspace;
    ScanParams sp;
    sp.part_s = reinterpret_cast<float *>(ws + pl.off_part_s);
    sp.part_i = reinterpret_cast<uint32_t *>(ws + pl.off_part_i);
    return launch_merge(pl, sp, nq, pl.k_pass, row_offset, out_scores, out_idx, k, 0, nullptr, nullptr, (hipStream_t)stream, nullptr, nullptr, 0,
                        pl.list_sm ? (size_t)pl.nq_pad * pl.k_pass : 0);
}

int ms_ip_topk(const float *db, int64_t n, int64_t row_offset, const float *q, int nq, int k, int mode,
               const float *inv_norm, const float *lengths, const float *qlen, float mincov, float *out_scores,
               int64_t *out_idx, void *workspace, size_t workspace_bytes, ms_stream_t stream) {
    int rc = check_search_args(db, n, q, nq, k, mode, inv_norm, lengths, qlen);
    if (rc) return rc;
    if (out_scores == nullptr || out_idx == nullptr) MS_FAIL(MS_ERR_ARG, "ms_ip_topk: NULL outputs");
    const ScanPlan pl = make_plan(n, nq, k, cu_count_cached());
    if (workspace == nullptr || workspace_bytes < pl.total)
        MS_FAIL(MS_ERR_WORKSPACE, "ms_ip_topk: workspace %zu < %zu bytes", workspace_bytes, pl.total);
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    ScanParams sp;
    rc = prepare_scan(pl, db, n, q, nq, mode, inv_norm, lengths, qlen, mincov, ws, st, &sp);
    if (rc) return rc;
    float *ub_s = reinterpret_cast<float *>(ws + pl.off_ub_s);
    uint32_t *ub_i = reinterpret_cast<uint32_t *>(ws + pl.off_ub_i);
    rc = run_prepass(pl, &sp, nq, ws, st);
    if (rc) return rc;
    hist_invalidate(workspace);     // (a staged ms_ip_topk_scan on this workspace must not take the counters for clean afterwards)
    // ceil(k / 64) passes; pass p returns ranks [64p, 64p + kp) using the last entry of pass
    // p-1 as an exclusive upper bound in the total order.
    // a handful of queries, one pass: the scan launch merges its own lists (ms_scan_body, last workgroup of a query group)
    char *blk = (k <= 64 && pl.qwb < 4 && nq <= fused_merge_setting() && pl.P <= 256 && (size_t)pl.k_pass * pl.P <= 4224)
                    ? sync_block_for(ws) : nullptr;
    if (blk != nullptr) {
        uint32_t *ticket = reinterpret_cast<uint32_t *>(blk);
        sp.fin_s = out_scores; sp.fin_i = out_idx; sp.fin_row_offset = row_offset; sp.fin_stride = k; sp.ticket = ticket;
        sp.k = pl.k_pass;
        return launch_scan(pl, sp, st);
    }
    for (int col0 = 0; col0 < k; col0 += 64) {
        const int kp = (k - col0) < 64 ? (k - col0) : 64;
        sp.k = kp;
        // the list stride inside the workspace is the pass's own k
        rc = launch_scan(pl, sp, st);
        if (rc) return rc;
        const bool more = col0 + 64 < k;
        rc = launch_merge(pl, sp, nq, kp, row_offset, out_scores, out_idx, k, col0, more ? ub_s : nullptr,
                          more ? ub_i : nullptr, st, nullptr, nullptr, 0, sp.list_sm ? (size_t)pl.nq_pad * kp : 0);
        if (rc) return rc;
        sp.ub_s = ub_s;
        sp.ub_i = ub_i;
    }
    return MS_OK;
}

// ---- prefiltered search (>= 3 query tiles, k <= 48) -------------------------------------------------------------------
// stages: 1 = queries + sample pass + bound, 2 = the scan launch, 4 = merge + exact re-scoring + the exact pass over the queries
// whose proof failed
namespace {
// candidates kept per query: twice k for short lists, at least 8-16 spare entries for long ones (the proof needs the rows within the
// error bound of the k-th best to fit; more spare entries = fewer queries for the exact pass on clustered data)
int pf_list_len(int k) { return k <= 5 ? 10 : (k <= 10 ? 20 : (k <= 24 ? 32 : (k <= MS_PREFILTER_MAX_K ? 64 : 0))); }
int pace_setting() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("MS_PF_PACE"); v = e ? atoi(e) : 1; }      // diagnostics: 0 = the query groups of a row stream run free
    return v;
}
int pf_rawq_setting() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("MS_PF_RAWQ"); v = e ? atoi(e) : 1; }      // diagnostics: 0 = always the prepared (normalised) copy of the queries
    return v;
}
int prefilter_setting() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("MS_PREFILTER"); v = e ? atoi(e) : 1; }      // diagnostics: 0 = always the fp32 scan
    return v;
}
uint32_t next_epoch() {                                // never 0: the gate word starts at zero
    static std::atomic<uint32_t> e{1};
    uint32_t v = e.fetch_add(1);
    if (v == 0) v = e.fetch_add(1);
    return v;
}
// Workspace of a prefiltered search: [the larger of the prefilter scan's and the exact scan's plans | candidate lists (approximate
// scores, rows) | per-query flags | the compacted batch of the exact pass: queries, bounds, lengths, map | its device plan]
struct PfLayout {
    ScanPlan pf, exact;
    size_t off_as, off_ai, off_flag, off_qn_c, off_lb_c, off_qlen_c, off_qmap, off_dp, off_xs, off_xi, total;
    int kp, exact_grid_max, exact_P_max;
    bool ok;
};
bool pf_format_ok(int f) { return f == MS_PF_BF16X3 || f == MS_PF_F16X2 || f == MS_PF_F16X1; }
float pf_err_coef(bool image, int format) {
    if (!image || format == MS_PF_BF16X3) return MS_PF_ERR;
    return format == MS_PF_F16X2 ? MS_PF_ERR_F16X2 : MS_PF_ERR_F16X1;
}
PfLayout pf_layout(int64_t n, int nq, int k, int mode, bool image, int format = MS_PF_BF16X3) {
    PfLayout L;
    const int cus = cu_count_cached();
    L.kp = pf_list_len(k);
    L.exact = make_plan(n, nq, k, cus);
    const bool ip = mode == MS_MODE_IP_PRENORM || mode == MS_MODE_IP_NORMQ;
    // without an image the rows are split in registers (round 3's kernel: inner-product modes only, the loader-wave form)
    // (one or two query tiles -- the reference's own CLI regime -- are HBM-bound: over the fp16 image the scan reads half the bytes of
    //  the fp32 rows; worth the fixed cost of the pipeline around it from a few million rows: pf_few_min_rows)
    const bool few_ok = image && format != MS_PF_BF16X3 && n >= pf_few_min_rows(nq);
    L.ok = prefilter_setting() && L.kp > 0 && n >= 65536 && (L.exact.qwb == 4 || few_ok) &&
           (image ? (ip || mode == MS_MODE_COSINE_UNIT) : (ip && loader_wave_setting() != 0));
    L.exact_grid_max = L.exact.grid; L.exact_P_max = L.exact.P;
    if (!L.ok) { L.total = L.exact.total; L.off_as = L.off_ai = L.off_flag = L.off_qn_c = L.off_lb_c = L.off_qlen_c = L.off_qmap = L.off_dp = L.off_xs = L.off_xi = 0; return L; }
    // two query tiles per wave (8 per workgroup) from 5 query tiles, while the lists leave room for it
    const int qpw = image ? ((L.exact.n_qtiles >= 5 && L.kp <= 32) ? 2 : 1) : 0;
    L.pf = make_plan(n, nq, L.kp, cus, qpw, (image && format != MS_PF_BF16X3) ? 64 : 32);
    // the exact pass runs over 1 .. nq queries, decomposed on the device: the launch grid and the merge's LDS cover every case
    size_t lists_max = 0;        // (its partial lists: nq_pad * P entries per rank, whichever decomposition the device picks)
    for (int qt = 1; qt <= L.exact.n_qtiles; ++qt) {
        ScanDevPlan d;
        ms_plan_core(n, qt * 32 < nq ? qt * 32 : nq, cus, &d);
        if (d.grid > L.exact_grid_max) L.exact_grid_max = d.grid;
        if (d.P > L.exact_P_max) L.exact_P_max = d.P;
        if ((size_t)d.nq_pad * d.P > lists_max) lists_max = (size_t)d.nq_pad * d.P;
    }
    if (L.exact_P_max > 256 || L.pf.P > 256 || nq >= (1 << 20)) {       // the merges behind the image scan and the exact pass stage <= 256 lists per query (a device with more than 256 CUs): ms_ip_topk
        L.ok = false;
        L.total = L.exact.total; L.off_as = L.off_ai = L.off_flag = L.off_qn_c = L.off_lb_c = L.off_qlen_c = L.off_qmap = L.off_dp = L.off_xs = L.off_xi = 0;
        return L;
    }
    size_t off = L.pf.total > L.exact.total ? L.pf.total : L.exact.total;
    const int nq_pad = L.pf.nq_pad > L.exact.nq_pad ? L.pf.nq_pad : L.exact.nq_pad;
    L.off_as = off;     off += ms_align_up((size_t)nq_pad * L.kp * sizeof(float), 256);
    L.off_ai = off;     off += ms_align_up((size_t)nq_pad * L.kp * sizeof(int64_t), 256);
    L.off_flag = off;   off += ms_align_up((size_t)nq_pad * sizeof(uint32_t), 256);
    L.off_qn_c = off;   off += ms_align_up((size_t)nq_pad * MS_DIM * sizeof(float), 256);
    L.off_lb_c = off;   off += ms_align_up((size_t)nq_pad * sizeof(float), 256);
    L.off_qlen_c = off; off += ms_align_up((size_t)nq_pad * sizeof(float), 256);
    L.off_qmap = off;   off += ms_align_up((size_t)nq_pad * sizeof(int), 256);
    L.off_dp = off;     off += 256;
    L.off_xs = off;     off += ms_align_up(lists_max * L.exact.k_pass * sizeof(float), 256);
    L.off_xi = off;     off += ms_align_up(lists_max * L.exact.k_pass * sizeof(uint32_t), 256);
    L.total = off;
    return L;
}
int pf_run(int stages, const float *db, const void *image, int format, int64_t n, int64_t row_offset, const float *q, int nq, int k, int mode,
           const float *lengths, const float *qlen, float mincov, float row_norm_bound, float *out_scores, int64_t *out_idx,
           void *workspace, size_t workspace_bytes, hipStream_t st) {
    int rc = check_search_args(db, n, q, nq, k, mode, nullptr, lengths, qlen);
    if (rc) return rc;
    if (mode == MS_MODE_COSINE_RAW) MS_FAIL(MS_ERR_ARG, "ms_ip_topk_prefiltered: MS_MODE_COSINE_RAW is not served (normalise the rows once: MS_MODE_COSINE_UNIT)");
    if (image != nullptr && !pf_format_ok(format)) MS_FAIL(MS_ERR_ARG, "ms_ip_topk_prefiltered: unknown pf_format %d", format);
    const PfLayout L = pf_layout(n, nq, k, mode, image != nullptr, format);
    if (workspace == nullptr || workspace_bytes < L.total)
        MS_FAIL(MS_ERR_WORKSPACE, "ms_ip_topk_prefiltered: workspace %zu < %zu bytes", workspace_bytes, L.total);
    char *blk = L.ok && row_norm_bound > 0.0f && row_norm_bound < INFINITY ? sync_block_for(workspace) : nullptr;
    if (blk == nullptr) {          // shapes the prefilter does not serve: the fp32 path, same stages
        if (stages == 7) return ms_ip_topk(db, n, row_offset, q, nq, k, mode, nullptr, lengths, qlen, mincov, out_scores, out_idx, workspace, workspace_bytes, st);
        if (stages == 1) return ms_ip_topk_prepare(db, n, q, nq, k, mode, nullptr, lengths, qlen, mincov, workspace, workspace_bytes, st);
        if (stages == 2) return ms_ip_topk_scan(db, n, q, nq, k, mode, nullptr, lengths, qlen, mincov, workspace, workspace_bytes, st);
        return ms_ip_topk_finish(n, row_offset, nq, k, out_scores, out_idx, workspace, workspace_bytes, st);
    }
    char *ws = (char *)workspace;
    const ScanPlan &pl = L.pf;
    ScanParams sp;
    // Raw queries over the fp16 image (MS_MODE_IP_NORMQ, MS_MODE_COSINE_UNIT; round 6): no query-preparation launch -- the sample pass and the
    // scan read the caller's array and normalise approximately in their set-up, the re-scoring launch normalises exactly (MS_PF_RAWQ=0: the
    // prepared copy as before)
    const float rawq_eps = (image != nullptr && format != MS_PF_BF16X3 && (mode == MS_MODE_IP_NORMQ || mode == MS_MODE_COSINE_UNIT) &&
                            ((uintptr_t)q & 15) == 0 && pf_rawq_setting()) ? (mode == MS_MODE_IP_NORMQ ? 1e-12f : 1e-8f) : 0.0f;
    if (stages & 1) {
        if (rawq_eps > 0.0f) fill_scan_params(pl, db, n, q, nq, nullptr, lengths, qlen, mincov, ws, mode, &sp);
        else rc = prepare_scan(pl, db, n, q, nq, mode, nullptr, lengths, qlen, mincov, ws, st, &sp);
        if (rc) return rc;
        if (rawq_eps > 0.0f) { sp.qn = q; sp.qnorm_eps = 0.0f; sp.qraw_eps = rawq_eps; }
        sp.prefilter = 1; sp.pf_image = image; sp.pf_format = format;
        rc = run_prepass(pl, &sp, nq, ws, st);
        if (rc) return rc;
        if (sp.hist != nullptr) hist_mark_clean(workspace, n, nq, L.kp);
    } else {
        fill_scan_params(pl, db, n, q, nq, nullptr, lengths, qlen, mincov, ws, mode, &sp);
        if (rawq_eps > 0.0f) { sp.qn = q; sp.qnorm_eps = 0.0f; sp.qraw_eps = rawq_eps; }
        sp.prefilter = 1; sp.pf_image = image; sp.pf_format = format;
        if (pl.prepass_tiles > 0) {
            sp.lb_s = reinterpret_cast<const float *>(ws + pl.off_lb_s);
            if (hist_setting()) {
                sp.hist = reinterpret_cast<uint32_t *>(ws + pl.off_hist);
                sp.hstep = reinterpret_cast<const float *>(ws + pl.off_hstep);
                if ((stages & 2) && !hist_take_clean(workspace, n, nq, L.kp))
                    MS_HIP_CHECK(hipMemsetAsync(sp.hist, 0, (size_t)pl.nq_pad * 16 * sizeof(uint32_t), st));
            }
        }
    }
    if ((stages & 1) && (stages & 2) && sp.hist != nullptr) (void)hist_take_clean(workspace, n, nq, L.kp);
    sp.k = pl.k_pass;
    {
        static const int dbg = [] { const char *e = getenv("MS_PF_DEBUG"); return e ? atoi(e) : 0; }();
        sp.debug_flags = dbg;
    }
    if (stages & 2) {
        if (image != nullptr && format != MS_PF_BF16X3 && pl.n_qgroups >= 2 && pl.n_qgroups <= 16 && pace_setting()) {
            // several query groups walk every row stream: they pace each other through progress words (ms_scan_pf16.h) so that a tile
            // one of them fetched is still in the XCD's L2 when the others want it
            static std::atomic<uint32_t> pace_epoch{1};
            sp.prog = reinterpret_cast<uint32_t *>(ws + pl.off_prog);
            sp.prog_epoch = pace_epoch.fetch_add(1) & 0xFFu;
        }
        rc = launch_scan(pl, sp, st);
        if (rc) return rc;
    }
    if (stages & 4) {
        if (out_scores == nullptr || out_idx == nullptr) MS_FAIL(MS_ERR_ARG, "ms_ip_topk_prefiltered: NULL outputs");
        float *as = reinterpret_cast<float *>(ws + L.off_as);
        int64_t *ai = reinterpret_cast<int64_t *>(ws + L.off_ai);
        uint32_t *gate = reinterpret_cast<uint32_t *>(blk + 256);
        const uint32_t epoch = next_epoch();
        const ScanPlan &px = L.exact;
        uint32_t *flag = reinterpret_cast<uint32_t *>(ws + L.off_flag);
        float *qn_c = reinterpret_cast<float *>(ws + L.off_qn_c), *lb_c = reinterpret_cast<float *>(ws + L.off_lb_c);
        float *qlen_c = reinterpret_cast<float *>(ws + L.off_qlen_c);
        int *qmap = reinterpret_cast<int *>(ws + L.off_qmap);
        ScanDevPlan *dp = reinterpret_cast<ScanDevPlan *>(ws + L.off_dp);
        PfCompact cp;
        cp.qn_c = qn_c; cp.lb_c = lb_c; cp.qlen_c = qlen_c; cp.qmap = qmap; cp.dp = dp; cp.gate = gate; cp.epoch = epoch; cp.n = n;
        cp.cus = cu_count_cached(); cp.nq = nq; cp.ticket_base = sync_take_tickets(blk, (uint32_t)nq);
        // ONE launch: merge of the per-stream candidate lists (sparse; stream-major behind the image scans) + exact re-scoring + proof + compaction
        PfRescore ra;
        ra.db = db; ra.qn = sp.qn; ra.as = as; ra.ai = ai; ra.lengths = lengths; ra.qlen = qlen; ra.out_s = out_scores; ra.out_i = out_idx; ra.flag = flag;
        ra.row_offset = row_offset; ra.err_coef = pf_err_coef(image != nullptr, format) * row_norm_bound; ra.mincov = mincov;
        ra.q_eps = rawq_eps;
        ra.k = k; ra.kp = L.kp; ra.fp16_range = (image != nullptr && format != MS_PF_BF16X3) ? 1 : 0; ra.cp = cp;
        const size_t block_lds = (size_t)MS_BLOCK_MERGE_SCRATCH + (((size_t)L.kp * pl.P + 3) & ~(size_t)3) * sizeof(uint2);
        if (block_lds > 156 * 1024 || pl.P > 256) MS_FAIL(MS_ERR_RANGE, "internal: %d candidate lists of %d entries exceed the merge", pl.P, L.kp);
        if (block_lds > 48 * 1024)
            MS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ms_merge_rescore_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)block_lds));
        hipLaunchKernelGGL(ms_merge_rescore_kernel, dim3(nq), dim3(256), block_lds, st, sp.part_s, sp.part_i, pl.P, as, ai,
                           sp.list_sm ? (size_t)pl.nq_pad * L.kp : (size_t)0, ra);
        MS_LAUNCH_CHECK("ms_merge_rescore_kernel");
        // The exact pass, for the flagged queries ONLY (the reference's semantics are per query: dbsearch.py:234-242): an fp32 scan
        // and a merge over the compacted batch, decomposed on the device (ScanDevPlan), both returning at once when no query was
        // flagged.  No sample pass: the k-th best exact score among a query's candidates is already a lower bound on its k-th
        // best (k rows score at least that), and a tight one.  Always the kernels for any number of query tiles (ms_scan_kernel).
        ScanParams sx;
        fill_scan_params(px, db, n, q, nq, nullptr, lengths, lengths != nullptr ? qlen_c : nullptr, mincov, ws, mode, &sx);
        sx.qn = qn_c;                        // (prepared -- normalised where the mode asks for it -- by the prefilter's stage 1)
        sx.qnorm_eps = 0.0f;
        sx.gate = gate; sx.gate_epoch = epoch;
        sx.lb_s = lb_c;
        sx.k = px.k_pass;
        sx.dev_plan = dp;
        sx.part_s = reinterpret_cast<float *>(ws + L.off_xs);
        sx.part_i = reinterpret_cast<uint32_t *>(ws + L.off_xi);
        ScanPlan pg = px;
        pg.grid = L.exact_grid_max; pg.P = L.exact_P_max; pg.qwb = 1;     // (qwb = 1: routes to ms_scan_kernel, whose decomposition is the device plan's)
        sx.qwb = 1;
        sx.list_sm = 0;                      // (ms_scan_kernel writes rank-major lists)
        // A handful of queries (the reference's own CLI regime over the fp16 image of a large database; round 6): the exact pass merges
        // inside its own launch -- the last workgroup of the query group, as ms_ip_topk does for such batches -- so that ONE gated launch
        // follows the re-scoring instead of two (each costs ~5 us even when it returns at once).  More queries: the last workgroup would
        // merge them one after the other (4.4 us each) while 255 others have finished; the merge launch stays.
        // (up to 8 queries whatever ms_ip_topk's own threshold is: the serial merge only runs for queries that were flagged -- rare -- while the
        //  second launch costs every call; MS_PF_FUSE_EXACT_MAX_NQ overrides)
        static const int fuse_max = [] { const char *e = getenv("MS_PF_FUSE_EXACT_MAX_NQ"); return e ? atoi(e) : 8; }();
        const bool fuse_exact = nq <= fuse_max && L.exact_P_max <= 256 && (size_t)px.k_pass * L.exact_P_max <= 4224;
        if (fuse_exact) {
            sx.fin_s = out_scores; sx.fin_i = out_idx; sx.fin_row_offset = row_offset; sx.fin_stride = k; sx.fin_qmap = qmap;
            sx.ticket = reinterpret_cast<uint32_t *>(blk);
        }
        rc = launch_scan(pg, sx, st);
        if (rc) return rc;
        if (!fuse_exact) {
            rc = launch_merge(pg, sx, nq, px.k_pass, row_offset, out_scores, out_idx, k, 0, nullptr, nullptr, st, dp, qmap);
            if (rc) return rc;
        }
    }
    return MS_OK;
}
}  // namespace

size_t ms_pf_image_bytes(int64_t n, int pf_format) {
    if (n < 0 || !pf_format_ok(pf_format)) return 0;
    if (pf_format == MS_PF_BF16X3) return (size_t)((n + 31) / 32) * 16384;
    return (size_t)((n + 63) / 64) * 16384 + 256;          // (+ the trailer the scan checks: magic, scale exponent, n)
}
float ms_pf_err_coef(int pf_format) { return pf_format_ok(pf_format) ? pf_err_coef(true, pf_format) : -1.0f; }

int ms_pf_build_image(const float *db, int64_t n, int pf_format, float row_norm_bound, void *image, ms_stream_t stream) {
    if (!pf_format_ok(pf_format)) MS_FAIL(MS_ERR_ARG, "ms_pf_build_image: unknown pf_format %d", pf_format);
    if (n < 0 || (n > 0 && db == nullptr) || image == nullptr) MS_FAIL(MS_ERR_ARG, "ms_pf_build_image: bad arguments");
    if (((uintptr_t)image & 15) != 0 || ((uintptr_t)db & 15) != 0) MS_FAIL(MS_ERR_ARG, "ms_pf_build_image: db and image must be 16-byte aligned");
    if (pf_format == MS_PF_BF16X3) {
        if (n == 0) return MS_OK;
        return ms_launch_pf_build_image(db, n, image, (hipStream_t)stream);
    }
    // fp16: every component of row * 2^sr must sit below 2^15 -- sr from the bound on the rows' norms
    if (!(row_norm_bound >= 9.094947e-13f && row_norm_bound <= 1.0995116e12f))
        MS_FAIL(MS_ERR_RANGE, "ms_pf_build_image: the fp16 image needs a row-norm bound in [2^-40, 2^40] (got %g): use MS_PF_BF16X3", (double)row_norm_bound);
    const int sr = 14 - ilogbf(row_norm_bound);
    return ms_launch_pf16_build_image(db, n, sr, image, (hipStream_t)stream);
}

size_t ms_ip_topk_prefiltered_workspace_bytes(int64_t n, int nq, int k) {
    if (n < 0 || nq < 1 || k < 1) return 0;
    size_t m = make_plan(n, nq, k, cu_count_cached()).total;
    for (int image = 0; image < 3; ++image) {       // no image, split-bf16 image (32-row tiles), fp16 image (64-row tiles)
        const size_t a = pf_layout(n, nq, k, image ? MS_MODE_COSINE_UNIT : MS_MODE_IP_PRENORM, image != 0, image == 2 ? MS_PF_F16X2 : MS_PF_BF16X3).total;
        if (a > m) m = a;
    }
    return m;
}

int ms_ip_topk_prefiltered(const float *db, const void *pf_image, int pf_format, int64_t n, int64_t row_offset, const float *q, int nq, int k, int mode,
                           const float *lengths, const float *qlen, float mincov, float row_norm_bound, float *out_scores,
                           int64_t *out_idx, void *workspace, size_t workspace_bytes, ms_stream_t stream) {
    if (k > 64) return ms_ip_topk(db, n, row_offset, q, nq, k, mode, nullptr, lengths, qlen, mincov, out_scores, out_idx, workspace, workspace_bytes, stream);
    return pf_run(7, db, pf_image, pf_format, n, row_offset, q, nq, k, mode, lengths, qlen, mincov, row_norm_bound, out_scores, out_idx, workspace,
                  workspace_bytes, (hipStream_t)stream);
}
int ms_ip_topk_prefiltered_prepare(const float *db, const void *pf_image, int pf_format, int64_t n, const float *q, int nq, int k, int mode,
                                   const float *lengths, const float *qlen, float mincov, float row_norm_bound, void *workspace,
                                   size_t workspace_bytes, ms_stream_t stream) {
    return pf_run(1, db, pf_image, pf_format, n, 0, q, nq, k, mode, lengths, qlen, mincov, row_norm_bound, nullptr, nullptr, workspace, workspace_bytes,
                  (hipStream_t)stream);
}
int ms_ip_topk_prefiltered_scan(const float *db, const void *pf_image, int pf_format, int64_t n, const float *q, int nq, int k, int mode,
                                const float *lengths, const float *qlen, float mincov, float row_norm_bound, void *workspace,
                                size_t workspace_bytes, ms_stream_t stream) {
    return pf_run(2, db, pf_image, pf_format, n, 0, q, nq, k, mode, lengths, qlen, mincov, row_norm_bound, nullptr, nullptr, workspace, workspace_bytes,
                  (hipStream_t)stream);
}
int ms_ip_topk_prefiltered_finish(const float *db, const void *pf_image, int pf_format, int64_t n, int64_t row_offset, const float *q, int nq, int k, int mode,
                                  const float *lengths, const float *qlen, float mincov, float row_norm_bound, float *out_scores,
                                  int64_t *out_idx, void *workspace, size_t workspace_bytes, ms_stream_t stream) {
    return pf_run(4, db, pf_image, pf_format, n, row_offset, q, nq, k, mode, lengths, qlen, mincov, row_norm_bound, out_scores, out_idx, workspace,
                  workspace_bytes, (hipStream_t)stream);
}
// Diagnostics for the tests (synchronises the device): the state the last prefiltered search on this workspace left behind --
// *flagged = how many of its queries needed the exact pass (0: the prefilter proved every answer).
int ms_debug_prefilter_state(void *workspace, unsigned int *gate_value, unsigned int *last_epoch, unsigned int *flagged) {
    char *blk = sync_block_for(workspace);
    if (blk == nullptr || hipDeviceSynchronize() != hipSuccess) return -1;
    unsigned int w[3] = {0, 0, 0};
    if (hipMemcpy(w, blk + 256, 12, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    *gate_value = w[0]; *last_epoch = w[1]; *flagged = w[2];
    return 0;
}

// Diagnostics for the tests (synchronises the device): overwrite the two counters of the compaction in the library-owned block of
// this workspace -- what an aborted launch, or a second stream on the same workspace, would leave behind.  The next prefiltered
// search on the workspace must then fail loudly (the gated launch traps: ms_gate_closed), never answer.
int ms_debug_prefilter_poison(void *workspace, unsigned int slot_counter, unsigned int ticket) {
    char *blk = sync_block_for(workspace);
    if (blk == nullptr || hipDeviceSynchronize() != hipSuccess) return -1;
    const unsigned int w[2] = {slot_counter, ticket};
    if (hipMemcpy(blk + 256 + 16, w, 8, hipMemcpyHostToDevice) != hipSuccess) return -1;
    return 0;
}

// Diagnostics (tools/pf_try.py): the candidate lists (approximate scores, rows) the last prefiltered search left in the workspace.
int ms_debug_prefilter_lists(void *workspace, int64_t n, int nq, int k, int image, float *as_host, int64_t *ai_host, int *kp_out) {
    // image: 0 = none, 1 = split-bf16 image, 2 = fp16 image (the layouts differ in their tile size only)
    const PfLayout L = pf_layout(n, nq, k, MS_MODE_IP_PRENORM, image != 0, image == 2 ? MS_PF_F16X2 : MS_PF_BF16X3);
    if (!L.ok || hipDeviceSynchronize() != hipSuccess) return -1;
    *kp_out = L.kp;
    if (hipMemcpy(as_host, (char *)workspace + L.off_as, (size_t)nq * L.kp * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (hipMemcpy(ai_host, (char *)workspace + L.off_ai, (size_t)nq * L.kp * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return 0;
}

int ms_topk_merge(const float *scores, const int64_t *idx, int S, int nq, int k, float *out_scores,
                  int64_t *out_idx, ms_stream_t stream) {
    if (S < 1 || nq < 1 || k < 1 || !scores || !idx || !out_scores || !out_idx)
        MS_FAIL(MS_ERR_ARG, "ms_topk_merge: need S >= 1, nq >= 1, k >= 1 and non-NULL buffers (S=%d)", S);
    if (S <= 64)
        hipLaunchKernelGGL(ms_kway_merge_kernel, dim3((nq + 63) / 64), dim3(64), 0, (hipStream_t)stream, scores, idx,
                           (int64_t)nq * k * 4, (int64_t)nq * k * 8, S, nq, k, out_scores, out_idx);
    else
        hipLaunchKernelGGL(ms_kway_merge_any_kernel, dim3((nq + 63) / 64), dim3(64), 0, (hipStream_t)stream, scores, idx,
                           (int64_t)nq * k * 4, (int64_t)nq * k * 8, S, nq, k, out_scores, out_idx);
    MS_LAUNCH_CHECK("ms_kway_merge_kernel");
    return MS_OK;
}

int ms_topk_merge_strided(const float *scores, const int64_t *idx, int64_t score_stride_bytes, int64_t idx_stride_bytes,
                          int S, int nq, int k, float *out_scores, int64_t *out_idx, ms_stream_t stream) {
    if (S < 1 || nq < 1 || k < 1 || !scores || !idx || !out_scores || !out_idx)
        MS_FAIL(MS_ERR_ARG, "ms_topk_merge_strided: need S >= 1, nq >= 1, k >= 1 and non-NULL buffers (S=%d)", S);
    if (score_stride_bytes % 4 != 0 || idx_stride_bytes % 8 != 0 || ((uintptr_t)idx & 7) != 0)
        MS_FAIL(MS_ERR_ARG, "ms_topk_merge_strided: strides / index pointer must keep float32 and int64 alignment");
    if (S <= 64)
        hipLaunchKernelGGL(ms_kway_merge_kernel, dim3((nq + 63) / 64), dim3(64), 0, (hipStream_t)stream, scores, idx,
                           score_stride_bytes, idx_stride_bytes, S, nq, k, out_scores, out_idx);
    else
        hipLaunchKernelGGL(ms_kway_merge_any_kernel, dim3((nq + 63) / 64), dim3(64), 0, (hipStream_t)stream, scores, idx,
                           score_stride_bytes, idx_stride_bytes, S, nq, k, out_scores, out_idx);
    MS_LAUNCH_CHECK("ms_kway_merge_kernel");
    return MS_OK;
}

}  // extern "C"
