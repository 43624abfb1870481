// Brute-force cosine / inner-product top-k over the 128-d domain-embedding database on gfx950.
//
// Reference arithmetic replaced (programs/Foldclass/dbsearch.py):
//   :75-81   search_query_against_db   cosine_similarity(db, q) * mask -> topk   (`.pt` DB)
//   :213-248 knn_exact_faiss           IndexFlat(IP).search per block + ResultHeap merge
//   :303-304 F.normalize(query_embeddings)
//
// Kernels
//   ms_normalize_rows_kernel / ms_row_inv_norms_kernel   one wave per 512-byte row
//   ms_scan_kernel<QW>   fused  S = D . Q^T  (fp32 MFMA 32x32x2)  +  per-query running top-k
//   ms_partial_merge_kernel   per query: merge the per-chunk lists, emit float32/int64 results
//   ms_kway_merge_kernel      public merge of S sorted lists (shards / blocks)
//
// Scan kernel layout (see DESIGN.md "scan kernel"):
//   * one workgroup = 4 waves; wave (qw, rw) owns query tile qw (32 queries, held for the
//     whole kernel as the MFMA B operand in 64 VGPRs) and row sub-tile rw of the staged tile;
//   * database rows stream HBM -> registers (16 B/lane, fully coalesced) -> LDS, XOR-swizzled
//     so that the ds_read_b128 A-operand reads (lane = row) are bank-conflict free;
//   * k-step s of the MFMA chain multiplies elements k = s (lanes 0-31) and k = 64 + s (lanes
//     32-63): every lane then reads 4 consecutive floats of its row per ds_read_b128.  The
//     accumulation order is therefore  s = 0..63: (k = s, k = 64 + s)  -- restated by
//     oracle/oracle.c:dot_ordered(order = 1), which reproduces these scores bit for bit;
//   * the 32x32 accumulator has the query on the lane (col = lane & 31) and 16 rows in
//     registers, so the top-k filter is a per-lane compare against that query's current
//     k-th best score; the rare survivors are inserted wave-cooperatively into a sorted
//     list in LDS (ms_wave_insert).  Rows are visited in ascending order, so "strictly
//     greater than the k-th best" is the exact (score desc, row asc) order.
#include "ms_common.h"

#include <math.h>

thread_local char ms_err_buf[512] = "";

// ------------------------------------------------------------------ normalisation ------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ __launch_bounds__(256) void ms_normalize_rows_kernel(float *x, int64_t n, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave0; r < n; r += nwaves) {
        float2 *p = reinterpret_cast<float2 *>(x + r * MS_DIM) + lane;
        float2 v = *p;
        const float ss = wave_sum(v.x * v.x + v.y * v.y);
        const float nrm = fmaxf(sqrtf(ss), eps);
        v.x = v.x / nrm;
        v.y = v.y / nrm;
        *p = v;
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

// ------------------------------------------------------------------ scan kernel --------
struct ScanParams {
    const float *db;        // [n,128]
    int64_t n;
    const float *qn;        // [nq_pad,128] prepared queries
    int nq;                 // real queries
    int nq_pad;
    int k;                  // entries per list this pass (<= 64)
    int mode;
    const float *inv_norm;  // [n] or NULL
    const float *lengths;   // [n] or NULL
    const float *qlen;      // [nq] or NULL
    float mincov;
    const float *ub_s;      // [nq_pad] exclusive upper bound of this pass, or NULL (first pass)
    const uint32_t *ub_i;
    float *part_s;          // [n_chunks, nq_pad, k]
    uint32_t *part_i;
    int rows_per_chunk;     // multiple of the tile height
    int n_chunks;
    int n_qgroups;
};

template <int QW>
__global__ __launch_bounds__(256, 2) void ms_scan_kernel(const ScanParams p) {
    constexpr int RW = 4 / QW;          // row sub-tiles per staged tile
    constexpr int TROWS = 32 * RW;      // rows per staged tile
    constexpr int LD_IT = 4 * RW;       // float4 loads per thread per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *tile = reinterpret_cast<f32x4 *>(smem);                       // TROWS x 32 float4, swizzled
    uint2 *lists = reinterpret_cast<uint2 *>(smem + (size_t)TROWS * 512); // [4 waves][32 queries][k]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qw = wave / RW, rw = wave % RW;
    const int r = lane & 31, h = lane >> 5;
    const int k = p.k;

    // block id -> (row chunk, query group).  Blocks that share a row chunk get ids that differ
    // by multiples of 8: the dispatcher deals ids round-robin over the 8 XCDs, so they land on
    // one XCD close in time and the chunk's rows are served by that XCD's L2 (speed only).
    const int bid = blockIdx.x;
    const int per_super = 8 * p.n_qgroups;
    const int super = bid / per_super, within = bid % per_super;
    int chunk = super * 8 + (within & 7);
    int qg = within >> 3;
    if (chunk >= p.n_chunks) return;   // last super-group may be ragged (uniform per block)

    const int64_t row_begin = (int64_t)chunk * p.rows_per_chunk;
    const int64_t row_end = (row_begin + p.rows_per_chunk < p.n) ? row_begin + p.rows_per_chunk : p.n;
    const int ntiles = (int)((row_end - row_begin + TROWS - 1) / TROWS);

    // this lane's query
    const int qidx = (qg * QW + qw) * 32 + r;
    const bool q_valid = qidx < p.nq;

    // B operand: query tile, resident for the whole kernel.  lane (q = r, h) holds
    // Q[q][64 h + s], s = 0..63.
    float qreg[64];
    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(p.qn + (size_t)qidx * MS_DIM + 64 * h);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const f32x4 v = src[t];
            qreg[4 * t + 0] = v.x; qreg[4 * t + 1] = v.y; qreg[4 * t + 2] = v.z; qreg[4 * t + 3] = v.w;
        }
    }
    const float my_qlen = (p.qlen != nullptr && q_valid) ? p.qlen[qidx] : 0.0f;
    float ubs = INFINITY;
    uint32_t ubi = 0;
    const bool has_ub = p.ub_s != nullptr;
    if (has_ub) { ubs = p.ub_s[qidx]; ubi = p.ub_i[qidx]; }

    // empty lists
    uint2 *my_lists = lists + (size_t)wave * 32 * k;
    for (int e = lane; e < 32 * k; e += 64) my_lists[e] = make_uint2(__float_as_uint(-INFINITY), MS_IDX_NONE);
    float tau = -INFINITY;

    // register prefetch of the first tile
    f32x4 pre[LD_IT];
    auto issue_loads = [&](int t) {
        const int64_t base_row = row_begin + (int64_t)t * TROWS;
#pragma unroll
        for (int it = 0; it < LD_IT; ++it) {
            const int f = it * 256 + tid;
            const int64_t grow = base_row + (f >> 5);
            if (grow < row_end)
                pre[it] = *(reinterpret_cast<const f32x4 *>(p.db + grow * MS_DIM) + (f & 31));
            else
                pre[it] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
    };
    issue_loads(0);

    for (int t = 0; t < ntiles; ++t) {
        // registers -> LDS (swizzled: float4 slot c4 of row rr lives at c4 ^ (rr & 15))
#pragma unroll
        for (int it = 0; it < LD_IT; ++it) {
            const int f = it * 256 + tid;
            const int rr = f >> 5, c4 = f & 31;
            tile[rr * 32 + (c4 ^ (rr & 15))] = pre[it];
        }
        __syncthreads();
        if (t + 1 < ntiles) issue_loads(t + 1);

        // ---- S[32 rows x 32 queries] = D_tile . Q_tile^T on the matrix cores ----
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        const f32x4 *arow = tile + (rw * 32 + r) * 32;
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) {
            const f32x4 a = arow[(16 * h + tt) ^ (r & 15)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qreg[4 * tt + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qreg[4 * tt + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qreg[4 * tt + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qreg[4 * tt + 3], acc, 0, 0, 0);
        }

        // ---- epilogue: scale / mask, filter against the k-th best, rare insert ----
        const int64_t sub_row0 = row_begin + (int64_t)t * TROWS + rw * 32;  // first row of this wave's sub-tile
        const bool sub_full = sub_row0 + 32 <= row_end;
        float sc[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float iv[4] = {1.0f, 1.0f, 1.0f, 1.0f};
            const int64_t rbase = sub_row0 + 8 * g + 4 * h;
            if (p.inv_norm != nullptr) {
                if (sub_full) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(p.inv_norm + rbase);
                    iv[0] = v.x; iv[1] = v.y; iv[2] = v.z; iv[3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) iv[j] = (rbase + j < row_end) ? p.inv_norm[rbase + j] : 0.0f;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) sc[4 * g + j] = (p.inv_norm != nullptr) ? acc[4 * g + j] * iv[j] : acc[4 * g + j];
            if (p.lengths != nullptr) {
                float ln[4];
                if (sub_full) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(p.lengths + rbase);
                    ln[0] = v.x; ln[1] = v.y; ln[2] = v.z; ln[3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) ln[j] = (rbase + j < row_end) ? p.lengths[rbase + j] : 0.0f;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float m = (my_qlen >= ln[j] * p.mincov) ? 1.0f : 0.0f;   // dbsearch.py:76
                    sc[4 * g + j] = sc[4 * g + j] * m;                              // dbsearch.py:78
                }
            }
        }

#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint64_t m[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float s = sc[4 * g + j];
                const uint32_t lrow = (uint32_t)(sub_row0 + 8 * g + 4 * h + j);
                bool pass = q_valid && (s > tau);
                if (!sub_full) pass = pass && (sub_row0 + 8 * g + 4 * h + j < row_end);
                if (has_ub) pass = pass && ((s < ubs) || (s == ubs && lrow > ubi));
                m[j] = __ballot(pass);
            }
            if ((m[0] | m[1] | m[2] | m[3]) == 0) continue;
            // ascending row order: rows 8g+4hh+j
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t mm = hh ? (uint32_t)(m[j] >> 32) : (uint32_t)m[j];
                    while (mm) {
                        const int b = __builtin_ctz(mm);
                        mm &= mm - 1;
                        const float cs = ms_readlane_f(sc[4 * g + j], b + 32 * hh);
                        const uint32_t crow = (uint32_t)(sub_row0 + 8 * g + 4 * hh + j);
                        const float nt = ms_wave_insert(my_lists + b * k, k, cs, crow, lane);
                        if (r == b) tau = nt;
                    }
                }
            }
        }
        __syncthreads();   // every wave is done with the tile before it is overwritten
    }

    // ---- combine the RW row-streams of each query tile inside the block ----
    if (RW > 1) {
        __syncthreads();
        // (qw, q) pairs: QW*32 lists to finish; wave w takes pairs w, w+4, ...
        for (int pair = wave; pair < QW * 32; pair += 4) {
            const int pqw = pair >> 5, pq = pair & 31;
            uint2 *dst = lists + ((size_t)(pqw * RW) * 32 + pq) * k;
            for (int srw = 1; srw < RW; ++srw) {
                const uint2 *src = lists + ((size_t)(pqw * RW + srw) * 32 + pq) * k;
                const uint2 e = (lane < k) ? src[lane] : make_uint2(0u, 0u);
                const uint2 last = dst[k - 1];
                const bool cand = lane < k && e.y != MS_IDX_NONE &&
                                  ms_better(__uint_as_float(e.x), e.y, __uint_as_float(last.x), last.y);
                const int c = __popcll(__ballot(cand));   // sorted source: survivors form a prefix
                for (int i = 0; i < c; ++i)
                    ms_wave_insert(dst, k, ms_readlane_f(__uint_as_float(e.x), i), ms_readlane_u(e.y, i), lane);
            }
        }
        __syncthreads();
    } else {
        __syncthreads();
    }

    // ---- write this block's lists: part[chunk][query][k] ----
    for (int e = tid; e < QW * 32 * k; e += 256) {
        const int lq = e / k, j = e % k;             // lq = qw*32 + q
        const int pqw = lq >> 5, pq = lq & 31;
        const uint2 v = lists[((size_t)(pqw * RW) * 32 + pq) * k + j];
        const size_t o = ((size_t)chunk * p.nq_pad + (size_t)qg * QW * 32 + lq) * k + j;
        p.part_s[o] = __uint_as_float(v.x);
        p.part_i[o] = v.y;
    }
}

// ------------------------------------------------------------------ partial merge ------
// One workgroup per query.  4 waves each reduce a quarter of the per-chunk lists into a sorted
// list in LDS, wave 0 merges the four, converts rows to int64 (+ row_offset) and records the
// exclusive upper bound of the next pass.
__global__ __launch_bounds__(256) void ms_partial_merge_kernel(const float *part_s, const uint32_t *part_i,
                                                              int n_chunks, int nq_pad, int k, int64_t row_offset,
                                                              float *out_s, int64_t *out_i, int out_stride,
                                                              int out_col0, float *ub_s, uint32_t *ub_i) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint2 *lists = reinterpret_cast<uint2 *>(smem);   // [4][k]
    const int q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint2 *mine = lists + wave * k;
    if (lane < k) mine[lane] = make_uint2(__float_as_uint(-INFINITY), MS_IDX_NONE);
    constexpr int UNROLL = 8;
    for (int c0 = wave * UNROLL; c0 < n_chunks; c0 += 4 * UNROLL) {
        float es[UNROLL];
        uint32_t ei[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int c = c0 + u;
            es[u] = -INFINITY;
            ei[u] = MS_IDX_NONE;
            if (c < n_chunks && lane < k) {
                const size_t o = ((size_t)c * nq_pad + q) * k + lane;
                es[u] = part_s[o];
                ei[u] = part_i[o];
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint2 last = mine[k - 1];
            const bool cand = ei[u] != MS_IDX_NONE && ms_better(es[u], ei[u], __uint_as_float(last.x), last.y);
            const int c = __popcll(__ballot(cand));
            for (int i = 0; i < c; ++i) ms_wave_insert(mine, k, ms_readlane_f(es[u], i), ms_readlane_u(ei[u], i), lane);
        }
    }
    __syncthreads();
    if (wave == 0) {
        for (int w = 1; w < 4; ++w) {
            const uint2 e = (lane < k) ? lists[w * k + lane] : make_uint2(0u, MS_IDX_NONE);
            const uint2 last = mine[k - 1];
            const bool cand = lane < k && e.y != MS_IDX_NONE &&
                              ms_better(__uint_as_float(e.x), e.y, __uint_as_float(last.x), last.y);
            const int c = __popcll(__ballot(cand));
            for (int i = 0; i < c; ++i)
                ms_wave_insert(mine, k, ms_readlane_f(__uint_as_float(e.x), i), ms_readlane_u(e.y, i), lane);
        }
        if (lane < k) {
            const uint2 e = mine[lane];
            const size_t o = (size_t)q * out_stride + out_col0 + lane;
            if (out_s != nullptr) {
                out_s[o] = __uint_as_float(e.x);
                out_i[o] = (e.y == MS_IDX_NONE) ? (int64_t)-1 : row_offset + (int64_t)e.y;
            }
            if (lane == k - 1 && ub_s != nullptr) {
                ub_s[q] = __uint_as_float(e.x);
                ub_i[q] = e.y;
            }
        }
    }
}

// ------------------------------------------------------------------ public k-way merge -
// One thread per query: classic k-way merge of S lists that are each sorted best-first.
__global__ __launch_bounds__(64) void ms_kway_merge_kernel(const float *scores, const int64_t *idx, int S, int nq,
                                                           int k, float *out_s, int64_t *out_i) {
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
            const size_t o = ((size_t)s * nq + q) * k + head[s];
            const int64_t ci = idx[o];
            if (ci < 0) { head[s] = k; continue; }   // padding: list exhausted
            const float cs = scores[o];
            if (best < 0 || cs > bs || (cs == bs && ci < bi)) { best = s; bs = cs; bi = ci; }
        }
        const size_t oo = (size_t)q * k + j;
        if (best < 0) { out_s[oo] = -INFINITY; out_i[oo] = -1; }
        else { out_s[oo] = bs; out_i[oo] = bi; ++head[best]; }
    }
}

// ------------------------------------------------------------------ host side ----------
namespace {

struct ScanPlan {
    int qw;            // query tiles per workgroup (1, 2 or 4)
    int n_qgroups;
    int nq_pad;
    int k_pass;        // list length per pass
    int rows_per_chunk;
    int n_chunks;
    int grid;
    size_t lds_bytes;
    // workspace carve (byte offsets)
    size_t off_qn, off_inv, off_part_s, off_part_i, off_ub_s, off_ub_i, total;
};

int cu_count_cached() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    return cus;
}

ScanPlan make_plan(int64_t n, int nq, int k, int cus) {
    ScanPlan pl;
    const int ntiles_q = (nq + 31) / 32;
    pl.qw = ntiles_q >= 3 ? 4 : (ntiles_q == 2 ? 2 : 1);
    pl.n_qgroups = (ntiles_q + pl.qw - 1) / pl.qw;
    pl.nq_pad = pl.n_qgroups * pl.qw * 32;
    pl.k_pass = k < 64 ? k : 64;
    const int trows = 32 * (4 / pl.qw);
    const int64_t tiles = (n + trows - 1) / trows;
    int64_t want = (2LL * cus + pl.n_qgroups - 1) / pl.n_qgroups;   // ~2 workgroups per CU in total
    if (want < 1) want = 1;
    if (want > tiles) want = tiles > 0 ? tiles : 1;
    const int64_t tiles_per_chunk = (tiles + want - 1) / want;
    pl.rows_per_chunk = (int)((tiles_per_chunk > 0 ? tiles_per_chunk : 1) * trows);
    pl.n_chunks = (int)((n + pl.rows_per_chunk - 1) / pl.rows_per_chunk);
    if (pl.n_chunks < 1) pl.n_chunks = 1;
    const int supers = (pl.n_chunks + 7) / 8;
    pl.grid = supers * 8 * pl.n_qgroups;
    pl.lds_bytes = (size_t)trows * 512 + (size_t)4 * 32 * pl.k_pass * sizeof(uint2);
    size_t off = 0;
    pl.off_qn = off;      off += ms_align_up((size_t)pl.nq_pad * MS_DIM * sizeof(float), 256);
    pl.off_inv = off;     off += ms_align_up((size_t)(n > 0 ? n : 1) * sizeof(float), 256);
    pl.off_part_s = off;  off += ms_align_up((size_t)pl.n_chunks * pl.nq_pad * pl.k_pass * sizeof(float), 256);
    pl.off_part_i = off;  off += ms_align_up((size_t)pl.n_chunks * pl.nq_pad * pl.k_pass * sizeof(uint32_t), 256);
    pl.off_ub_s = off;    off += ms_align_up((size_t)pl.nq_pad * sizeof(float), 256);
    pl.off_ub_i = off;    off += ms_align_up((size_t)pl.nq_pad * sizeof(uint32_t), 256);
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
    if (mode != MS_MODE_IP_PRENORM && mode != MS_MODE_COSINE_RAW) MS_FAIL(MS_ERR_ARG, "ms_ip_topk: unknown mode %d", mode);
    if (mode == MS_MODE_IP_PRENORM && (inv_norm || lengths || qlen))
        MS_FAIL(MS_ERR_ARG, "ms_ip_topk: inv_norm / lengths / qlen are only valid in MS_MODE_COSINE_RAW");
    if ((lengths == nullptr) != (qlen == nullptr)) MS_FAIL(MS_ERR_ARG, "ms_ip_topk: lengths and qlen go together");
    return MS_OK;
}

int launch_scan(const ScanPlan &pl, const ScanParams &sp, hipStream_t st) {
    switch (pl.qw) {
        case 4: {
            MS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ms_scan_kernel<4>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes));
            hipLaunchKernelGGL(ms_scan_kernel<4>, dim3(pl.grid), dim3(256), pl.lds_bytes, st, sp);
            break;
        }
        case 2: {
            MS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ms_scan_kernel<2>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes));
            hipLaunchKernelGGL(ms_scan_kernel<2>, dim3(pl.grid), dim3(256), pl.lds_bytes, st, sp);
            break;
        }
        default: {
            MS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ms_scan_kernel<1>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes));
            hipLaunchKernelGGL(ms_scan_kernel<1>, dim3(pl.grid), dim3(256), pl.lds_bytes, st, sp);
            break;
        }
    }
    MS_LAUNCH_CHECK("ms_scan_kernel");
    return MS_OK;
}

// Prepare queries (+ inverse norms if absent) and fill ScanParams for the first pass.
int prepare_scan(const ScanPlan &pl, const float *db, int64_t n, const float *q, int nq, int mode,
                 const float *inv_norm, const float *lengths, const float *qlen, float mincov, char *ws,
                 hipStream_t st, ScanParams *sp) {
    float *qn = reinterpret_cast<float *>(ws + pl.off_qn);
    hipLaunchKernelGGL(ms_prepare_queries_kernel, dim3((pl.nq_pad + 3) / 4), dim3(256), 0, st, q, nq, pl.nq_pad,
                       mode == MS_MODE_COSINE_RAW ? 1 : 0, 1e-8f, qn);
    MS_LAUNCH_CHECK("ms_prepare_queries_kernel");
    const float *inv = inv_norm;
    if (mode == MS_MODE_COSINE_RAW && inv == nullptr && n > 0) {
        float *inv_ws = reinterpret_cast<float *>(ws + pl.off_inv);
        const int64_t blocks = (n + 3) / 4;
        hipLaunchKernelGGL(ms_row_inv_norms_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, st,
                           db, n, 1e-8f, inv_ws);
        MS_LAUNCH_CHECK("ms_row_inv_norms_kernel");
        inv = inv_ws;
    }
    sp->db = db; sp->n = n; sp->qn = qn; sp->nq = nq; sp->nq_pad = pl.nq_pad; sp->k = pl.k_pass; sp->mode = mode;
    sp->inv_norm = inv; sp->lengths = lengths; sp->qlen = qlen; sp->mincov = mincov;
    sp->ub_s = nullptr; sp->ub_i = nullptr;
    sp->part_s = reinterpret_cast<float *>(ws + pl.off_part_s);
    sp->part_i = reinterpret_cast<uint32_t *>(ws + pl.off_part_i);
    sp->rows_per_chunk = pl.rows_per_chunk; sp->n_chunks = pl.n_chunks; sp->n_qgroups = pl.n_qgroups;
    return MS_OK;
}

}  // namespace

extern "C" {

int ms_version(void) { return 100; }
const char *ms_last_error(void) { return ms_err_buf; }

int ms_device_count(void) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

int ms_device_cu_count(void) { return cu_count_cached(); }

int ms_l2_normalize_rows(float *x, int64_t n, int d, float eps, ms_stream_t stream) {
    if (d != MS_DIM) MS_FAIL(MS_ERR_ARG, "ms_l2_normalize_rows: d must be %d (got %d)", MS_DIM, d);
    if (n < 0 || (n > 0 && x == nullptr)) MS_FAIL(MS_ERR_ARG, "ms_l2_normalize_rows: bad arguments");
    if (n == 0) return MS_OK;
    const int64_t blocks = (n + 3) / 4;
    hipLaunchKernelGGL(ms_normalize_rows_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0,
                       (hipStream_t)stream, x, n, eps);
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

int ms_ip_topk_scan(const float *db, int64_t n, const float *q, int nq, int k, int mode, const float *inv_norm,
                    const float *lengths, const float *qlen, float mincov, void *workspace, size_t workspace_bytes,
                    ms_stream_t stream) {
    int rc = check_search_args(db, n, q, nq, k, mode, inv_norm, lengths, qlen);
    if (rc) return rc;
    if (k > 64) MS_FAIL(MS_ERR_ARG, "ms_ip_topk_scan: k <= 64 only (use ms_ip_topk)");
    const ScanPlan pl = make_plan(n, nq, k, cu_count_cached());
    if (workspace == nullptr || workspace_bytes < pl.total)
        MS_FAIL(MS_ERR_WORKSPACE, "ms_ip_topk_scan: workspace %zu < %zu bytes", workspace_bytes, pl.total);
    ScanParams sp;
    rc = prepare_scan(pl, db, n, q, nq, mode, inv_norm, lengths, qlen, mincov, (char *)workspace, (hipStream_t)stream, &sp);
    if (rc) return rc;
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
    hipLaunchKernelGGL(ms_partial_merge_kernel, dim3(nq), dim3(256), 4 * pl.k_pass * sizeof(uint2), (hipStream_t)stream,
                       reinterpret_cast<const float *>(ws + pl.off_part_s),
                       reinterpret_cast<const uint32_t *>(ws + pl.off_part_i), pl.n_chunks, pl.nq_pad, pl.k_pass,
                       row_offset, out_scores, out_idx, k, 0, (float *)nullptr, (uint32_t *)nullptr);
    MS_LAUNCH_CHECK("ms_partial_merge_kernel");
    return MS_OK;
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
    // ceil(k / 64) passes; pass p returns ranks [64p, 64p + kp) using the last entry of pass
    // p-1 as an exclusive upper bound in the total order.
    for (int col0 = 0; col0 < k; col0 += 64) {
        const int kp = (k - col0) < 64 ? (k - col0) : 64;
        sp.k = kp;
        // the list stride inside the workspace is the pass's own k
        rc = launch_scan(pl, sp, st);
        if (rc) return rc;
        const bool more = col0 + 64 < k;
        hipLaunchKernelGGL(ms_partial_merge_kernel, dim3(nq), dim3(256), 4 * kp * sizeof(uint2), st, sp.part_s,
                           sp.part_i, pl.n_chunks, pl.nq_pad, kp, row_offset, out_scores, out_idx, k, col0,
                           more ? ub_s : (float *)nullptr, more ? ub_i : (uint32_t *)nullptr);
        MS_LAUNCH_CHECK("ms_partial_merge_kernel");
        sp.ub_s = ub_s;
        sp.ub_i = ub_i;
    }
    return MS_OK;
}

int ms_topk_merge(const float *scores, const int64_t *idx, int S, int nq, int k, float *out_scores,
                  int64_t *out_idx, ms_stream_t stream) {
    if (S < 1 || S > 64 || nq < 1 || k < 1 || !scores || !idx || !out_scores || !out_idx)
        MS_FAIL(MS_ERR_ARG, "ms_topk_merge: need 1 <= S <= 64, nq >= 1, k >= 1 and non-NULL buffers (S=%d)", S);
    hipLaunchKernelGGL(ms_kway_merge_kernel, dim3((nq + 63) / 64), dim3(64), 0, (hipStream_t)stream, scores, idx, S, nq,
                       k, out_scores, out_idx);
    MS_LAUNCH_CHECK("ms_kway_merge_kernel");
    return MS_OK;
}

}  // extern "C"
