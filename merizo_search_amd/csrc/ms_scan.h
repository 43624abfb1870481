// Scan kernels of the search path and their launch templates, shared by the translation units that
// instantiate them (one per list length, so that they compile in parallel): ms_scan_kl5.hip, ms_scan_kl16.hip,
// ms_scan_kl10.hip, ms_scan_kl32.hip, ms_scan_kl32ub.hip.  ms_search.hip holds everything else.
#pragma once
#include "ms_common.h"

#include <math.h>
#include <stdlib.h>
#include <type_traits>

// ------------------------------------------------------------------ scan kernel --------
// The decomposition of one scan launch: query tiles x row streams.  Computed on the host for an ordinary search (make_plan) and ON
// THE DEVICE for the exact pass behind a prefiltered search, whose batch -- the queries whose proof failed -- is only known there
// (the last workgroup of ms_rescore_kernel writes a ScanDevPlan, the gated scan and merge read it).
struct ScanDevPlan {
    int nq, nq_pad, n_qtiles, qwb, n_qgroups, n_sgroups, n_streams, rows_per_stream, P, grid;
    int pad_[6];
};
__host__ __device__ inline void ms_plan_core(int64_t n, int nq, int cus, ScanDevPlan *d) {
    d->nq = nq;
    d->n_qtiles = (nq + 31) / 32;
    d->qwb = d->n_qtiles >= 3 ? 4 : (d->n_qtiles == 2 ? 2 : 1);
    d->n_qgroups = (d->n_qtiles + d->qwb - 1) / d->qwb;
    d->nq_pad = d->n_qgroups * d->qwb * 32;
    const int64_t tiles = (n + 31) / 32;
    // one wave per (query tile, stream): one wave on each of the 4 * cus SIMDs
    int64_t want = ((int64_t)4 * cus) / ((int64_t)d->n_qgroups * d->qwb);
    if (want < 1) want = 1;
    if (want > tiles) want = tiles > 0 ? tiles : 1;
    const int64_t tiles_per_stream = (tiles + want - 1) / want;
    d->rows_per_stream = (int)((tiles_per_stream > 0 ? tiles_per_stream : 1) * 32);
    d->n_streams = (int)((n + d->rows_per_stream - 1) / d->rows_per_stream);
    if (d->n_streams < 1) d->n_streams = 1;
    const int spb = 4 / d->qwb;
    d->n_sgroups = (d->n_streams + spb - 1) / spb;
    d->P = d->qwb == 4 ? d->n_streams : d->n_sgroups;
    d->grid = ((d->n_sgroups + 7) / 8) * 8 * d->n_qgroups;
}

struct ScanParams {
    const float *db;        // [n,128]
    int64_t n;
    const float *qn;        // prepared queries [nq_pad,128], or the caller's [nq,128] array (inner-product mode, 16-byte aligned)
    int nq;                 // real queries
    int nq_pad;
    int k;                  // ranks wanted this pass (<= 2*KL <= 64)
    const float *inv_norm;  // [n] or NULL
    const float *lengths;   // [n] or NULL
    const float *qlen;      // [nq] or NULL
    float mincov;
    float qnorm_eps;        // > 0: qn is the caller's RAW query array; every wave L2-normalises its query tile itself (x / max(|x|, eps),
                            //      the arithmetic of ms_normalize_rows_kernel) -- the kernels for 1-2 query tiles only
    float qraw_eps = 0.0f;  // > 0 (fp16-image scans of the prefilter only, round 6): qn is the caller's RAW query array and every wave normalises its
                            //      query tile in its set-up, APPROXIMATELY (q * (1 / max(|q|, eps)): a few ulp from F.normalize, far inside the scan's
                            //      error bound; the sample pass and the scan run the same sequence) -- the exact normalisation the answer needs is
                            //      done by the re-scoring launch, one query per workgroup (ms_rescore_body): no query-preparation launch at all
    int debug_flags = 0;    // diagnostics (MS_PF_DEBUG, fp16-image scan): 1 = every threshold +inf (the rare path is compiled in and never taken: WRONG results)
    int unit_rows;          // MS_MODE_COSINE_UNIT: the rows are L2-normalised already (no inv_norm array); lengths / qlen mask as usual
    const float *ub_s;      // [nq_pad] exclusive upper bound of this pass (total order), or NULL
    const uint32_t *ub_i;
    const float *lb_s;      // [nq_pad] inclusive lower bound on the k-th best score (from the sample pass), or NULL
    uint32_t *hist;         // [nq_pad][16] candidates counted per score bucket during the full pass (loader-wave form), or NULL
    const float *hstep;     // [nq_pad] bucket width of a query (bucket j starts at lb + j * step); 0 = no histogram for it
    int max_tiles;          // > 0: sample pass, every stream stops after this many tiles
    // A handful of queries (1-2 query tiles, ms_ip_topk only): the LAST workgroup of a query group to finish merges the lists
    // inside the scan launch -- final results straight into fin_s / fin_i, no merge launch (NULL: off)
    float *fin_s;
    int64_t *fin_i;
    int64_t fin_row_offset;
    int fin_stride;
    const int *fin_qmap = nullptr;   // ... of query q of the batch into output row fin_qmap[q] (the exact pass behind a prefiltered search of a handful of queries)
    int prefilter = 0;      // loader-wave form only: score with three bf16 matrix instructions on the split operands (approximate scores;
                            // the caller re-scores the survivors exactly: ms_ip_topk_prefiltered)
    const ScanDevPlan *dev_plan = nullptr;   // ms_scan_kernel only: nq, the streams and the grid come from device memory (exact pass
                                             // over the flagged queries of a prefiltered search); workgroups past its grid return at once
    const void *pf_image = nullptr;   // prefilter, ms_scan_pf.h / ms_scan_pf16.h: the image of db (ms_pf_build_image), or NULL (split in registers)
    int pf_format = 0;                // ... and its arithmetic: MS_PF_BF16X3 (32-row tiles), MS_PF_F16X2 / MS_PF_F16X1 (64-row tiles)
    int list_sm = 0;                  // the launch writes its per-stream lists STREAM-MAJOR ([stream][query][rank]: a workgroup's lists are one contiguous block;
                                      // the image scans always, the loader-wave kernel when its merge can read them: ScanPlan::list_sm) instead of rank-major
    uint32_t *prog = nullptr;         // fp16-image scan with 2..16 query groups per row stream: [n_streams][16] progress words (8-bit epoch << 24 | 24-bit tile),
    uint32_t prog_epoch = 0;          // by which the workgroups of a stream keep within one L2 window of each other (ms_scan_pf16.h); NULL: off
    int qpw = 1;                      // ... and the waves per workgroup of that kernel, one query tile each, in fours (1: 4 waves, 2: 8)
    const uint32_t *gate = nullptr;   // NULL, or: the launch does nothing unless *gate == gate_epoch (the exact pipeline behind a
    uint32_t gate_epoch = 0;          // prefiltered search runs only when the prefilter could not prove its answer)
    uint32_t *ticket;       // [n_qgroups] arrival counters in library-owned memory, zero between launches (the last arriver resets its counter)
    float *part_s;          // [nq_pad][k][P]  rank-major per query, P partial lists
    uint32_t *part_i;
    int rows_per_stream;    // multiple of 32
    int n_streams;          // row streams (one wave each per query tile)
    int n_qtiles;           // 32-query tiles
    int qwb;                // query tiles per workgroup: 4, 2 or 1 (the other 4/qwb waves take other streams)
    int n_qgroups;
    int n_sgroups;          // stream groups = workgroups per query group
    int P;                  // partial lists per query written by this launch
#ifdef MS_STAMP
    unsigned long long *stamps;   // diagnostic builds only: per compute wave {cycles, 100 MHz ticks, tiles, 0}
#endif
};

#ifdef MS_STAMP
// diagnostic builds: phase stamps of ms_scan_body (100 MHz ticks, absolute), 8 words per (workgroup, wave); the sample pass
// writes into the second half of the buffer
#define MS_BODY_STAMP(slot)                                                                                             \
    do {                                                                                                               \
        if (lane == 0 && p.stamps != nullptr && bid < 4096)                                                            \
            (p.stamps + (MAXONLY ? 8 * 8 * 4096 : 0) + ((size_t)bid * 8 + wave) * 8)[slot] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define MS_BODY_STAMP(slot) do { } while (0)
#endif

// value of `x` in the partner lane (lane ^ 32): one v_permlane32_swap + one select, no LDS
__device__ __forceinline__ uint32_t ms_xor32_u(uint32_t x, int h) {
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return h ? r[0] : r[1];
}
__device__ __forceinline__ float ms_xor32_f(float x, int h) { return __uint_as_float(ms_xor32_u(__float_as_uint(x), h)); }

// Head-advance merge of the P partial lists (rank-major [k][P], each sorted best first) of query q by ONE wave: the [k][P]
// block is staged in `ent` (LDS, k * P entries), then k rounds of "best list head wins and its list advances" -- a wave-level
// arg-max per round, no block barrier, no pool.  Used by ms_head_merge_kernel (one wave per query) and, for a handful of
// queries, by the last workgroup of the scan launch itself (ms_scan_body).
template <int PER, bool STAGED = false>      // lists per lane: P <= 64 * PER; STAGED: `ent` is filled already
__device__ __forceinline__ void ms_head_merge_wave(uint2 *ent, const float *part_s, const uint32_t *part_i, int P, int k,
                                                   int64_t row_offset, float *out_s, int64_t *out_i, int out_stride, int out_col0,
                                                   float *ub_s, uint32_t *ub_i, int q, int lane) {
    const float *ps = part_s + (size_t)q * k * P;
    const uint32_t *pi = part_i + (size_t)q * k * P;
    if (!STAGED) {
#pragma unroll 8
        for (int e = lane; e < k * P; e += 64) ent[e] = make_uint2(__float_as_uint(ps[e]), pi[e]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (one wave: its own LDS writes are in order; no barrier)
    }
    float hs[PER];
    uint32_t hi[PER];
    int dep[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int pp = lane + 64 * u;
        hs[u] = -INFINITY; hi[u] = MS_IDX_NONE; dep[u] = 0;
        if (pp < P) { const uint2 e = ent[pp]; hs[u] = __uint_as_float(e.x); hi[u] = e.y; }
    }
    const size_t o0 = (size_t)q * out_stride + out_col0;
    for (int round = 0; round < k; ++round) {
        float bs = hs[0];
        uint32_t bi = hi[0];
        int bu = 0;
#pragma unroll
        for (int u = 1; u < PER; ++u)
            if (ms_better(hs[u], hi[u], bs, bi)) { bs = hs[u]; bi = hi[u]; bu = u; }
        // wave arg-max: 4 DPP steps inside each row of 16 lanes (xor 1, xor 2, half-row mirror, row
        // mirror: max is idempotent, so mirrors all-reduce as well as a butterfly), then the 4 row
        // results through SGPRs -- no LDS-crossbar shuffles in the round
        float ws = bs;
        uint32_t wi = bi;
#define MS_DPP_STEP(CTRL)                                                                                         \
        {                                                                                                         \
            const float os = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(ws), CTRL, 0xF, 0xF, false)); \
            const uint32_t oi = __builtin_amdgcn_update_dpp(0u, wi, CTRL, 0xF, 0xF, false);                       \
            if (ms_better(os, oi, ws, wi)) { ws = os; wi = oi; }                                                  \
        }
        MS_DPP_STEP(0xB1)      // quad_perm [1,0,3,2]
        MS_DPP_STEP(0x4E)      // quad_perm [2,3,0,1]
        MS_DPP_STEP(0x141)     // row_half_mirror
        MS_DPP_STEP(0x140)     // row_mirror
#undef MS_DPP_STEP
        {
            float rs = ms_readlane_f(ws, 0);
            uint32_t ri = ms_readlane_u(wi, 0);
#pragma unroll
            for (int row = 1; row < 4; ++row) {
                const float os = ms_readlane_f(ws, 16 * row);
                const uint32_t oi = ms_readlane_u(wi, 16 * row);
                if (ms_better(os, oi, rs, ri)) { rs = os; ri = oi; }
            }
            ws = rs; wi = ri;
        }
        if (wi == MS_IDX_NONE) {                                // every list is exhausted: pad the tail
            if (lane == 0) {
                for (int r2 = round; r2 < k; ++r2) { out_s[o0 + r2] = -INFINITY; out_i[o0 + r2] = -1; }
                if (ub_s != nullptr) { ub_s[q] = -INFINITY; ub_i[q] = MS_IDX_NONE; }
            }
            break;
        }
        if (lane == 0) {
            out_s[o0 + round] = ws;
            out_i[o0 + round] = row_offset + (int64_t)wi;
            if (round == k - 1 && ub_s != nullptr) { ub_s[q] = ws; ub_i[q] = wi; }
        }
        if (bi == wi) {                                         // rows are unique across lists: exactly one lane
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                if (u != bu) continue;
                const int d = ++dep[u];
                hs[u] = -INFINITY; hi[u] = MS_IDX_NONE;
                if (d < k) { const uint2 e = ent[(size_t)d * P + lane + 64 * u]; hs[u] = __uint_as_float(e.x); hi[u] = e.y; }
            }
        }
    }
}

__device__ __forceinline__ float ms_wave_sum_xor(float v) {       // the butterfly of ms_normalize_rows_kernel (same order, same bits)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// (MS_BODY_DMA_AUX: the rows of a few-query search are read ONCE, by one workgroup -- the non-temporal policy, `nt`, of the guide's
//  weight streams: 6.4 -> 6.5-6.8 TB/s there)
#ifndef MS_BODY_DMA_AUX
#define MS_BODY_DMA_AUX 2
#endif
// One wave = one (query tile, row stream) pair, one wave per SIMD; waves never synchronise
// with each other inside the scan.  The loop over 32-row tiles is software-pipelined around
// the dependent chain of 64 v_mfma_f32_32x32x2_f32 of tile t (4096 cycles of matrix pipe):
//   before the chain   s_waitcnt vmcnt(0): tile t+1 has landed in LDS (issued a tile ago);
//                      LDS-DMA of tile t+2 (global_load_lds_dwordx4, 16 x 1 KiB) into the slot
//                      tile t just vacated -- lane-linear destination, XOR-swizzled SOURCE;
//   in the MFMA gaps   16 ds_read_b128 pull tile t+1 into 64 VGPRs (A-fragment order), and the
//                      filter of tile t-1 runs: one compare per score against the query's k-th
//                      best so far;
//   after the chain    the rare insertion steps for tile t-1.
// The running top-k of query q lives in the REGISTERS of its two lanes (q, q+32): lane q holds
// ranks 0..KL-1, lane q+32 ranks KL..2KL-1, sorted.  An insertion step handles one database
// row for all 32 queries at once (SIMD over queries): the candidate goes to lane q, lane q's
// displaced last entry (one compare tells which) goes to lane q+32, both lanes update their
// sorted half with one compare and four selects per entry.  No atomics, no cross-wave traffic.
// (ms_scan_loader_kernel below is the form used for >= 3 query tiles; this one serves 1-2 query
// tiles -- the HBM-bound regime -- and, with MAXONLY, the sample pass.)
template <int KL>
struct ScanState {
    float ls[KL];
    uint32_t li[KL];
    float tau;     // scores must be > tau to matter: max(k-th best so far, floor)
    float floor;   // largest float below the sample pass's lower bound (-inf without one)
};

// Shared lower bound of the full pass (loader-wave form).  A stream's own list hardly ever tightens the sample's bound
// (a query has ~2 rows above it per stream), so every wave would keep visiting the insertion path for rows that cannot
// reach the top k of the WHOLE shard: ~270 per query at C2, where ~40 would do with a bound that follows the scan.
// Exchange, cheap on both sides: every candidate a wave accepts is counted in one of 16 score buckets of its query
// (one no-return agent-scope atomic add; bucket j = scores >= lb + j * step, the last one open-ended), and every 16
// tiles a wave reads its queries' 16 counters (sc1 loads: counters only grow, a stale value is just a weaker bound)
// and raises its threshold to the highest bucket edge that already has k rows at or above it.  Rows are visited once
// per query, so the counts are of distinct rows and the edge is a valid inclusive lower bound on the k-th best.
struct ScanHist {
    uint32_t *counters;    // this lane's query: 16 counters (NULL: off)
    float base, step, inv_step;
};
__device__ __forceinline__ float ms_next_below(float x) {        // largest float below a finite x
    uint32_t u = __float_as_uint(x);
    u = (x > 0.0f) ? u - 1u : ((x < 0.0f) ? u + 1u : 0x80000001u);
    return __uint_as_float(u);
}
__device__ __forceinline__ float ms_hist_edge(const ScanHist &hg, int j) { return fmaf((float)j, hg.step, hg.base); }
__device__ __forceinline__ void ms_hist_count(const ScanHist &hg, float v) {        // v >= base (it passed the filter)
    int j = (int)((v - hg.base) * hg.inv_step);
    j = j > 15 ? 15 : (j < 0 ? 0 : j);
    if (v < ms_hist_edge(hg, j)) j -= 1;       // the bucket is decided by the same expression the readers evaluate
    j = j < 0 ? 0 : j;
    __hip_atomic_fetch_add(hg.counters + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// insertion steps for one 32-row tile whose (scaled, masked) scores are sc[16]
// one insertion step: the candidate of row `crow` (score v in lanes of half hh whose bit is set
// in mm) goes into the lists of all 32 queries at once
template <int KL, bool HIST = false>
__device__ __forceinline__ void ms_row_insert(ScanState<KL> &st, float v, uint32_t mm, int hh, uint32_t crow, int r, int h,
                                              const ScanHist *hg = nullptr) {
    // candidate of this lane pair (or -inf); re-checked against the current tau
    const bool mine = (h == hh) && ((mm >> r) & 1u) && (v > st.tau);
    if (HIST) {
        if (mine && hg->counters != nullptr) ms_hist_count(*hg, v);
    }
    const float c = mine ? v : -INFINITY;
    const float pc = ms_xor32_f(c, h);
    const float cand = (h == hh) ? c : pc;
    // lane q+32 receives lane q's last entry if the candidate displaces it
    const float pl_s = ms_xor32_f(st.ls[KL - 1], h);
    const uint32_t pl_i = ms_xor32_u(st.li[KL - 1], h);
    const bool spill = (h == 1) && (cand > pl_s);
    // Sorted insert without a serial compare-exchange chain: with ge[e] = (old ls[e] >= x),
    //     new ls[e] = ge[e] ? ls[e] : (ge[e-1] ? x : ls[e-1])        (ge[-1] = true)
    // so every entry needs one compare and two selects per array, all on OLD values (updated in
    // place from the last entry down).  New rows lose ties (>=: ascending row order).  A spilled
    // entry is not below anything in lane q+32's half (the pair's list is sorted), so it goes to
    // position 0: its compare value is NaN (ge false everywhere).  No candidate: x = -inf, ge true.
    const float x_cmp = spill ? __builtin_nanf("") : cand;
    const float ins_s = spill ? pl_s : cand;
    const uint32_t ins_i = spill ? pl_i : crow;
    bool ge_hi = st.ls[KL - 1] >= x_cmp;
#pragma unroll
    for (int e = KL - 1; e >= 1; --e) {
        const bool ge_lo = st.ls[e - 1] >= x_cmp;
        const float ns = ge_lo ? ins_s : st.ls[e - 1];
        const uint32_t ni = ge_lo ? ins_i : st.li[e - 1];
        st.ls[e] = ge_hi ? st.ls[e] : ns;
        st.li[e] = ge_hi ? st.li[e] : ni;
        ge_hi = ge_lo;
    }
    st.ls[0] = ge_hi ? st.ls[0] : ins_s;
    st.li[0] = ge_hi ? st.li[0] : ins_i;
    const float worst = ms_xor32_f(st.ls[KL - 1], h);   // lane q+32's last = the pair's k-th best
    st.tau = fmaxf(h ? st.ls[KL - 1] : worst, st.floor);
}

// The same step with a candidate (v, row) of ITS OWN per lane (the split-image prefilter scan empties its per-lane candidate
// buffers with it): lanes of half hh contribute, every query at most one candidate per call.  Candidates arrive in no particular
// row order, so entries that tie in score keep no particular order either -- fine for approximate scores, which the exact
// re-scoring re-ranks (ms_rescore_kernel), not for the fp32 scan.
template <int KL>
__device__ __forceinline__ void ms_lane_insert(ScanState<KL> &st, float v, uint32_t row, int hh, int h) {
    const bool mine = (h == hh) && (v > st.tau);
    const float c = mine ? v : -INFINITY;
    const float pc = ms_xor32_f(c, h);
    const uint32_t prow = ms_xor32_u(row, h);
    const float cand = (h == hh) ? c : pc;
    const uint32_t crow = (h == hh) ? row : prow;
    const float pl_s = ms_xor32_f(st.ls[KL - 1], h);
    const uint32_t pl_i = ms_xor32_u(st.li[KL - 1], h);
    const bool spill = (h == 1) && (cand > pl_s);
    const float x_cmp = spill ? __builtin_nanf("") : cand;
    const float ins_s = spill ? pl_s : cand;
    const uint32_t ins_i = spill ? pl_i : crow;
    bool ge_hi = st.ls[KL - 1] >= x_cmp;
#pragma unroll
    for (int e = KL - 1; e >= 1; --e) {
        const bool ge_lo = st.ls[e - 1] >= x_cmp;
        const float ns = ge_lo ? ins_s : st.ls[e - 1];
        const uint32_t ni = ge_lo ? ins_i : st.li[e - 1];
        st.ls[e] = ge_hi ? st.ls[e] : ns;
        st.li[e] = ge_hi ? st.li[e] : ni;
        ge_hi = ge_lo;
    }
    st.ls[0] = ge_hi ? st.ls[0] : ins_s;
    st.li[0] = ge_hi ? st.li[0] : ins_i;
    const float worst = ms_xor32_f(st.ls[KL - 1], h);   // lane q+32's last = the pair's k-th best
    st.tau = fmaxf(h ? st.ls[KL - 1] : worst, st.floor);
}

// The same step when BOTH lanes of a pair already hold the pair's next candidate (cand, crow) -- the append-and-flush rare path of
// the fp32 scan (ms_scan_loader_kernel, lists of 16 / 32 entries per lane) feeds a query's buffered candidates in ascending row order,
// so "new rows lose ties" holds exactly as in ms_row_insert and the lists come out the same, entry for entry.
typedef uint32_t ms_u32x2 __attribute__((ext_vector_type(2)));
template <int KL>
__device__ __forceinline__ void ms_pair_insert(ScanState<KL> &st, float cand, uint32_t crow, int h) {
    const float c = (cand > st.tau) ? cand : -INFINITY;
    const float pl_s = ms_xor32_f(st.ls[KL - 1], h);
    const uint32_t pl_i = ms_xor32_u(st.li[KL - 1], h);
    const bool spill = (h == 1) && (c > pl_s);
    const float x_cmp = spill ? __builtin_nanf("") : c;
    const float ins_s = spill ? pl_s : c;
    const uint32_t ins_i = spill ? pl_i : crow;
    bool ge_hi = st.ls[KL - 1] >= x_cmp;
#pragma unroll
    for (int e = KL - 1; e >= 1; --e) {
        const bool ge_lo = st.ls[e - 1] >= x_cmp;
        const float ns = ge_lo ? ins_s : st.ls[e - 1];
        const uint32_t ni = ge_lo ? ins_i : st.li[e - 1];
        st.ls[e] = ge_hi ? st.ls[e] : ns;
        st.li[e] = ge_hi ? st.li[e] : ni;
        ge_hi = ge_lo;
    }
    st.ls[0] = ge_hi ? st.ls[0] : ins_s;
    st.li[0] = ge_hi ? st.li[0] : ins_i;
    const float worst = ms_xor32_f(st.ls[KL - 1], h);   // lane q+32's last = the pair's k-th best
    st.tau = fmaxf(h ? st.ls[KL - 1] : worst, st.floor);
}

// insertion steps for one 32-row tile whose (scaled, masked) scores are sc[16]; rows in ascending
// order: row 8 g + 4 hh + j lives in lanes of half hh, register 4 g + j
#ifndef MS_STATIC_INSERT_MAX_KL
#define MS_STATIC_INSERT_MAX_KL 32
#endif
template <int KL, bool HIST = false, bool LOOP = false>      // LOOP: the runtime-loop form whatever the list length (small code)
__device__ __forceinline__ void ms_tile_insert(ScanState<KL> &st, const float (&sc)[16], const uint64_t (&m)[16],
                                               int64_t sub_row0, int r, int h, const ScanHist *hg = nullptr) {
    if (!LOOP && KL <= (HIST ? 16 : MS_STATIC_INSERT_MAX_KL)) {
        // short lists (k <= 10, the common case): one static copy of the step per row
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if ((m[4 * g] | m[4 * g + 1] | m[4 * g + 2] | m[4 * g + 3]) == 0) continue;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint64_t mj = m[4 * g + j];
                    const uint32_t mm = hh ? (uint32_t)(mj >> 32) : (uint32_t)mj;
                    if (mm == 0) continue;
                    ms_row_insert<KL, HIST>(st, sc[4 * g + j], mm, hh, (uint32_t)(sub_row0 + 8 * g + 4 * hh + j), r, h, hg);
                }
            }
        }
    } else {
        // long lists: a single copy of the (32-slot) step inside a runtime loop keeps the code size and the build time
        // down; the loop visits only the rows that have a candidate (the masks are wave-uniform: scalar work)
        uint32_t rows = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row0 = 8 * (i >> 2) + (i & 3);            // register i: rows row0 (lanes 0-31) and row0 + 4 (lanes 32-63)
            rows |= ((uint32_t)m[i] != 0u ? 1u : 0u) << row0;
            rows |= ((uint32_t)(m[i] >> 32) != 0u ? 1u : 0u) << (row0 + 4);
        }
        while (rows != 0u) {
            const int row = __builtin_ctz(rows);
            rows &= rows - 1u;
            const int reg = (row & 3) + 4 * (row >> 3), hh = (row >> 2) & 1;
            uint64_t mj = m[0];
            float v = sc[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) { mj = (reg == i) ? m[i] : mj; v = (reg == i) ? sc[i] : v; }
            const uint32_t mm = hh ? (uint32_t)(mj >> 32) : (uint32_t)mj;
            ms_row_insert<KL, HIST>(st, v, mm, hh, (uint32_t)(sub_row0 + row), r, h, hg);
        }
    }
}

template <int KL, bool AUX, bool UB, bool MAXONLY>
__device__ __forceinline__ void ms_scan_body(const ScanParams &p_in) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (ms_gate_closed(p_in.gate, p_in.gate_epoch)) return;          // (uniform: a scalar load)
    ScanParams p = p_in;
    if (p_in.dev_plan != nullptr) {                                     // (uniform: scalar loads)
        const ScanDevPlan d = *p_in.dev_plan;
        if ((int)blockIdx.x >= d.grid) return;
        p.nq = d.nq; p.nq_pad = d.nq_pad; p.n_qtiles = d.n_qtiles; p.qwb = d.qwb; p.n_qgroups = d.n_qgroups; p.n_sgroups = d.n_sgroups;
        p.n_streams = d.n_streams; p.rows_per_stream = d.rows_per_stream; p.P = d.P;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    // the wave index is uniform across the wave: say so, or every row / stream / loop quantity
    // below becomes 64-bit per-lane arithmetic
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // two private 32 x 32 float4 tile slots per wave
    f32x4 *slot0 = reinterpret_cast<f32x4 *>(smem + wave * 32768);
    // cosine mode: 1/|row| (32 floats) + target lengths (32 floats) of a tile, 4 slots (tile index & 3)
    float *aux0 = reinterpret_cast<float *>(smem + 4 * 32768 + wave * 1024);

    // block id -> (stream group, query group); groups sharing rows get ids 8 apart (same XCD, L2 reuse)
    const int bid = blockIdx.x;
    MS_BODY_STAMP(0);
    const int per_super = 8 * p.n_qgroups;
    const int super = bid / per_super, within = bid % per_super;
    const int sgroup = super * 8 + (within & 7);
    const int qg = within >> 3;
    if (sgroup >= p.n_sgroups) return;
    const int spb = 4 / p.qwb;
    const int qw = wave % p.qwb, sw = wave / p.qwb;
    const int stream = sgroup * spb + sw;
    const int qtile = qg * p.qwb + qw;
    const bool active = stream < p.n_streams && qtile < p.n_qtiles;
    ScanState<KL> st;
#pragma unroll
    for (int j = 0; j < KL; ++j) { st.ls[j] = -INFINITY; st.li[j] = MS_IDX_NONE; }
    st.floor = -INFINITY;
#ifdef MS_DEBUG_NO_INSERT
    st.tau = INFINITY;
#else
    st.tau = -INFINITY;
#endif

    // MAXONLY (sample pass): no lists, only this lane's best row so far (its half of every tile)
    float smax = -INFINITY;
    uint32_t srow = MS_IDX_NONE;

    if (active) {
        const int64_t row_begin = (int64_t)stream * p.rows_per_stream;
        int64_t row_end = (row_begin + p.rows_per_stream < p.n) ? row_begin + p.rows_per_stream : p.n;
        if (p.max_tiles > 0 && row_begin + (int64_t)p.max_tiles * 32 < row_end) row_end = row_begin + (int64_t)p.max_tiles * 32;
        const int qidx = qtile * 32 + r;
        const bool q_valid = qidx < p.nq;
        if (p.lb_s != nullptr) {
            // s >= lb  <=>  s > nextbelow(lb): at least k sampled rows score >= lb, so anything
            // below it cannot reach the top k; rows that tie with it still can
            const float lb = p.lb_s[qidx];
            st.floor = (lb == -INFINITY) ? -INFINITY : nextafterf(lb, -INFINITY);
#ifndef MS_DEBUG_NO_INSERT
            st.tau = st.floor;
#endif
        }
        // padding queries of the last tile never pass the filter: threshold +inf (one compare per score)
        if (!q_valid) { st.floor = INFINITY; st.tau = INFINITY; }

        // LDS-DMA of one tile into slot (t & 1).  Instruction `it` fills float4 slots 64 it .. 64 it + 63,
        // i.e. rows 2 it and 2 it + 1; slot (row, cs) must hold logical float4 column cs ^ (row & 15).
        // Per-lane byte offset inside the tile for instruction it:
        //     (2 it + h) * 512 + 16 * ((r ^ h) ^ (2 it & 15))  =  it * 1024 [scalar] + off8[it & 7]
        uint32_t off8[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) off8[c] = (uint32_t)(h * 512 + 16 * ((r ^ h) ^ (2 * c)));
        // cosine mode: one more LDS-DMA piece per tile (4 B per lane): lanes 0-31 fetch 1/|row| of the
        // tile's rows, lanes 32-63 their lengths (clamped to the last database row), into aux slot t & 3
        auto issue_aux_dma = [&](int t) {
            if (!AUX) return;
            int64_t row = row_begin + (int64_t)t * 32 + r;
            if (row >= p.n) row = p.n - 1;
            const float *base = ((h == 1 || p.inv_norm == nullptr) && p.lengths != nullptr) ? p.lengths : p.inv_norm;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + row),
                                             (__attribute__((address_space(3))) void *)(aux0 + (t & 3) * 64), 4, 0, 0);
        };
        auto issue_dma = [&](int t) {
            issue_aux_dma(t);
            const int64_t row0 = row_begin + (int64_t)t * 32;
            f32x4 *dst = slot0 + (t & 1) * 1024;
            const char *tile_src = reinterpret_cast<const char *>(p.db) + row0 * 512;
            if (row0 + 32 <= p.n) {
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const char *src = tile_src + it * 1024 + off8[it & 7];
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(dst + it * 64), 16, 0, MS_BODY_DMA_AUX);
                }
            } else {   // last tile of the database: clamp rows past the end (their scores are discarded)
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    int64_t row = row0 + 2 * it + h;
                    if (row >= p.n) row = p.n - 1;
                    const char *src = reinterpret_cast<const char *>(p.db) + row * 512 + 16 * ((r ^ h) ^ ((2 * it) & 15));
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(dst + it * 64), 16, 0, MS_BODY_DMA_AUX);
                }
            }
        };

        // tile 0 is requested before the queries are read (and, in MS_MODE_IP_NORMQ, normalised): the first row fetch and the
        // query preparation overlap
        const bool any_full_tile = ((row_end - row_begin) >> 5) > 0;
        if (any_full_tile) issue_dma(0);
        // B operand: lane (q = r, h) holds Q[q][64 h + s], s = 0..63, for the whole kernel
        float qreg[64];
        if (p.qnorm_eps > 0.0f) {
            // MS_MODE_IP_NORMQ with 1-2 query tiles: F.normalize of the raw queries (dbsearch.py:303-304) inside the scan launch.
            // The wave normalises the rows of its tile one by one exactly as ms_normalize_rows_kernel does (float2 per lane, the
            // same butterfly sum, sqrt, max, divide: the same bits) into its own second tile slot, which no DMA has touched yet,
            // and reads its B operand back from there.
            float *qn_lds = reinterpret_cast<float *>(slot0 + 1024);
            const int rows = (p.nq - qtile * 32) < 32 ? (p.nq - qtile * 32) : 32;
            for (int row0 = 0; row0 < 32; row0 += 16) {          // 16 rows' loads in flight (a row at a time is a chain of L2 round trips)
                float2 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    v[u] = make_float2(0.0f, 0.0f);
                    if (row0 + u < rows) v[u] = *(reinterpret_cast<const float2 *>(p.qn + (size_t)(qtile * 32 + row0 + u) * MS_DIM) + lane);
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    if (row0 + u < rows) {
                        const float ss = ms_wave_sum_xor(v[u].x * v[u].x + v[u].y * v[u].y);
                        const float nrm = fmaxf(sqrtf(ss), p.qnorm_eps);
                        v[u].x = v[u].x / nrm;
                        v[u].y = v[u].y / nrm;
                    }
                    *(reinterpret_cast<float2 *>(qn_lds + (row0 + u) * MS_DIM) + lane) = v[u];
                }
                if (row0 + 16 >= rows) {                          // the remaining rows are padding: zeros
                    for (int row = row0 + 16; row < 32; ++row) *(reinterpret_cast<float2 *>(qn_lds + row * MS_DIM) + lane) = make_float2(0.0f, 0.0f);
                    break;
                }
            }
            const f32x4 *src = reinterpret_cast<const f32x4 *>(qn_lds + r * MS_DIM + 64 * h);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const f32x4 v = src[t];
                qreg[4 * t + 0] = v.x; qreg[4 * t + 1] = v.y; qreg[4 * t + 2] = v.z; qreg[4 * t + 3] = v.w;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            // p.qn may be the caller's own [nq,128] array (inner-product mode): rows past nq read as zeros
            const f32x4 *src = reinterpret_cast<const f32x4 *>(p.qn + (size_t)(q_valid ? qidx : 0) * MS_DIM + 64 * h);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                f32x4 v = src[t];
                if (!q_valid) v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                qreg[4 * t + 0] = v.x; qreg[4 * t + 1] = v.y; qreg[4 * t + 2] = v.z; qreg[4 * t + 3] = v.w;
            }
        }
        MS_BODY_STAMP(1);
        // cosine mode, branch-free: without a lengths array the mask test is +inf >= x * 0
        float my_qlen = 0.0f;
        if (AUX) my_qlen = (p.qlen != nullptr && q_valid) ? p.qlen[qidx] : 0.0f;
        const float qlen_eff = (AUX && p.lengths == nullptr) ? INFINITY : my_qlen;
        const float mincov_eff = (AUX && p.lengths == nullptr) ? 0.0f : p.mincov;
        float ubs = INFINITY;
        uint32_t ubi = 0;
        if (UB) { ubs = p.ub_s[qidx]; ubi = p.ub_i[qidx]; }
        // Scores of registers 4g..4g+3 of a finished tile -> sc (cosine mode: * 1/|row|, * length
        // mask) and pass masks.  Branch-free so that it can sit between the MFMAs of the next
        // tile; CHECK_ROWS (row < row_end) is only needed for the last tile of a stream, which is
        // filtered in the drain.
        auto filter_group = [&](const f32x16 &acc, int64_t sub_row0, int g, bool check_rows, float (&sc)[16],
                                uint64_t (&m)[16]) {
            const int64_t rbase = sub_row0 + 8 * g + 4 * h;
            f32x4 inv4 = {1.0f, 1.0f, 1.0f, 1.0f}, len4 = {0.0f, 0.0f, 0.0f, 0.0f};
            if (AUX) {
                // the group's 4 row scales and 4 row lengths: two ds_read_b128.  Tile index from its first
                // row; before the first tile (sub_row0 < row_begin, scores are -inf) any slot will do
                const int tix = (int)((sub_row0 - row_begin) >> 5) & 3;
                const f32x4 *ax = reinterpret_cast<const f32x4 *>(aux0 + tix * 64 + 8 * g + 4 * h);
                inv4 = p.unit_rows ? f32x4{1.0f, 1.0f, 1.0f, 1.0f} : ax[0];
                len4 = ax[8];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s = acc[4 * g + j];
                if (AUX) {
                    float sv = s * inv4[j];                                                      // 1 / max(|row|, 1e-8)
                    const float mk = (qlen_eff >= len4[j] * mincov_eff) ? 1.0f : 0.0f;           // dbsearch.py:76
                    sv = sv * mk;                                                                // dbsearch.py:78
                    s = (sub_row0 >= row_begin) ? sv : -INFINITY;   // "tile -1" of the pipeline has no aux data
                }
                sc[4 * g + j] = s;
                if (MAXONLY) {      // strict >: the lowest row wins ties; -inf / NaN scores never enter
                    bool ok = s > smax;
                    if (check_rows) ok = ok && (rbase + j < row_end);
                    smax = ok ? s : smax;
                    srow = ok ? (uint32_t)(rbase + j) : srow;
                    m[4 * g + j] = 0;
                    continue;
                }
                bool pass = s > st.tau;
                if (check_rows) pass = pass && (rbase + j < row_end);
                if (UB) {
                    const uint32_t lrow = (uint32_t)(rbase + j);
                    pass = pass && ((s < ubs) || (s == ubs && lrow > ubi));
                }
                m[4 * g + j] = __ballot(pass);
            }
        };

        // In the pipeline (full tiles, no upper bound, not the sample pass) the filter is one compare per
        // TILE: final scores of group g -> sc and the lane's running maximum; the per-score ballots are
        // taken only when some lane's maximum passes its threshold.
        auto scale_max_group = [&](const f32x16 &acc, int64_t sub_row0, int g, float (&sc)[16], float &mx) {
            f32x4 inv4 = {1.0f, 1.0f, 1.0f, 1.0f}, len4 = {0.0f, 0.0f, 0.0f, 0.0f};
            if (AUX) {
                const int tix = (int)((sub_row0 - row_begin) >> 5) & 3;
                const f32x4 *ax = reinterpret_cast<const f32x4 *>(aux0 + tix * 64 + 8 * g + 4 * h);
                inv4 = p.unit_rows ? f32x4{1.0f, 1.0f, 1.0f, 1.0f} : ax[0];
                len4 = ax[8];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s = acc[4 * g + j];
                if (AUX) {
                    float sv = s * inv4[j];
                    const float mk = (qlen_eff >= len4[j] * mincov_eff) ? 1.0f : 0.0f;
                    sv = sv * mk;
                    s = (sub_row0 >= row_begin) ? sv : -INFINITY;
                }
                sc[4 * g + j] = s;
            }
            mx = fmaxf(mx, fmaxf(fmaxf(sc[4 * g], sc[4 * g + 1]), fmaxf(sc[4 * g + 2], sc[4 * g + 3])));
        };

        // one pipeline stage: MFMA chain of tile t from `areg`; each fragment register is
        // refilled with tile t+1 right after its 4 MFMAs were issued; the filter of tile t-1
        // (scores in `prev`) is spread over the first MFMA gaps; its insertion steps follow.
        f32x4 areg[16];
        const int nfull = (int)((row_end - row_begin) >> 5);       // full 32-row tiles: the pipelined loop
        const int rem = (int)((row_end - row_begin) & 31);          // partial last tile: handled after it
        auto stage = [&](int t, const f32x16 &prev, f32x16 &out) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // slot t&1 (tile t) fully read into areg
            const f32x4 *src = slot0 + ((t + 1) & 1) * 1024 + r * 32;
            const int64_t prev_row0 = row_begin + (int64_t)(t - 1) * 32;
            // LDS-DMA of tile t+2 (past the end: a harmless re-read of the last tile) into the slot tile t
            // vacated, one piece per MFMA group so that its issue time hides behind the matrix pipe
            const int tnext = (t + 2 < nfull) ? t + 2 : nfull - 1;
            const char *dma_src = reinterpret_cast<const char *>(p.db) + (row_begin + (int64_t)tnext * 32) * 512;
            f32x4 *dma_dst = slot0 + (t & 1) * 1024;
            constexpr bool FAST = !UB && !MAXONLY;
            float sc[16];
            uint64_t m[16];
            float mx = -INFINITY;
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
            for (int tt = 0; tt < 16; ++tt) {
                const f32x4 a = areg[tt];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qreg[4 * tt + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qreg[4 * tt + 1], acc, 0, 0, 0);
                // tile t+1 was issued during the previous chain: it only has to have landed by the middle
                // of this one.  vmcnt(8): everything but the 8 pieces of tile t+2 issued so far.
                if (tt == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(dma_src + tt * 1024 + off8[tt & 7]),
                                                 (__attribute__((address_space(3))) void *)(dma_dst + tt * 64), 16, 0, MS_BODY_DMA_AUX);
                if (tt == 15) issue_aux_dma(tnext);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qreg[4 * tt + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qreg[4 * tt + 3], acc, 0, 0, 0);
                if (tt >= 8) {   // refill the fragment registers of groups 2(tt-8), 2(tt-8)+1 (already consumed) with tile t+1
                    const int f0 = 2 * (tt - 8);
                    areg[f0] = src[(16 * h + f0) ^ (r & 15)];
                    areg[f0 + 1] = src[(16 * h + f0 + 1) ^ (r & 15)];
                }
                if (tt >= 2 && tt < 6) {
                    if (FAST) scale_max_group(prev, prev_row0, tt - 2, sc, mx);
                    else filter_group(prev, prev_row0, tt - 2, false, sc, m);
                }
            }
            out = acc;
            if (FAST) {
                if (__ballot(mx > st.tau) != 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) m[i] = __ballot(sc[i] > st.tau);
                    ms_tile_insert<KL>(st, sc, m, prev_row0, r, h);
                }
            } else {
                uint64_t any = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) any |= m[i];
                if (any != 0) ms_tile_insert<KL>(st, sc, m, prev_row0, r, h);
            }
        };

        f32x16 last;   // scores of the tile whose candidates are not inserted yet
#pragma unroll
        for (int i = 0; i < 16; ++i) last[i] = -INFINITY;
        int64_t last_row0 = row_begin;
        if (nfull > 0) {
            f32x16 acc0, acc1;
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc0[i] = -INFINITY; acc1[i] = -INFINITY; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (tile 0 was requested in front of the query load)
            MS_BODY_STAMP(2);
#pragma unroll
            for (int tt = 0; tt < 16; ++tt) areg[tt] = slot0[r * 32 + ((16 * h + tt) ^ (r & 15))];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue_dma(1);                   // into slot 1; tile 1 may be past the end: clamped reads, never used
            int t = 0;
            for (; t + 1 < nfull; t += 2) {
                stage(t, acc0, acc1);       // acc0 = scores of tile t-1 (or -inf), acc1 <- tile t
                stage(t + 1, acc1, acc0);   // acc1 = tile t, acc0 <- tile t+1
            }
            if (t < nfull) {                // odd tail
                stage(t, acc0, acc1);
                acc0 = acc1;
            }
            asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");   // stray prefetches done
            last = acc0;
            last_row0 = row_begin + (int64_t)(nfull - 1) * 32;
        }
        if (nfull > 0) {   // filter + insert the last full tile
            float sc[16];
            uint64_t m[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) filter_group(last, last_row0, g, true, sc, m);
            uint64_t any = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) any |= m[i];
            if (any != 0) ms_tile_insert<KL>(st, sc, m, last_row0, r, h);
        }
        if (rem > 0) {                      // partial last tile of the stream, not pipelined
            issue_dma(nfull);               // -> slot nfull & 1; rows past the database end are clamped
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const f32x4 *src = slot0 + (nfull & 1) * 1024 + r * 32;
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
            for (int tt = 0; tt < 16; ++tt) {
                const f32x4 a = src[(16 * h + tt) ^ (r & 15)];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qreg[4 * tt + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qreg[4 * tt + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qreg[4 * tt + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qreg[4 * tt + 3], acc, 0, 0, 0);
            }
            float sc[16];
            uint64_t m[16];
            const int64_t tail_row0 = row_begin + (int64_t)nfull * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g) filter_group(acc, tail_row0, g, true, sc, m);
            uint64_t any = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) any |= m[i];
            if (any != 0) ms_tile_insert<KL>(st, sc, m, tail_row0, r, h);
        }
    }
    MS_BODY_STAMP(3);
    if (MAXONLY && active) {
        // the stream's list = the two half-tile maxima of the lane pair (distinct rows), best first,
        // in lane q; lane q+32 stays empty
        const float ps2 = ms_xor32_f(smax, h);
        const uint32_t pr2 = ms_xor32_u(srow, h);
        if (h == 0) {
            const bool mine_first = ms_better(smax, srow, ps2, pr2);
            st.ls[0] = mine_first ? smax : ps2; st.li[0] = mine_first ? srow : pr2;
            st.ls[1] = mine_first ? ps2 : smax; st.li[1] = mine_first ? pr2 : srow;
            if (st.li[0] == MS_IDX_NONE) st.ls[0] = -INFINITY;
            if (st.li[1] == MS_IDX_NONE) st.ls[1] = -INFINITY;
        }
    }
    const int KLc = KL;
    float (&ls)[KL] = st.ls;
    uint32_t (&li)[KL] = st.li;
    (void)KLc;

    // ---- write the lists.  qwb == 4: one partial list per (stream, query).  qwb < 4: the
    //      4/qwb streams of a query tile inside this workgroup are merged through LDS first. ----
    constexpr int K2 = 2 * KL;
    if (p.qwb == 4) {
        if (!active) return;
        const int qidx = qtile * 32 + r;
#pragma unroll
        for (int j = 0; j < KL; ++j) {
            const int rank = h * KL + j;
            if (rank < p.k) {
                const size_t o = ((size_t)qidx * p.k + rank) * p.P + stream;
                p.part_s[o] = ls[j];
                p.part_i[o] = li[j];
            }
        }
        return;
    }
    __syncthreads();                                   // every wave is done with its tile slot
    if (MAXONLY) {
        // Sample pass: a stream's list is two maxima.  The 4/qwb streams of a query tile leave them in LDS, one thread per query
        // sorts the 2 * spb <= 8 entries (rows are distinct) and writes all p.k ranks (the general merge below took 5.4 us of an
        // 18.6 us launch: ~20 LDS round trips behind four barriers).
        uint2 *two = reinterpret_cast<uint2 *>(smem);      // [qw][32 queries][spb][2]
        if (h == 0) {
            uint2 *mine = two + (((size_t)qw * 32 + r) * spb + sw) * 2;
            const bool act = active;
            mine[0] = act ? make_uint2(__float_as_uint(ls[0]), li[0]) : make_uint2(__float_as_uint(-INFINITY), MS_IDX_NONE);
            mine[1] = act ? make_uint2(__float_as_uint(ls[1]), li[1]) : make_uint2(__float_as_uint(-INFINITY), MS_IDX_NONE);
        }
        __syncthreads();
        if (tid < p.qwb * 32) {
            const int pqw = tid >> 5, pq = tid & 31;
            const int qt = qg * p.qwb + pqw;
            if (qt < p.n_qtiles) {
                uint2 e[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) e[u] = (u < 2 * spb) ? two[((size_t)pqw * 32 + pq) * spb * 2 + u] : make_uint2(__float_as_uint(-INFINITY), MS_IDX_NONE);
                // insertion sort of 8 under the total order (static indices: registers)
#pragma unroll
                for (int a = 1; a < 8; ++a) {
#pragma unroll
                    for (int b2 = a; b2 >= 1; --b2) {
                        const bool sw_ = ms_better(__uint_as_float(e[b2].x), e[b2].y, __uint_as_float(e[b2 - 1].x), e[b2 - 1].y);
                        const uint2 hi_ = sw_ ? e[b2] : e[b2 - 1], lo_ = sw_ ? e[b2 - 1] : e[b2];
                        e[b2 - 1] = hi_; e[b2] = lo_;
                    }
                }
                const size_t o0 = (size_t)(qt * 32 + pq) * p.k * p.P + sgroup;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (u < p.k) { p.part_s[o0 + (size_t)u * p.P] = __uint_as_float(e[u].x); p.part_i[o0 + (size_t)u * p.P] = e[u].y; }
                for (int u = 8; u < p.k; ++u) { p.part_s[o0 + (size_t)u * p.P] = -INFINITY; p.part_i[o0 + (size_t)u * p.P] = MS_IDX_NONE; }
            }
        }
        MS_BODY_STAMP(4);
        MS_BODY_STAMP(5);
        return;
    }
    uint2 *lists = reinterpret_cast<uint2 *>(smem);    // [qw][sw][32 queries][K2]
    {
        uint2 *mine = lists + ((size_t)(qw * spb + sw) * 32 + r) * K2 + h * KL;
#pragma unroll
        for (int j = 0; j < KL; ++j) mine[j] = make_uint2(__float_as_uint(ls[j]), li[j]);
    }
    __syncthreads();
    // The 4/qwb stream lists of a query (sorted, K2 entries each, empty slots last) -> its best p.k: one thread per query advances the
    // best of the list heads p.k times and writes as it goes (rows are distinct: a total order).  One barrier; the general
    // rank-by-binary-search merge that stood here (four barriers, ~80 LDS round trips per thread) took 6 us of every launch.
    if (tid < p.qwb * 32) {
        const int pqw = tid >> 5, pq = tid & 31;
        const int qt = qg * p.qwb + pqw;
        if (qt < p.n_qtiles) {
            const uint2 none = make_uint2(__float_as_uint(-INFINITY), MS_IDX_NONE);
            const uint2 *L0 = lists + ((size_t)(pqw * spb) * 32 + pq) * K2;      // list u of the query: + u * 32 * K2
            uint2 head[4];
            int pos[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { pos[u] = 0; head[u] = (u < spb) ? L0[(size_t)u * 32 * K2] : none; }
            const size_t o0 = (size_t)(qt * 32 + pq) * p.k * p.P + sgroup;
            for (int rank = 0; rank < p.k; ++rank) {
                int best = 0;
                uint2 bh = head[0];
#pragma unroll
                for (int u = 1; u < 4; ++u)
                    if (ms_better(__uint_as_float(head[u].x), head[u].y, __uint_as_float(bh.x), bh.y)) { bh = head[u]; best = u; }
                p.part_s[o0 + (size_t)rank * p.P] = __uint_as_float(bh.x);
                p.part_i[o0 + (size_t)rank * p.P] = bh.y;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (u == best && bh.y != MS_IDX_NONE) {
                        pos[u] += 1;
                        head[u] = pos[u] < K2 ? L0[(size_t)u * 32 * K2 + pos[u]] : none;
                    }
                }
            }
        }
    }
    MS_BODY_STAMP(4);
    if (!MAXONLY && p.fin_s != nullptr) {
        // One launch per search for a handful of queries: the last workgroup of this query group to get here merges.
        // Hand-off by the book (cdna_hip_programming.md Guideline 16): every storing wave drains its stores, workgroup
        // barrier, ONE lane releases at agent scope and takes a ticket; the last arriver acquires, then everybody loads.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        volatile uint32_t *last_flag = reinterpret_cast<volatile uint32_t *>(smem);       // (the list area is free again)
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t tk = __hip_atomic_fetch_add(p.ticket + qg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool last = tk == (uint32_t)p.n_sgroups - 1u;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            *last_flag = last ? 1u : 0u;
        }
        __syncthreads();
        const bool i_am_last = *last_flag != 0u;
        __syncthreads();
        if (i_am_last) {
            MS_BODY_STAMP(6);
            // one query at a time: all 256 threads stage its [k][P] block (k * P <= 4224 entries, 33 KiB), then merge it as a
            // workgroup (ms_block_merge)
            const int q_first = qg * p.qwb * 32;
            const int q_count = (p.nq - q_first) < p.qwb * 32 ? (p.nq - q_first) : p.qwb * 32;
            const int kP = p.k * p.P;
            uint2 *ent = reinterpret_cast<uint2 *>(smem + MS_BLOCK_MERGE_SCRATCH);
            for (int q0 = 0; q0 < q_count; ++q0) {
                const size_t base = (size_t)(q_first + q0) * kP;
                if ((kP & 3) == 0) {
                    const f32x4 *ps4 = reinterpret_cast<const f32x4 *>(p.part_s + base);
                    const uint4 *pi4 = reinterpret_cast<const uint4 *>(p.part_i + base);
#pragma unroll 4
                    for (int e = tid; e < (kP >> 2); e += 256) {
                        const f32x4 sv = ps4[e];
                        const uint4 iv = pi4[e];
                        uint2 *dst = ent + 4 * e;
                        dst[0] = make_uint2(__float_as_uint(sv.x), iv.x); dst[1] = make_uint2(__float_as_uint(sv.y), iv.y);
                        dst[2] = make_uint2(__float_as_uint(sv.z), iv.z); dst[3] = make_uint2(__float_as_uint(sv.w), iv.w);
                    }
                } else {
#pragma unroll 4
                    for (int e = tid; e < kP; e += 256) ent[e] = make_uint2(__float_as_uint(p.part_s[base + e]), p.part_i[base + e]);
                }
                __syncthreads();
                MS_BODY_STAMP(7);
                const uint2 *fin = ms_block_merge(ent, smem, p.P, p.k, tid);
                const int q_out = p.fin_qmap != nullptr ? p.fin_qmap[q_first + q0] : q_first + q0;      // (uniform)
                if (fin == nullptr) {        // (uniform across the workgroup) the shapes it declines: head-advance merge by one wave
                    if (wave == 0)
                        ms_head_merge_wave<4, true>(ent, p.part_s, p.part_i, p.P, p.k, p.fin_row_offset, p.fin_s, p.fin_i, p.fin_stride, 0,
                                                    nullptr, nullptr, q_out, lane);
                } else if (tid < p.k) {
                    const uint2 v = fin[tid];
                    const size_t o = (size_t)q_out * p.fin_stride + tid;
                    p.fin_s[o] = __uint_as_float(v.x);
                    p.fin_i[o] = v.y == MS_IDX_NONE ? (int64_t)-1 : p.fin_row_offset + (int64_t)v.y;
                }
                __syncthreads();
            }
            if (tid == 0) p.ticket[qg] = 0u;             // (visible to the next launch: kernel boundary)
        }
    }
    MS_BODY_STAMP(5);
}

// ------------------------------------------------------------------ scan, loader-wave form
// MFMA-bound batches (>= 3 query tiles, qwb == 4): the 4 compute waves of a workgroup scan the SAME rows for 4 query
// tiles; a FIFTH wave does nothing but LDS-DMA: one copy of each tile into a ring of LDR_R slots shared by the
// workgroup, up to LDR_D tiles in flight (counted s_waitcnt vmcnt), published through ONE counter in LDS (`landed`:
// tiles are published in order); the compute waves look at it in the middle of chain t and report what they have
// consumed (one ds_add per tile).  No barrier: a wave delayed by insertions may trail the others by LDR_R - 1 tiles
// before the loader has to wait for it.  (The kernel needs <= 256 registers per wave so that the loader can share a
// SIMD with a compute wave: __launch_bounds__(320, 2).)
//
// What a tile costs (tools/probes/mfma_price_probe.hip, profiles/r03_mfma_price_probe_v1.log): the dependent chain of
// 64 v_mfma_f32_32x32x2_f32 runs at exactly 64 cycles per instruction (4096 per tile), but EVERY vector instruction the
// wave issues between them costs the chain 8-15 cycles (the fp32 matrix instruction runs at the vector rate and the
// two do not overlap inside one wave), and a second wave on the SIMD is starved while the chain runs
// (tools/probes/pair_probe.hip), so the work cannot be handed to a partner either.  The stage is therefore written
// to issue as few vector instructions as possible, and hipcc is not allowed to place them (it clusters ~60 of them
// behind the last MFMAs of the chain): MFMAs and fillers are volatile asm statements in program order.
//   * tile image in LDS is CHUNK-major, [16-byte chunk c = 0..31][row r = 0..31]: lane (r, h) reads its 16 fragments
//     at base + 512 f (immediate offsets, f = 0..15, base = slot + 8192 h + 16 r): conflict free (the 16 lanes of a
//     ds_read_b128 group have distinct r mod 16) with NO address arithmetic (the row-major XOR-swizzled image needs
//     one v_xor per read).  The loader's LDS-DMA piece `it` gathers chunks 2 it, 2 it + 1 of all 32 rows (32 bytes
//     per row; the four pieces that share a 128-byte line follow each other);
//   * filter of tile t-1: 8 v_max3 fold the lane's 16 scores into one maximum, ONE compare per tile;
//   * slot base of tile t+1: one v_add; loader flag: one ds_read_b32 + v_readfirstlane; consumed counter: ds_add
//     under an EXEC mask set by scalar moves.  11 vector instructions per tile in inner-product mode.
#ifndef MS_HIST_PERIOD
#define MS_HIST_PERIOD 16        // tiles between two looks at the shared bound's counters (a power of two)
#endif
#ifndef MS_LDR_R
#define MS_LDR_R 8
#endif
#ifndef MS_LDR_D
#define MS_LDR_D 3
#endif
constexpr int LDR_R = MS_LDR_R;  // ring slots (tiles)
constexpr int LDR_D = MS_LDR_D;  // tiles the loader keeps in flight before publishing the oldest
constexpr int LDR_AUX = 2 * LDR_R;   // aux (row scale / length) ring: a tile's aux data is read up to two stages after its slot was
                                     // released, while the loader may run LDR_R - 1 tiles ahead of the slowest wave
constexpr int LDR_LDS = LDR_R * 16384 + LDR_AUX * 256 + 64;
// Append-and-flush rare path (lists of 16 / 32 entries per lane, k > 20): candidate buffers, LDR_CAND entries of 8 bytes per lane
// and compute wave, behind the counters
#ifndef MS_APPEND_MIN_KL
#define MS_APPEND_MIN_KL 32
#endif
constexpr int LDR_CAND = 8;
constexpr int LDR_LDS_APPEND = LDR_LDS + 4 * LDR_CAND * 512;

// LDS-DMA pieces as inline asm; each statement overwrites M0 and declares it as a clobber:
// 64 lanes x 16 B (or 4 B) from global memory to LDS bytes lds_addr + lane * size.  M0 is a reserved
// register for hipcc, which therefore warns about the clobber; the clobber is what makes it re-load M0
// before any later use of its own (LDS-DMA builtins, v_readlane / movrel with M0), so it stays.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// SGPR base + 32-bit lane offset + immediate.  The immediate is added to the global address AND to the LDS address
// (tools/probes/glds_offset_probe.hip), so M0 carries the destination minus the immediate.
template <int IMM>
__device__ __forceinline__ void ms_glds_s16(uint32_t lds_addr, uint32_t lane_off, uint64_t sbase) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3" ::"s"(lds_addr - (uint32_t)IMM), "v"(lane_off), "s"(sbase), "i"(IMM) : "memory", "m0");
}
template <int IMM>      // the same with the non-temporal policy (rows that ONE workgroup reads once)
__device__ __forceinline__ void ms_glds_s16_nt(uint32_t lds_addr, uint32_t lane_off, uint64_t sbase) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3 nt" ::"s"(lds_addr - (uint32_t)IMM), "v"(lane_off), "s"(sbase), "i"(IMM) : "memory", "m0");
}
template <int IMM>      // the same with sc1: past this CU's L1 (counters other workgroups are adding to)
__device__ __forceinline__ void ms_glds_s16_sc1(uint32_t lds_addr, uint32_t lane_off, uint64_t sbase) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3 sc1" ::"s"(lds_addr - (uint32_t)IMM), "v"(lane_off), "s"(sbase), "i"(IMM) : "memory", "m0");
}
__device__ __forceinline__ void ms_glds_v4(uint32_t lds_addr, const void *lane_ptr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(lds_addr), "v"(lane_ptr) : "memory", "m0");
}
#pragma clang diagnostic pop

template <bool AUX, int N>
__device__ __forceinline__ void ms_vmcnt_tiles() {   // wait until at most N tiles' worth of DMA pieces are in flight
    constexpr int P = AUX ? 17 : 16;
    if (N * P == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (N * P == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (N * P == 17) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
    else if (N * P == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else if (N * P == 34) asm volatile("s_waitcnt vmcnt(34)" ::: "memory");
    else if (N * P == 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
    else if (N * P == 51) asm volatile("s_waitcnt vmcnt(51)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#ifdef MS_ABL_NOFLAG
#define MS_ABL_NOFLAG_ 1
#else
#define MS_ABL_NOFLAG_ 0
#endif
// Pinned instructions of the compute waves' stage (volatile asm statements keep their program order).
#define MS_MFMA(ACC, A, B) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
#define MS_MFMA_Z(ACC, A, B) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=v"(ACC) : "v"(A), "v"(B))   // C = 0 inline: a new chain
#define MS_MAX3(M, A, B) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(M) : "v"(A), "v"(B))
#define MS_FRAG_READ(DST, BASE, IMM) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(BASE), "i"(IMM) : "memory")

// Prefilter (PF): a float x is split into hi = its upper 16 bits (a bf16 by truncation) and lo = the upper 16 bits of
// x - hi (exact in fp32); x = hi + lo + r with |x - hi| < 2^-7 |x| and |r| < 2^-14 |x|.  Eight consecutive floats of a row
// (two fragments) become the two 8 x bf16 operands of v_mfma_f32_32x32x16_bf16; the same split of the query is the B side.
// hi.hi + hi.lo + lo.hi misses lo.lo and the r terms: |approx - exact| <= 3 * 2^-14 |x||q| plus the fp32 accumulation of 384
// products: < 2.5e-4 |x||q| (MS_PF_ERR).  bf16 x bf16 products are exact in fp32.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MS_PF_ERR 2.5e-4f
__device__ __forceinline__ void ms_split8(const f32x4 &x0, const f32x4 &x1, bf16x8 &hi, bf16x8 &lo) {
    typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    u32x4_ H, L;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t a = __float_as_uint(x[2 * i]), b = __float_as_uint(x[2 * i + 1]);
        H[i] = __builtin_amdgcn_perm(b, a, 0x07060302u);            // upper halves of b : a
        // x - hi for the pair in one packed subtraction (v_pk_add_f32 with the second operand negated)
        const f32x2_ xv = {x[2 * i], x[2 * i + 1]};
        const f32x2_ tv = {__uint_as_float(a & 0xFFFF0000u), __uint_as_float(b & 0xFFFF0000u)};
        const f32x2_ lv = xv - tv;
        L[i] = __builtin_amdgcn_perm(__float_as_uint(lv.y), __float_as_uint(lv.x), 0x07060302u);
    }
    hi = __builtin_bit_cast(bf16x8, H);
    lo = __builtin_bit_cast(bf16x8, L);
}

// SAMPLE: the sample pass in this form -- no lists, every lane keeps the maximum of its half of the rows of the first
// max_tiles FULL tiles of its stream; the two maxima of a lane pair (distinct rows) are the stream's entry for the
// bound selection (ms_sample_bound_kernel looks at values only).
// AUXM: 0 = inner-product mode; 1 = cosine mode on raw rows (1/|row| and the length mask applied to every score inside the
// chain: ~50 more vector instructions per tile); 2 = cosine mode on UNIT rows (MS_MODE_COSINE_UNIT): the scores are final
// as they leave the matrix pipe, and a masked row's score (+-0) can only matter to a query whose threshold is negative,
// so the in-chain filter is the inner-product one and the mask is applied in the rare path (all of a tile's scores are
// re-derived there); a wave with a negative threshold somewhere visits the rare path for every tile until it is gone.
// PF: the prefilter's scan (inner-product modes): same loader, ring, lists, bounds and merges, on approximate scores.
template <int KL, int AUXM, bool SAMPLE, bool PF = false>
__global__ __launch_bounds__(320, 2) void ms_scan_loader_kernel(const ScanParams p) {
    static_assert(!PF || AUXM == 0, "the prefilter exists for the inner-product modes");
    if (ms_gate_closed(p.gate, p.gate_epoch)) return;          // (uniform: a scalar load)
    constexpr bool AUX = AUXM != 0;
    constexpr bool SCALE_IN_CHAIN = AUXM == 1 || (AUXM == 2 && SAMPLE);      // (the sample pass needs every score final)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // 0..3 compute, 4 loader
    const int r = lane & 31, h = lane >> 5;
    // smem: LDR_R tile slots of 16 KiB, then the aux ring, then the counters
    float *auxring = reinterpret_cast<float *>(smem + LDR_R * 16384);                 // LDR_AUX x 64 floats
    // counters, as explicit LDS (address space 3) pointers: a volatile access through a generic
    // pointer compiles to flat_load + s_waitcnt vmcnt(0), hundreds of cycles in the middle of a chain
    typedef volatile __attribute__((address_space(3))) uint32_t lds_flag_t;
    lds_flag_t *landed = (lds_flag_t *)(smem + LDR_R * 16384 + LDR_AUX * 256);        // [1] tiles published by the loader (in order)
    lds_flag_t *consumed = landed + 8;                                                 // [4] tiles read by compute wave w

    const int bid = blockIdx.x;
    const int per_super = 8 * p.n_qgroups;
    const int super = bid / per_super, within = bid % per_super;
    const int stream = super * 8 + (within & 7);      // qwb == 4: one stream per workgroup
    const int qg = within >> 3;
    if (stream >= p.n_streams) return;
    const int64_t row_begin = (int64_t)stream * p.rows_per_stream;
    const int64_t row_end = (row_begin + p.rows_per_stream < p.n) ? row_begin + p.rows_per_stream : p.n;
    const int nfull = (int)((row_end - row_begin) >> 5);
    const int rem = (int)((row_end - row_begin) & 31);
    // tiles the loader delivers: the whole stream (the last tile may be partial), or the first max_tiles FULL tiles (sample pass)
    const int ntl = SAMPLE ? (nfull < p.max_tiles ? nfull : p.max_tiles) : nfull + (rem > 0 ? 1 : 0);

    if (tid < 16) {
        uint32_t v = 0;
        if (tid >= 8 && tid < 12) v = ((qg * 4 + (tid - 8)) < p.n_qtiles) ? 0u : 0xFFFFFFFFu;   // padding query tiles never block the loader
        landed[tid] = v;
    }
    __syncthreads();
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem);

    if (wave == 4) {
        // ---------------- loader ----------------
        // The loader shares a SIMD (and its vector issue port) with a compute wave that issues MFMAs back to back, so it
        // is written to need as few instructions as possible: raised priority, and every LDS-DMA piece is one s_mov m0 +
        // one global_load_lds with an SGPR base, ONE lane-offset register and an immediate (inline asm).
        __builtin_amdgcn_s_setprio(3);
        // piece `it` -> LDS bytes [it * 1024, it * 1024 + 1024) of the slot = chunks 2 it (lanes 0-31) and 2 it + 1 (lanes 32-63)
        // of rows r = lane & 31: source byte r * 512 + (2 it + h) * 16 = voff + 32 it
        const uint32_t voff_full = (uint32_t)(r * 512 + h * 16);
        const uint32_t aux_lds = ring_lds + LDR_R * 16384;
#ifdef MS_STAMP
        unsigned long long lst_poll = 0, lst_issue = 0, lst_vm = 0, lst_t0 = __builtin_amdgcn_s_memtime();
#define LST(acc) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - lst_t0; lst_t0 = now_; }
#else
#define LST(acc)
#endif
        for (int t = 0; t < ntl; ++t) {
            if (t >= LDR_R) {                          // slot t % R is free once everybody has read tile t - R
#ifdef MS_ABL_EXEC1
                const uint32_t need = (uint32_t)(t - LDR_R + 1);
#else
                const uint32_t need = (uint32_t)(t - LDR_R + 1) * 64u;      // (every lane of a compute wave adds: units of 64)
#endif
                for (uint32_t spins = 0;; ++spins) {
                    const uint32_t c0 = consumed[0], c1 = consumed[1], c2 = consumed[2], c3 = consumed[3];
                    const uint32_t m01 = c0 < c1 ? c0 : c1, m23 = c2 < c3 ? c2 : c3;
                    if (__builtin_amdgcn_readfirstlane(m01 < m23 ? m01 : m23) >= need) break;
                    if (spins > (1u << 24)) __builtin_trap();     // never a silent hang
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            LST(lst_poll)
            const int64_t row0 = row_begin + (int64_t)t * 32;
            const uint32_t slot_lds = ring_lds + (uint32_t)(t % LDR_R) * 16384u;
            if (AUX) {
                int64_t row = row0 + r;
                if (row >= p.n) row = p.n - 1;
                const float *base = ((h == 1 || p.inv_norm == nullptr) && p.lengths != nullptr) ? p.lengths : p.inv_norm;
                ms_glds_v4(aux_lds + (uint32_t)(t % LDR_AUX) * 256u, base + row);
            }
            // last tile of the database: rows past the end re-read the last row (their scores are discarded)
            uint32_t voff = voff_full;
            if (row0 + 32 > p.n) {
                const int last = (int)(p.n - 1 - row0);
                voff = (uint32_t)((r < last ? r : last) * 512 + h * 16);
            }
            const uint64_t b = (uint64_t)(uintptr_t)p.db + (uint64_t)row0 * 512u;
            const uint32_t b_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);        // (the builtin returns int:
            const uint32_t b_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));  //  no sign extension)
            const uint64_t sb = ((uint64_t)b_hi << 32) | (uint64_t)b_lo;
#ifdef MS_ABL_NODMA
#define MS_PIECE(IT)
#else
            // (prefilter: 64 idle cycles behind every piece -- a burst of 16 VMEM instructions holds the SIMD's vector issue port
            //  against the compute wave it shares with, whose stage is vector-issue bound there: 4-6 % of the call; the fp32 stage,
            //  bound by the matrix pipe, is faster with the burst)
#define MS_PIECE(IT) ms_glds_s16<32 * (IT)>(slot_lds + (IT) * 1024, voff, sb); if (PF) __builtin_amdgcn_s_sleep(1);
#endif
            MS_PIECE(0) MS_PIECE(1) MS_PIECE(2) MS_PIECE(3) MS_PIECE(4) MS_PIECE(5) MS_PIECE(6) MS_PIECE(7)
            MS_PIECE(8) MS_PIECE(9) MS_PIECE(10) MS_PIECE(11) MS_PIECE(12) MS_PIECE(13) MS_PIECE(14) MS_PIECE(15)
#undef MS_PIECE
            LST(lst_issue)
            if (t >= LDR_D - 1) {                      // tile t - (D-1) has landed: publish it
                ms_vmcnt_tiles<AUX, LDR_D - 1>();
                if (lane == 0) landed[0] = (uint32_t)(t - (LDR_D - 1) + 1);
            }
            LST(lst_vm)
        }
#ifdef MS_STAMP
        if (!SAMPLE && lane == 0 && p.stamps != nullptr && (size_t)bid * 64 + 64 <= 4 * 4 * 65536) {
            unsigned long long *o = p.stamps + ((size_t)bid * 8 + 4) * 8;
            o[0] = lst_poll; o[1] = lst_issue; o[2] = lst_vm; o[3] = (unsigned long long)ntl;
        }
#endif
        // drain: publish the last D-1 tiles
        if (ntl >= 2) {
            ms_vmcnt_tiles<AUX, 1>();
            if (lane == 0) landed[0] = (uint32_t)(ntl - 1);
        }
        if (ntl >= 1) {
            ms_vmcnt_tiles<AUX, 0>();
            if (lane == 0) landed[0] = (uint32_t)ntl;
        }
        return;
    }

    // ---------------- compute waves ----------------
    const int qtile = qg * 4 + wave;
    if (qtile >= p.n_qtiles) return;
    ScanState<SAMPLE ? 1 : KL> st;
#pragma unroll
    for (int j = 0; j < (SAMPLE ? 1 : KL); ++j) { st.ls[j] = -INFINITY; st.li[j] = MS_IDX_NONE; }
    st.floor = -INFINITY;
#ifdef MS_DEBUG_NO_INSERT
    st.tau = INFINITY;
#else
    st.tau = -INFINITY;
#endif
    const int qidx = qtile * 32 + r;
    const bool q_valid = qidx < p.nq;
    if (!SAMPLE && p.lb_s != nullptr) {
        const float lb = p.lb_s[qidx];
        st.floor = (lb == -INFINITY) ? -INFINITY : nextafterf(lb, -INFINITY);
#ifndef MS_DEBUG_NO_INSERT
        st.tau = st.floor;
#endif
    }
    if (!q_valid) { st.floor = INFINITY; st.tau = INFINITY; }   // padding queries never pass the filter
    ScanHist hg;
    hg.counters = nullptr; hg.base = 0.0f; hg.step = 0.0f; hg.inv_step = 0.0f;
    const bool hist_on = !SAMPLE && p.hist != nullptr && p.lb_s != nullptr;      // (uniform)
    if (hist_on && q_valid) {
        const float stp = p.hstep[qidx], lb = p.lb_s[qidx];
        if (stp > 0.0f && lb > -INFINITY) { hg.counters = p.hist + (size_t)qidx * 16; hg.base = lb; hg.step = stp; hg.inv_step = 1.0f / stp; }
    }
    float qreg[64];
    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(p.qn + (size_t)(q_valid ? qidx : 0) * MS_DIM + 64 * h);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            f32x4 v = src[t];
            if (!q_valid) v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};     // (p.qn may be the caller's own [nq,128] array)
            qreg[4 * t + 0] = v.x; qreg[4 * t + 1] = v.y; qreg[4 * t + 2] = v.z; qreg[4 * t + 3] = v.w;
        }
    }
    bf16x8 qhi[8], qlo[8];            // PF: the query tile as split bf16 operands, k block b = elements 64 h + 8 b .. + 7
    if (PF) {
#pragma unroll
        for (int b = 0; b < 8; ++b)
            ms_split8(f32x4{qreg[8 * b], qreg[8 * b + 1], qreg[8 * b + 2], qreg[8 * b + 3]},
                      f32x4{qreg[8 * b + 4], qreg[8 * b + 5], qreg[8 * b + 6], qreg[8 * b + 7]}, qhi[b], qlo[b]);
    }
    float my_qlen = 0.0f;
    if (AUX) my_qlen = (p.qlen != nullptr && q_valid) ? p.qlen[qidx] : 0.0f;
    const float qlen_eff = (AUX && p.lengths == nullptr) ? INFINITY : my_qlen;      // no lengths: +inf >= x * 0
    const float mincov_eff = (AUX && p.lengths == nullptr) ? 0.0f : p.mincov;

    auto wait_landed = [&](uint32_t need) -> uint32_t {          // until the loader has published `need` tiles; returns the counter
        uint32_t seen;
        for (uint32_t spins = 0; (seen = (uint32_t)__builtin_amdgcn_readfirstlane(landed[0])) < need; ++spins) {
            if (spins > (1u << 24)) __builtin_trap();             // never a silent hang
            __builtin_amdgcn_s_sleep(1);
        }
        return seen;
    };
    auto filter_group = [&](const f32x16 &acc, int t, int g, bool check_rows, float (&sc)[16], uint64_t (&m)[16]) {
        const int64_t sub_row0 = row_begin + (int64_t)t * 32;
        const int64_t rbase = sub_row0 + 8 * g + 4 * h;
        f32x4 inv4 = {1.0f, 1.0f, 1.0f, 1.0f}, len4 = {0.0f, 0.0f, 0.0f, 0.0f};
        if (AUX) {          // the group's 4 row scales and 4 row lengths: two ds_read_b128
            const f32x4 *ax = reinterpret_cast<const f32x4 *>(auxring + (t & (LDR_AUX - 1)) * 64 + 8 * g + 4 * h);
            if (AUXM == 1) inv4 = ax[0];
            len4 = ax[8];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = acc[4 * g + j];
            if (AUX) {
                float sv = s * inv4[j];                                                      // 1 / max(|row|, 1e-8) (unit rows: 1)
                const float mk = (qlen_eff >= len4[j] * mincov_eff) ? 1.0f : 0.0f;           // dbsearch.py:76
                sv = sv * mk;                                                                // dbsearch.py:78
                s = (t >= 0) ? sv : -INFINITY;     // "tile -1" of the pipeline has no aux data
            }
            sc[4 * g + j] = s;
            bool pass = s > st.tau;
            if (check_rows) pass = pass && (rbase + j < row_end);
            m[4 * g + j] = __ballot(pass);
        }
    };
    // cosine mode: final scores of group g of the previous tile, in place (the inner-product mode needs no such step)
    auto scale_group = [&](f32x16 &acc, int t, int g) {
        const f32x4 *ax = reinterpret_cast<const f32x4 *>(auxring + (t & (LDR_AUX - 1)) * 64 + 8 * g + 4 * h);
        const f32x4 inv4 = (AUXM == 1) ? ax[0] : f32x4{1.0f, 1.0f, 1.0f, 1.0f}, len4 = ax[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float sv = acc[4 * g + j] * inv4[j];
            const float mk = (qlen_eff >= len4[j] * mincov_eff) ? 1.0f : 0.0f;
            sv = sv * mk;
            acc[4 * g + j] = (t >= 0) ? sv : -INFINITY;
        }
    };

    f32x4 areg[16];
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 hc0 = {0, 0, 0, 0}, hc1 = hc0, hc2 = hc0, hc3 = hc0;      // the shared bound's counters of this lane's query, as last fetched
    const uint32_t lin0 = ring_lds + (uint32_t)(8192 * h + 16 * r);         // fragment f of this lane in slot s: lin0 + 16384 s + 512 f
    uint32_t landed_addr = ring_lds + LDR_R * 16384 + LDR_AUX * 256;
    uint32_t cons_addr = landed_addr + 32 + 4 * wave, two = 2, flag = 0;
    asm volatile("" : "+v"(cons_addr), "+v"(two), "+v"(flag), "+v"(landed_addr));     // kept in vector registers across the loop
    float smax = -INFINITY;                                                 // SAMPLE: this lane's best score so far
    bool neg_tau = AUXM == 2 && (__ballot(st.tau < 0.0f) != 0);             // unit rows: some query of this wave still has a negative threshold
    uint32_t landed_seen = 0;                                               // the loader's counter as this wave last saw it (scalar)
    // Rare path of a tile whose lane maximum passed: the candidates go into the lists.
    //   lists of 5 / 10 entries per lane: one sorted-insertion step per database ROW that holds a candidate (ms_tile_insert);
    //   lists of 32 (APPEND): such a step costs 3,300 cycles per visit there (stamps: 24 % of the tiles at k = 64), so a lane whose score passes only counts it in the shared bound's histogram and APPENDS (score, row) to a
    //   buffer of its own in LDS; when some lane holds more than LDR_CAND - 4 entries (checked once per group of four score
    //   registers), and at the end of the stream, the buffers are emptied: per round every query takes the candidate with the
    //   smaller row of its two lanes' buffer heads (rows ascend within a buffer: a two-way merge, so ties still go to the lower
    //   row) and ONE insertion step serves all 32 queries (ms_pair_insert).  Thresholds move at flushes and with the shared bound.
    constexpr bool APPEND = KL >= MS_APPEND_MIN_KL && !SAMPLE && !PF;
    typedef __attribute__((address_space(3))) ms_u32x2 lds_cand_t;
    uint32_t ccnt = 0;
#ifdef MS_STAMP_FLUSH
    unsigned long long stamp_flush = 0, stamp_nflush = 0, stamp_rounds = 0;
#endif
    auto cand_slot = [&](uint32_t c) __attribute__((always_inline)) -> lds_cand_t * {
        return (lds_cand_t *)((__attribute__((address_space(3))) char *)smem + LDR_LDS + (wave * LDR_CAND) * 512) + c * 64 + lane;
    };
    auto flush = [&]() __attribute__((always_inline)) {
        uint32_t head = 0;
#ifdef MS_STAMP_FLUSH
        const unsigned long long f0_ = __builtin_amdgcn_s_memtime();
        stamp_nflush += 1;
#endif
#pragma unroll 1
        while (__ballot(head < ccnt) != 0) {
#ifdef MS_STAMP_FLUSH
            stamp_rounds += 1;
#endif
            const bool have = head < ccnt;
            const ms_u32x2 e = *cand_slot(have ? head : 0);
            const float v = have ? __uint_as_float(e.x) : -INFINITY;
            const uint32_t row = have ? e.y : MS_IDX_NONE;
            const float pv = ms_xor32_f(v, h);
            const uint32_t prow = ms_xor32_u(row, h);
            const bool take = row < prow;                 // (the two halves hold distinct rows; both exhausted: nobody)
            head += take ? 1u : 0u;
            ms_pair_insert<SAMPLE ? 1 : KL>(st, take ? v : pv, take ? row : prow, h);
        }
        ccnt = 0;
#ifdef MS_STAMP_FLUSH
        stamp_flush += __builtin_amdgcn_s_memtime() - f0_;
#endif
    };
    auto rare = [&](f32x16 &prev, int tp) __attribute__((always_inline)) {
        if (AUXM == 2) {            // unit rows: the length mask (dbsearch.py:76,78) is applied here, to the whole tile
            scale_group(prev, tp, 0); scale_group(prev, tp, 1); scale_group(prev, tp, 2); scale_group(prev, tp, 3);
        }
        if constexpr (APPEND) {
            const uint32_t row0 = (uint32_t)(row_begin + (int64_t)tp * 32) + (uint32_t)(4 * h);
#define MS_APPEND_GROUP(G)                                                                                              \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                             \
                const float s_ = prev[4 * (G) + j];                                                                     \
                const bool pass = s_ > st.tau;                                                                          \
                if (__ballot(pass) != 0) {                                                                              \
                    if (pass) {                                                                                         \
                        float s2_ = s_;                                                                                 \
                        asm volatile("" : "+v"(s2_));       /* (or hipcc computes all 16 bucket addresses up front: spills) */ \
                        if (hg.counters != nullptr) ms_hist_count(hg, s2_);                                             \
                        *cand_slot(ccnt) = ms_u32x2{__float_as_uint(s2_), row0 + (uint32_t)(8 * (G) + j)};              \
                        ccnt += 1;                                                                                      \
                    }                                                                                                   \
                }                                                                                                       \
            }
#pragma unroll 1
            for (int g = 0; g < 4; ++g) {          // (a runtime loop: ONE copy of the flush)
                int g_ = g;
                asm volatile("" : "+s"(g_));
                if (__builtin_expect(__ballot(ccnt > (uint32_t)(LDR_CAND - 4)) != 0, 0)) flush();
                if (g_ == 0) { MS_APPEND_GROUP(0) } else if (g_ == 1) { MS_APPEND_GROUP(1) } else if (g_ == 2) { MS_APPEND_GROUP(2) } else { MS_APPEND_GROUP(3) }
            }
#undef MS_APPEND_GROUP
        } else {
            float sc[16];
            uint64_t m[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { sc[i] = prev[i]; m[i] = __ballot(sc[i] > st.tau); }
            ms_tile_insert<SAMPLE ? 1 : KL, true>(st, sc, m, row_begin + (int64_t)tp * 32, r, h, &hg);
        }
        if (AUXM == 2) neg_tau = __ballot(st.tau < 0.0f) != 0;
    };

#ifdef MS_STAMP
    unsigned long long stamp_wait = 0, stamp_nwait = 0, stamp_ins = 0, stamp_nins = 0;
#endif
    // One stage = the chain of tile t (from areg) into `out`; in its gaps: the filter of tile t-1 (`prev`, raw scores;
    // cosine mode scales them in place first), the look at the loader's counter and the refill of areg with tile t+1.
    // After the chain: the rare insertion steps of tile t-1.
    // Synchronisation with the loader is per PAIR of tiles (the loop runs two stages per iteration): before the first
    // stage of a pair the wave makes sure that both tiles the pair will refill from have landed -- from the counter value
    // it cached, or from the copy of the counter the previous stage left in `flag` (one v_readfirstlane, only when the
    // cached value does not cover the pair), or by waiting; the second stage reports both tiles as consumed (every lane
    // adds to the counter: it counts in units of 64, and no EXEC juggling sits in front of the chain) and re-reads the
    // loader's counter into `flag` for the next pair.  The chain itself contains no branch and no scalar dependency.
    auto ensure_landed = [&](int t) {            // before the first stage of a pair: tiles t+1 and t+2 (as far as they exist)
        const uint32_t need = (uint32_t)((t + 3 < ntl) ? t + 3 : ntl);
        if (__builtin_expect(!MS_ABL_NOFLAG_ && landed_seen < need, 0)) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(flag) :: "memory");
            landed_seen = __builtin_amdgcn_readfirstlane(flag);
            if (landed_seen < need) {                                         // normally long since published
#ifdef MS_STAMP
                const unsigned long long w0 = __builtin_amdgcn_s_memtime();
#endif
                landed_seen = wait_landed(need);
#ifdef MS_STAMP
                stamp_wait += __builtin_amdgcn_s_memtime() - w0;
                stamp_nwait += 1;
#endif
            }
        }
    };
    auto stage = [&](auto first_c, int t, f32x16 &prev, f32x16 &out) {
        constexpr bool FIRST = decltype(first_c)::value;
        // Tile t is in areg (the statement names areg so that no use of it is placed above the wait) -- except, where the
        // chain's gaps hold no other LDS reads (LATE_WAIT), its last four fragments: they were requested behind the last
        // MFMAs of the previous chain, are not needed before group 12 of this one, and waiting for them here would expose
        // their LDS latency (~50 cycles per tile).  LDS operations return in order, so "all but the 4 youngest" is exact.
        constexpr bool LATE_WAIT = !SCALE_IN_CHAIN;
#define MS_AREG_ALL "+v"(areg[0]), "+v"(areg[1]), "+v"(areg[2]), "+v"(areg[3]), "+v"(areg[4]), "+v"(areg[5]), "+v"(areg[6]), "+v"(areg[7]), \
                    "+v"(areg[8]), "+v"(areg[9]), "+v"(areg[10]), "+v"(areg[11]), "+v"(areg[12]), "+v"(areg[13]), "+v"(areg[14]), "+v"(areg[15])
        if (LATE_WAIT) asm volatile("s_waitcnt lgkmcnt(4)" : MS_AREG_ALL, "+v"(flag) :: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : MS_AREG_ALL, "+v"(flag) :: "memory");
        if (FIRST) {
            // (the look at the loader's counter for this pair sits in front of the stage, in `ensure_landed`)
        } else {
#ifdef MS_ABL_EXEC1
            asm volatile("s_mov_b64 exec, 1\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(cons_addr), "v"(two) : "memory");
#else
            asm volatile("ds_add_u32 %0, %1" ::"v"(cons_addr), "v"(two) : "memory");     // tiles t-1 and t are in registers
#endif
        }
        const uint32_t slot_off = (uint32_t)((t + 1) % LDR_R) * 16384u;
        float mx;
        uint32_t rbase;
#define MS_GROUP(TT, ZERO_C)                                                                                  \
        if (ZERO_C) { MS_MFMA_Z(out, areg[TT].x, qreg[4 * (TT) + 0]); } else { MS_MFMA(out, areg[TT].x, qreg[4 * (TT) + 0]); } \
        MS_MFMA(out, areg[TT].y, qreg[4 * (TT) + 1]); MS_MFMA(out, areg[TT].z, qreg[4 * (TT) + 2]); MS_MFMA(out, areg[TT].w, qreg[4 * (TT) + 3]);
#define MS_REFILL(TT) MS_FRAG_READ(areg[2 * ((TT) - 8)], rbase, 512 * (2 * ((TT) - 8))); MS_FRAG_READ(areg[2 * ((TT) - 8) + 1], rbase, 512 * (2 * ((TT) - 8) + 1));
        MS_GROUP(0, true)
        MS_GROUP(1, false)
        if (SCALE_IN_CHAIN) { __builtin_amdgcn_sched_barrier(0); scale_group(prev, t - 1, 0); scale_group(prev, t - 1, 1); __builtin_amdgcn_sched_barrier(0); }
        MS_GROUP(2, false)
        if (SCALE_IN_CHAIN) { __builtin_amdgcn_sched_barrier(0); scale_group(prev, t - 1, 2); scale_group(prev, t - 1, 3); __builtin_amdgcn_sched_barrier(0); }
        MS_GROUP(3, false)
        // the lane's maximum over the 16 scores of tile t-1 (the running maximum of the whole sample in SAMPLE mode)
#ifdef MS_ABL_NOMAX
        mx = prev[0];
#define MS_MAX3_(M, A, B)
#else
#define MS_MAX3_(M, A, B) MS_MAX3(M, A, B)
#endif
        if (SAMPLE) {
            MS_MAX3(smax, prev[0], prev[1]); MS_MAX3(smax, prev[2], prev[3]); MS_MAX3(smax, prev[4], prev[5]); MS_MAX3(smax, prev[6], prev[7]);
        } else {
#ifndef MS_ABL_NOMAX
            asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(prev[0]), "v"(prev[1]), "v"(prev[2]));
#endif
            MS_MAX3_(mx, prev[3], prev[4]); MS_MAX3_(mx, prev[5], prev[6]); MS_MAX3_(mx, prev[7], prev[8]);
        }
        MS_GROUP(4, false)
        if (SAMPLE) {
            MS_MAX3(smax, prev[8], prev[9]); MS_MAX3(smax, prev[10], prev[11]); MS_MAX3(smax, prev[12], prev[13]); MS_MAX3(smax, prev[14], prev[15]);
        } else {
            MS_MAX3_(mx, prev[9], prev[10]); MS_MAX3_(mx, prev[11], prev[12]); MS_MAX3_(mx, prev[13], prev[14]);
#ifndef MS_ABL_NOMAX
            asm volatile("v_max_f32 %0, %0, %1" : "+v"(mx) : "v"(prev[15]));
#endif
        }
        if (!FIRST && !MS_ABL_NOFLAG_) asm volatile("ds_read_b32 %0, %1" : "=v"(flag) : "v"(landed_addr) : "memory");     // for the next pair
        MS_GROUP(5, false)
        MS_GROUP(6, false)
        asm volatile("v_add_u32 %0, %1, %2" : "=v"(rbase) : "s"(slot_off), "v"(lin0));
        MS_GROUP(7, false)
        // fragments of groups already consumed <- tile t+1 (past the last tile: a stale slot, never used)
        MS_GROUP(8, false) MS_REFILL(8)
        MS_GROUP(9, false) MS_REFILL(9)
        MS_GROUP(10, false) MS_REFILL(10)
        MS_GROUP(11, false) MS_REFILL(11)
        if (LATE_WAIT) {    // fragments 12..15 of THIS tile: everything older than this stage's own LDS operations (8 refill reads; the
                            // second stage of a pair also issued its counter add and re-read) has returned
            if (FIRST) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(areg[12]), "+v"(areg[13]), "+v"(areg[14]), "+v"(areg[15]) :: "memory");
            else if (MS_ABL_NOFLAG_) asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(areg[12]), "+v"(areg[13]), "+v"(areg[14]), "+v"(areg[15]) :: "memory");
            else asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(areg[12]), "+v"(areg[13]), "+v"(areg[14]), "+v"(areg[15]) :: "memory");
        }
        MS_GROUP(12, false) MS_REFILL(12)
        MS_GROUP(13, false) MS_REFILL(13)
        MS_GROUP(14, false) MS_REFILL(14)
        MS_GROUP(15, false) MS_REFILL(15)
#undef MS_GROUP
#undef MS_REFILL
#undef MS_MAX3_
#undef MS_AREG_ALL
        __builtin_amdgcn_sched_barrier(0);
#ifdef MS_ABL_NOCMP
        asm volatile("" ::"v"(mx));
        if (false) {
            if (false) {
#else
        if (!SAMPLE) {
            if (__builtin_expect(__ballot(mx > st.tau) != 0 || (AUXM == 2 && neg_tau), 0)) {
#endif
#ifdef MS_STAMP
                const unsigned long long i0 = __builtin_amdgcn_s_memtime();
#endif
                // Compiler-scheduled code follows, and the fragment registers still have LDS reads in flight (issued by the asm
                // statements above): everything has to have landed before hipcc may touch -- move, spill to AGPRs -- any of them
                // (with 32-entry lists it did, a copy taken before the data was there put stale fragments into the next tile:
                // one run in twenty lost a row somewhere).
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(areg[0]), "+v"(areg[1]), "+v"(areg[2]), "+v"(areg[3]), "+v"(areg[4]), "+v"(areg[5]),
                             "+v"(areg[6]), "+v"(areg[7]), "+v"(areg[8]), "+v"(areg[9]), "+v"(areg[10]), "+v"(areg[11]), "+v"(areg[12]),
                             "+v"(areg[13]), "+v"(areg[14]), "+v"(areg[15]), "+v"(flag) :: "memory");
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(hc0), "+v"(hc1), "+v"(hc2), "+v"(hc3) :: "memory");     // (a counter fetch may be in flight too)
                rare(prev, t - 1);
#ifdef MS_STAMP
                stamp_ins += __builtin_amdgcn_s_memtime() - i0;
                stamp_nins += 1;
#endif
            }
        }
    };

    // 32-entry lists (k <= 64) leave no room for a whole tile of fragments in registers (64 of them): RING_FRAGS keeps FOUR
    // fragment registers instead -- group g of the chain waits for its fragment (LDS reads return in order: "all but the
    // two youngest"), and requests the one three groups ahead, the first three of the next tile during the last three groups.
    // Same synchronisation with the loader as above, except that a tile is reported as consumed when its last fragment
    // has been requested (group 12 of its own chain) instead of before its chain starts.
    constexpr bool RING_FRAGS = KL > 16 && !SAMPLE && !PF;
    uint32_t rb0 = lin0, rb1 = lin0;          // lane base of the slot of the current / next tile (the two stages of a pair swap them)
    auto stage_ring = [&](auto first_c, int t, f32x16 &prev, f32x16 &out) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value;
        uint32_t &rcur = FIRST ? rb0 : rb1;
        uint32_t &rnext = FIRST ? rb1 : rb0;
        const uint32_t slot_off = (uint32_t)((t + 1) % LDR_R) * 16384u;
        asm volatile("v_add_u32 %0, %1, %2" : "=v"(rnext) : "s"(slot_off), "v"(lin0));
        float mx;
#define MS_RWAIT(TT) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(areg[(TT) & 3]), "+v"(flag) :: "memory")
#define MS_RREAD(TT)                                                                                           \
        if ((TT) + 3 < 16) { MS_FRAG_READ(areg[((TT) + 3) & 3], rcur, 512 * (((TT) + 3) & 15)); }             \
        else { MS_FRAG_READ(areg[((TT) + 3) & 3], rnext, 512 * (((TT) + 3) & 15)); }
#define MS_RGROUP(TT, ZERO_C)                                                                                  \
        MS_RWAIT(TT);                                                                                          \
        if (ZERO_C) { MS_MFMA_Z(out, areg[(TT) & 3].x, qreg[4 * (TT) + 0]); } else { MS_MFMA(out, areg[(TT) & 3].x, qreg[4 * (TT) + 0]); } \
        MS_MFMA(out, areg[(TT) & 3].y, qreg[4 * (TT) + 1]); MS_MFMA(out, areg[(TT) & 3].z, qreg[4 * (TT) + 2]); MS_MFMA(out, areg[(TT) & 3].w, qreg[4 * (TT) + 3]); \
        MS_RREAD(TT)
        MS_RGROUP(0, true)
        MS_RGROUP(1, false)
        if (SCALE_IN_CHAIN) { __builtin_amdgcn_sched_barrier(0); scale_group(prev, t - 1, 0); scale_group(prev, t - 1, 1); __builtin_amdgcn_sched_barrier(0); }
        MS_RGROUP(2, false)
        if (SCALE_IN_CHAIN) { __builtin_amdgcn_sched_barrier(0); scale_group(prev, t - 1, 2); scale_group(prev, t - 1, 3); __builtin_amdgcn_sched_barrier(0); }
        MS_RGROUP(3, false)
        asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(prev[0]), "v"(prev[1]), "v"(prev[2]));
        MS_MAX3(mx, prev[3], prev[4]); MS_MAX3(mx, prev[5], prev[6]); MS_MAX3(mx, prev[7], prev[8]);
        MS_RGROUP(4, false)
        MS_MAX3(mx, prev[9], prev[10]); MS_MAX3(mx, prev[11], prev[12]); MS_MAX3(mx, prev[13], prev[14]);
        asm volatile("v_max_f32 %0, %0, %1" : "+v"(mx) : "v"(prev[15]));
        MS_RGROUP(5, false)
        MS_RGROUP(6, false)
        MS_RGROUP(7, false)
        MS_RGROUP(8, false)
        MS_RGROUP(9, false)
        MS_RGROUP(10, false)
        MS_RGROUP(11, false)
        MS_RGROUP(12, false)
        if (!FIRST) {       // the last fragments of tiles t-1 and t have been requested: both slots may be refilled; then the loader's
                            // counter for the next pair (both behind the fragment reads in the LDS queue)
            asm volatile("ds_add_u32 %0, %1" ::"v"(cons_addr), "v"(two) : "memory");
            asm volatile("ds_read_b32 %0, %1" : "=v"(flag) : "v"(landed_addr) : "memory");
        }
        MS_RGROUP(13, false)
        MS_RGROUP(14, false)
        MS_RGROUP(15, false)
#undef MS_RGROUP
#undef MS_RREAD
#undef MS_RWAIT
        __builtin_amdgcn_sched_barrier(0);
        if (__builtin_expect(__ballot(mx > st.tau) != 0 || (AUXM == 2 && neg_tau), 0)) {
#ifdef MS_STAMP
            const unsigned long long i0 = __builtin_amdgcn_s_memtime();
#endif
            // (as in `stage`: the prefetched fragments of the next tile must have landed before compiler-scheduled code runs)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(areg[0]), "+v"(areg[1]), "+v"(areg[2]), "+v"(areg[3]), "+v"(flag) :: "memory");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(hc0), "+v"(hc1), "+v"(hc2), "+v"(hc3) :: "memory");
            rare(prev, t - 1);
#ifdef MS_STAMP
            stamp_ins += __builtin_amdgcn_s_memtime() - i0;
            stamp_nins += 1;
#endif
        }
    };
    // PF stage (compiler-scheduled): tile t is read from its slot during its own stage -- two fragments per k block, split,
    // three bf16 matrix instructions -- 24 of them per tile instead of 64 fp32 ones at a quarter of the cycles each; the
    // filter, the lists and the synchronisation with the loader are those of stage_ring.
    auto stage_pf = [&](auto first_c, int t, f32x16 &prev, f32x16 &out) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value;
        const char *slot = smem + (size_t)(t % LDR_R) * 16384 + 8192 * h + 16 * r;
        f32x16 acc;
#ifdef MS_STAMP
        const unsigned long long pf_t0 = __builtin_amdgcn_s_memtime();
#endif
        if constexpr (KL <= 16) {
            // all 16 fragments requested up front (their LDS latency overlaps the first blocks), two independent chains (even /
            // odd k blocks) so that a matrix instruction never waits for the previous one's result
            f32x4 fr[16];
#pragma unroll
            for (int f = 0; f < 16; ++f) fr[f] = *reinterpret_cast<const f32x4 *>(slot + 512 * f);
            if constexpr (KL <= 10) __builtin_amdgcn_sched_barrier(0);      // (hipcc would otherwise request four at a time and wait for each
                                                                            //  batch; with 16-entry lists the registers do not allow it)
            f32x16 acc_a, acc_b;
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc_a[i] = 0.0f; acc_b[i] = 0.0f; }
#pragma unroll
            for (int b = 0; b < 8; b += 2) {
                bf16x8 ah, al, bh2, bl2;
                ms_split8(fr[2 * b], fr[2 * b + 1], ah, al);
                ms_split8(fr[2 * b + 2], fr[2 * b + 3], bh2, bl2);
                acc_a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qhi[b], acc_a, 0, 0, 0);
                acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh2, qhi[b + 1], acc_b, 0, 0, 0);
                acc_a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qlo[b], acc_a, 0, 0, 0);
                acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh2, qlo[b + 1], acc_b, 0, 0, 0);
                acc_a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, qhi[b], acc_a, 0, 0, 0);
                acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl2, qhi[b + 1], acc_b, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = acc_a[i] + acc_b[i];
        } else {
            // 32-entry lists: no room for a tile of fragments; two k blocks at a time.  The SAME arithmetic, instruction for
            // instruction, as above (the sample pass runs the branch above: its bound must hold for these scores bit for bit)
            f32x16 acc_a, acc_b;
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc_a[i] = 0.0f; acc_b[i] = 0.0f; }
#pragma unroll
            for (int b = 0; b < 8; b += 2) {
                const f32x4 x0 = *reinterpret_cast<const f32x4 *>(slot + 1024 * b), x1 = *reinterpret_cast<const f32x4 *>(slot + 1024 * b + 512);
                const f32x4 x2 = *reinterpret_cast<const f32x4 *>(slot + 1024 * b + 1024), x3 = *reinterpret_cast<const f32x4 *>(slot + 1024 * b + 1536);
                bf16x8 ah, al, bh2, bl2;
                ms_split8(x0, x1, ah, al);
                ms_split8(x2, x3, bh2, bl2);
                acc_a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qhi[b], acc_a, 0, 0, 0);
                acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh2, qhi[b + 1], acc_b, 0, 0, 0);
                acc_a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qlo[b], acc_a, 0, 0, 0);
                acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh2, qlo[b + 1], acc_b, 0, 0, 0);
                acc_a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, qhi[b], acc_a, 0, 0, 0);
                acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl2, qhi[b + 1], acc_b, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = acc_a[i] + acc_b[i];
        }
        out = acc;
#ifdef MS_STAMP
        asm volatile("s_nop 0" : "+v"(out));
        stamp_ins += __builtin_amdgcn_s_memtime() - pf_t0;       // (diagnostics: cycles of the score computation of this stage)
        stamp_nins += 1;
#endif
        if (!FIRST) {       // tiles t-1 and t have been read (LDS operations execute in order); the loader's counter for the next pair.
            // The read is a plain volatile one here: in compiler-scheduled code an asm statement whose destination register is
            // filled later, behind the compiler's back, is a bug waiting for the register allocator (the data of such a read
            // landed in a register that had been given to the next tile's base address by then).
            asm volatile("ds_add_u32 %0, %1" ::"v"(cons_addr), "v"(two) : "memory");
            flag = landed[0];
        }
        if (SAMPLE) {
#pragma unroll
            for (int i = 0; i < 16; ++i) smax = (prev[i] > smax) ? prev[i] : smax;       // (NaN scores never enter)
        } else {
            float mx = prev[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) mx = fmaxf(mx, prev[i]);
            if (__builtin_expect(__ballot(mx > st.tau) != 0, 0)) {
                float sc[16];
                uint64_t m[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) { sc[i] = prev[i]; m[i] = __ballot(sc[i] > st.tau); }
                ms_tile_insert<SAMPLE ? 1 : KL, true>(st, sc, m, row_begin + (int64_t)(t - 1) * 32, r, h, &hg);
            }
        }
    };
    auto run_stage = [&](auto first_c, int t, f32x16 &prev, f32x16 &out) __attribute__((always_inline)) {
        if constexpr (PF) stage_pf(first_c, t, prev, out);
        else if constexpr (RING_FRAGS) stage_ring(first_c, t, prev, out);
        else stage(first_c, t, prev, out);
    };

#ifdef MS_STAMP
    const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long stamp_cm = stamp_c0, stamp_rm = stamp_r0;
#endif
    if (ntl > 0) {
        f32x16 acc0, acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc0[i] = -INFINITY; acc1[i] = -INFINITY; }
        landed_seen = wait_landed(1);
        if (PF) {
            // (every stage reads its own tile)
        } else if (RING_FRAGS) {
            MS_FRAG_READ(areg[0], lin0, 0); MS_FRAG_READ(areg[1], lin0, 512); MS_FRAG_READ(areg[2], lin0, 1024);
        } else {
#pragma unroll
            for (int f = 0; f < 16; ++f) areg[f] = *reinterpret_cast<const f32x4 *>(smem + (lin0 - ring_lds) + 512 * f);
        }
        // every tile, the partial last one included, goes through the pipeline; its rows past
        // row_end are rejected by the filter of the last stage / the drain below
        int t = 0;
        for (; t + 1 < ntl; t += 2) {
#ifdef MS_STAMP
            if (t == (ntl / 4) * 2) { stamp_cm = __builtin_amdgcn_s_memtime(); stamp_rm = __builtin_amdgcn_s_memrealtime(); }
#endif
            // shared bound (ScanHist): every 16th tile this query's 16 bucket counters are fetched (sc1: past this CU's L1)
            // while two tiles are multiplied, then the threshold is raised.  The two tests are evaluated separately on purpose
            // (kept apart by the empty asm): carried from one to the other, hipcc keeps the flag in a vector register.
            // (asm: four loads, asynchronous, no wait here.  The insertion path -- compiler-scheduled code -- may run while they are
            //  in flight, and a register the hardware fills behind the compiler's back is not safe there: it starts with a
            //  wait for them, see `stage`.  The prefilter's stage is compiler-scheduled throughout and uses atomic loads.)
            if (hist_on && (t & (MS_HIST_PERIOD - 1)) == MS_HIST_PERIOD / 2) {
                const uint32_t *hp = hg.counters != nullptr ? hg.counters : p.hist;
                if (PF) {
                    uint32_t c_[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) c_[j] = __hip_atomic_load(hp + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    hc0 = u32x4{c_[0], c_[1], c_[2], c_[3]}; hc1 = u32x4{c_[4], c_[5], c_[6], c_[7]};
                    hc2 = u32x4{c_[8], c_[9], c_[10], c_[11]}; hc3 = u32x4{c_[12], c_[13], c_[14], c_[15]};
                } else {
                    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                                 "global_load_dwordx4 %2, %4, off offset:32 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:48 sc1"
                                 : "=&v"(hc0), "=&v"(hc1), "=&v"(hc2), "=&v"(hc3) : "v"(hp) : "memory");
                }
            }
            ensure_landed(t);
            run_stage(std::true_type{}, t, acc0, acc1);
            run_stage(std::false_type{}, t + 1, acc1, acc0);
            int t2 = t;
            asm volatile("" : "+s"(t2));
            if (hist_on && (t2 & (MS_HIST_PERIOD - 1)) == MS_HIST_PERIOD / 2) {
                // the highest bucket edge with at least k rows at or above it (counted by all waves so far) bounds the k-th best
                if (!PF) {     // (compiler-scheduled arithmetic follows: the counters and the stage's last fragment reads must have landed)
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(hc0), "+v"(hc1), "+v"(hc2), "+v"(hc3) :: "memory");
                    if constexpr (RING_FRAGS)
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(areg[0]), "+v"(areg[1]), "+v"(areg[2]), "+v"(areg[3]), "+v"(flag) :: "memory");
                    else
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(areg[0]), "+v"(areg[1]), "+v"(areg[2]), "+v"(areg[3]), "+v"(areg[4]), "+v"(areg[5]),
                                     "+v"(areg[6]), "+v"(areg[7]), "+v"(areg[8]), "+v"(areg[9]), "+v"(areg[10]), "+v"(areg[11]), "+v"(areg[12]),
                                     "+v"(areg[13]), "+v"(areg[14]), "+v"(areg[15]), "+v"(flag) :: "memory");
                }
                const uint32_t c[16] = {hc0.x, hc0.y, hc0.z, hc0.w, hc1.x, hc1.y, hc1.z, hc1.w, hc2.x, hc2.y, hc2.z, hc2.w, hc3.x, hc3.y, hc3.z, hc3.w};
                uint32_t cum = 0;
                int n_lt = 0;
#pragma unroll
                for (int j = 15; j >= 0; --j) { cum += c[j]; n_lt += (cum < (uint32_t)p.k) ? 1 : 0; }
                const int J = 15 - n_lt;
                if (hg.counters != nullptr && J >= 1) {
                    st.floor = fmaxf(st.floor, ms_next_below(ms_hist_edge(hg, J)));
                    st.tau = fmaxf(st.tau, st.floor);
                }
                if (AUXM == 2) neg_tau = __ballot(st.tau < 0.0f) != 0;
            }
        }
        if (t < ntl) {
            ensure_landed(t);
            run_stage(std::true_type{}, t, acc0, acc1);
            acc0 = acc1;
        }
        // the last chain's result is read by compiler-scheduled code next: wait out the matrix pipe (the hazard
        // recognizer does not see inside the asm statements)
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt lgkmcnt(0)" : "+v"(acc0) :: "memory");
        if (lane == 0) consumed[wave] = 0xFFFFFFFFu;
        if (SAMPLE) {
            if (AUX) { scale_group(acc0, ntl - 1, 0); scale_group(acc0, ntl - 1, 1); scale_group(acc0, ntl - 1, 2); scale_group(acc0, ntl - 1, 3); }
#pragma unroll
            for (int i = 0; i < 16; ++i) smax = (acc0[i] > smax) ? acc0[i] : smax;       // (NaN scores never enter)
        } else {
            if constexpr (APPEND) flush();      // (the last tile's rows come after every buffered one)
            float sc[16];
            uint64_t m[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) filter_group(acc0, ntl - 1, g, true, sc, m);
            uint64_t any = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) any |= m[i];
            if (any != 0) ms_tile_insert<SAMPLE ? 1 : KL, true>(st, sc, m, row_begin + (int64_t)(ntl - 1) * 32, r, h, &hg);
        }
    }
#ifdef MS_STAMP
    if (!SAMPLE && lane == 0 && p.stamps != nullptr && (size_t)bid * 64 + 64 <= 4 * 4 * 65536) {
        unsigned long long *o = p.stamps + ((size_t)bid * 8 + wave) * 8;
        o[0] = __builtin_amdgcn_s_memtime() - stamp_c0;
        o[1] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
        o[6] = stamp_cm - stamp_c0; o[7] = stamp_rm - stamp_r0;      // first half of the stream (cycles, 100 MHz ticks)
#ifdef MS_STAMP_FLUSH
        o[6] = stamp_flush; o[7] = (stamp_nflush << 32) | stamp_rounds;
#endif
        o[2] = (unsigned long long)ntl;
        o[3] = (stamp_nwait << 40) | stamp_wait;
        o[4] = stamp_ins; o[5] = stamp_nins;
    }
#endif
    if (SAMPLE) {
        // the stream's entry: the two half-tile maxima of the lane pair (scores of distinct rows), larger first; rows are
        // not recorded (the bound selection reads values only), a distinct placeholder keeps the slots "occupied"
        const float other = ms_xor32_f(smax, h);
        if (h == 0) {
            const float hi = (other > smax) ? other : smax, lo = (other > smax) ? smax : other;
            const size_t o = p.list_sm ? ((size_t)stream * p.nq_pad + qidx) * p.k : ((size_t)qidx * p.k + 0) * p.P + stream;
            const size_t o1 = p.list_sm ? o + 1 : o + p.P;
            p.part_s[o] = hi;
            p.part_i[o] = (hi > -INFINITY) ? (uint32_t)(2 * stream) : MS_IDX_NONE;
            if (p.k > 1) {
                p.part_s[o1] = lo;
                p.part_i[o1] = (lo > -INFINITY) ? (uint32_t)(2 * stream + 1) : MS_IDX_NONE;
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < (SAMPLE ? 1 : KL); ++j) {
        const int rank = h * KL + j;
        if (rank < p.k) {
            // (stream-major when the merge behind this launch reads it: a workgroup's lists are then one contiguous block that the L2
            //  writes back as whole lines -- rank-major, every 4-byte entry shares its line with the other streams' workgroups on other
            //  XCDs: 26.5 MB of HBM writes for a 2.6 MB payload at C2; ms_scan_pf16.h)
            const size_t o = p.list_sm ? ((size_t)stream * p.nq_pad + qidx) * p.k + rank : ((size_t)qidx * p.k + rank) * p.P + stream;
            p.part_s[o] = st.ls[j];
            p.part_i[o] = st.li[j];
        }
    }
}

// The full scan and the sample pass run the same body; two symbols so that profiles tell them apart.
template <int KL, bool AUX, bool UB>
__global__ __launch_bounds__(256, 1) void ms_scan_kernel(const ScanParams p) { ms_scan_body<KL, AUX, UB, false>(p); }
template <int KL, bool AUX>
__global__ __launch_bounds__(256, 1) void ms_scan_sample_kernel(const ScanParams p) { ms_scan_body<KL, AUX, false, true>(p); }

// ------------------------------------------------------------------ launch plan + launch templates
struct ScanPlan {
    int n_qtiles, qwb, n_qgroups, nq_pad, nq_real;
    int k_pass;            // ranks per pass (<= 64)
    int kl;                // list entries per lane: smallest of {5,10,32} with 2*kl >= k_pass
    int rows_per_stream, n_streams, n_sgroups, P;
    int grid;
    int prepass_tiles;     // tiles per stream scanned by the sample pass (0 = no sample pass)
    int qpw;               // split-image prefilter scan (ms_scan_pf.h): query tiles per wave (0: any other kernel)
    int list_sm;           // the scan of this plan writes stream-major lists (the image scans; the loader-wave kernel when P <= 256 and the block merge takes them)
    size_t lds_bytes;
    // workspace carve (byte offsets)
    size_t off_qn, off_inv, off_part_s, off_part_i, off_ub_s, off_ub_i, off_lb_s, off_lb_i, off_scr_s, off_scr_i, off_hist, off_hstep, off_prog, total;
};

inline int loader_wave_setting() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("MS_LOADER_WAVE"); v = e ? atoi(e) : 1; }
    return v;
}

template <int KL, bool AUX, bool UB>
int launch_scan_variant(const ScanPlan &pl, const ScanParams &sp, hipStream_t st) {
    if (!UB && sp.max_tiles > 0) {      // sample pass
        MS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ms_scan_sample_kernel<KL, AUX>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes));
        hipLaunchKernelGGL((ms_scan_sample_kernel<KL, AUX>), dim3(pl.grid), dim3(256), pl.lds_bytes, st, sp);
        MS_LAUNCH_CHECK("ms_scan_sample_kernel");
        return MS_OK;
    }
    if constexpr (!UB) {                // loader-wave form: its compute waves must fit 256 registers WITHOUT spills (the pinned stage
                                        // cannot tolerate a spill of a register an LDS read is still filling): 32-entry lists get there
                                        // with four fragment registers instead of a tile's sixteen (RING_FRAGS)
        if constexpr (!AUX) {
            if (sp.qwb == 4 && sp.prefilter) {          // the prefilter's scan (ms_ip_topk_prefiltered)
                MS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ms_scan_loader_kernel<KL, 0, false, true>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDR_LDS));
                hipLaunchKernelGGL((ms_scan_loader_kernel<KL, 0, false, true>), dim3(pl.grid), dim3(320), LDR_LDS, st, sp);
                MS_LAUNCH_CHECK("ms_scan_loader_kernel (prefilter)");
                return MS_OK;
            }
        }
        if (sp.qwb == 4 && loader_wave_setting()) {     // MFMA-bound batches
#define MS_LAUNCH_LOADER(AUXM)                                                                                           \
            constexpr int lds_ = KL >= MS_APPEND_MIN_KL ? LDR_LDS_APPEND : LDR_LDS;       /* (the candidate buffers of the append-and-flush rare path) */ \
            MS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ms_scan_loader_kernel<KL, AUXM, false>),     \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, lds_));                         \
            hipLaunchKernelGGL((ms_scan_loader_kernel<KL, AUXM, false>), dim3(pl.grid), dim3(320), lds_, st, sp);
            if constexpr (!AUX) { MS_LAUNCH_LOADER(0) }
            else if (sp.unit_rows) { MS_LAUNCH_LOADER(2) }
            else { MS_LAUNCH_LOADER(1) }
#undef MS_LAUNCH_LOADER
            MS_LAUNCH_CHECK("ms_scan_loader_kernel");
            return MS_OK;
        }
    }
    MS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ms_scan_kernel<KL, AUX, UB>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes));
    hipLaunchKernelGGL((ms_scan_kernel<KL, AUX, UB>), dim3(pl.grid), dim3(256), pl.lds_bytes, st, sp);
    MS_LAUNCH_CHECK("ms_scan_kernel");
    return MS_OK;
}

// Sample pass in the loader-wave form (qwb == 4), values only: one instantiation per mode, whatever the list length.
template <int AUXM>
int launch_sample_loader_variant(const ScanPlan &pl, const ScanParams &sp, hipStream_t st) {
    if constexpr (AUXM == 0) {
        if (sp.prefilter) {
            MS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ms_scan_loader_kernel<5, 0, true, true>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDR_LDS));
            hipLaunchKernelGGL((ms_scan_loader_kernel<5, 0, true, true>), dim3(pl.grid), dim3(320), LDR_LDS, st, sp);
            MS_LAUNCH_CHECK("ms_scan_loader_kernel (prefilter sample)");
            return MS_OK;
        }
    }
    MS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ms_scan_loader_kernel<5, AUXM, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDR_LDS));
    hipLaunchKernelGGL((ms_scan_loader_kernel<5, AUXM, true>), dim3(pl.grid), dim3(320), LDR_LDS, st, sp);
    MS_LAUNCH_CHECK("ms_scan_loader_kernel (sample)");
    return MS_OK;
}

template <int KL, bool UB>
int launch_scan_kl(const ScanPlan &pl, const ScanParams &sp, hipStream_t st) {
    const bool aux = sp.inv_norm != nullptr || sp.lengths != nullptr;
    return aux ? launch_scan_variant<KL, true, UB>(pl, sp, st) : launch_scan_variant<KL, false, UB>(pl, sp, st);
}

// One non-template entry point per list length (defined in ms_scan_kl*.hip).
int ms_launch_scan_kl5(const ScanPlan &pl, const ScanParams &sp, hipStream_t st);
int ms_launch_scan_kl10(const ScanPlan &pl, const ScanParams &sp, hipStream_t st);
int ms_launch_scan_kl16(const ScanPlan &pl, const ScanParams &sp, hipStream_t st);
int ms_launch_scan_kl32(const ScanPlan &pl, const ScanParams &sp, hipStream_t st);
int ms_launch_scan_kl32ub(const ScanPlan &pl, const ScanParams &sp, hipStream_t st);
int ms_launch_sample_loader(const ScanPlan &pl, const ScanParams &sp, hipStream_t st);     // ms_scan_kl5.hip
