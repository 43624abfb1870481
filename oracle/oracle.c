/*
 * oracle.c -- CPU restatement of the Foldclass embed-and-search hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP kernels in
 * merizo_search_amd/csrc.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product path never calls it and has no CPU fallback.
 *
 * Every function restates, in plain C / fp32, what the reference computes.  Citations are
 * file:line under /root/reference/merizo_search/programs/Foldclass/ unless stated otherwise.
 *
 * Parity status: PINNED for the EGNN encoder, F.normalize and the `.pt` cosine+mask+topk
 * path -- checked against golden vectors produced by importing the reference in the build
 * container (oracle/gen_golden.py -> tests/golden/, tests/test_oracle_golden.py).
 * The faiss path (knn_exact_faiss, dbsearch.py:213-248) is restated from its call site
 * because faiss (un-pinned upstream: README.md:16-18) is not installed anywhere here;
 * at that boundary parity is UNPINNED against faiss itself and instead cross-checked
 * against the pinned torch path on pre-normalised data (tests/test_oracle_golden.py).
 *
 * Tie policy (torch.topk / faiss heaps leave it implementation-defined): score descending,
 * then index ascending; -0.0 == +0.0.
 */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_DIM 128          /* FoldClassNet(128): dbsearch.py:40 */
#define ORC_MDIM 256         /* m_dim = width*2: nndef_fold_egnn_embed.py:46 */
#define ORC_EIN 257          /* edge_input_dim = 2*dim + 1: my_egnn_nocoords.py:14 */
#define ORC_EHID 514         /* edge_input_dim * 2: my_egnn_nocoords.py:19 */
#define ORC_NIN 384          /* dim + m_dim: my_egnn_nocoords.py:31 */
#define ORC_NHID 256         /* dim * 2: my_egnn_nocoords.py:31 */

/* Canonical per-layer weight blob = the reference state_dict tensors of one EGNN layer,
 * flattened row-major in state_dict order:
 *   edge_mlp.0.weight [514,257], .bias [514], edge_mlp.2.weight [256,514], .bias [256],
 *   edge_gate.0.weight [1,256], .bias [1], node_mlp.0.weight [256,384], .bias [256],
 *   node_mlp.2.weight [128,256], .bias [128]                                          */
#define OFF_W1 0
#define OFF_B1 (OFF_W1 + ORC_EHID * ORC_EIN)
#define OFF_W2 (OFF_B1 + ORC_EHID)
#define OFF_B2 (OFF_W2 + ORC_MDIM * ORC_EHID)
#define OFF_WG (OFF_B2 + ORC_MDIM)
#define OFF_BG (OFF_WG + ORC_MDIM)
#define OFF_WN1 (OFF_BG + 1)
#define OFF_BN1 (OFF_WN1 + ORC_NHID * ORC_NIN)
#define OFF_WN2 (OFF_BN1 + ORC_NHID)
#define OFF_BN2 (OFF_WN2 + ORC_DIM * ORC_NHID)
#define ORC_LAYER_FLOATS (OFF_BN2 + ORC_DIM) /* 396165; x2 layers = 792330 parameters */

int orc_layer_floats(void) { return ORC_LAYER_FLOATS; }

#ifdef _OPENMP
#include <omp.h>
int orc_num_threads(void) { return omp_get_max_threads(); }
#else
int orc_num_threads(void) { return 1; }
#endif

static inline float silu_f(float x) { return x / (1.0f + expf(-x)); }     /* nn.SiLU */
static inline float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); } /* nn.Sigmoid */

/* out[o] = b[o] + sum_k wT[k][o] * x[k]; per-output accumulation runs k = 0..K-1 in order.
 * wT is the [K,O] transpose of the nn.Linear weight so the o loop vectorises without
 * reassociating any sum. */
static void linear_T(const float *wT, const float *b, const float *x, int K, int O, float *out) {
    for (int o = 0; o < O; ++o) out[o] = b[o];
    for (int k = 0; k < K; ++k) {
        const float xk = x[k];
        const float *w = wT + (size_t)k * O;
        for (int o = 0; o < O; ++o) out[o] += w[o] * xk;
    }
}

static float *transpose_alloc(const float *w, int O, int K) { /* w[O][K] -> wT[K][O] */
    float *t = (float *)malloc(sizeof(float) * (size_t)O * K);
    if (!t) return NULL;
    for (int o = 0; o < O; ++o)
        for (int k = 0; k < K; ++k) t[(size_t)k * O + o] = w[(size_t)o * K + k];
    return t;
}

/* One EGNN layer, literal: my_egnn_nocoords.py:44-74.
 *   rel_coors / dist            :48-49  (sqrt of sum of squares, then dist*dist at :58)
 *   edge_input = [h_i, h_j, d2] :51-58  (all j including j == i; no radius cut-off)
 *   m_ij = edge_mlp(edge_input) :63     Linear(257,514) SiLU Linear(514,256) SiLU  (:18-23)
 *   m_ij *= edge_gate(m_ij)     :64     Linear(256,1) Sigmoid                      (:25-28)
 *   m_i = sum_j m_ij            :69
 *   out = node_mlp([h, m_i]) + h :71-72 Linear(384,256) SiLU Linear(256,128)       (:30-34) */
int orc_egnn_layer(const float *wl, const float *h, const float *coords, int n, float *h_out) {
    float *w1T = transpose_alloc(wl + OFF_W1, ORC_EHID, ORC_EIN);
    float *w2T = transpose_alloc(wl + OFF_W2, ORC_MDIM, ORC_EHID);
    float *wn1T = transpose_alloc(wl + OFF_WN1, ORC_NHID, ORC_NIN);
    float *wn2T = transpose_alloc(wl + OFF_WN2, ORC_DIM, ORC_NHID);
    if (!w1T || !w2T || !wn1T || !wn2T) return -1;
    const float *b1 = wl + OFF_B1, *b2 = wl + OFF_B2, *wg = wl + OFF_WG, *bn1 = wl + OFF_BN1, *bn2 = wl + OFF_BN2;
    const float bg = wl[OFF_BG];
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < n; ++i) {
        float x[ORC_EIN], hid[ORC_EHID], m[ORC_MDIM], mi[ORC_MDIM], nin[ORC_NIN], nh[ORC_NHID], o[ORC_DIM];
        for (int c = 0; c < ORC_MDIM; ++c) mi[c] = 0.0f;
        memcpy(x, h + (size_t)i * ORC_DIM, sizeof(float) * ORC_DIM);
        for (int j = 0; j < n; ++j) {
            memcpy(x + ORC_DIM, h + (size_t)j * ORC_DIM, sizeof(float) * ORC_DIM);
            const float dx = coords[3 * i] - coords[3 * j], dy = coords[3 * i + 1] - coords[3 * j + 1],
                        dz = coords[3 * i + 2] - coords[3 * j + 2];
            const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
            x[2 * ORC_DIM] = dist * dist;
            linear_T(w1T, b1, x, ORC_EIN, ORC_EHID, hid);
            for (int c = 0; c < ORC_EHID; ++c) hid[c] = silu_f(hid[c]);
            linear_T(w2T, b2, hid, ORC_EHID, ORC_MDIM, m);
            float g = bg;
            for (int c = 0; c < ORC_MDIM; ++c) {
                m[c] = silu_f(m[c]);
                g += wg[c] * m[c];
            }
            g = sigmoid_f(g);
            for (int c = 0; c < ORC_MDIM; ++c) mi[c] += m[c] * g;
        }
        memcpy(nin, h + (size_t)i * ORC_DIM, sizeof(float) * ORC_DIM);
        memcpy(nin + ORC_DIM, mi, sizeof(float) * ORC_MDIM);
        linear_T(wn1T, bn1, nin, ORC_NIN, ORC_NHID, nh);
        for (int c = 0; c < ORC_NHID; ++c) nh[c] = silu_f(nh[c]);
        linear_T(wn2T, bn2, nh, ORC_NHID, ORC_DIM, o);
        for (int c = 0; c < ORC_DIM; ++c) h_out[(size_t)i * ORC_DIM + c] = o[c] + h[(size_t)i * ORC_DIM + c];
    }
    free(w1T); free(w2T); free(wn1T); free(wn2T);
    return 0;
}

/* FoldClassNet.forward over a ragged batch: nndef_fold_egnn_embed.py:50-62.
 *   seq_feats = pe[:, :nres, :]  :54 (PositionalEncoder.forward :27-30; the coordinates'
 *                                    values are NOT features, only their count is used)
 *   2 x EGNN                     :57
 *   embed = mean(dim=1)          :61
 * weights: 2 consecutive layer blobs; pe: the [max_len,128] table as DATA.
 * layer_out (optional) receives the per-layer node features [2][sumN][128].
 * Returns -2 if a structure is longer than the table (the reference would fail the
 * broadcast at my_egnn_nocoords.py:51-53 for N > 3000). */
int orc_egnn_embed(const float *weights, const float *pe, int max_len, const float *coords, const int *offsets,
                   int nb, float *out, float *layer_out) {
    const int total = offsets[nb];
    for (int b = 0; b < nb; ++b) {
        const int n = offsets[b + 1] - offsets[b];
        if (n <= 0 || n > max_len) return -2;
        float *h0 = (float *)malloc(sizeof(float) * (size_t)n * ORC_DIM);
        float *h1 = (float *)malloc(sizeof(float) * (size_t)n * ORC_DIM);
        if (!h0 || !h1) return -1;
        memcpy(h0, pe, sizeof(float) * (size_t)n * ORC_DIM);
        for (int l = 0; l < 2; ++l) {
            int rc = orc_egnn_layer(weights + (size_t)l * ORC_LAYER_FLOATS, h0, coords + 3 * (size_t)offsets[b], n, h1);
            if (rc) return rc;
            if (layer_out)
                memcpy(layer_out + ((size_t)l * total + offsets[b]) * ORC_DIM, h1, sizeof(float) * (size_t)n * ORC_DIM);
            float *t = h0; h0 = h1; h1 = t;
        }
        for (int c = 0; c < ORC_DIM; ++c) { /* mean over residues, in residue order */
            float s = 0.0f;
            for (int i = 0; i < n; ++i) s += h0[(size_t)i * ORC_DIM + c];
            out[(size_t)b * ORC_DIM + c] = s / (float)n;
        }
        free(h0); free(h1);
    }
    return 0;
}

/* F.normalize(x) (dbsearch.py:303-304, eps 1e-12) and the per-operand normalisation inside
 * F.cosine_similarity (dbsearch.py:78, eps 1e-8): x / max(||x||_2, eps), in place. */
int orc_l2_normalize_rows(float *x, int64_t n, int d, float eps) {
#pragma omp parallel for
    for (int64_t r = 0; r < n; ++r) {
        float *p = x + r * d;
        float ss = 0.0f;
        for (int c = 0; c < d; ++c) ss += p[c] * p[c];
        float nrm = sqrtf(ss);
        if (nrm < eps) nrm = eps;
        for (int c = 0; c < d; ++c) p[c] = p[c] / nrm;
    }
    return 0;
}

/* ---- top-k under the total order (score desc, index asc); -0.0 == +0.0 ------------- */
static inline int better(float sa, int64_t ia, float sb, int64_t ib) {
    return (sa > sb) || (sa == sb && ia < ib);
}

/* Insert (s,i) into a list sorted best-first holding `cnt` valid entries (cnt <= k). */
static inline int topk_insert(float *ls, int64_t *li, int cnt, int k, float s, int64_t i) {
    if (cnt == k && !better(s, i, ls[k - 1], li[k - 1])) return cnt;
    int p = (cnt < k) ? cnt : k - 1;
    while (p > 0 && better(s, i, ls[p - 1], li[p - 1])) {
        ls[p] = ls[p - 1];
        li[p] = li[p - 1];
        --p;
    }
    ls[p] = s;
    li[p] = i;
    return cnt < k ? cnt + 1 : k;
}

/* Inner product of two 128-vectors as a k-ordered fmaf chain.
 * order 0: k = 0,1,...,d-1.
 * order 1: the order of the gfx950 scan kernel (k = s, then half + s for s = 0..half-1):
 *          with it the HIP scores are reproduced bit-for-bit (scan kernel).
 * faiss/BLAS leave the summation order unspecified, so any fixed order restates a8. */
static inline float dot_ordered(const float *a, const float *b, int d, int order) {
    float acc = 0.0f;
    if (order == 1) {
        const int half = d / 2;
        for (int s = 0; s < half; ++s) {
            acc = fmaf(a[s], b[s], acc);
            acc = fmaf(a[half + s], b[half + s], acc);
        }
    } else {
        for (int c = 0; c < d; ++c) acc = fmaf(a[c], b[c], acc);
    }
    return acc;
}

/* search_query_against_db, dbsearch.py:75-81, for nq queries against a RAW `.pt` database:
 *   mask   = (len(q_seq) >= lengths * mincov).float()            :76
 *   scores = F.cosine_similarity(db, q, dim=-1) * mask           :78
 *            (both operands normalised first, x / max(||x||, 1e-8), then multiply-sum)
 *   topk(scores, k)                                              :79
 * Masked rows score (+-)0.0, they are NOT removed.  lengths/qlen may be NULL (no mask).
 * Rows are numbered row_offset + r.  Returns -3 if k > n (torch.topk raises). */
int orc_cosine_topk(const float *db, int64_t n, int d, int64_t row_offset, const float *q, int nq, int k,
                    const float *lengths, const float *qlen, float mincov, float *out_s, int64_t *out_i) {
    if (k > n) return -3;
    float *dbn = (float *)malloc(sizeof(float) * (size_t)n * d);
    float *qn = (float *)malloc(sizeof(float) * (size_t)nq * d);
    if (!dbn || !qn) return -1;
    memcpy(dbn, db, sizeof(float) * (size_t)n * d);
    memcpy(qn, q, sizeof(float) * (size_t)nq * d);
    orc_l2_normalize_rows(dbn, n, d, 1e-8f);
    orc_l2_normalize_rows(qn, nq, d, 1e-8f);
#pragma omp parallel for schedule(dynamic, 1)
    for (int qi = 0; qi < nq; ++qi) {
        float *ls = out_s + (size_t)qi * k;
        int64_t *li = out_i + (size_t)qi * k;
        int cnt = 0;
        for (int64_t r = 0; r < n; ++r) {
            float s = 0.0f;
            const float *a = dbn + r * d, *b = qn + (size_t)qi * d;
            for (int c = 0; c < d; ++c) s += a[c] * b[c];
            if (lengths) {
                const float mask = (qlen[qi] >= lengths[r] * mincov) ? 1.0f : 0.0f;
                s = s * mask;
            }
            cnt = topk_insert(ls, li, cnt, k, s, row_offset + r);
        }
    }
    free(dbn); free(qn);
    return 0;
}

/* One block of knn_exact_faiss, dbsearch.py:234-242: IndexFlat(d, METRIC_INNER_PRODUCT)
 * .add(block) / .search(xq, k) and `I += i0`.  xq and the block are used as given (the
 * reference passes F.normalize'd queries and a pre-normalised *_norm.db memmap).
 * If the block has fewer than k rows the tail is (-inf, -1) like faiss.
 *
 * Every (row, query) score is exactly dot_ordered(): one fmaf chain in the chosen k order.
 * For speed (this is also bench.py's cpu_baseline) ORC_QB queries are scored at a time with
 * the query block transposed: the 8 lanes of each AVX2 register are 8 QUERIES, and
 * _mm256_fmadd_ps is one IEEE fma per lane, so blocking never changes a score.  OpenMP
 * splits the query blocks over threads. */
#define ORC_QB 32
int orc_ip_topk(const float *db, int64_t n, int d, int64_t row_offset, const float *q, int nq, int k, int order,
                float *out_s, int64_t *out_i) {
    const int nblk = (nq + ORC_QB - 1) / ORC_QB;
    int *korder = (int *)malloc(sizeof(int) * (size_t)d);
    if (!korder) return -1;
    if (order == 1) {
        const int half = d / 2;
        for (int s = 0; s < half; ++s) { korder[2 * s] = s; korder[2 * s + 1] = half + s; }
    } else {
        for (int c = 0; c < d; ++c) korder[c] = c;
    }
    int fail = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < nblk; ++b) {
        const int q0 = b * ORC_QB, qb = (nq - q0 < ORC_QB) ? nq - q0 : ORC_QB;
        float *qT = (float *)aligned_alloc(64, sizeof(float) * (size_t)d * ORC_QB); /* [chain step][query] */
        int cnt[ORC_QB];
        float thr[ORC_QB];
        if (!qT) { fail = 1; continue; }
        for (int c = 0; c < d; ++c)
            for (int j = 0; j < ORC_QB; ++j)
                qT[(size_t)c * ORC_QB + j] = (j < qb) ? q[(size_t)(q0 + j) * d + korder[c]] : 0.0f;
        for (int j = 0; j < ORC_QB; ++j) { cnt[j] = 0; thr[j] = -INFINITY; }
        for (int64_t r = 0; r < n; ++r) {
            const float *row = db + r * d;
            float acc[ORC_QB] __attribute__((aligned(32)));
            __m256 a0 = _mm256_setzero_ps(), a1 = a0, a2 = a0, a3 = a0;
            for (int c = 0; c < d; ++c) {
                const __m256 a = _mm256_set1_ps(row[korder[c]]);
                const float *qq = qT + (size_t)c * ORC_QB;
                a0 = _mm256_fmadd_ps(a, _mm256_load_ps(qq), a0);
                a1 = _mm256_fmadd_ps(a, _mm256_load_ps(qq + 8), a1);
                a2 = _mm256_fmadd_ps(a, _mm256_load_ps(qq + 16), a2);
                a3 = _mm256_fmadd_ps(a, _mm256_load_ps(qq + 24), a3);
            }
            _mm256_store_ps(acc, a0); _mm256_store_ps(acc + 8, a1);
            _mm256_store_ps(acc + 16, a2); _mm256_store_ps(acc + 24, a3);
            for (int j = 0; j < qb; ++j) {
                /* rows arrive in ascending order, so a tie with the k-th best never wins */
                if (cnt[j] == k && !(acc[j] > thr[j])) continue;
                float *ls = out_s + (size_t)(q0 + j) * k;
                int64_t *li = out_i + (size_t)(q0 + j) * k;
                cnt[j] = topk_insert(ls, li, cnt[j], k, acc[j], row_offset + r);
                if (cnt[j] == k) thr[j] = ls[k - 1];
            }
        }
        for (int j = 0; j < qb; ++j) {
            float *ls = out_s + (size_t)(q0 + j) * k;
            int64_t *li = out_i + (size_t)(q0 + j) * k;
            for (int c = cnt[j]; c < k; ++c) { ls[c] = -INFINITY; li[c] = -1; }
        }
        free(qT);
    }
    free(korder);
    return fail ? -1 : 0;
}

/* faiss.ResultHeap(nq, k, keep_max=True).add_result(D, I) ... .finalize(), dbsearch.py:224,
 * 240,245: merge S per-block (or per-shard) result lists [S,nq,k] into the best k per query,
 * sorted best-first.  Entries with index < 0 are padding. */
int orc_topk_merge(const float *s, const int64_t *idx, int S, int nq, int k, float *out_s, int64_t *out_i) {
    for (int qi = 0; qi < nq; ++qi) {
        float *ls = out_s + (size_t)qi * k;
        int64_t *li = out_i + (size_t)qi * k;
        int cnt = 0;
        for (int b = 0; b < S; ++b)
            for (int e = 0; e < k; ++e) {
                const size_t o = ((size_t)b * nq + qi) * k + e;
                if (idx[o] < 0) continue;
                cnt = topk_insert(ls, li, cnt, k, s[o], idx[o]);
            }
        for (; cnt < k; ++cnt) { ls[cnt] = -INFINITY; li[cnt] = -1; }
    }
    return 0;
}

/* knn_exact_faiss, dbsearch.py:213-248, with db_iterator (dbutil.py:33-35): stream the
 * database in blocks of `block` rows, per-block top-k, running merge. */
int orc_knn_exact_blockwise(const float *db, int64_t n, int d, const float *q, int nq, int k, int64_t block, int order,
                            float *out_s, int64_t *out_i) {
    float *ps = (float *)malloc(sizeof(float) * (size_t)2 * nq * k);
    int64_t *pi = (int64_t *)malloc(sizeof(int64_t) * (size_t)2 * nq * k);
    if (!ps || !pi) return -1;
    for (size_t t = 0; t < (size_t)nq * k; ++t) { ps[t] = -INFINITY; pi[t] = -1; }
    for (int64_t i0 = 0; i0 < n; i0 += block) {
        const int64_t ni = (n - i0 < block) ? n - i0 : block;
        orc_ip_topk(db + i0 * d, ni, d, i0, q, nq, k, order, ps + (size_t)nq * k, pi + (size_t)nq * k);
        orc_topk_merge(ps, pi, 2, nq, k, out_s, out_i);
        memcpy(ps, out_s, sizeof(float) * (size_t)nq * k);
        memcpy(pi, out_i, sizeof(int64_t) * (size_t)nq * k);
    }
    if (n == 0) { memcpy(out_s, ps, sizeof(float) * (size_t)nq * k); memcpy(out_i, pi, sizeof(int64_t) * (size_t)nq * k); }
    free(ps); free(pi);
    return 0;
}
