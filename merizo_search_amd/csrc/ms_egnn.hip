// Foldclass structure encoder (2-layer EGNN, 128-d) on gfx950 for a ragged batch of CA traces.
//
// Reference arithmetic replaced (programs/Foldclass/):
//   nndef_fold_egnn_embed.py:50-62  FoldClassNet.forward: pe[:N] -> 2 x EGNN -> mean over residues
//   my_egnn_nocoords.py:44-74       EGNN.forward: all-pairs message / gate / aggregate / update
//
// The reference materialises edge_input [N,N,257] and the hidden layer [N,N,514].  Here the
// first edge Linear is split per node (my_egnn_nocoords.py:58,63 restructured):
//     W1 . [h_i, h_j, d2] + b1  =  (W1a . h_i + b1)  +  W1b . h_j  +  w_c * d2
// so only the second edge Linear (514 -> 256, the dominant 2*514*256 FLOP per edge) runs over
// the N^2 edges, as an fp32-MFMA GEMM whose A operand is produced on the fly:
//
//   ms_egnn_proj_kernel   Ap = h.W1a^T + b1, Bp = h.W1b^T           [nodes x 520], float4-transposed
//   ms_egnn_edge_kernel   per tile of 128 consecutive edges (i*N + j) of one structure:
//                           H = SiLU(Ap_i + Bp_j + w_c d2_ij)      VALU, staged through LDS
//                           M = SiLU(H . W2^T + b2)                32x32x2 fp32 MFMA, K = 520
//                           g = sigmoid(M . w_g + b_g); M *= g     in-register, per edge row
//                           per-residue partial sums of M over the tile's rows -> records
//   ms_egnn_node_kernel   m_i = sum of records; h' = Wn2.SiLU(Wn1.[h, m_i] + bn1) + bn2 + h
//   ms_egnn_pool_kernel   mean over residues
//
// No atomics: every sum has a fixed order, results are bit-reproducible run to run.
#include "ms_common.h"

#include <math.h>
#include <stdlib.h>

namespace {

constexpr int DIM = 128;
constexpr int MD = 256;          // m_dim
constexpr int EIN = 257;         // edge input width
constexpr int EH = 514;          // edge hidden width
constexpr int EHP = 520;         // padded to a multiple of 8 (zero weights in the pad)
constexpr int KQ = EHP / 4;      // 130 float4 quads over the hidden width
constexpr int NGRP = EHP / 8;    // 65 MFMA operand groups (8 k each: one float4 per lane half)
constexpr int STAGE_G = 5;       // groups per pipeline stage
constexpr int NSTAGE = NGRP / STAGE_G;   // 13
constexpr int NIN = 384;
constexpr int NHID = 256;
constexpr int TILE_E = 128;      // edges per workgroup tile (4 waves x 32 rows)
constexpr int LAYER_FLOATS = 396165;

// canonical blob offsets (state_dict order, see include/merizo_search_amd.h)
constexpr int C_W1 = 0;
constexpr int C_B1 = C_W1 + EH * EIN;
constexpr int C_W2 = C_B1 + EH;
constexpr int C_B2 = C_W2 + MD * EH;
constexpr int C_WG = C_B2 + MD;
constexpr int C_BG = C_WG + MD;
constexpr int C_WN1 = C_BG + 1;
constexpr int C_BN1 = C_WN1 + NHID * NIN;
constexpr int C_WN2 = C_BN1 + NHID;
constexpr int C_BN2 = C_WN2 + DIM * NHID;
static_assert(C_BN2 + DIM == LAYER_FLOATS, "blob layout");

// prepared (kernel-layout) per-layer offsets, in floats
constexpr int P_W1AT = 0;                          // [128 k][520 c]
constexpr int P_W1BT = P_W1AT + DIM * EHP;         // [128 k][520 c]
constexpr int P_B1 = P_W1BT + DIM * EHP;           // [520]
constexpr int P_WC = P_B1 + EHP;                   // [520]  distance column of W1
constexpr int P_W2F = P_WC + EHP;                  // [65 g][8 nt][64 lanes][4]  MFMA B fragments
constexpr int P_B2 = P_W2F + NGRP * 8 * 64 * 4;    // [256]
constexpr int P_WG = P_B2 + MD;                    // [256]
constexpr int P_BG = P_WG + MD;                    // [4] (1 used)
constexpr int P_WN1T = P_BG + 4;                   // [384 k][256 o]
constexpr int P_BN1 = P_WN1T + NIN * NHID;         // [256]
constexpr int P_WN2T = P_BN1 + NHID;               // [256 k][128 o]
constexpr int P_BN2 = P_WN2T + NHID * DIM;         // [128]
constexpr int P_W2S = P_BN2 + DIM;                 // [33 blocks][8 nt][3 parts][64 lanes][8 bf16]: W2 split into bf16 hi / mid / lo,
constexpr int KB16 = 33;                           //   in B-fragment order of v_mfma_f32_32x32x16_bf16 (K = 514 padded to 528)
constexpr int W2S_BLOCK_BYTES = 8 * 3 * 64 * 16;   // 24 KiB per k block of 16
constexpr int P_LAYER = P_W2S + KB16 * W2S_BLOCK_BYTES / 4;
static_assert(P_LAYER % 4 == 0 && P_W2F % 4 == 0 && P_WC % 4 == 0 && P_B1 % 4 == 0 && P_W2S % 4 == 0, "float4 alignment");

// x = hi + mid + lo EXACTLY, each part a bf16 (the upper 16 bits of what is left: 8 + 8 + 8 significant bits).  A product x * w
// over the parts has nine terms; hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid are kept -- what is dropped (mid.lo, lo.mid,
// lo.lo) is below 2^-23 |x w|, the size of one fp32 rounding.  bf16 x bf16 products are exact in the fp32 accumulator.
#ifndef MS_EGNN_SPLIT_PK
#define MS_EGNN_SPLIT_PK 0          // round 6: packed residual subtractions in split3_pair -- 16 fewer vector instructions per block pair and 1.2 % SLOWER (three
#endif                             // same-box passes: 57.7-57.8 against 57.0-57.1 ms per 1,000 domains, profiles/r06_egnn_split_pk_ab.log): not the default
typedef __bf16 bf16x8_e __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4_e __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split3_pair(float x0, float x1, uint32_t &hi, uint32_t &mid, uint32_t &lo) {
    const uint32_t a = __float_as_uint(x0), b = __float_as_uint(x1);
    hi = __builtin_amdgcn_perm(b, a, 0x07060302u);                  // upper halves of x1 : x0
#if MS_EGNN_SPLIT_PK
    // Round 6 (VERDICT r05 #5a), measured and NOT the default: both residual subtractions of the pair as ONE packed fp32 instruction each
    // (v_pk_add_f32 with the second operand negated) -- 9 vector instructions per pair instead of 11, the same bits (x - trunc(x) is exact
    // either way), but the packed add wants its operands in aligned register pairs and issues no faster than two scalar ones here
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ r = f32x2_{x0, x1} - f32x2_{__uint_as_float(a & 0xFFFF0000u), __uint_as_float(b & 0xFFFF0000u)};
    const uint32_t c = __float_as_uint(r.x), d = __float_as_uint(r.y);
    mid = __builtin_amdgcn_perm(d, c, 0x07060302u);
    const f32x2_ s = r - f32x2_{__uint_as_float(c & 0xFFFF0000u), __uint_as_float(d & 0xFFFF0000u)};
    lo = __builtin_amdgcn_perm(__float_as_uint(s.y), __float_as_uint(s.x), 0x07060302u);
#else
    const float r0 = x0 - __uint_as_float(a & 0xFFFF0000u), r1 = x1 - __uint_as_float(b & 0xFFFF0000u);
    const uint32_t c = __float_as_uint(r0), d = __float_as_uint(r1);
    mid = __builtin_amdgcn_perm(d, c, 0x07060302u);
    const float s0 = r0 - __uint_as_float(c & 0xFFFF0000u), s1 = r1 - __uint_as_float(d & 0xFFFF0000u);
    lo = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
#endif
}

__device__ __forceinline__ float silu_f(float x) {
    // x / (1 + exp(-x)); v_exp_f32 + v_rcp_f32 (about 1 ulp each)
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// Two / four SiLUs at a time: the same operations per element as silu_f (multiply by -log2(e), v_exp_f32, add 1,
// v_rcp_f32, multiply), with the three non-transcendental ones as PACKED fp32 instructions (v_pk_mul_f32 /
// v_pk_add_f32: one issue slot for two elements).  In the edge kernel every VALU instruction competes for issue
// slots with the co-resident workgroup's MFMA chain (about 13 cycles each), so instruction count
// is what matters there.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 silu2_f(f32x2 x) {
    const f32x2 t = x * -1.4426950408889634f;
    f32x2 e;
    e.x = __builtin_amdgcn_exp2f(t.x);
    e.y = __builtin_amdgcn_exp2f(t.y);
    const f32x2 d = e + 1.0f;
    f32x2 r;
    r.x = __builtin_amdgcn_rcpf(d.x);
    r.y = __builtin_amdgcn_rcpf(d.y);
    return x * r;
}

// ---------------------------------------------------------------- weight preparation ---
__global__ __launch_bounds__(256) void ms_egnn_prepare_kernel(const float *__restrict__ blob, float *__restrict__ prep) {
    const int layer = blockIdx.y;
    const float *w = blob + (size_t)layer * LAYER_FLOATS;
    float *p = prep + (size_t)layer * P_LAYER;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < P_W2S; t += gridDim.x * blockDim.x) {
        float v = 0.0f;
        if (t < P_W1BT) {                       // W1aT[k][c] = W1[c][k]
            const int k = t / EHP, c = t % EHP;
            if (c < EH) v = w[C_W1 + c * EIN + k];
        } else if (t < P_B1) {                  // W1bT[k][c] = W1[c][128 + k]
            const int u = t - P_W1BT, k = u / EHP, c = u % EHP;
            if (c < EH) v = w[C_W1 + c * EIN + DIM + k];
        } else if (t < P_WC) {
            const int c = t - P_B1;
            if (c < EH) v = w[C_B1 + c];
        } else if (t < P_W2F) {
            const int c = t - P_WC;
            if (c < EH) v = w[C_W1 + c * EIN + 2 * DIM];
        } else if (t < P_B2) {                  // W2f[g][nt][kh*32+n][u] = W2[32nt+n][8g+4kh+u]
            const int u4 = t - P_W2F;
            const int u = u4 & 3, lane = (u4 >> 2) & 63, nt = (u4 >> 8) & 7, g = u4 >> 11;
            const int k = 8 * g + 4 * (lane >> 5) + u, n = 32 * nt + (lane & 31);
            if (k < EH) v = w[C_W2 + n * EH + k];
        } else if (t < P_WG) {
            v = w[C_B2 + (t - P_B2)];
        } else if (t < P_BG) {
            v = w[C_WG + (t - P_WG)];
        } else if (t < P_WN1T) {
            if (t == P_BG) v = w[C_BG];
        } else if (t < P_BN1) {                 // Wn1T[k][o] = Wn1[o][k]
            const int u = t - P_WN1T, k = u / NHID, o = u % NHID;
            v = w[C_WN1 + o * NIN + k];
        } else if (t < P_WN2T) {
            v = w[C_BN1 + (t - P_BN1)];
        } else if (t < P_BN2) {                 // Wn2T[k][o] = Wn2[o][k]
            const int u = t - P_WN2T, k = u / DIM, o = u % DIM;
            v = w[C_WN2 + o * NHID + k];
        } else {
            v = w[C_BN2 + (t - P_BN2)];
        }
        p[t] = v;
    }
    // W2S[b][nt][part][lane = 32 h + n][j]: part of W2[32 nt + n][16 b + 8 h + j]; one thread per (b, nt, lane): 8 values -> 3 x 16 bytes
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < KB16 * 8 * 64; t += gridDim.x * blockDim.x) {
        const int lane = t & 63, nt = (t >> 6) & 7, b = t >> 9;
        const int n = 32 * nt + (lane & 31), k0 = 16 * b + 8 * (lane >> 5);
        u32x4_e hi, mid, lo;
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            const int ka = k0 + 2 * jp, kb = ka + 1;
            const float x0 = ka < EH ? w[C_W2 + n * EH + ka] : 0.0f, x1 = kb < EH ? w[C_W2 + n * EH + kb] : 0.0f;
            uint32_t h_, m_, l_;
            split3_pair(x0, x1, h_, m_, l_);
            hi[jp] = h_; mid[jp] = m_; lo[jp] = l_;
        }
        u32x4_e *dst = reinterpret_cast<u32x4_e *>(p + P_W2S) + ((size_t)(b * 8 + nt) * 3) * 64 + lane;
        dst[0] = hi; dst[64] = mid; dst[128] = lo;
    }
}

// ---------------------------------------------------------------- batch plan -----------
// Per structure d: n, C = ceil(n/32)+1 (records per residue), exclusive prefixes of edge
// tiles and of records.  One workgroup, blocked scan.
__global__ __launch_bounds__(1024) void ms_egnn_plan_kernel(const int32_t *__restrict__ offsets, int nb,
                                                           int32_t *__restrict__ tile_pre, int32_t *__restrict__ rec_pre) {
    __shared__ int64_t s_t[1024];
    __shared__ int64_t s_r[1024];
    const int tid = threadIdx.x;
    const int per = (nb + 1023) / 1024;
    const int d0 = tid * per, d1 = (d0 + per < nb) ? d0 + per : nb;
    int64_t st = 0, sr = 0;
    for (int d = d0; d < d1; ++d) {
        const int64_t n = offsets[d + 1] - offsets[d];
        st += (n * n + TILE_E - 1) / TILE_E;
        sr += n * ((n + 31) / 32 + 1);
    }
    s_t[tid] = st; s_r[tid] = sr;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int64_t a = 0, b = 0;
        if (tid >= off) { a = s_t[tid - off]; b = s_r[tid - off]; }
        __syncthreads();
        s_t[tid] += a; s_r[tid] += b;
        __syncthreads();
    }
    int64_t pt = s_t[tid] - st, pr = s_r[tid] - sr;   // exclusive
    for (int d = d0; d < d1; ++d) {
        const int64_t n = offsets[d + 1] - offsets[d];
        tile_pre[d] = (int32_t)pt; rec_pre[d] = (int32_t)pr;
        pt += (n * n + TILE_E - 1) / TILE_E;
        pr += n * ((n + 31) / 32 + 1);
    }
    if (tid == 1023) { tile_pre[nb] = (int32_t)s_t[1023]; rec_pre[nb] = (int32_t)s_r[1023]; }
}

__device__ __forceinline__ int find_segment(const int32_t *__restrict__ pre, int n, int x) {
    // largest d in [0, n) with pre[d] <= x  (pre is non-decreasing, pre[0] = 0)
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (pre[mid] <= x) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// node -> structure, and h0 = pe[position in structure]  (nndef_fold_egnn_embed.py:27-30,54)
__global__ __launch_bounds__(256) void ms_egnn_init_nodes_kernel(const int32_t *__restrict__ offsets, int nb, int total,
                                                                 const float *__restrict__ pe,
                                                                 int32_t *__restrict__ node_dom, float *__restrict__ h0) {
    const int g = blockIdx.x * 8 + (threadIdx.x >> 5);    // 32 threads (one float4 each) per node
    if (g >= total) return;
    const int d = find_segment(offsets, nb, g);
    const int pos = g - offsets[d];
    const int c4 = threadIdx.x & 31;
    if (c4 == 0) node_dom[g] = d;
    reinterpret_cast<f32x4 *>(h0 + (size_t)g * DIM)[c4] = reinterpret_cast<const f32x4 *>(pe + (size_t)pos * DIM)[c4];
}

// ---------------------------------------------------------------- node projections -----
// ApT4[kq][g] (float4) = b1[4kq..] + sum_k h[g][k] W1[4kq..][k];  BpT4 likewise with W1[:,128+k], no bias.
// NJ nodes per thread (16 NJ nodes per workgroup); a thread owns one 4-channel quad per 16 gridDim.y quads, so every pair of
// weight float4 it loads feeds 8 NJ FMAs.  Large batches: NJ = 4, gridDim.y = 1 (each weight load reused 32 times, the quads
// looped over).  Small batches (a query of a few domains) are latency-bound -- 6 workgroups each walking 9 x 128 dependent
// weight loads took 132 us for 383 residues -- so they run NJ = 1 with the quads spread over gridDim.y = 9 workgroups.
// Per output the k order of the fmaf chain is the same in every configuration.
template <int NJ>
__global__ __launch_bounds__(256) void ms_egnn_proj_kernel(const float *__restrict__ prep, const float *__restrict__ h,
                                                          int total, f32x4 *__restrict__ ApT4, f32x4 *__restrict__ BpT4) {
    constexpr int NODES = 16 * NJ;
    __shared__ float hs[NODES][DIM + 1];
    const int g0 = blockIdx.x * NODES;
    const int tid = threadIdx.x;
    for (int e = tid; e < NODES * DIM; e += 256) {
        const int n = e >> 7, k = e & 127;
        hs[n][k] = (g0 + n < total) ? h[(size_t)(g0 + n) * DIM + k] : 0.0f;
    }
    __syncthreads();
    const int n = tid & 15, cq0 = (tid >> 4) + 16 * blockIdx.y;
    const f32x4 *w1a = reinterpret_cast<const f32x4 *>(prep + P_W1AT);
    const f32x4 *w1b = reinterpret_cast<const f32x4 *>(prep + P_W1BT);
    const f32x4 *b1 = reinterpret_cast<const f32x4 *>(prep + P_B1);
    for (int cq = cq0; cq < KQ; cq += 16 * gridDim.y) {
        f32x4 a[NJ], b[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { a[j] = b1[cq]; b[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
        constexpr int PF = 8;            // weight rows fetched ahead of their use
        f32x4 qa[PF], qb[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) { qa[u] = w1a[u * KQ + cq]; qb[u] = w1b[u * KQ + cq]; }
#pragma unroll 1      // (left to itself hipcc unrolls all 16 rounds and hoists their 256 weight loads: the NJ = 1 instantiation took 512 registers and 548 bytes of scratch)
        for (int k0 = 0; k0 < DIM; k0 += PF) {
            f32x4 ca[PF], cb[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) { ca[u] = qa[u]; cb[u] = qb[u]; }
            if (k0 + PF < DIM) {
#pragma unroll
                for (int u = 0; u < PF; ++u) { qa[u] = w1a[(k0 + PF + u) * KQ + cq]; qb[u] = w1b[(k0 + PF + u) * KQ + cq]; }
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const f32x4 wa = ca[u], wb = cb[u];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const float x = hs[n + 16 * j][k0 + u];
                    a[j].x = fmaf(x, wa.x, a[j].x); a[j].y = fmaf(x, wa.y, a[j].y); a[j].z = fmaf(x, wa.z, a[j].z); a[j].w = fmaf(x, wa.w, a[j].w);
                    b[j].x = fmaf(x, wb.x, b[j].x); b[j].y = fmaf(x, wb.y, b[j].y); b[j].z = fmaf(x, wb.z, b[j].z); b[j].w = fmaf(x, wb.w, b[j].w);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int node = g0 + n + 16 * j;
            if (node < total) {
                ApT4[(size_t)cq * total + node] = a[j];
                BpT4[(size_t)cq * total + node] = b[j];
            }
        }
    }
}

// ---------------------------------------------------------------- fused edge kernel ----
struct EdgeParams {
    const float *prep;            // this layer's prepared weights
    const float *coords;          // [total,3]
    const int32_t *offsets;       // [nb+1]
    const int32_t *tile_pre;      // [nb+1]
    const int32_t *rec_pre;       // [nb+1]
    const f32x4 *ApT4;            // [130][total]
    const f32x4 *BpT4;
    float *part;                  // [records][256]
    int nb;
    int total;
    int n_tiles;                  // tiles of this launch (split form with 8 waves: a workgroup takes two; the last one may have one)
#ifdef MS_STAMP
    unsigned long long *stamps;   // diagnostic builds: per tile phase cycles
#endif
};

// What bounds it (tools/probes/valu_price_probe.hip, profiles/r03_valu_price_probe.log): next to v_mfma_f32_32x32x2_f32 every
// vector instruction is ADDITIVE -- a cluster of N of them costs the matrix pipe ~8 + 4.6 N cycles (9 per v_exp / v_rcp, packed
// fp32 at the scalar price), whichever wave of the SIMD issues them; ds_write_b128 and VMEM instructions cost 3 / 8 cycles when
// spaced one per MFMA and 40 / 64 each in a burst.  So: (1) H is computed by the lane that needs it as its MFMA A operand --
// lane (r, h) of wave w owns edge row 32 w + r and the k half h, exactly the fragment v_mfma_f32_32x32x2_f32 wants -- and never
// goes through LDS (the first version wrote it with 5 ds_write_b128 per thread and stage in a burst, and needed a second
// barrier per stage); (2) the W2 stage (40 KiB, already in B-fragment order in HBM) arrives by LDS-DMA into a ring of two
// slots, one stage ahead, one barrier per stage; (3) the 25 memory instructions of a stage (projections of the next stage,
// its distance weights, 10 DMA pieces) are issued one per block of 4 MFMAs; (4) the SiLU arithmetic of the next stage is ONE
// cluster behind the stage's 160 MFMAs, kept there by a scheduling fence (hipcc would spread it through the MFMAs).
constexpr int W_STAGE_F4 = STAGE_G * 8 * 64;                      // float4 per W2 stage (40 KiB)
constexpr int EDGE_LDS = 2 * W_STAGE_F4 * 16;                     // two ring slots: 80 KiB per workgroup, two workgroups per CU
#ifndef MS_EGNN_W8
#define MS_EGNN_W8 0            // round 5 A/B (profiles/r05_egnn_variants_ab.log): the 8-wave workgroup is correct and no faster (58.2-58.3 against 57.4-57.9 ms)
#endif
#ifndef MS_EGNN_BPREFETCH
#define MS_EGNN_BPREFETCH 1
#endif
#ifndef MS_EGNN_LOAD_NT
#define MS_EGNN_LOAD_NT 4       // the channel tile of a k block behind which the projections of block b + 2 are requested: their registers are free
                                // from tile 4 on, and three tiles more of lead cover the L2 round trip (tile 7, round 4: 59.1 -> 57.7 ms)
#endif
// split form, MS_EGNN_W8=1 (round 5, built, parity-green, NOT the default): ONE workgroup of EIGHT waves per CU -- two 128-edge tiles side
// by side (waves 0-3 / 4-7) over ONE W2S ring of six 24 KiB slots, a barrier every SECOND k block, every W2S block crossing L2 -> LDS
// once per CU instead of twice.  Round 4's stamps had the waves of a SIMD waiting at their barriers 28 % of the time; halving the
// barriers and sharing the ring changed NOTHING (same-box A/B, profiles/r05_egnn_variants_ab.log) -- the wait is not the barrier's.
// The default stays round 4's form: two workgroups of four waves per CU, a ring of three slots each, a barrier per block.
constexpr bool EGNN_W8 = MS_EGNN_W8 != 0;
constexpr int EDGE_RING_SPLIT = EGNN_W8 ? 6 : 3;
#ifndef MS_EGNN_DYNPRIO
#define MS_EGNN_DYNPRIO 0
#endif
constexpr int EDGE_LDS_SPLIT = EDGE_RING_SPLIT * W2S_BLOCK_BYTES; // split form: slots of one k block (24 KiB) each: 144 KiB (72 KiB with 3)

// SPLIT (round 4): the same GEMM on the bf16 matrix instruction -- v_mfma_f32_32x32x16_bf16 at 16 x the rate of the fp32 one --
// with both operands split three ways (split3_pair): per 16 k and 32 channels SIX matrix instructions (hi.hi, hi.mid, mid.hi,
// hi.lo, lo.hi, mid.mid) of 32 cycles instead of eight fp32 ones of 64: 0.375 of the matrix time, fp32-grade results (what is
// dropped is below one fp32 rounding per product; accumulation is fp32 either way).  W2 is split once (ms_egnn_prepare_weights: the
// image P_W2S, in B-fragment order, 24 KiB per k block, streamed through a ring of three slots by LDS-DMA); H is split by the lane
// that computes it (~45 vector instructions per 8 values, in the shadow of the 48 matrix instructions of a k block: next to the
// bf16 matrix instruction vector instructions are NOT additive).  One barrier per k block.
template <bool SPLIT>
__global__ __launch_bounds__((SPLIT && EGNN_W8) ? 512 : 256, 2) void ms_egnn_edge_kernel(const EdgeParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *Wring = reinterpret_cast<f32x4 *>(smem);                 // [2 slots][5 g][8 nt][64 lanes]
    constexpr bool W8 = SPLIT && EGNN_W8;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7 with eight waves: DMA pieces are dealt over all of them
    const int wave = W8 ? (wave_all & 3) : wave_all;                // this wave's 32 rows inside ITS tile
    const int T_raw = W8 ? 2 * (int)blockIdx.x + (wave_all >> 2) : (int)blockIdx.x;
    const bool tile_valid = T_raw < p.n_tiles;                      // (an odd number of tiles: the last workgroup's second half only loads and synchronises)
    const int T = tile_valid ? T_raw : p.n_tiles - 1;
    const int d = find_segment(p.tile_pre, p.nb, T);
    const int off = p.offsets[d];
    const int n = p.offsets[d + 1] - off;
    const int tt = T - p.tile_pre[d];
    const int64_t nn = (int64_t)n * n;
    const int64_t e0 = (int64_t)tt * TILE_E;

    // this lane's edge row and k half: the A fragment of its wave's 32 x 32 x 2 MFMAs
    const int R = 32 * wave + (lane & 31), kh = lane >> 5;
    const int64_t e = e0 + R;
    int i = 0, j = 0;
    if (e < nn) { i = (int)(e / n); j = (int)(e - (int64_t)i * n); }
    const int gi = off + i, gj = off + j;
    const f32x4 *wc4 = reinterpret_cast<const f32x4 *>(p.prep + P_WC);
    const f32x4 *w2f = reinterpret_cast<const f32x4 *>(p.prep + P_W2F);

    // Memory instructions of the stage loop as inline asm with a SCALAR base and one 32-bit lane offset each (hipcc forms the
    // 64-bit lane addresses with a v_lshl_add_u64 per access: 25 more vector instructions per stage, ~12 cycles apiece alone in
    // an MFMA gap).  hipcc does not count asm loads: the waits are written out below (`wait_loads`).
    // one 1 KiB piece of W2 stage s: pieces 10 w .. 10 w + 9 belong to wave w
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem);
    const uint32_t voff_w = (uint32_t)(lane * sizeof(f32x4));
    auto dma_w = [&](int s, int piece) {
        const int pidx = wave * 10 + piece;
        const uint64_t base = (uint64_t)(uintptr_t)w2f + ((uint64_t)s * W_STAGE_F4 + (uint64_t)pidx * 64) * sizeof(f32x4);
        const uint32_t dst = lds0 + (uint32_t)(((s & 1) * W_STAGE_F4 + pidx * 64) * sizeof(f32x4));
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff_w), "s"(base) : "memory", "m0");
    };
    float d2;
    {
        // rel_coors, dist = norm(rel), then dist * dist: my_egnn_nocoords.py:48-49,58
        const float dx = p.coords[3 * gi] - p.coords[3 * gj];
        const float dy = p.coords[3 * gi + 1] - p.coords[3 * gj + 1];
        const float dz = p.coords[3 * gi + 2] - p.coords[3 * gj + 2];
        const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
        d2 = dist * dist;
    }
    f32x16 acc[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;

#ifdef MS_STAMP
    unsigned long long tH = 0, tB1 = 0, tM = 0, tB2 = 0, tc = __builtin_amdgcn_s_memtime();
    const unsigned long long t_begin = tc;
#define EST(acc) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc += n_ - tc; tc = n_; }
#else
#define EST(acc)
#endif
    if constexpr (!SPLIT) {
    // The projections and distance weights of the next stage are BUFFER loads -- a scalar resource descriptor per (stage, k quad) and
    // one 32-bit lane offset, as cheap to address as the asm form they replace (hipcc forms 64-bit lane addresses with a
    // v_lshl_add_u64 per global_load) -- and, unlike asm loads, VISIBLE to the compiler: their destination registers are in flight
    // across the compiler-scheduled matrix instructions of a stage, and a register the hardware fills behind the compiler's back
    // is only safe as long as nothing makes hipcc move it (the scan lost a row that way in round 3: DESIGN.md 5.5).
    f32x4 pa[STAGE_G], pb[STAGE_G], pc[STAGE_G], hv[STAGE_G];
    typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
    const uint32_t voff_a = (uint32_t)(((size_t)kh * p.total + gi) * sizeof(f32x4));
    const uint32_t voff_b = (uint32_t)(((size_t)kh * p.total + gj) * sizeof(f32x4));
    const uint32_t voff_c = (uint32_t)(kh * sizeof(f32x4));
    auto buf_load = [&](const f32x4 *base, uint32_t voff) -> f32x4 {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0x7fffffff, 0x00027000);
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, 0, 0));
    };
    auto load_a = [&](int s, int g) { pa[g] = buf_load(p.ApT4 + (size_t)(2 * (STAGE_G * s + g)) * p.total, voff_a); };
    auto load_b = [&](int s, int g) { pb[g] = buf_load(p.BpT4 + (size_t)(2 * (STAGE_G * s + g)) * p.total, voff_b); };
    auto load_c = [&](int s, int g) { pc[g] = buf_load(wc4 + (size_t)(2 * (STAGE_G * s + g)), voff_c); };
    // the W2 pieces of the next stage (LDS-DMA, inline asm: hipcc does not count them) have landed too
    auto wait_loads = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
#pragma unroll
    for (int pc_ = 0; pc_ < 10; ++pc_) dma_w(0, pc_);
#pragma unroll
    for (int g = 0; g < STAGE_G; ++g) { load_a(0, g); load_b(0, g); load_c(0, g); }
    wait_loads();
    // H = SiLU(Ap_i + Bp_j + w_c * d2) of one stage, straight into the A-fragment registers
    auto compute_h = [&]() {
#pragma unroll
        for (int g = 0; g < STAGE_G; ++g) {
            const f32x4 zsum = pa[g] + pb[g];
            const f32x2 d2v = {d2, d2};
            const f32x2 z01 = __builtin_elementwise_fma(f32x2{pc[g].x, pc[g].y}, d2v, f32x2{zsum.x, zsum.y});
            const f32x2 z23 = __builtin_elementwise_fma(f32x2{pc[g].z, pc[g].w}, d2v, f32x2{zsum.z, zsum.w});
            const f32x2 h01 = silu2_f(z01), h23 = silu2_f(z23);
            hv[g] = f32x4{h01.x, h01.y, h23.x, h23.y};
        }
    };
    compute_h();

    for (int s = 0; s < NSTAGE; ++s) {
        // W2 stage s has landed for every wave (its pieces were issued a stage ago), and every wave is done reading the other slot
        __syncthreads();
        EST(tB1)
        const f32x4 *Wl = Wring + (s & 1) * W_STAGE_F4 + lane;
        const bool more = s + 1 < NSTAGE;
        f32x4 bcur = Wl[0];
#pragma unroll
        for (int blk = 0; blk < STAGE_G * 8; ++blk) {
            const int g = blk >> 3, nt = blk & 7;
            const f32x4 a = hv[g];
            const f32x4 bnext = Wl[((blk + 1 < STAGE_G * 8) ? blk + 1 : blk) * 64];     // B fragment of the next block
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bcur.x, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bcur.y, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bcur.z, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bcur.w, acc[nt], 0, 0, 0);
            // one memory instruction of the next stage per block
            if (more) {
                if (blk < 5) load_a(s + 1, blk);
                else if (blk < 10) load_b(s + 1, blk - 5);
                else if (blk < 15) load_c(s + 1, blk - 10);
                else if (blk < 25) dma_w(s + 1, blk - 15);
            }
            bcur = bnext;
            __builtin_amdgcn_sched_barrier(0);
        }
        EST(tM)
        if (more) { wait_loads(); compute_h(); }        // (issued ~150 MFMAs ago: landed long since)
        __builtin_amdgcn_sched_barrier(0);
        EST(tH)
    }
    } else {
        // ---------------- split-bf16 form ----------------
        const u32x4_e *Sring = reinterpret_cast<const u32x4_e *>(smem);      // [3 slots][8 nt][3 parts][64 lanes] of 16 bytes
        const u32x4_e *w2s = reinterpret_cast<const u32x4_e *>(p.prep + P_W2S);
        constexpr int BLK_V = 8 * 3 * 64;                                    // 16-byte vectors per k block
        constexpr int PPW = W8 ? 3 : 6;                                      // 1 KiB pieces of a block per wave (24 in all)
        auto dma_s = [&](int b, int piece) {                                 // pieces PPW w .. PPW w + PPW - 1 of block b -> slot b % ring
            const int pidx = wave_all * PPW + piece;
            const uint64_t base = (uint64_t)(uintptr_t)w2s + ((uint64_t)b * BLK_V + (uint64_t)pidx * 64) * 16;
            const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(((b % EDGE_RING_SPLIT) * BLK_V + pidx * 64) * 16));
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff_w), "s"(base) : "memory", "m0");
        };
        // lane (r, h) owns edge row 32 w + r and the k half h of every block: k = 16 b + 8 h + j = quads 4 b + 2 h, 4 b + 2 h + 1.
        // The last block's upper half (k = 520 .. 527) has no projections: it reads the lower half's (W2S is zero there).
        const uint32_t total16 = (uint32_t)p.total * 16u;
        const uint32_t va = (uint32_t)(((size_t)(2 * kh) * p.total + gi) * 16), vb = (uint32_t)(((size_t)(2 * kh) * p.total + gj) * 16);
        const uint32_t va0 = (uint32_t)((size_t)gi * 16), vb0 = (uint32_t)((size_t)gj * 16), vc = (uint32_t)(2 * kh * 16);
        auto buf_load2 = [&](const f32x4 *base, uint32_t voff, uint32_t soff) -> f32x4 {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0x7fffffff, 0x00027000);
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0));
        };
        f32x4 ld[6];
        auto load6 = [&](int b) {
            const bool last = b == KB16 - 1;
            const f32x4 *ba = p.ApT4 + (size_t)(4 * b) * p.total, *bb = p.BpT4 + (size_t)(4 * b) * p.total, *bc = wc4 + 4 * b;
            ld[0] = buf_load2(ba, last ? va0 : va, 0); ld[1] = buf_load2(ba, last ? va0 : va, total16);
            ld[2] = buf_load2(bb, last ? vb0 : vb, 0); ld[3] = buf_load2(bb, last ? vb0 : vb, total16);
            ld[4] = buf_load2(bc, last ? 0u : vc, 0);  ld[5] = buf_load2(bc, last ? 0u : vc, 16);
        };
        // H = SiLU(Ap_i + Bp_j + w_c d2) of this lane's 8 k values, split into the three A operands
        auto make_a = [&](bf16x8_e &ah, bf16x8_e &am, bf16x8_e &al) {
            u32x4_e hi, mid, lo;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const f32x4 zsum = ld[half] + ld[2 + half];
                const f32x2 d2v = {d2, d2};
                const f32x2 z01 = __builtin_elementwise_fma(f32x2{ld[4 + half].x, ld[4 + half].y}, d2v, f32x2{zsum.x, zsum.y});
                const f32x2 z23 = __builtin_elementwise_fma(f32x2{ld[4 + half].z, ld[4 + half].w}, d2v, f32x2{zsum.z, zsum.w});
                const f32x2 h01 = silu2_f(z01), h23 = silu2_f(z23);
                uint32_t h_, m_, l_;
                split3_pair(h01.x, h01.y, h_, m_, l_); hi[2 * half] = h_; mid[2 * half] = m_; lo[2 * half] = l_;
                split3_pair(h23.x, h23.y, h_, m_, l_); hi[2 * half + 1] = h_; mid[2 * half + 1] = m_; lo[2 * half + 1] = l_;
            }
            ah = __builtin_bit_cast(bf16x8_e, hi); am = __builtin_bit_cast(bf16x8_e, mid); al = __builtin_bit_cast(bf16x8_e, lo);
        };
#if MS_EGNN_DYNPRIO
        uint32_t *prio_cnt = reinterpret_cast<uint32_t *>(smem + EDGE_LDS_SPLIT);      // (64 bytes behind the ring)
        if (tid == 0) *prio_cnt = 0u;
        __syncthreads();
#endif
        // prologue: the first blocks of the ring are requested before anything else (8 waves: four blocks ahead; 4 waves: two)
        constexpr int AHEAD = W8 ? 4 : 2;
#pragma unroll
        for (int b0 = 0; b0 < AHEAD; ++b0)
#pragma unroll
            for (int pc_ = 0; pc_ < PPW; ++pc_) dma_s(b0, pc_);
        load6(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bf16x8_e ah, am, al, nh, nm, nl;
        make_a(ah, am, al);
        nh = ah; nm = am; nl = al;
        load6(1);
        for (int b = 0; b < KB16; ++b) {
            // W2S block b has landed for every wave (each waited for its own pieces in the middle of the last iteration), and every
            // wave is done reading slot (b - 1) % 3, which block b + 2 is about to overwrite.
            // Eight waves, six slots: ONE barrier per PAIR of blocks -- before blocks 2p, 2p + 1 every wave has waited for its pieces of
            // both (they were requested four blocks ahead; the vmcnt(0) in the middle of block 2p - 1 covers everything requested up to
            // block 2p - 2, i.e. blocks <= 2p + 2) and is done with the pair before; during block b the pieces of block b + 4 go to
            // slot (b + 4) % 6 = (b - 2) % 6, a slot of the PREVIOUS pair, which nobody reads any more.
#if MS_EGNN_DYNPRIO
            if constexpr (!W8) {
                // Round 6 experiment: the wave that arrives LAST at a block's barrier -- the one its workgroup waited for -- takes the higher
                // issue priority for the next block's chain, the others the lower one (arrival order from a counter in LDS: 4 adds per block)
                uint32_t ord_ = 0;
                if (lane == 0) ord_ = __hip_atomic_fetch_add(prio_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                ord_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)ord_) & 3u;
                __syncthreads();
                if (ord_ == 3u) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
            } else
#endif
            if (!W8 || (b & 1) == 0) __syncthreads();
            EST(tB1)
            const u32x4_e *Sl = Sring + (b % EDGE_RING_SPLIT) * BLK_V + lane;
            // (round 5: the three B fragments of channel tile nt + 1 are requested BEFORE the six matrix instructions of tile nt -- hipcc
            //  left every ds_read_b128 right in front of its first use with an s_waitcnt behind it: eight exposed LDS round trips per
            //  block and wave; MS_EGNN_BPREFETCH=0 at build time keeps that form)
            u32x4_e bq[3] = {Sl[0], Sl[64], Sl[128]};
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
#if MS_EGNN_BPREFETCH
                const u32x4_e c0 = bq[0], c1 = bq[1], c2 = bq[2];
                if (nt + 1 < 8) {
                    bq[0] = Sl[((nt + 1) * 3 + 0) * 64]; bq[1] = Sl[((nt + 1) * 3 + 1) * 64]; bq[2] = Sl[((nt + 1) * 3 + 2) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8_e bh = __builtin_bit_cast(bf16x8_e, c0), bm = __builtin_bit_cast(bf16x8_e, c1), bl = __builtin_bit_cast(bf16x8_e, c2);
#else
                const bf16x8_e bh = __builtin_bit_cast(bf16x8_e, Sl[(nt * 3 + 0) * 64]), bm = __builtin_bit_cast(bf16x8_e, Sl[(nt * 3 + 1) * 64]),
                               bl = __builtin_bit_cast(bf16x8_e, Sl[(nt * 3 + 2) * 64]);
#endif
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[nt], 0, 0, 0);
                if (nt == 3 && b + 1 < KB16) {
                    // the next block's projections were requested a block ago; everything older than that -- this wave's pieces of
                    // W2S block b + 1 -- has landed with them (vector-memory operations complete in order)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    make_a(nh, nm, nl);
                }
                if constexpr (W8) {
                    if (nt >= 5 && b + 4 < KB16) dma_s(b + 4, nt - 5);
                } else {
                    if (nt >= 5 && b + 2 < KB16) { dma_s(b + 2, 2 * (nt - 5)); dma_s(b + 2, 2 * (nt - 5) + 1); }
                }
                if (nt == MS_EGNN_LOAD_NT && b + 2 < KB16) load6(b + 2);
            }
            EST(tM)
            ah = nh; am = nm; al = nl;
        }
    }
#ifdef MS_STAMP
    const unsigned long long t_loop_end = tc;
#endif
    const int wave_ = wave;
    (void)wave_;

    __builtin_amdgcn_s_setprio(3);
    // ---- epilogue.  acc[nt][r]: edge row (r&3) + 8(r>>2) + 4(lane>>5) of this wave's 32 rows,
    //      channel 32 nt + (lane & 31).
    const int c = lane & 31, hh = lane >> 5;
    float gate_dot[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) gate_dot[r] = 0.0f;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
        const float b2 = p.prep[P_B2 + 32 * nt + c];
        const float wg = p.prep[P_WG + 32 * nt + c];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const f32x2 m = silu2_f(f32x2{acc[nt][r], acc[nt][r + 1]} + b2);          // edge_mlp[2] bias + SiLU (:21-22)
            acc[nt][r] = m.x;
            acc[nt][r + 1] = m.y;
            const f32x2 gd = __builtin_elementwise_fma(f32x2{wg, wg}, m, f32x2{gate_dot[r], gate_dot[r + 1]});
            gate_dot[r] = gd.x;
            gate_dot[r + 1] = gd.y;
        }
    }
    const float bg = p.prep[P_BG];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        // sum over the 32 channel lanes of this half: cyclic doubling inside each row of 16 lanes with DPP rotations
        // (row_ror 8, 4, 2, 1: no LDS crossbar), then the other row of the half
        float v = gate_dot[r];
        v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x128, 0xF, 0xF, false));
        v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x124, 0xF, 0xF, false));
        v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x122, 0xF, 0xF, false));
        v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x121, 0xF, 0xF, false));
        v += __shfl_xor(v, 16);
        gate_dot[r] = sigmoid_f(v + bg);                      // edge_gate (:25-28)
    }
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {                          // m_ij * gate (:64), two rows per instruction
            const f32x2 mg = f32x2{acc[nt][r], acc[nt][r + 1]} * f32x2{gate_dot[r], gate_dot[r + 1]};
            acc[nt][r] = mg.x;
            acc[nt][r + 1] = mg.y;
        }

    // per-residue partial sums over this wave's 32 consecutive edges (m_i = sum_j m_ij, :69)
    const int64_t u0 = e0 + 32 * wave;                 // first edge of the unit
    if (tile_valid && u0 < nn) {
        const int unit = tt * 4 + wave;                // 32-edge unit index inside the structure
        const int C = (n + 31) / 32 + 1;
        const int i_first = (int)(u0 / n);
        int64_t last_e = u0 + 31;
        if (last_e > nn - 1) last_e = nn - 1;
        const int i_last = (int)(last_e / n);
        for (int is = i_first; is <= i_last; ++is) {
            const int64_t lo64 = (int64_t)is * n - u0, hi64 = (int64_t)(is + 1) * n - u0;
            const int lo = lo64 < 0 ? 0 : (int)lo64;
            const int hi = hi64 > 32 ? 32 : (int)hi64;
            const int q = unit - (int)(((int64_t)is * n) >> 5);
            float *dst = p.part + ((size_t)p.rec_pre[d] + (size_t)is * C + q) * MD;
            // rows of this residue as 0/1 weights, once per residue; then 8 packed FMAs per channel tile
            // (even and odd rows accumulate separately and are added last)
            f32x2 wsel[8];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int row0 = (r & 3) + 8 * (r >> 2) + 4 * hh, row1 = ((r + 1) & 3) + 8 * ((r + 1) >> 2) + 4 * hh;
                wsel[r >> 1] = f32x2{(row0 >= lo && row0 < hi) ? 1.0f : 0.0f, (row1 >= lo && row1 < hi) ? 1.0f : 0.0f};
            }
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
                f32x2 sum2 = {0.0f, 0.0f};
#pragma unroll
                for (int r = 0; r < 16; r += 2)
                    sum2 = __builtin_elementwise_fma(f32x2{acc[nt][r], acc[nt][r + 1]}, wsel[r >> 1], sum2);
                float sum = sum2.x + sum2.y;
                sum += __shfl_xor(sum, 32);
                if (hh == 0) dst[32 * nt + c] = sum;
            }
        }
    }
#ifdef MS_STAMP
    if ((tid & 255) == 0 && tile_valid && p.stamps != nullptr && T < 32768) {
        unsigned long long *o = p.stamps + 8 * (size_t)T;
        o[0] = tH; o[1] = tB1; o[2] = tM; o[3] = tB2; o[4] = __builtin_amdgcn_s_memtime() - t_loop_end; o[5] = 1; o[6] = t_begin;
    }
#endif
}

// ---------------------------------------------------------------- node update ----------
struct NodeParams {
    const float *prep;
    const float *h_in;            // [total,128]
    const float *part;            // [records][256]
    const int32_t *offsets;
    const int32_t *rec_pre;
    const int32_t *node_dom;
    float *h_out;                 // [total,128]
    int total;
};

#ifdef MS_STAMP
unsigned long long *ms_egnn_stamp_buffer() {
    static unsigned long long *buf = nullptr;
    if (buf == nullptr) {
        if (hipMalloc(reinterpret_cast<void **>(&buf), 8 * 32768 * 8) != hipSuccess) return nullptr;
        (void)hipMemset(buf, 0, 8 * 32768 * 8);
    }
    return buf;
}
#endif

// NB nodes per workgroup: 16 for large batches (every weight column a thread loads feeds 16 FMAs), 4 for a query of a few
// domains (more workgroups, shorter inner work: the kernel is then a chain of L2 round trips for the weights).  The weight
// columns are fetched PF rows ahead of their use.  Per output the fmaf chain runs over k in ascending order in every case.
template <int NB>
__global__ __launch_bounds__(256) void ms_egnn_node_kernel(const NodeParams p) {
    __shared__ float xs[NB][NIN];      // [h, m_i]
    __shared__ float hid[NB][NHID];
    constexpr int PF = NB <= 4 ? 16 : 8;
    const int tid = threadIdx.x;
    const int g0 = blockIdx.x * NB;
    for (int e = tid; e < NB * DIM; e += 256) {
        const int nd = e >> 7, k = e & 127;
        xs[nd][k] = (g0 + nd < p.total) ? p.h_in[(size_t)(g0 + nd) * DIM + k] : 0.0f;
    }
    // m_i: sum this residue's records in record order (the NB residues side by side: independent loads in flight)
    {
        const float *src[NB];
        int cnt[NB], cmax = 0;
        // (float64 accumulator: a residue of a 2000-residue chain has 63 records, and the reference's own fp32 sums -- torch's
        //  blocked reductions -- are within 1e-7 of the float64 truth there; a sequential fp32 sum is not: round 5, SURVEY.md 8c's
        //  bar of 1e-6 at N = 2000)
        double m[NB];
#pragma unroll
        for (int nd = 0; nd < NB; ++nd) {
            const int g = g0 + nd;
            src[nd] = p.part; cnt[nd] = 0; m[nd] = 0.0;
            if (g < p.total) {
                const int d = p.node_dom[g];
                const int off = p.offsets[d];
                const int n = p.offsets[d + 1] - off;
                const int i = g - off;
                const int C = (n + 31) / 32 + 1;
                const int first = (int)(((int64_t)i * n) >> 5);
                const int last = (int)(((int64_t)i * n + n - 1) >> 5);
                src[nd] = p.part + ((size_t)p.rec_pre[d] + (size_t)i * C) * MD + tid;
                cnt[nd] = last - first + 1;
            }
            cmax = cnt[nd] > cmax ? cnt[nd] : cmax;
        }
        for (int q = 0; q < cmax; ++q) {
#pragma unroll
            for (int nd = 0; nd < NB; ++nd)
                if (q < cnt[nd]) m[nd] += (double)src[nd][(size_t)q * MD];
        }
#pragma unroll
        for (int nd = 0; nd < NB; ++nd) xs[nd][DIM + tid] = (float)m[nd];
    }
    __syncthreads();
    {   // node_mlp[0] + SiLU (:31-32): thread = output channel, all NB nodes
        float a[NB];
        const float b = p.prep[P_BN1 + tid];
#pragma unroll
        for (int nd = 0; nd < NB; ++nd) a[nd] = b;
        const float *w = p.prep + P_WN1T + tid;
        float wq[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) wq[u] = w[(size_t)u * NHID];
        for (int k0 = 0; k0 < NIN; k0 += PF) {
            float wc[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) wc[u] = wq[u];
            if (k0 + PF < NIN) {
#pragma unroll
                for (int u = 0; u < PF; ++u) wq[u] = w[(size_t)(k0 + PF + u) * NHID];
            }
#pragma unroll
            for (int u = 0; u < PF; ++u)
#pragma unroll
                for (int nd = 0; nd < NB; ++nd) a[nd] = fmaf(wc[u], xs[nd][k0 + u], a[nd]);
        }
#pragma unroll
        for (int nd = 0; nd < NB; ++nd) hid[nd][tid] = silu_f(a[nd]);
    }
    __syncthreads();
    {   // node_mlp[2] + residual (:33, :72): thread = (output channel, half of the nodes)
        const int o = tid & 127, half = tid >> 7;
        float a[NB / 2];
        const float b = p.prep[P_BN2 + o];
#pragma unroll
        for (int nd = 0; nd < NB / 2; ++nd) a[nd] = b;
        const float *w = p.prep + P_WN2T + o;
        float wq[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) wq[u] = w[(size_t)u * DIM];
        for (int k0 = 0; k0 < NHID; k0 += PF) {
            float wc[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) wc[u] = wq[u];
            if (k0 + PF < NHID) {
#pragma unroll
                for (int u = 0; u < PF; ++u) wq[u] = w[(size_t)(k0 + PF + u) * DIM];
            }
#pragma unroll
            for (int u = 0; u < PF; ++u)
#pragma unroll
                for (int nd = 0; nd < NB / 2; ++nd) a[nd] = fmaf(wc[u], hid[half * (NB / 2) + nd][k0 + u], a[nd]);
        }
#pragma unroll
        for (int nd = 0; nd < NB / 2; ++nd) {
            const int node = half * (NB / 2) + nd;
            if (g0 + node < p.total) p.h_out[(size_t)(g0 + node) * DIM + o] = a[nd] + xs[node][o];
        }
    }
}

// embed = mean over residues, in residue order (nndef_fold_egnn_embed.py:61); the rows are fetched eight at a time (a loop of
// dependent L2 round trips otherwise: 36 us for a 163-residue domain), the additions keep the residue order
__global__ __launch_bounds__(128) void ms_egnn_pool_kernel(const float *__restrict__ h, const int32_t *__restrict__ offsets,
                                                          float *__restrict__ out) {
    const int d = blockIdx.x, c = threadIdx.x;
    const int off = offsets[d], n = offsets[d + 1] - off;
    const float *src = h + (size_t)off * DIM + c;
    double s = 0.0;          // (float64: 2000 sequential fp32 additions lose what the reference's blocked mean keeps)
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(i + u) * DIM];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; i < n; ++i) s += (double)src[(size_t)i * DIM];
    out[(size_t)d * DIM + c] = (float)(s / (double)n);
}

struct EgnnCarve {
    size_t off_h0, off_h1, off_ap, off_bp, off_part, off_tile_pre, off_rec_pre, off_node_dom, total;
};

EgnnCarve egnn_carve(int nb, int64_t total, int64_t rec_bound) {
    EgnnCarve c;
    size_t o = 0;
    c.off_h0 = o; o += ms_align_up((size_t)total * DIM * sizeof(float), 256);
    c.off_h1 = o; o += ms_align_up((size_t)total * DIM * sizeof(float), 256);
    c.off_ap = o; o += ms_align_up((size_t)KQ * total * sizeof(f32x4), 256);
    c.off_bp = o; o += ms_align_up((size_t)KQ * total * sizeof(f32x4), 256);
    c.off_part = o; o += ms_align_up((size_t)rec_bound * MD * sizeof(float), 256);
    c.off_tile_pre = o; o += ms_align_up((size_t)(nb + 1) * sizeof(int32_t), 256);
    c.off_rec_pre = o; o += ms_align_up((size_t)(nb + 1) * sizeof(int32_t), 256);
    c.off_node_dom = o; o += ms_align_up((size_t)total * sizeof(int32_t), 256);
    c.total = o;
    return c;
}

}  // namespace

extern "C" {

size_t ms_egnn_weight_floats(void) { return 2 * (size_t)LAYER_FLOATS; }
size_t ms_egnn_prepared_bytes(void) { return 2 * (size_t)P_LAYER * sizeof(float); }

int ms_egnn_prepare_weights(const float *weights, void *prepared, ms_stream_t stream) {
    if (weights == nullptr || prepared == nullptr) MS_FAIL(MS_ERR_ARG, "ms_egnn_prepare_weights: NULL argument");
    hipLaunchKernelGGL(ms_egnn_prepare_kernel, dim3(256, 2), dim3(256), 0, (hipStream_t)stream, weights, (float *)prepared);
    MS_LAUNCH_CHECK("ms_egnn_prepare_kernel");
    return MS_OK;
}

size_t ms_egnn_workspace_bytes(int nb, int64_t total_residues, int64_t sum_sq) {
    if (nb < 1 || total_residues < 1 || sum_sq < 1) return 0;
    // records per structure = N * (ceil(N/32) + 1) <= N^2/32 + 2N
    return egnn_carve(nb, total_residues, sum_sq / 32 + 2 * total_residues + 1).total;
}

#ifdef MS_STAMP
int ms_debug_egnn_stamps(unsigned long long *host, int words) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpy(host, ms_egnn_stamp_buffer(), (size_t)words * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

int ms_egnn_embed(const void *prepared, const float *pe, int pe_len, const float *coords, const int32_t *offsets,
                  const int32_t *offsets_host, int nb, float *out, void *workspace, size_t workspace_bytes,
                  ms_stream_t stream) {
    if (!prepared || !pe || !coords || !offsets || !offsets_host || !out || nb < 1)
        MS_FAIL(MS_ERR_ARG, "ms_egnn_embed: NULL argument or nb < 1");
    if (offsets_host[0] != 0) MS_FAIL(MS_ERR_ARG, "ms_egnn_embed: offsets[0] must be 0");
    int64_t total = 0, sum_sq = 0, tiles = 0, recs = 0;
    for (int d = 0; d < nb; ++d) {
        const int64_t n = (int64_t)offsets_host[d + 1] - offsets_host[d];
        if (n < 1) MS_FAIL(MS_ERR_ARG, "ms_egnn_embed: structure %d is empty", d);
        if (n > pe_len)
            MS_FAIL(MS_ERR_RANGE, "ms_egnn_embed: structure %d has %lld residues, positional table has %d", d, (long long)n, pe_len);
        total += n; sum_sq += n * n;
        tiles += (n * n + TILE_E - 1) / TILE_E;
        recs += n * ((n + 31) / 32 + 1);
    }
    if (tiles >= 0x7FFFFFFF || recs >= 0x7FFFFFFF || total >= 0x7FFFFFFF)
        MS_FAIL(MS_ERR_RANGE, "ms_egnn_embed: batch too large (sum N^2 = %lld); split it", (long long)sum_sq);
    const EgnnCarve cv = egnn_carve(nb, total, sum_sq / 32 + 2 * total + 1);
    if (workspace == nullptr || workspace_bytes < cv.total)
        MS_FAIL(MS_ERR_WORKSPACE, "ms_egnn_embed: workspace %zu < %zu bytes", workspace_bytes, cv.total);
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    float *h0 = (float *)(ws + cv.off_h0), *h1 = (float *)(ws + cv.off_h1);
    f32x4 *ap = (f32x4 *)(ws + cv.off_ap), *bp = (f32x4 *)(ws + cv.off_bp);
    float *part = (float *)(ws + cv.off_part);
    int32_t *tile_pre = (int32_t *)(ws + cv.off_tile_pre), *rec_pre = (int32_t *)(ws + cv.off_rec_pre);
    int32_t *node_dom = (int32_t *)(ws + cv.off_node_dom);
    const float *prep = (const float *)prepared;

    hipLaunchKernelGGL(ms_egnn_plan_kernel, dim3(1), dim3(1024), 0, st, offsets, nb, tile_pre, rec_pre);
    MS_LAUNCH_CHECK("ms_egnn_plan_kernel");
    hipLaunchKernelGGL(ms_egnn_init_nodes_kernel, dim3((unsigned)((total + 7) / 8)), dim3(256), 0, st, offsets, nb,
                       (int)total, pe, node_dom, h0);
    MS_LAUNCH_CHECK("ms_egnn_init_nodes_kernel");
    // the edge GEMM: split-bf16 matrix instructions (default; fp32-grade results) or MS_EGNN_SPLIT=0: the fp32 ones
    static const int split_form = [] { const char *e = getenv("MS_EGNN_SPLIT"); return e ? atoi(e) : 1; }();
    const size_t edge_lds = split_form ? (size_t)EDGE_LDS_SPLIT + (MS_EGNN_DYNPRIO ? 64 : 0) : (size_t)EDGE_LDS;
    if (split_form)
        MS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ms_egnn_edge_kernel<true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)edge_lds));
    else
        MS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ms_egnn_edge_kernel<false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)edge_lds));
    float *hin = h0, *hout = h1;
    for (int layer = 0; layer < 2; ++layer) {
        const float *lp = prep + (size_t)layer * P_LAYER;
        if (total > 8192)
            hipLaunchKernelGGL(ms_egnn_proj_kernel<4>, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, lp, hin, (int)total, ap, bp);
        else        // a few structures: spread the quads over 9 workgroups per 16 nodes (latency, not throughput)
            hipLaunchKernelGGL(ms_egnn_proj_kernel<1>, dim3((unsigned)((total + 15) / 16), 9), dim3(256), 0, st, lp, hin, (int)total, ap, bp);
        MS_LAUNCH_CHECK("ms_egnn_proj_kernel");
        EdgeParams ep;
        ep.prep = lp; ep.coords = coords; ep.offsets = offsets; ep.tile_pre = tile_pre; ep.rec_pre = rec_pre;
        ep.ApT4 = ap; ep.BpT4 = bp; ep.part = part; ep.nb = nb; ep.total = (int)total; ep.n_tiles = (int)tiles;
#ifdef MS_STAMP
        ep.stamps = ms_egnn_stamp_buffer();
#endif
        if (split_form && EGNN_W8) hipLaunchKernelGGL(ms_egnn_edge_kernel<true>, dim3((unsigned)((tiles + 1) / 2)), dim3(512), edge_lds, st, ep);
        else if (split_form) hipLaunchKernelGGL(ms_egnn_edge_kernel<true>, dim3((unsigned)tiles), dim3(256), edge_lds, st, ep);
        else hipLaunchKernelGGL(ms_egnn_edge_kernel<false>, dim3((unsigned)tiles), dim3(256), edge_lds, st, ep);
        MS_LAUNCH_CHECK("ms_egnn_edge_kernel");
        NodeParams np;
        np.prep = lp; np.h_in = hin; np.part = part; np.offsets = offsets; np.rec_pre = rec_pre; np.node_dom = node_dom;
        np.h_out = hout; np.total = (int)total;
        if (total > 8192) hipLaunchKernelGGL(ms_egnn_node_kernel<16>, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, st, np);
        else hipLaunchKernelGGL(ms_egnn_node_kernel<4>, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, st, np);
        MS_LAUNCH_CHECK("ms_egnn_node_kernel");
        float *t = hin; hin = hout; hout = t;
    }
    hipLaunchKernelGGL(ms_egnn_pool_kernel, dim3(nb), dim3(128), 0, st, hin, offsets, out);
    MS_LAUNCH_CHECK("ms_egnn_pool_kernel");
    return MS_OK;
}

}  // extern "C"
