// Diagnostic probe (not part of the library): what a 32-row x 32-query x 128-k tile costs on the fp32 matrix pipe of
// gfx950 in the loop shapes the scan kernel can take, with NO iteration draining the pipe (accumulators ping-pong or
// run on; the first MFMA of a chain takes C = 0 as an inline constant), and what vector / LDS instructions placed
// between the MFMAs of a dependent chain cost.  One wave per SIMD on every CU unless the mode says otherwise.
//
//   hipcc --offload-arch=gfx950 -O3 -o mfma_price_probe mfma_price_probe.hip && ./mfma_price_probe
//
// Nominal: v_mfma_f32_32x32x2_f32 = 64 cycles issue = 64 cycles dependent latency (MI355X_MICROARCH.md cycle constants),
// so a tile (64 of them) = 4096 cycles; v_mfma_f32_16x16x4_f32 = 32 / 40 cycles, 128 per tile.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA_Z(acc, a, b) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b))
#define MFMA16(acc, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define VFILL(x, y) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(y))
#define VMAX3(m, a, b) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(a), "v"(b))

enum {
    M_RUNON = 0,        // one dependent chain that never ends (acc runs on): the bare rate
    M_PINGPONG = 1,     // chain of 64 into acc A (C = 0 inline), next into acc B, ...; the previous tile's 16 scores are folded
                        // into a running maximum by 8 v_max3 in the gaps of the next chain (the scan's filter)
    M_TWOHALF = 2,      // two interleaved chains of 32 (K halves) that run on
    M_PP_READ_XOR = 3,  // M_PINGPONG + 16 ds_read_b128 of the next tile with one v_xor per address (the shipped refill)
    M_PP_READ_IMM = 4,  // M_PINGPONG + 16 ds_read_b128 with immediate offsets (no address arithmetic)
    M_PP_FILL = 5,      // M_PINGPONG + NF v_xor per MFMA
    M_TWO_QT = 7,       // two query tiles per wave sharing the A fragments: two independent chains of 64, interleaved (128 MFMA)
};

template <int MODE, int NF>
__global__ __launch_bounds__(256, 1) void probe(const float *in, float *out, unsigned long long *stamps, int iters) {
    __shared__ __attribute__((aligned(16))) f32x4 tile[4][2048];     // two 16 KiB slots per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    float q[64], q2[(MODE == M_TWO_QT) ? 64 : 1];
    f32x4 a[16];
#pragma unroll
    for (int i = 0; i < 64; ++i) q[i] = in[lane * 64 + i];
    if (MODE == M_TWO_QT) {
#pragma unroll
        for (int i = 0; i < 64; ++i) q2[i] = in[((lane * 64 + i) * 7 + 3) & 4095];
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) tile[wave][lane * 32 + i] = f32x4{in[(lane * 16 + i) & 4095], in[(i * 64 + lane + 1) & 4095], in[(i + 2 + lane * 3) & 4095], in[(i * 5 + lane + 3) & 4095]};
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = tile[wave][r * 32 + ((16 * h + i) ^ (r & 15))];
    f32x16 accA, accB;
#pragma unroll
    for (int i = 0; i < 16; ++i) { accA[i] = 0.0f; accB[i] = 0.0f; }
    float mx = -1e30f;
    uint32_t fill = lane, fill2 = lane * 3;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)&tile[wave][0];
    const uint32_t frag0 = lds0 + (uint32_t)(r * 512 + 256 * h + 16 * (r & 15));          // XOR form: address = frag0 ^ 16 f
    const uint32_t lin0 = lds0 + (uint32_t)(16 * (512 * h + r));                           // chunk-major image: address = lin0 + 512 f
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();

    // one tile: 16 groups of 4 MFMAs from a[tt] and q[4 tt ..]
#define GROUP(ACC, FIRST, tt)                                                        \
    {                                                                                \
        if (FIRST) { MFMA_Z(ACC, a[tt].x, q[4 * (tt) + 0]); } else { MFMA(ACC, a[tt].x, q[4 * (tt) + 0]); } \
        FILLERS MFMA(ACC, a[tt].y, q[4 * (tt) + 1]);                                 \
        FILLERS MFMA(ACC, a[tt].z, q[4 * (tt) + 2]);                                 \
        FILLERS MFMA(ACC, a[tt].w, q[4 * (tt) + 3]);                                 \
        FILLERS                                                                      \
    }
#define FILLERS                                          \
    if (MODE == M_PP_FILL) {                             \
        if (NF >= 1) VFILL(fill, fill2);                 \
        if (NF >= 2) VFILL(fill2, fill);                 \
        if (NF >= 3) VFILL(fill, fill2);                 \
        if (NF >= 4) VFILL(fill2, fill);                 \
        if (NF >= 6) { VFILL(fill, fill2); VFILL(fill2, fill); } \
        if (NF >= 8) { VFILL(fill, fill2); VFILL(fill2, fill); } \
    }
    // previous tile's scores -> running maximum, 2 v_max3 per group in groups 2..5
#define FOLD(PREV, g) { VMAX3(mx, PREV[4 * (g)], PREV[4 * (g) + 1]); VMAX3(mx, PREV[4 * (g) + 2], PREV[4 * (g) + 3]); }
#define REFILL_XOR(FR, f)                                                                                              \
    {                                                                                                              \
        uint32_t ad = FR ^ (uint32_t)(16 * (f));                                                                \
        asm volatile("ds_read_b128 %0, %1" : "=v"(a[f]) : "v"(ad) : "memory");                                     \
    }
#define REFILL_IMM(LN, f) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[f]) : "v"(LN), "i"(512 * (f)) : "memory")
#define PP_TILE(ACC, PREV, FR, LN)                                                               \
    {                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                               \
        _Pragma("unroll") for (int tt = 0; tt < 16; ++tt) {                              \
            if (tt == 0) GROUP(ACC, true, tt) else GROUP(ACC, false, tt)                 \
            if (tt >= 2 && tt < 6) FOLD(PREV, tt - 2)                                    \
            if (MODE == M_PP_READ_XOR && tt >= 8) { REFILL_XOR(FR, 2 * (tt - 8)) REFILL_XOR(FR, 2 * (tt - 8) + 1) } \
            if (MODE == M_PP_READ_IMM && tt >= 8) { REFILL_IMM(LN, 2 * (tt - 8)); REFILL_IMM(LN, 2 * (tt - 8) + 1); } \
        }                                                                                \
    }

    for (int it = 0; it < iters; it += 2) {
        // the slot the refill reads changes from tile to tile, as in the scan's ring: the addresses are not loop invariants
        const uint32_t slot_off = (uint32_t)(it & 2) << 13;
        const uint32_t fragt = frag0 + slot_off, lint = lin0 + slot_off, fragu = frag0 + (slot_off ^ 16384u), linu = lin0 + (slot_off ^ 16384u);
        if (MODE == M_RUNON) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int tt = 0; tt < 16; ++tt) GROUP(accA, false, tt)
        } else if (MODE == M_TWOHALF) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int tt = 0; tt < 8; ++tt) {
                    MFMA(accA, a[tt].x, q[4 * tt + 0]); MFMA(accB, a[tt + 8].x, q[4 * tt + 32]);
                    MFMA(accA, a[tt].y, q[4 * tt + 1]); MFMA(accB, a[tt + 8].y, q[4 * tt + 33]);
                    MFMA(accA, a[tt].z, q[4 * tt + 2]); MFMA(accB, a[tt + 8].z, q[4 * tt + 34]);
                    MFMA(accA, a[tt].w, q[4 * tt + 3]); MFMA(accB, a[tt + 8].w, q[4 * tt + 35]);
                }
        } else if (MODE == M_TWO_QT) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int tt = 0; tt < 16; ++tt) {
                    MFMA(accA, a[tt].x, q[4 * tt + 0]); MFMA(accB, a[tt].x, q2[4 * tt + 0]);
                    MFMA(accA, a[tt].y, q[4 * tt + 1]); MFMA(accB, a[tt].y, q2[4 * tt + 1]);
                    MFMA(accA, a[tt].z, q[4 * tt + 2]); MFMA(accB, a[tt].z, q2[4 * tt + 2]);
                    MFMA(accA, a[tt].w, q[4 * tt + 3]); MFMA(accB, a[tt].w, q2[4 * tt + 3]);
                }
        } else {
            PP_TILE(accA, accB, fragt, lint)
            PP_TILE(accB, accA, fragu, linu)
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = mx + __uint_as_float((fill ^ fill2) & 0x3FFFFFu);
#pragma unroll
    for (int i = 0; i < 16; ++i) s += accA[i] + accB[i] + a[i].x;
    out[blockIdx.x * 256 + tid] = s;
    if (lane == 0) { stamps[(blockIdx.x * 4 + wave) * 2] = c1 - c0; stamps[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0; }
}

// 16x16x4: the tile as 2 x 2 blocks of 16 x 16, four independent accumulators that run on, 128 MFMAs per tile
__global__ __launch_bounds__(256, 1) void probe16(const float *in, float *out, unsigned long long *stamps, int iters) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float q[64], a[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) { q[i] = in[lane * 64 + i]; a[i] = in[(lane * 37 + i) & 4095]; }
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            MFMA16(acc0, a[s], q[s]);
            MFMA16(acc1, a[s], q[32 + s]);
            MFMA16(acc2, a[32 + s], q[s]);
            MFMA16(acc3, a[32 + s], q[32 + s]);
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + tid] = acc0[0] + acc1[1] + acc2[2] + acc3[3];
    if (lane == 0) { stamps[(blockIdx.x * 4 + wave) * 2] = c1 - c0; stamps[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0; }
}

// two waves per SIMD, each running its own never-ending chain (512-thread workgroups, one per CU)
__global__ __launch_bounds__(512, 2) void probe2w(const float *in, float *out, unsigned long long *stamps, int iters) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float q[64], a[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) { q[i] = in[lane * 64 + i]; a[i] = in[(lane * 37 + i) & 4095]; }
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 64; ++s) MFMA(acc, a[s], q[s]);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0) { stamps[(blockIdx.x * 8 + wave) * 2] = c1 - c0; stamps[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0; }
}

static float *g_in, *g_out;
static unsigned long long *g_st;
static const int BLOCKS = 256;

static void report(const char *name, int waves, int iters, double mfma_per_tile, double tiles_per_iter) {
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)waves * 2);
    hipMemcpy(h.data(), g_st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cpt, ghz;
    for (int i = 0; i < waves; ++i) { cpt.push_back((double)h[2 * i] / (iters * tiles_per_iter)); ghz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1); }
    std::sort(cpt.begin(), cpt.end()); std::sort(ghz.begin(), ghz.end());
    const double med = cpt[cpt.size() / 2];
    printf("%-78s %8.1f cycles/tile (min %.1f max %.1f) = %6.2f per MFMA = %5.1f%% of nominal | clock %.3f GHz\n", name, med, cpt.front(), cpt.back(),
           med / mfma_per_tile, 4096.0 / med * 100.0, ghz[ghz.size() / 2]);
    fflush(stdout);
}

template <int MODE, int NF>
static void run(const char *name, int iters) {
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<MODE, NF>), dim3(BLOCKS), dim3(256), 0, 0, g_in, g_out, g_st, iters);
    const bool two = MODE == M_TWO_QT;
    report(name, BLOCKS * 4, iters, two ? 128.0 : 64.0, two ? 0.5 : 1.0);   // M_TWO_QT: one iteration pair = 2 x (2 tiles) -> per (32 x 32) tile
}

int main() {
    const int iters = 4000;
    hipMalloc(&g_in, 4096 * 4 + 64); hipMalloc(&g_out, BLOCKS * 512 * 4); hipMalloc(&g_st, BLOCKS * 8 * 2 * 8);
    std::vector<float> h(4096 + 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.0f - 0.5f;
    hipMemcpy(g_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    // warm the clocks
    for (int w = 0; w < 40; ++w) hipLaunchKernelGGL((probe<M_RUNON, 0>), dim3(BLOCKS), dim3(256), 0, 0, g_in, g_out, g_st, iters);
    hipDeviceSynchronize();
    printf("tile = 32 rows x 32 queries x 128 k; nominal 4096 matrix-pipe cycles; %d tiles per wave, one wave per SIMD on %d CUs\n", iters, BLOCKS);
    run<M_RUNON, 0>("32x32x2: one dependent chain that runs on (no reset, no drain)", iters);
    run<M_PINGPONG, 0>("32x32x2: chains of 64 ping-pong (C = 0 inline), 8 v_max3 of the previous tile in the gaps", iters);
    run<M_TWOHALF, 0>("32x32x2: two interleaved K-half chains that run on", iters);
    run<M_TWO_QT, 0>("32x32x2: two query tiles per wave, shared A, two interleaved chains (per 32x32 tile)", iters);
    run<M_PP_READ_XOR, 0>("ping-pong + 16 ds_read_b128 refill, one v_xor per address (shipped form)", iters);
    run<M_PP_READ_IMM, 0>("ping-pong + 16 ds_read_b128 refill, immediate offsets (chunk-major image)", iters);
    run<M_PP_FILL, 1>("ping-pong + 1 v_xor per MFMA", iters);
    run<M_PP_FILL, 2>("ping-pong + 2 v_xor per MFMA", iters);
    run<M_PP_FILL, 4>("ping-pong + 4 v_xor per MFMA", iters);
    run<M_PP_FILL, 6>("ping-pong + 6 v_xor per MFMA", iters);
    run<M_PP_FILL, 8>("ping-pong + 8 v_xor per MFMA", iters);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(probe16, dim3(BLOCKS), dim3(256), 0, 0, g_in, g_out, g_st, iters);
    report("16x16x4: four independent accumulators that run on (128 MFMA per tile)", BLOCKS * 4, iters, 128.0, 1.0);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(probe2w, dim3(BLOCKS), dim3(512), 0, 0, g_in, g_out, g_st, iters);
    report("32x32x2: TWO waves per SIMD, each its own chain that runs on (cycles per wave-tile)", BLOCKS * 8, iters, 64.0, 1.0);
    // sustained: ~2 s of back-to-back launches of the run-on chain, wall-clocked
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const int big = 20000, launches = 60;
        for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((probe<M_RUNON, 0>), dim3(BLOCKS), dim3(256), 0, 0, g_in, g_out, g_st, big);
        hipEventRecord(e0);
        for (int w = 0; w < launches; ++w) hipLaunchKernelGGL((probe<M_RUNON, 0>), dim3(BLOCKS), dim3(256), 0, 0, g_in, g_out, g_st, big);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)launches * BLOCKS * 4 * (double)big * 64 * 4096.0;
        std::vector<unsigned long long> hh(4);
        hipMemcpy(hh.data(), g_st, 32, hipMemcpyDeviceToHost);
        printf("sustained %.2f s of the run-on chain on all SIMDs (wall clock): %.1f TFLOP/s = %.1f%% of 157.3; in-kernel clock %.3f GHz, %.2f cycles/MFMA\n",
               ms / 1e3, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 157.3e12 * 100, (double)hh[0] / hh[1] * 0.1, (double)hh[0] / (64.0 * big));
    }
    return 0;
}
