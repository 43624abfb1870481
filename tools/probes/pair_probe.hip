// Diagnostic probe (not part of the library): the "paired waves" form of the scan loop.  Per SIMD one MFMA wave (A) that
// issues nothing but the dependent v_mfma_f32_32x32x2_f32 chain, the LDS reads that refill its A fragments (immediate
// offsets, chunk-major tile image, 4-slot ring unrolled x4) and four ds_write_b128 that hand the previous tile's raw
// scores to its partner; and one filter wave (B) that polls, reads the scores back, folds them into a running maximum
// (8 v_max3), spends NVB more vector instructions per tile (insertion steps / cosine scaling stand-in) and issues NDMA
// LDS-DMA pieces per tile (the loader's duty split over the four B waves).  Question: how many cycles per tile does A
// need with B beside it on the same SIMD?  (mfma_price_probe: a vector instruction inside A's own chain costs 8-15 cycles.)
//
//   hipcc --offload-arch=gfx950 -O3 -o pair_probe pair_probe.hip && ./pair_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA_Z(acc, a, b) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b))
#define VMAX3(m, a, b) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(a), "v"(b))
#define VFILL(x, y) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(y))

constexpr int TILE_RING = 4, SCORE_RING = 4;
constexpr int LDS_TILES = TILE_RING * 16384;                 // 64 KiB
constexpr int LDS_SCORES = 4 * SCORE_RING * 4096;            // 64 KiB: [pair][slot][4 groups][64 lanes] float4
constexpr int LDS_TOTAL = LDS_TILES + LDS_SCORES + 256;

template <int NVB, int NDMA, bool WITH_B, int PRIO>
__global__ __launch_bounds__(512, 2) void probe_pair(const float *in, const float *gbuf, float *out, unsigned long long *stamps, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const int pair = wave & 3;
    const uint32_t score0 = lds0 + LDS_TILES + pair * (SCORE_RING * 4096) + lane * 16;      // + slot * 4096 + g * 1024
    const uint32_t cnt0 = lds0 + LDS_TILES + LDS_SCORES;                                      // produced[4], consumed[4] (dwords)
    if (PRIO > 0 && wave >= 4) __builtin_amdgcn_s_setprio(PRIO);       // before A starts its chain: a starved wave cannot even raise its priority
    // fill the tile ring with valid floats, zero the counters
    for (int i = tid; i < LDS_TILES / 4; i += 512) reinterpret_cast<float *>(smem)[i] = in[i & 4095];
    if (tid < 64) reinterpret_cast<uint32_t *>(smem + LDS_TILES + LDS_SCORES)[tid] = 0;
    __syncthreads();
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned long long c0 = 0, c1 = 0, r0 = 0, r1 = 0;
    unsigned long long *prog = stamps + (size_t)gridDim.x * 8 * 4;      // block 0: B progress stamps [4][8], A start / end [4][2]
    float result = 0.0f;
    if (wave < 4) {
        // ------------------------------------------------ A: the MFMA wave
        float q[64];
        f32x4 a[16];
#pragma unroll
        for (int i = 0; i < 64; ++i) q[i] = in[lane * 64 + i];
        const uint32_t lin0 = lds0 + (uint32_t)(16 * (512 * h + r));                          // chunk-major image: + slot * 16384 + 512 f
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[i]) : "v"(lin0), "i"(512 * i) : "memory");
        f32x16 accA, accB;
#pragma unroll
        for (int i = 0; i < 16; ++i) { accA[i] = 0.0f; accB[i] = 0.0f; }
        const uint32_t prod_addr = cnt0 + 4 * pair, cons_addr = cnt0 + 16 + 4 * pair;
        uint32_t one = 1, flag = 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
#define A_GROUP(ACC, FIRST, tt)                                                                                   \
    {                                                                                                             \
        if (FIRST) { MFMA_Z(ACC, a[tt].x, q[4 * (tt) + 0]); } else { MFMA(ACC, a[tt].x, q[4 * (tt) + 0]); }        \
        MFMA(ACC, a[tt].y, q[4 * (tt) + 1]); MFMA(ACC, a[tt].z, q[4 * (tt) + 2]); MFMA(ACC, a[tt].w, q[4 * (tt) + 3]); \
    }
#define A_STORE(PREV, g, SSLOT)                                                                                   \
    {                                                                                                             \
        const f32x4 v_ = {PREV[4 * (g)], PREV[4 * (g) + 1], PREV[4 * (g) + 2], PREV[4 * (g) + 3]};                \
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(score0), "v"(v_), "i"((SSLOT) * 4096 + (g) * 1024) : "memory"); \
    }
        // tile t: chain into ACC; the previous tile's scores (PREV) go to score slot SSLOT in groups 1..4, then the produced
        // counter is bumped (lane 0 only, EXEC set by scalar moves); the partner's consumed counter is read at group 5 and
        // looked at at group 8 (back-pressure); groups 8..15 refill a[] from tile slot TSLOT
#define A_TILE(ACC, PREV, TSLOT, SSLOT)                                                                           \
    {                                                                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                        \
        _Pragma("unroll") for (int tt = 0; tt < 16; ++tt) {                                                       \
            if (tt == 0) A_GROUP(ACC, true, tt) else A_GROUP(ACC, false, tt)                                      \
            if (WITH_B && tt >= 1 && tt < 5) A_STORE(PREV, tt - 1, SSLOT)                                         \
            if (WITH_B && tt == 5) {                                                                              \
                asm volatile("s_mov_b64 exec, 1\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(prod_addr), "v"(one) : "memory"); \
                asm volatile("ds_read_b32 %0, %1" : "=v"(flag) : "v"(cons_addr) : "memory");                      \
            }                                                                                                     \
            if (WITH_B && tt == 8) {                                                                              \
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(flag)::"memory");                                      \
                if (__builtin_amdgcn_readfirstlane(flag) + SCORE_RING < (uint32_t)(t_done + 1)) stalls += 1;      \
            }                                                                                                     \
            if (tt >= 8) {                                                                                        \
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[2 * (tt - 8)]) : "v"(lin0), "i"((TSLOT) * 16384 + 512 * (2 * (tt - 8))) : "memory"); \
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[2 * (tt - 8) + 1]) : "v"(lin0), "i"((TSLOT) * 16384 + 512 * (2 * (tt - 8) + 1)) : "memory"); \
            }                                                                                                     \
        }                                                                                                         \
        t_done += 1;                                                                                              \
    }
        int t_done = 0, stalls = 0;
        for (int it = 0; it < iters; it += 4) {
            A_TILE(accA, accB, 1, 3)
            A_TILE(accB, accA, 2, 0)
            A_TILE(accA, accB, 3, 1)
            A_TILE(accB, accA, 0, 2)
        }
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        c1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int i = 0; i < 16; ++i) result += accA[i] + accB[i] + a[i].x;
        result += (float)stalls;
        if (lane == 0) stamps[(blockIdx.x * 8 + wave) * 4 + 3] = (unsigned long long)stalls;
        if (lane == 0 && blockIdx.x == 0) { prog[32 + wave * 2] = c0; prog[32 + wave * 2 + 1] = c1; }
    } else if (WITH_B) {
        // ------------------------------------------------ B: the filter / loader wave
        const uint32_t prod_addr = cnt0 + 4 * pair, cons_addr = cnt0 + 16 + 4 * pair;
        float mx = -1e30f;
        uint32_t fill = lane, fill2 = 3 * lane, one = 1;
        const char *gsrc = reinterpret_cast<const char *>(gbuf) + (size_t)blockIdx.x * 65536 + lane * 16;
        c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long polls = 0;
        for (int t = 0; t < iters; ++t) {
            if ((t & 1023) == 0 && lane == 0 && blockIdx.x == 0) prog[(wave - 4) * 8 + (t >> 10)] = __builtin_amdgcn_s_memtime();
            // loader duty: NDMA pieces of 1 KiB into tile slot (t + 3) % 4 (this wave's quarter of the tile)
#pragma unroll
            for (int p = 0; p < NDMA; ++p) {
                const uint32_t dst = lds0 + (uint32_t)(((t + 3) & 3) * 16384 + (pair * 4 + p) * 1024);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(gsrc + ((t & 15) * 4096 + p * 1024)) : "memory");
            }
            if (NDMA > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * NDMA) : "memory");
            // filter duty: wait for the partner's tile t
            uint32_t v;
            for (;;) {
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(prod_addr) : "memory");
                if (__builtin_amdgcn_readfirstlane(v) >= (uint32_t)(t + 1)) break;
                polls += 1;
                __builtin_amdgcn_s_sleep(4);
            }
            f32x4 s0, s1, s2, s3;
            const int sslot = (t + 3) & 3;          // A's tile t (t_done = t) wrote slot SSLOT of its unrolled position
            asm volatile("ds_read_b128 %0, %1" : "=v"(s0) : "v"(score0 + sslot * 4096) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(s1) : "v"(score0 + sslot * 4096) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(s2) : "v"(score0 + sslot * 4096) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:3072\n\ts_waitcnt lgkmcnt(0)" : "=v"(s3) : "v"(score0 + sslot * 4096) : "memory");
            asm volatile("" : "+v"(s0), "+v"(s1), "+v"(s2));
            VMAX3(mx, s0.x, s0.y); VMAX3(mx, s0.z, s0.w); VMAX3(mx, s1.x, s1.y); VMAX3(mx, s1.z, s1.w);
            VMAX3(mx, s2.x, s2.y); VMAX3(mx, s2.z, s2.w); VMAX3(mx, s3.x, s3.y); VMAX3(mx, s3.z, s3.w);
#pragma unroll
            for (int i = 0; i < NVB; i += 2) { VFILL(fill, fill2); VFILL(fill2, fill); }
            asm volatile("s_mov_b64 exec, 1\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(cons_addr), "v"(one) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        c1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        result = mx + __uint_as_float((fill ^ fill2) & 0x3FFFFFu);
        if (lane == 0) stamps[(blockIdx.x * 8 + wave) * 4 + 3] = polls;
    }
    out[blockIdx.x * 512 + tid] = result;
    if (lane == 0) {
        stamps[(blockIdx.x * 8 + wave) * 4 + 0] = c1 - c0;
        stamps[(blockIdx.x * 8 + wave) * 4 + 1] = r1 - r0;
        stamps[(blockIdx.x * 8 + wave) * 4 + 2] = hwid;
    }
}

static float *g_in, *g_gbuf, *g_out;
static unsigned long long *g_st;
static const int BLOCKS = 256;

template <int NVB, int NDMA, bool WITH_B, int PRIO = 0>
static void run(const char *name, int iters) {
    auto k = probe_pair<NVB, NDMA, WITH_B, PRIO>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(BLOCKS), dim3(512), LDS_TOTAL, 0, g_in, g_gbuf, g_out, g_st, iters);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: %s\n", name, hipGetErrorString(e)); return; }
    std::vector<unsigned long long> h((size_t)BLOCKS * 8 * 4 + 64);
    hipMemcpy(h.data(), g_st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ca, cb, ghz;
    double stalls = 0, polls = 0;
    for (int b = 0; b < BLOCKS; ++b)
        for (int w = 0; w < 8; ++w) {
            const unsigned long long *s = &h[((size_t)b * 8 + w) * 4];
            if (w < 4) { ca.push_back((double)s[0] / iters); ghz.push_back((double)s[0] / (double)s[1] * 0.1); stalls += (double)s[3]; }
            else if (WITH_B) { cb.push_back((double)s[0] / iters); polls += (double)s[3]; }
        }
    std::sort(ca.begin(), ca.end()); std::sort(ghz.begin(), ghz.end());
    printf("%-62s A: %7.1f cycles/tile (min %.1f max %.1f) = %5.1f%% of nominal", name, ca[ca.size() / 2], ca.front(), ca.back(), 4096.0 / ca[ca.size() / 2] * 100.0);
    if (WITH_B) { std::sort(cb.begin(), cb.end()); printf(" | B: %7.1f, polls/tile %.2f, A stalls/tile %.4f", cb[cb.size() / 2], polls / (BLOCKS * 4.0 * iters), stalls / (BLOCKS * 4.0 * iters)); }
    printf(" | clock %.3f GHz\n", ghz[ghz.size() / 2]);
    if (WITH_B) {
        const unsigned long long *pg = &h[(size_t)BLOCKS * 8 * 4];
        printf("  block 0 pair 0: A runs [0, %llu]; B reaches tile 0 / 1024 / 2048 / 3072 at %lld / %lld / %lld / %lld\n", pg[33] - pg[32],
               (long long)(pg[0] - pg[32]), (long long)(pg[1] - pg[32]), (long long)(pg[2] - pg[32]), (long long)(pg[3] - pg[32]));
    }
    static bool shown = NVB + NDMA + PRIO != 0;
    if (!shown) {
        shown = true;
        printf("  block 0 wave -> SIMD (HW_ID bits 5:4):");
        for (int w = 0; w < 8; ++w) printf(" w%d:%llu", w, (h[(size_t)w * 4 + 2] >> 4) & 3);
        printf("\n");
    }
    fflush(stdout);
}

int main() {
    const int iters = 4000;
    hipMalloc(&g_in, 4096 * 4 + 64); hipMalloc(&g_out, BLOCKS * 512 * 4); hipMalloc(&g_st, (size_t)BLOCKS * 8 * 4 * 8 + 64 * 8);
    hipMalloc(&g_gbuf, (size_t)BLOCKS * 65536 + 65536 + 4096);
    std::vector<float> h(4096 + 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.0f - 0.5f;
    hipMemcpy(g_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> g(((size_t)BLOCKS * 65536 + 65536 + 4096) / 4);
    for (size_t i = 0; i < g.size(); ++i) g[i] = h[i & 4095];
    hipMemcpy(g_gbuf, g.data(), g.size() * 4, hipMemcpyHostToDevice);
    printf("paired waves: per SIMD one MFMA wave (A) + one filter / loader wave (B); %d tiles per wave; nominal 4096 cycles per tile\n", iters);
    run<0, 0, false>("A alone (chain + 16 ds_read_b128 imm), B idle", iters);
    run<0, 0, true>("A + B: scores through LDS, 8 v_max3 in B", iters);
    run<0, 4, true>("A + B: + 4 LDS-DMA pieces per tile in each B", iters);
    run<32, 4, true>("A + B: + 32 more VALU per tile in B", iters);
    run<128, 4, true>("A + B: + 128 more VALU per tile in B", iters);
    run<512, 4, true>("A + B: + 512 more VALU per tile in B", iters);
    run<0, 4, true, 1>("B at s_setprio 1: 4 DMA pieces, 8 v_max3", iters);
    run<0, 4, true, 3>("B at s_setprio 3: 4 DMA pieces, 8 v_max3", iters);
    run<32, 4, true, 3>("B at s_setprio 3: + 32 more VALU per tile in B", iters);
    run<128, 4, true, 3>("B at s_setprio 3: + 128 more VALU per tile in B", iters);
    run<512, 4, true, 3>("B at s_setprio 3: + 512 more VALU per tile in B", iters);
    return 0;
}
