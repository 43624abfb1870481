// Diagnostic probe (not part of the library): what one instruction of each kind costs a wave that is issuing
// v_mfma_f32_32x32x2_f32 back to back into EIGHT independent accumulators (the encoder's edge kernel: no MFMA waits for
// its predecessor), one wave per SIMD.  NF fillers of one kind follow every MFMA; cost = (cycles per MFMA - 64) / NF.
//   hipcc --offload-arch=gfx950 -O3 -o valu_price_probe valu_price_probe.hip && ./valu_price_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
enum { F_NONE, F_XOR, F_FMA, F_PKFMA, F_PKMUL, F_EXP, F_RCP, F_CNDMASK, F_DSREAD128, F_DSWRITE128, F_MAX3, F_DEP_FMA, F_GLDS, F_GLOAD, F_SALU };

template <int KIND, int NF>
__global__ __launch_bounds__(256, 1) void probe(const float *in, float *out, unsigned long long *stamps, int iters) {
    __shared__ __attribute__((aligned(16))) f32x4 lds[4][1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = in[lane * 8 + i]; b[i] = in[(lane * 8 + i + 77) & 4095]; }
    f32x16 acc[8];
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[n][i] = 0.0f;
    // filler operands: independent values so that fillers do not wait for each other (F_DEP_FMA: one dependent chain)
    float x[8], y[8];
    f32x2 p[8], q2[8];
    f32x4 v4[4];
    uint32_t u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = in[(lane + i * 64) & 4095] * 0.001f + 1.0f; y[i] = 1.0001f; p[i] = f32x2{x[i], y[i]}; q2[i] = f32x2{1.0001f, 0.9999f}; u[i] = lane + i; }
#pragma unroll
    for (int i = 0; i < 4; ++i) v4[i] = f32x4{x[i], y[i], x[i + 4], y[i + 4]};
    for (int i = 0; i < 16; ++i) lds[wave][lane * 16 + i] = f32x4{x[i & 7], 0, 0, 0};
    __syncthreads();
    const uint32_t laddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)&lds[wave][0] + lane * 16;
    const uint32_t lds_s = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)&lds[wave][0]);
    const uint32_t goff = lane * 16;
    const unsigned long long gbase = (unsigned long long)(uintptr_t)in;
    uint32_t sal = 0;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#define FILL(i)                                                                                                         \
    {                                                                                                                   \
        if (KIND == F_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[(i) & 7]) : "v"(u[((i) + 1) & 7]));             \
        if (KIND == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[(i) & 7]) : "v"(y[(i) & 7]));                \
        if (KIND == F_DEP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[0]) : "v"(y[0]));                        \
        if (KIND == F_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[(i) & 7]) : "v"(q2[(i) & 7]));          \
        if (KIND == F_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[(i) & 7]) : "v"(q2[(i) & 7]));              \
        if (KIND == F_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(i) & 7]));                                          \
        if (KIND == F_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[(i) & 7]));                                          \
        if (KIND == F_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[(i) & 7]) : "v"(y[(i) & 7]));       \
        if (KIND == F_MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[(i) & 7]) : "v"(y[(i) & 7]), "v"(y[((i) + 1) & 7])); \
        if (KIND == F_DSREAD128) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v4[(i) & 3]) : "v"(laddr), "i"(1024 * ((i) & 7)) : "memory"); \
        if (KIND == F_GLDS) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_s + 1024u * ((i) & 7)), "v"(goff), "s"(gbase) : "memory"); \
        if (KIND == F_GLOAD) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v4[(i) & 3]) : "v"(goff), "s"(gbase), "i"(1024 * ((i) & 3)) : "memory"); \
        if (KIND == F_SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sal));                                            \
        if (KIND == F_DSWRITE128) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(laddr), "v"(v4[(i) & 3]), "i"(1024 * ((i) & 7)) : "memory"); \
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                MFMA(acc[n], a[kk], b[n]);
#pragma unroll
                for (int f = 0; f < NF; ++f) FILL(n * NF + f)
            }
            if (KIND == F_DSREAD128 || KIND == F_DSWRITE128) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (KIND == F_GLDS || KIND == F_GLOAD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int n = 0; n < 8; ++n) s += acc[n][0] + acc[n][7];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i] + p[i].x + p[i].y + __uint_as_float(u[i] & 0xFFFF);
    s += v4[0].x + v4[1].y + v4[2].z + v4[3].w + (float)sal;
    out[blockIdx.x * 256 + tid] = s;
    if (lane == 0) { stamps[(blockIdx.x * 4 + wave) * 2] = c1 - c0; stamps[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0; }
}

static float *g_in, *g_out;
static unsigned long long *g_st;
static const int BLOCKS = 256;
static double g_base = 64.0;

template <int KIND, int NF>
static void run(const char *name, int iters) {
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<KIND, NF>), dim3(BLOCKS), dim3(256), 0, 0, g_in, g_out, g_st, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)BLOCKS * 8);
    hipMemcpy(h.data(), g_st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> c;
    for (int i = 0; i < BLOCKS * 4; ++i) c.push_back((double)h[2 * i] / ((double)iters * 64.0));
    std::sort(c.begin(), c.end());
    const double med = c[c.size() / 2];
    if (KIND == F_NONE) g_base = med;
    printf("%-44s NF=%d: %7.2f cycles per MFMA -> %6.2f cycles per filler\n", name, NF, med, NF ? (med - g_base) / NF : 0.0);
    fflush(stdout);
}

int main() {
    const int iters = 3000;
    hipMalloc(&g_in, 4096 * 4 + 64); hipMalloc(&g_out, BLOCKS * 256 * 4); hipMalloc(&g_st, (size_t)BLOCKS * 8 * 8);
    std::vector<float> h(4096 + 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.0f - 0.5f;
    hipMemcpy(g_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int w = 0; w < 30; ++w) hipLaunchKernelGGL((probe<F_NONE, 0>), dim3(BLOCKS), dim3(256), 0, 0, g_in, g_out, g_st, iters);
    hipDeviceSynchronize();
    printf("fillers beside v_mfma_f32_32x32x2_f32 into 8 independent accumulators, one wave per SIMD (nominal 64 cycles per MFMA)\n");
    run<F_NONE, 0>("no fillers", iters);
#define BOTH(K, N) run<K, 1>(N, iters); run<K, 2>(N, iters); run<K, 4>(N, iters); run<K, 8>(N, iters);
    if (getenv("PROBE_ALL")) {
    BOTH(F_XOR, "v_xor_b32")
    BOTH(F_FMA, "v_fma_f32 (independent)")
    BOTH(F_DEP_FMA, "v_fma_f32 (one dependent chain)")
    BOTH(F_PKFMA, "v_pk_fma_f32")
    BOTH(F_PKMUL, "v_pk_mul_f32")
    BOTH(F_EXP, "v_exp_f32")
    BOTH(F_RCP, "v_rcp_f32")
    BOTH(F_CNDMASK, "v_cndmask_b32")
    BOTH(F_MAX3, "v_max3_f32")
    BOTH(F_DSREAD128, "ds_read_b128")
    }
    BOTH(F_DSWRITE128, "ds_write_b128")
    BOTH(F_GLDS, "global_load_lds_dwordx4 (+ s_mov m0)")
    BOTH(F_GLOAD, "global_load_dwordx4 (L2-resident)")
    BOTH(F_SALU, "s_add_u32")
    return 0;
}
