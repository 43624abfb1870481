// Diagnostic probe (not part of the library): cycles per v_mfma_f32_32x32x2_f32 in the shapes the
// scan kernel uses.  One wave per SIMD on every CU.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_chain_probe mfma_chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(const float *in, float *out, unsigned long long *stamps, int iters) {
    __shared__ f32x4 tile[4][1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    float q[64];
    f32x4 a[16];
    for (int i = 0; i < 64; ++i) q[i] = in[lane * 64 + i];
    for (int i = 0; i < 16; ++i) { tile[wave][lane * 16 + i] = f32x4{in[i], in[i + 1], in[i + 2], in[i + 3]}; }
    __syncthreads();
    for (int i = 0; i < 16; ++i) a[i] = tile[wave][r * 32 + ((16 * h + i) ^ (r & 15))];
    f32x16 acc0, acc1, sum;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0; acc1[i] = 0; sum[i] = 0; }
    const f32x4 *src = &tile[wave][r * 32];
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2 || MODE == 3) {          // single dependent chain of 64
            if (MODE >= 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (int i = 0; i < 16; ++i) acc0[i] = 0;
#pragma unroll
            for (int tt = 0; tt < 16; ++tt) {
                const f32x4 v = a[tt];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, q[4 * tt + 0], acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, q[4 * tt + 1], acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, q[4 * tt + 2], acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, q[4 * tt + 3], acc0, 0, 0, 0);
                if (MODE == 2 && tt >= 8) {                 // refill as the scan does, pinned after each group
                    const int f0 = 2 * (tt - 8);
                    a[f0] = src[(16 * h + f0) ^ (r & 15)];
                    a[f0 + 1] = src[(16 * h + f0 + 1) ^ (r & 15)];
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (MODE == 3 && tt >= 8) {                 // refill, compiler-scheduled
                    const int f0 = 2 * (tt - 8);
                    a[f0] = src[(16 * h + f0) ^ (r & 15)];
                    a[f0 + 1] = src[(16 * h + f0 + 1) ^ (r & 15)];
                }
            }
            for (int i = 0; i < 16; ++i) sum[i] = fmaxf(sum[i], acc0[i]);
        } else {                                             // two interleaved chains of 32
            for (int i = 0; i < 16; ++i) { acc0[i] = 0; acc1[i] = 0; }
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) {
                const f32x4 v = a[tt], w = a[tt + 8];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, q[4 * tt + 0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, q[4 * tt + 32], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, q[4 * tt + 1], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, q[4 * tt + 33], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, q[4 * tt + 2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, q[4 * tt + 34], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, q[4 * tt + 3], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, q[4 * tt + 35], acc1, 0, 0, 0);
            }
            for (int i = 0; i < 16; ++i) sum[i] = fmaxf(sum[i], acc0[i] + acc1[i]);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += sum[i];
    out[blockIdx.x * 256 + tid] = s;
    if (lane == 0) { stamps[(blockIdx.x * 4 + wave) * 2] = c1 - c0; stamps[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0; }
}

typedef float f32x4v __attribute__((ext_vector_type(4)));
// same flops per "tile" (32 rows x 32 queries x 128 k) with v_mfma_f32_16x16x4_f32: 4 independent 16x16
// accumulators, 32 k-steps of 4 -> 128 instructions per tile
__global__ __launch_bounds__(256, 1) void probe16(const float *in, float *out, unsigned long long *stamps, int iters) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float q[64], a[64];
    for (int i = 0; i < 64; ++i) { q[i] = in[lane * 64 + i]; a[i] = in[(lane * 37 + i) & 4095]; }
    f32x4v acc[4], sum;
    for (int i = 0; i < 4; ++i) sum[i] = 0;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        for (int b = 0; b < 4; ++b) for (int i = 0; i < 4; ++i) acc[b][i] = 0;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], q[s], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], q[32 + s], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[32 + s], q[s], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[32 + s], q[32 + s], acc[3], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) sum[i] = fmaxf(sum[i], acc[0][i] + acc[1][i] + acc[2][i] + acc[3][i]);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + tid] = sum[0] + sum[1] + sum[2] + sum[3];
    if (lane == 0) { stamps[(blockIdx.x * 4 + wave) * 2] = c1 - c0; stamps[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0; }
}

// the single chain with the accumulator forced into AGPRs (inline asm, "+a" constraint)
__global__ __launch_bounds__(256, 1) void probe_agpr(const float *in, float *out, unsigned long long *stamps, int iters) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float q[64], a[64];
    for (int i = 0; i < 64; ++i) { q[i] = in[lane * 64 + i]; a[i] = in[(lane * 37 + i) & 4095]; }
    f32x16 acc, sum;
    for (int i = 0; i < 16; ++i) sum[i] = 0;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        for (int i = 0; i < 16; ++i) acc[i] = 0;
#pragma unroll
        for (int s = 0; s < 64; ++s) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a[s]), "v"(q[s]));
        for (int i = 0; i < 16; ++i) sum[i] = fmaxf(sum[i], acc[i]);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sres = 0;
    for (int i = 0; i < 16; ++i) sres += sum[i];
    out[blockIdx.x * 256 + tid] = sres;
    if (lane == 0) { stamps[(blockIdx.x * 4 + wave) * 2] = c1 - c0; stamps[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0; }
}

template <int MODE>
void run(const char *name, float *in, float *out, unsigned long long *st, int blocks, int iters) {
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, st, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), st, blocks * 8 * 8, hipMemcpyDeviceToHost);
    std::vector<double> cpm, ghz;
    for (int i = 0; i < blocks * 4; ++i) { cpm.push_back((double)h[2 * i] / (64.0 * iters)); ghz.push_back((double)h[2 * i] / h[2 * i + 1] * 0.1); }
    std::sort(cpm.begin(), cpm.end()); std::sort(ghz.begin(), ghz.end());
    printf("%-44s cycles/MFMA median %.2f max %.2f | clock %.3f GHz\n", name, cpm[cpm.size() / 2], cpm.back(), ghz[ghz.size() / 2]);
}

int main() {
    float *in, *out; unsigned long long *st;
    const int blocks = 256, iters = 2000;
    hipMalloc(&in, 64 * 64 * 4 + 64); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&st, blocks * 8 * 8);
    std::vector<float> h(64 * 64 + 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.0f - 0.5f;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0>("single chain of 64", in, out, st, blocks, iters);
    run<1>("two interleaved chains of 32", in, out, st, blocks, iters);
    run<2>("single chain + 16 ds_read_b128 (pinned)", in, out, st, blocks, iters);
    run<3>("single chain + 16 ds_read_b128 (compiler)", in, out, st, blocks, iters);
    {
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(probe16, dim3(blocks), dim3(256), 0, 0, in, out, st, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * 8);
        hipMemcpy(h.data(), st, blocks * 8 * 8, hipMemcpyDeviceToHost);
        printf("16x16x4, four interleaved accumulators:      cycles per tile (128 MFMA) %.1f (32x32x2: 64 MFMA x the figure above; nominal 4096) | clock %.3f GHz\n",
               (double)h[0] / iters, (double)h[0] / h[1] * 0.1);
    }
    {
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(probe_agpr, dim3(blocks), dim3(256), 0, 0, in, out, st, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * 8);
        hipMemcpy(h.data(), st, blocks * 8 * 8, hipMemcpyDeviceToHost);
        printf("single chain, accumulator in AGPRs:           cycles/MFMA %.2f | clock %.3f GHz\n", (double)h[0] / (64.0 * iters), (double)h[0] / h[1] * 0.1);
    }
    // sustained rate: ~2 s of back-to-back launches of the pure chain, wall-clocked with events
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const int big = 20000, launches = 300;
        for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, in, out, st, big);
        hipEventRecord(e0);
        for (int w = 0; w < launches; ++w) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, in, out, st, big);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)launches * blocks * 4 * (double)big * 64 * 4096.0;
        std::vector<unsigned long long> h(blocks * 8);
        hipMemcpy(h.data(), st, blocks * 8 * 8, hipMemcpyDeviceToHost);
        printf("sustained %.2f s of the single chain on all SIMDs: %.1f TFLOP/s (%.1f%% of 157.3), in-kernel clock %.3f GHz, %.2f cycles/MFMA\n",
               ms / 1e3, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 157.3e12 * 100, (double)h[0] / h[1] * 0.1, (double)h[0] / (64.0 * big));
    }
    return 0;
}
