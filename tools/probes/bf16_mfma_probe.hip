// Probe: cycles per v_mfma_f32_32x32x16_bf16 in dependent chains of 1 / 2 / 4 accumulators, operands in registers, one wave per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/bf16_mfma_probe tools/probes/bf16_mfma_probe.hip && /tmp/bf16_mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(256, 1) void k(unsigned long long *out, float *sink, int iters) {
    u32x4 av = {0x3f803f80u + threadIdx.x, 0x3f803f81u, 0x3f823f80u, 0x3f803f83u}, bv = {0x3f803f80u, 0x3f813f80u, 0x3f803f82u, 0x3f833f80u};
    bf16x8 a = __builtin_bit_cast(bf16x8, av), b = __builtin_bit_cast(bf16x8, bv);
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 24 / CH; ++j)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int c = 0; c < CH; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
int main() {
    unsigned long long *o; float *sink; hipMalloc(&o, 1024 * 8); hipMalloc(&sink, 1024 * 256 * 4);
    unsigned long long h[1024];
    const int iters = 2000;
#define RUN(CH) { hipLaunchKernelGGL(k<CH>, dim3(256), dim3(256), 0, 0, o, sink, iters); hipDeviceSynchronize(); hipLaunchKernelGGL(k<CH>, dim3(256), dim3(256), 0, 0, o, sink, iters); hipDeviceSynchronize(); \
    hipMemcpy(h, o, 256 * 8, hipMemcpyDeviceToHost); double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; printf("%d chain(s): %.1f cycles per v_mfma_f32_32x32x16_bf16\n", CH, m / 256 / iters / 24); }
    RUN(1) RUN(2) RUN(4)
    return 0;
}
