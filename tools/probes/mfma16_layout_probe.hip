// Probe: operand / result lane layout of v_mfma_f32_16x16x32_bf16 (gfx950).  Assumed: A (16 x 32): lane l holds row l % 16, k = 8 (l / 16) + j;
// B (32 x 16): lane l holds column l % 16, k = 8 (l / 16) + j; D (16 x 16): lane l holds column l % 16, rows 4 (l / 16) + r.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma16_layout_probe tools/probes/mfma16_layout_probe.hip && /tmp/mfma16_layout_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void k(const float *A, const float *B, float *D) {
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A[(l % 16) * 32 + 8 * (l / 16) + j]; b[j] = (__bf16)B[(8 * (l / 16) + j) * 16 + (l % 16)]; }
    f32x4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + (l % 16)] = d[r];
}
int main() {
    float hA[16 * 32], hB[32 * 16], hD[256], ref[256];
    for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7 + 3) % 13 - 6); hB[i] = (float)((i * 5 + 1) % 11 - 5); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int kk = 0; kk < 32; ++kk) s += hA[i * 32 + kk] * hB[kk * 16 + j]; ref[i * 16 + j] = s; }
    float *A, *B, *D; hipMalloc(&A, 2048); hipMalloc(&B, 2048); hipMalloc(&D, 1024);
    hipMemcpy(A, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(B, hB, 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A, B, D); hipMemcpy(hD, D, 1024, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
    printf("16x16x32 bf16 layout as assumed: %s (%d of 256 differ)\n", bad ? "NO" : "YES", bad);
    return bad != 0;
}
