// Instantiates the prefilter's fp16-image scan for one list length (see ms_scan_pf16.h).
#include "ms_scan_pf16.h"

int ms_launch_scan_pf16_kl5(const ScanPlan &pl, const ScanParams &sp, hipStream_t st) {
    return sp.qpw == 2 ? launch_scan_pf16_any<5, 8>(pl, sp, st) : launch_scan_pf16_any<5, 4>(pl, sp, st);
}

template <int NW, bool MASK, int NQP>
static int launch_sample_pf16(const ScanPlan &pl, const ScanParams &sp, hipStream_t st) {
    MS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ms_scan_pf16_kernel<5, NW, true, MASK, NQP>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)PF2_LDS));
    hipLaunchKernelGGL((ms_scan_pf16_kernel<5, NW, true, MASK, NQP>), dim3(pl.grid), dim3(64 * NW), PF2_LDS, st, sp);
    MS_LAUNCH_CHECK("ms_scan_pf16_kernel (sample)");
    return MS_OK;
}
template <int NW>
static int launch_sample_pf16_any(const ScanPlan &pl, const ScanParams &sp, hipStream_t st) {
    const bool two = sp.pf_format == MS_PF_F16X2;
    if (sp.lengths != nullptr) return two ? launch_sample_pf16<NW, true, 2>(pl, sp, st) : launch_sample_pf16<NW, true, 1>(pl, sp, st);
    return two ? launch_sample_pf16<NW, false, 2>(pl, sp, st) : launch_sample_pf16<NW, false, 1>(pl, sp, st);
}

int ms_launch_sample_pf16(const ScanPlan &pl, const ScanParams &sp, hipStream_t st) {
    return sp.qpw == 2 ? launch_sample_pf16_any<8>(pl, sp, st) : launch_sample_pf16_any<4>(pl, sp, st);
}

// ---- the image itself: one wave per half tile; lane (r, h) reads its half row (256 B), scales it by 2^sr (exact), rounds to fp16 (to
//      nearest even, clamped to the fp16 range) and writes eight 16-byte fragments, each store instruction of the wave one contiguous KiB
__global__ __launch_bounds__(256) void ms_pf16_build_image_kernel(const float *db, int64_t n, char *image, int64_t ntiles, int sr) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const float scale = __uint_as_float((uint32_t)(127 + sr) << 23);
    for (int64_t U = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); U < 2 * ntiles; U += (int64_t)gridDim.x * 4) {
        const int64_t T = U >> 1;
        const int half = (int)(U & 1);
        const int64_t row = T * 64 + 32 * half + r;
        f32x4 x[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            x[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (row < n) x[t] = *reinterpret_cast<const f32x4 *>(db + row * MS_DIM + 64 * h + 4 * t);
        }
        char *dst = image + T * 16384 + 16 * lane;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const f32x4 a0 = x[2 * b] * scale, a1 = x[2 * b + 1] * scale;
            const float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#ifdef MS_PF16_ABL_BF16
                o[j] = __builtin_bit_cast(_Float16, (__bf16)fminf(fmaxf(v[j], -65504.0f), 65504.0f));      // (diagnostic build: a bf16 image in the same layout)
#elif defined(MS_PF16_ABL_F16R8)
                o[j] = (_Float16)(float)(__bf16)fminf(fmaxf(v[j], -65504.0f), 65504.0f);                   // (diagnostic build: fp16 values with 8 significant bits)
#else
                o[j] = (_Float16)fminf(fmaxf(v[j], -65504.0f), 65504.0f);
#endif
            }
            *reinterpret_cast<f16x8 *>(dst + 1024 * (2 * b + half)) = o;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {          // the trailer: {magic, sr, n, 0 ...}
        uint32_t *tr = reinterpret_cast<uint32_t *>(image + ntiles * 16384);
        uint32_t w = 0u;
        if (threadIdx.x == 0) w = MS_PF16_MAGIC;
        if (threadIdx.x == 1) w = (uint32_t)sr;
        if (threadIdx.x == 2) w = (uint32_t)(n & 0xFFFFFFFFll);
        if (threadIdx.x == 3) w = (uint32_t)(n >> 32);
        tr[threadIdx.x] = w;
    }
}

int ms_launch_pf16_build_image(const float *db, int64_t n, int sr, void *image, hipStream_t st) {
    const int64_t ntiles = (n + 63) / 64;
    int64_t blocks = (2 * ntiles + 3) / 4;
    if (blocks < 1) blocks = 1;                      // (n == 0: one workgroup writes the trailer of an image without tiles)
    hipLaunchKernelGGL(ms_pf16_build_image_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st, db, n, (char *)image, ntiles, sr);
    MS_LAUNCH_CHECK("ms_pf16_build_image_kernel");
    return MS_OK;
}
