// Instantiates the prefilter's split-image scan for one list length (see ms_scan_pf.h).
#include "ms_scan_pf.h"

int ms_launch_scan_pf2_kl5(const ScanPlan &pl, const ScanParams &sp, hipStream_t st) {
    if (sp.lengths != nullptr) return sp.qpw == 2 ? launch_scan_pf2<5, 8, true>(pl, sp, st) : launch_scan_pf2<5, 4, true>(pl, sp, st);
    return sp.qpw == 2 ? launch_scan_pf2<5, 8, false>(pl, sp, st) : launch_scan_pf2<5, 4, false>(pl, sp, st);
}

template <int NW, bool MASK>
static int launch_sample_pf2(const ScanPlan &pl, const ScanParams &sp, hipStream_t st) {
    MS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ms_scan_pf2_kernel<5, NW, true, MASK>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)PF2_LDS));
    hipLaunchKernelGGL((ms_scan_pf2_kernel<5, NW, true, MASK>), dim3(pl.grid), dim3(64 * NW), PF2_LDS, st, sp);
    MS_LAUNCH_CHECK("ms_scan_pf2_kernel (sample)");
    return MS_OK;
}

int ms_launch_sample_pf2(const ScanPlan &pl, const ScanParams &sp, hipStream_t st) {
    if (sp.lengths != nullptr) return sp.qpw == 2 ? launch_sample_pf2<8, true>(pl, sp, st) : launch_sample_pf2<4, true>(pl, sp, st);
    return sp.qpw == 2 ? launch_sample_pf2<8, false>(pl, sp, st) : launch_sample_pf2<4, false>(pl, sp, st);
}

// ---- the image itself: one wave per tile; lane (r, h) reads its half row (256 B) and writes sixteen 16-byte fragments, each
//      store instruction of the wave one contiguous KiB
__global__ __launch_bounds__(256) void ms_pf_build_image_kernel(const float *db, int64_t n, char *image, int64_t ntiles) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    for (int64_t T = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); T < ntiles; T += (int64_t)gridDim.x * 4) {
        const int64_t row = T * 32 + r;
        f32x4 x[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            x[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (row < n) x[t] = *reinterpret_cast<const f32x4 *>(db + row * MS_DIM + 64 * h + 4 * t);
        }
        char *dst = image + T * 16384 + 16 * lane;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            bf16x8 hi, lo;
            ms_split8(x[2 * b], x[2 * b + 1], hi, lo);
            *reinterpret_cast<bf16x8 *>(dst + 1024 * (2 * b)) = hi;
            *reinterpret_cast<bf16x8 *>(dst + 1024 * (2 * b + 1)) = lo;
        }
    }
}

int ms_launch_pf_build_image(const float *db, int64_t n, void *image, hipStream_t st) {
    const int64_t ntiles = (n + 31) / 32;
    const int64_t blocks = (ntiles + 3) / 4;
    hipLaunchKernelGGL(ms_pf_build_image_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st, db, n, (char *)image, ntiles);
    MS_LAUNCH_CHECK("ms_pf_build_image_kernel");
    return MS_OK;
}
