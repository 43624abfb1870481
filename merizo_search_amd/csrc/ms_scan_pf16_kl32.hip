// Instantiates the prefilter's fp16-image scan for one list length (see ms_scan_pf16.h).
#include "ms_scan_pf16.h"

int ms_launch_scan_pf16_kl32(const ScanPlan &pl, const ScanParams &sp, hipStream_t st) {
    // (32-entry lists do not fit 256 registers: four waves, one per SIMD)
    return launch_scan_pf16_any<32, 4>(pl, sp, st);
}
