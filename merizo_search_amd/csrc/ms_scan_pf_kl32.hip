// Instantiates the prefilter's split-image scan for one list length (see ms_scan_pf.h).
#include "ms_scan_pf.h"

int ms_launch_scan_pf2_kl32(const ScanPlan &pl, const ScanParams &sp, hipStream_t st) {
    // (32-entry lists do not fit 256 registers: four waves, one per SIMD)
    return sp.lengths != nullptr ? launch_scan_pf2<32, 4, true>(pl, sp, st) : launch_scan_pf2<32, 4, false>(pl, sp, st);
}
