// Instantiates the prefilter's split-image scan for one list length (see ms_scan_pf.h).
#include "ms_scan_pf.h"

int ms_launch_scan_pf2_kl16(const ScanPlan &pl, const ScanParams &sp, hipStream_t st) {
    if (sp.lengths != nullptr) return sp.qpw == 2 ? launch_scan_pf2<16, 8, true>(pl, sp, st) : launch_scan_pf2<16, 4, true>(pl, sp, st);
    return sp.qpw == 2 ? launch_scan_pf2<16, 8, false>(pl, sp, st) : launch_scan_pf2<16, 4, false>(pl, sp, st);
}
