// Instantiates the prefilter's split-image scan for one list length (see ms_scan_pf.h).
#include "ms_scan_pf.h"

int ms_launch_scan_pf2_kl16(const ScanPlan &pl, const ScanParams &sp, hipStream_t st) {
    return sp.qpw == 2 ? launch_scan_pf2<16, 8>(pl, sp, st) : launch_scan_pf2<16, 4>(pl, sp, st);
}
