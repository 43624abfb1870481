// Instantiates the prefilter's fp16-image scan for one list length (see ms_scan_pf16.h).
#include "ms_scan_pf16.h"

int ms_launch_scan_pf16_kl16(const ScanPlan &pl, const ScanParams &sp, hipStream_t st) {
    return sp.qpw == 2 ? launch_scan_pf16_any<16, 8>(pl, sp, st) : launch_scan_pf16_any<16, 4>(pl, sp, st);
}
