// Instantiates the scan kernels of one list length (see ms_scan.h).
#include "ms_scan.h"

int ms_launch_scan_kl16(const ScanPlan &pl, const ScanParams &sp, hipStream_t st) {
    return launch_scan_kl<16, false>(pl, sp, st);
}
