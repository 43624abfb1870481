// Instantiates the scan kernels of one list length (see ms_scan.h).
#include "ms_scan.h"

int ms_launch_scan_kl32(const ScanPlan &pl, const ScanParams &sp, hipStream_t st) {
    return launch_scan_kl<32, false>(pl, sp, st);
}
