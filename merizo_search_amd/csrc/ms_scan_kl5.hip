// Instantiates the scan kernels of one list length (see ms_scan.h).
#include "ms_scan.h"

int ms_launch_scan_kl5(const ScanPlan &pl, const ScanParams &sp, hipStream_t st) {
    return launch_scan_kl<5, false>(pl, sp, st);
}

int ms_launch_sample_loader(const ScanPlan &pl, const ScanParams &sp, hipStream_t st) {
    const bool aux = sp.inv_norm != nullptr || sp.lengths != nullptr;
    if (!aux) return launch_sample_loader_variant<0>(pl, sp, st);
    return sp.unit_rows ? launch_sample_loader_variant<2>(pl, sp, st) : launch_sample_loader_variant<1>(pl, sp, st);
}
