/* A plain-C caller of the C ABI (no Python, no torch): what a maintainer binding the library from
 * another language sees.  Builds a small database, runs ms_l2_normalize_rows + ms_ip_topk +
 * ms_topk_merge on the GPU and checks them against a brute-force scan written here.
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I<repo>/include abi_smoke.c \
 *       -L<repo>/merizo_search_amd -lmerizo_search_amd -L/opt/rocm/lib -lamdhip64 -lm -o abi_smoke */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "merizo_search_amd.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_MS(x) do { int r_ = (x); if (r_ != MS_OK) { fprintf(stderr, "%s failed (%d): %s\n", #x, r_, ms_last_error()); return 3; } } while (0)

static float frand(unsigned *s) { *s = *s * 1664525u + 1013904223u; return (float)((*s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }

int main(void) {
    const int64_t n = 20000;
    const int nq = 37, k = 10, d = MS_DIM;
    unsigned seed = 12345u;
    float *db = (float *)malloc((size_t)n * d * sizeof(float)), *q = (float *)malloc((size_t)nq * d * sizeof(float));
    for (int64_t i = 0; i < n * d; ++i) db[i] = frand(&seed);
    for (int i = 0; i < nq * d; ++i) q[i] = frand(&seed);

    if (ms_version() < 100 || ms_device_count() < 1) { fprintf(stderr, "no device / old library\n"); return 1; }
    float *d_db, *d_q, *d_s, *d_s2;
    int64_t *d_i, *d_i2;
    void *d_ws;
    const size_t ws_bytes = ms_ip_topk_workspace_bytes(n, nq, k);
    CHECK_HIP(hipMalloc((void **)&d_db, (size_t)n * d * 4));
    CHECK_HIP(hipMalloc((void **)&d_q, (size_t)nq * d * 4));
    CHECK_HIP(hipMalloc((void **)&d_s, (size_t)2 * nq * k * 4));
    CHECK_HIP(hipMalloc((void **)&d_i, (size_t)2 * nq * k * 8));
    CHECK_HIP(hipMalloc((void **)&d_s2, (size_t)nq * k * 4));
    CHECK_HIP(hipMalloc((void **)&d_i2, (size_t)nq * k * 8));
    CHECK_HIP(hipMalloc(&d_ws, ws_bytes));
    CHECK_HIP(hipMemcpy(d_db, db, (size_t)n * d * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_q, q, (size_t)nq * d * 4, hipMemcpyHostToDevice));

    /* unit rows on the device, as the faiss-layout database stores them */
    CHECK_MS(ms_l2_normalize_rows(d_db, n, d, 1e-12f, NULL));
    CHECK_MS(ms_l2_normalize_rows(d_q, nq, d, 1e-12f, NULL));
    /* two shards + merge == one scan */
    const int64_t half = n / 2 + 7;
    CHECK_MS(ms_ip_topk(d_db, half, 0, d_q, nq, k, MS_MODE_IP_PRENORM, NULL, NULL, NULL, 0.0f, d_s, d_i, d_ws, ws_bytes, NULL));
    CHECK_MS(ms_ip_topk(d_db + half * d, n - half, half, d_q, nq, k, MS_MODE_IP_PRENORM, NULL, NULL, NULL, 0.0f, d_s + nq * k,
                        d_i + nq * k, d_ws, ws_bytes, NULL));
    CHECK_MS(ms_topk_merge(d_s, d_i, 2, nq, k, d_s2, d_i2, NULL));
    CHECK_MS(ms_ip_topk(d_db, n, 0, d_q, nq, k, MS_MODE_IP_PRENORM, NULL, NULL, NULL, 0.0f, d_s, d_i, d_ws, ws_bytes, NULL));
    CHECK_HIP(hipDeviceSynchronize());

    float *s = (float *)malloc((size_t)nq * k * 4), *s2 = (float *)malloc((size_t)nq * k * 4);
    int64_t *ix = (int64_t *)malloc((size_t)nq * k * 8), *ix2 = (int64_t *)malloc((size_t)nq * k * 8);
    CHECK_HIP(hipMemcpy(s, d_s, (size_t)nq * k * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(ix, d_i, (size_t)nq * k * 8, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(s2, d_s2, (size_t)nq * k * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(ix2, d_i2, (size_t)nq * k * 8, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(db, d_db, (size_t)n * d * 4, hipMemcpyDeviceToHost));     /* the normalised rows */
    CHECK_HIP(hipMemcpy(q, d_q, (size_t)nq * d * 4, hipMemcpyDeviceToHost));

    if (memcmp(s, s2, (size_t)nq * k * 4) != 0 || memcmp(ix, ix2, (size_t)nq * k * 8) != 0) { fprintf(stderr, "sharded != unsharded\n"); return 4; }
    /* brute force on the host: indices must agree, scores within 1e-5 (different summation order) */
    int bad = 0;
    for (int qi = 0; qi < nq && !bad; ++qi) {
        float best_s[16]; int64_t best_i[16];
        for (int j = 0; j < k; ++j) { best_s[j] = -INFINITY; best_i[j] = -1; }
        for (int64_t r = 0; r < n; ++r) {
            double acc = 0.0;
            for (int c = 0; c < d; ++c) acc += (double)db[r * d + c] * (double)q[qi * d + c];
            float v = (float)acc;
            if (v > best_s[k - 1]) {
                int p = k - 1;
                while (p > 0 && best_s[p - 1] < v) { best_s[p] = best_s[p - 1]; best_i[p] = best_i[p - 1]; --p; }
                best_s[p] = v; best_i[p] = r;
            }
        }
        for (int j = 0; j < k; ++j) {
            if (fabsf(best_s[j] - s[qi * k + j]) > 1e-5f) { fprintf(stderr, "score q%d rank %d: %g vs %g\n", qi, j, s[qi * k + j], best_s[j]); bad = 1; }
            if (best_i[j] != ix[qi * k + j] && fabsf(best_s[j] - (j + 1 < k ? best_s[j + 1] : -1.0f)) > 1e-6f &&
                fabsf(best_s[j] - (j > 0 ? best_s[j - 1] : 2.0f)) > 1e-6f) { fprintf(stderr, "index q%d rank %d: %lld vs %lld\n", qi, j, (long long)ix[qi * k + j], (long long)best_i[j]); bad = 1; }
        }
    }
    /* argument errors come back as codes + messages, never as crashes */
    if (ms_ip_topk(NULL, n, 0, d_q, nq, k, MS_MODE_IP_PRENORM, NULL, NULL, NULL, 0.0f, d_s, d_i, d_ws, ws_bytes, NULL) == MS_OK || strlen(ms_last_error()) == 0) bad = 1;
    if (ms_ip_topk(d_db, n, 0, d_q, nq, k, MS_MODE_IP_PRENORM, NULL, NULL, NULL, 0.0f, d_s, d_i, d_ws, 16, NULL) != MS_ERR_WORKSPACE) bad = 1;
    if (bad) return 5;
    printf("abi_smoke ok: %d queries x %lld rows, k=%d, %d CUs\n", nq, (long long)n, k, ms_device_cu_count());
    return 0;
}
