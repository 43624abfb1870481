// Where does the immediate offset of global_load_lds_dwordx4 go: the global address only, or the LDS address too?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned *src, unsigned *out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned *l = reinterpret_cast<unsigned *>(smem);
    for (int i = threadIdx.x; i < 1024; i += 64) l[i] = 0xFFFFFFFFu;
    __syncthreads();
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem);
    const unsigned voff = threadIdx.x * 16;
    const unsigned long long sb = (unsigned long long)src;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:32\n\ts_waitcnt vmcnt(0)" ::"s"(lds0), "v"(voff), "s"(sb) : "memory", "m0");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = l[i];
}
int main() {
    unsigned *src, *out;
    hipMalloc(&src, 8192); hipMalloc(&out, 4096);
    std::vector<unsigned> h(2048);
    for (int i = 0; i < 2048; ++i) h[i] = i;
    hipMemcpy(src, h.data(), 8192, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, src, out);
    std::vector<unsigned> o(1024);
    hipMemcpy(o.data(), out, 4096, hipMemcpyDeviceToHost);
    int first = -1;
    for (int i = 0; i < 1024; ++i) if (o[i] != 0xFFFFFFFFu) { first = i; break; }
    printf("first LDS dword written: %d (byte %d), holds source dword %u (byte %u)\n", first, first * 4, o[first], o[first] * 4);
    printf("=> LDS address %s the immediate; global address %s it\n", first == 8 ? "INCLUDES" : "does not include", o[first] == 8 ? "includes" : "does NOT include");
    return 0;
}
