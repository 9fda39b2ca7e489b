// probe: do kernels that take a whole CU's LDS (160 KB per work-group, like smc_trace) from different
// HIP streams run concurrently on gfx950?  Each launch: 64 work-groups spinning ~50 ms.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(768) void spin(unsigned long long ticks, int *out)
{
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { __builtin_amdgcn_s_sleep(32); }
    if (threadIdx.x == 0) out[blockIdx.x] = (int)lds[5];
}
int main()
{
    const int nstream = 4, lds = 160 * 1024 - 64;
    hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    int *o; hipMalloc(&o, 4096 * 4);
    std::vector<hipStream_t> st(nstream);
    for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int mode = 0; mode < 3; mode++) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < nstream; k++) {
            hipStream_t s = (mode == 0) ? st[0] : st[k];
            if (mode == 2) hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            hipLaunchKernelGGL(spin, dim3(64), dim3(768), lds, s, 5000000ULL /*50 ms*/, o + 64 * k);
        }
        hipDeviceSynchronize();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("%s: %d launches x 64 WG x 50 ms -> %.1f ms\n", mode == 0 ? "one stream      " : (mode == 1 ? "four streams    " : "four streams+attr"), nstream, ms);
    }
    return 0;
}
