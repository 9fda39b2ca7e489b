// What a one-shot process pays before its first useful HIP call: runtime start-up, the first stream, the first kernel launch of a
// library (code-object load).   hipcc --offload-arch=gfx950 -O2 -o hip_init hip_init.cpp && ./hip_init [path/to/libpnr_hip.so]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <dlfcn.h>
__global__ void nop(int *p) { if (p) *p = 1; }
int main(int argc, char **argv)
{
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = clk::now();
    void *lib = argc > 1 ? dlopen(argv[1], RTLD_NOW) : nullptr;
    const auto t1 = clk::now();
    int n = 0;
    hipGetDeviceCount(&n);
    const auto t2 = clk::now();
    hipSetDevice(0);
    hipStream_t s;
    hipStreamCreate(&s);
    const auto t3 = clk::now();
    int *d = nullptr;
    hipMalloc(&d, 8);
    const auto t4 = clk::now();
    hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, s, d);
    hipStreamSynchronize(s);
    const auto t5 = clk::now();
    printf("dlopen %.1f ms | hipGetDeviceCount (runtime start-up) %.1f ms | hipSetDevice + hipStreamCreate %.1f ms | first hipMalloc %.1f ms | first launch + sync %.1f ms | total %.1f ms (%d devices, lib %s)\n",
           ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, t5), ms(t0, t5), n, lib ? "loaded" : "none");
    return 0;
}
