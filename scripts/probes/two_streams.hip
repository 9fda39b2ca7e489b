// Can a kernel on stream B run WHILE a kernel on stream A spins on a flag that B's kernel sets?  (Needs the two streams on different
// hardware queues.)  Every spin has a 200 ms watchdog.  usage: two_streams [streams_created_before]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void spinner(int *flag, int *result, long long limit)
{
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
        __builtin_amdgcn_s_sleep(32);
        if (wall_clock64() - t0 > limit) { *result = -1; return; }
    }
    *result = 1;
}
__global__ void setter(int *flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
int main(int argc, char **argv)
{
    const int pre = argc > 1 ? atoi(argv[1]) : 0;
    hipStream_t dummy[16], a, b;
    for (int i = 0; i < pre && i < 16; i++) hipStreamCreateWithFlags(&dummy[i], hipStreamNonBlocking);
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    int *flag, *res;
    hipMalloc(&flag, 4); hipMalloc(&res, 4);
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeWallClockRate, 0); // kHz
    for (int rep = 0; rep < 4; rep++) {
        hipMemset(flag, 0, 4); hipMemset(res, 0, 4);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(spinner, dim3(1), dim3(64), 0, a, flag, res, (long long)clk * 200);
        hipLaunchKernelGGL(setter, dim3(1), dim3(64), 0, b, flag);
        hipDeviceSynchronize();
        int r = 0;
        hipMemcpy(&r, res, 4, hipMemcpyDeviceToHost);
        printf("pre %d rep %d: wall clock %d kHz, spinner %s\n", pre, rep, clk, r == 1 ? "released by the other stream's kernel" : "TIMED OUT (same queue?)");
    }
    return 0;
}
