// probe: calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the SMC stash access pattern
// (one dword per lane, 256 B contiguous per wave-instruction), on a known byte count (1 GiB each way).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void rd(const float *__restrict__ p, float *out, long long n)
{
    float acc = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 123.456f) out[0] = acc;
}
__global__ void wr(float *__restrict__ p, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = (float)i;
}
int main()
{
    const long long n = 1LL << 28; // 1 GiB of f32
    float *a, *o;
    hipMalloc(&a, n * 4); hipMalloc(&o, 4);
    hipMemset(a, 0, n * 4);
    hipLaunchKernelGGL(wr, dim3(4096), dim3(256), 0, 0, a, n);
    hipLaunchKernelGGL(rd, dim3(4096), dim3(256), 0, 0, a, o, n);
    hipDeviceSynchronize();
    printf("calib: each kernel moved %lld bytes\n", n * 4);
    return 0;
}
