// probe: LDS gather throughput on gfx950 for the SMC corner fetches.  768 threads (12 waves), 128 KB
// byte box, per lane pseudo-random addresses (like particles scattered in the box).
// modes: 0 = 8 x ds_read_u8, 1 = 4 x unaligned ds_read_u16 (any address), 2 = 4 x ds_read_u16 at even
// addresses, 3 = 4 x ds_read_b32 (aligned), 4 = mode 1 but lanes clustered (+-8 voxels around a centre)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(768) void k(unsigned *out, int reps, unsigned long long *cyc)
{
    extern __shared__ unsigned char s[];
    const int NB = 128 * 1024;
    for (int i = threadIdx.x; i < NB; i += blockDim.x) s[i] = (unsigned char)(i * 7 + 3);
    __syncthreads();
    unsigned a = threadIdx.x * 2654435761u;
    unsigned acc = 0;
    const int bx = 52, sxy = 52 * 52;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
        a = a * 1664525u + 1013904223u;
        unsigned base;
        if (MODE == 4) {
            unsigned x = 20 + ((a >> 8) & 15), y = 20 + ((a >> 12) & 15), z = 16 + ((a >> 16) & 15);
            base = z * sxy + y * bx + x + (r & 7);
        } else {
            base = (a >> 8) % (NB - 2 * sxy - 8);
        }
        if (MODE == 2) base &= ~1u;
        if (MODE == 3) base &= ~3u;
        const unsigned char *p = s + base;
        if (MODE == 0) {
            acc += p[0] + p[1] + p[bx] + p[bx + 1] + p[sxy] + p[sxy + 1] + p[sxy + bx] + p[sxy + bx + 1];
        } else if (MODE == 3) {
            acc += *(const unsigned *)p + *(const unsigned *)(p + 52) + *(const unsigned *)(p + sxy) + *(const unsigned *)(p + sxy + 52);
        } else {
            unsigned short v0, v1, v2, v3;
            __builtin_memcpy(&v0, p, 2); __builtin_memcpy(&v1, p + bx, 2);
            __builtin_memcpy(&v2, p + sxy, 2); __builtin_memcpy(&v3, p + sxy + bx, 2);
            acc += v0 + v1 + v2 + v3;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char *name, int ninst)
{
    unsigned *o; unsigned long long *c, h;
    hipMalloc(&o, 768 * 4); hipMalloc(&c, 8);
    const int reps = 20000;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(768), 128 * 1024, 0, o, reps, c);
    hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-34s %8.1f cycles per sample-group per wave (12 waves resident) -> %6.1f per LDS instr, CU-wide %.2f instr/clk\n", name,
           (double)h / reps, (double)h / reps / ninst, 12.0 * ninst * reps / (double)h);
    hipFree(o); hipFree(c);
}
int main()
{
    run<0>("8 x ds_read_u8 random", 8);
    run<1>("4 x ds_read_u16 unaligned random", 4);
    run<2>("4 x ds_read_u16 even random", 4);
    run<3>("4 x ds_read_b32 aligned random", 4);
    run<4>("4 x ds_read_u16 unaligned clustered", 4);
    return 0;
}
