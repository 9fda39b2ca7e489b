// probe: are unaligned 16-bit LDS reads (odd byte address) correct on gfx950, and what do they cost
// relative to two ds_read_u8?  (decides the corner-fetch form of the SMC kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const int *idx, unsigned *out, int n, int mode, int reps, unsigned long long *cyc)
{
    __shared__ unsigned char s[65536];
    for (int i = threadIdx.x; i < 65536; i += blockDim.x) s[i] = (unsigned char)(i * 7 + 3);
    __syncthreads();
    int a = idx[threadIdx.x];
    unsigned acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
        if (mode == 0) {
            acc += s[a] + 256u * s[a + 1];
        } else {
            unsigned short v;
            __builtin_memcpy(&v, s + a, 2);
            acc += v;
        }
        a = (a + 12345 + (acc & 1)) & 32767;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    const int n = 256, reps = 4096;
    std::vector<int> h(n);
    for (int i = 0; i < n; i++) h[i] = (i * 977 + 1) & 32767; // odd and even addresses
    int *d; unsigned *o; unsigned long long *c;
    hipMalloc(&d, n * 4); hipMalloc(&o, n * 4); hipMalloc(&c, 8);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<unsigned> r0(n), r1(n);
    unsigned long long c0, c1;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(n), 0, 0, d, o, n, 0, reps, c);
        hipMemcpy(r0.data(), o, n * 4, hipMemcpyDeviceToHost); hipMemcpy(&c0, c, 8, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k, dim3(1), dim3(n), 0, 0, d, o, n, 1, reps, c);
        hipMemcpy(r1.data(), o, n * 4, hipMemcpyDeviceToHost); hipMemcpy(&c1, c, 8, hipMemcpyDeviceToHost);
    }
    int bad = 0;
    for (int i = 0; i < n; i++) bad += r0[i] != r1[i];
    printf("mismatches %d ; cycles/iter u8x2 %.1f  u16 %.1f\n", bad, (double)c0 / reps, (double)c1 / reps);
    return bad != 0;
}
