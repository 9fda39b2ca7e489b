// How far behind its writer can a reader run and still be served by the 256 MB Infinity Cache?  (a sigma-pipelined
// SMC step whose write -> read distance stays below the cache size.)  The stash pattern of smc_phased.hip: a region of S megabytes
// is written as ph_sample writes it (waves storing rows of 256 B, non-temporal or plain), then D megabytes of OTHER stash stores follow
// (the next sigma's / the other trace group's sampling), then the region is read twice the way ph_sums reads it (one wave per 1.4 MB
// chain group, 32 loads in flight per lane): pass 1 and, right behind it, pass 2.
//   table 1: read rate of pass 1 and pass 2 by S and D                    -> the window in which a re-read is served on-die
//   table 2: the same with the D megabytes written CONCURRENTLY (second stream) while the region is read -> what the product would see
//   hipcc --offload-arch=gfx950 -O3 -o mall_window mall_window.hip && ./mall_window
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void wr(float *p, size_t rows_per_wave)
{
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    float *q = p + wave * rows_per_wave * 64 + (threadIdx.x & 63);
    for (size_t r = 0; r < rows_per_wave; r++) {
        const float v = (float)(r & 255);
        if (NT) __builtin_nontemporal_store(v, q + r * 64); else q[r * 64] = v;
    }
}

__global__ __launch_bounds__(64) void rd(const float *p, size_t rows_per_wave, float *out)
{
    const float *q = p + (size_t)blockIdx.x * rows_per_wave * 64 + threadIdx.x;
    float acc = 0.f;
    for (size_t r = 0; r + 32 <= rows_per_wave; r += 32) {
        float v[32];
#pragma unroll
        for (int j = 0; j < 32; j++) v[j] = q[(r + j) * 64];
#pragma unroll
        for (int j = 0; j < 32; j++) acc += v[j];
    }
    if (acc == -1.f) out[blockIdx.x] = acc;
}

int main()
{
    const size_t rows = 5632; // one chain group of the long templates: 5625 rows of 256 B
    const size_t MB = 1 << 20, cap = 1024 * MB;
    float *region, *other, *flush, *out;
    CK(hipMalloc(&region, cap)); CK(hipMalloc(&other, cap)); CK(hipMalloc(&flush, 2 * cap)); CK(hipMalloc(&out, 1 << 22));
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    auto grid_w = [&](size_t bytes) { return (unsigned)(bytes / (rows * 256) / 4); };
    auto grid_r = [&](size_t bytes) { return (unsigned)(bytes / (rows * 256)); };
    for (int concurrent = 0; concurrent < 2; concurrent++) {
        printf(concurrent ? "\n# table 2: the D megabytes of other stores run CONCURRENTLY with the two read passes (second stream)\n"
                          : "# table 1: region written, then D megabytes of other stores, then read twice\n");
        printf("stores  S [MB]  D [MB]   pass 1 [GB/s]   pass 2 [GB/s]%s\n", concurrent ? "   writer beside them [GB/s]" : "");
        for (int nt = 1; nt >= 0; nt--)
            for (size_t S : {64, 128, 192, 256, 384, 512})
                for (size_t D : {0, 64, 128, 256, 512}) {
                    if (concurrent && D == 0) continue;
                    const size_t sb = S * MB / (rows * 256 * 4) * (rows * 256 * 4), db = D * MB / (rows * 256 * 4) * (rows * 256 * 4);
                    float best1 = 0, best2 = 0, bestw = 0;
                    for (int rep = 0; rep < 3; rep++) {
                        // start from a cache full of unrelated lines
                        hipLaunchKernelGGL(wr<false>, dim3(grid_w(2 * cap)), dim3(256), 0, s0, flush, rows);
                        if (nt) hipLaunchKernelGGL(wr<true>, dim3(grid_w(sb)), dim3(256), 0, s0, region, rows);
                        else hipLaunchKernelGGL(wr<false>, dim3(grid_w(sb)), dim3(256), 0, s0, region, rows);
                        if (!concurrent && db) {
                            if (nt) hipLaunchKernelGGL(wr<true>, dim3(grid_w(db)), dim3(256), 0, s0, other, rows);
                            else hipLaunchKernelGGL(wr<false>, dim3(grid_w(db)), dim3(256), 0, s0, other, rows);
                        }
                        CK(hipStreamSynchronize(s0));
                        float mw = 0;
                        if (concurrent) {
                            CK(hipEventRecord(e0, s1));
                            // enough other stores to last through both passes: D per pass
                            if (nt) hipLaunchKernelGGL(wr<true>, dim3(grid_w(2 * db)), dim3(256), 0, s1, other, rows);
                            else hipLaunchKernelGGL(wr<false>, dim3(grid_w(2 * db)), dim3(256), 0, s1, other, rows);
                            CK(hipEventRecord(e1, s1));
                        }
                        hipEvent_t a, b, c;
                        CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&c));
                        CK(hipEventRecord(a, s0));
                        hipLaunchKernelGGL(rd, dim3(grid_r(sb)), dim3(64), 0, s0, region, rows, out);
                        CK(hipEventRecord(b, s0));
                        hipLaunchKernelGGL(rd, dim3(grid_r(sb)), dim3(64), 0, s0, region, rows, out);
                        CK(hipEventRecord(c, s0));
                        CK(hipDeviceSynchronize());
                        float m1, m2;
                        CK(hipEventElapsedTime(&m1, a, b)); CK(hipEventElapsedTime(&m2, b, c));
                        if (concurrent) { CK(hipEventElapsedTime(&mw, e0, e1)); if (2 * db / mw / 1e6f > bestw) bestw = 2 * db / mw / 1e6f; }
                        if (sb / m1 / 1e6f > best1) best1 = sb / m1 / 1e6f;
                        if (sb / m2 / 1e6f > best2) best2 = sb / m2 / 1e6f;
                        CK(hipEventDestroy(a)); CK(hipEventDestroy(b)); CK(hipEventDestroy(c));
                    }
                    if (concurrent) printf("%-6s  %5zu   %5zu   %10.0f      %10.0f      %10.0f\n", nt ? "nt" : "plain", S, D, best1, best2, bestw);
                    else printf("%-6s  %5zu   %5zu   %10.0f      %10.0f\n", nt ? "nt" : "plain", S, D, best1, best2);
                }
    }
    return 0;
}
