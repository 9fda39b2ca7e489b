// Does the 256 MB Infinity Cache keep lines a kernel has just WRITTEN (plain or non-temporal stores), so that a kernel reading them
// next is served on-die?  The sample stash (smc_phased.hip) is written by ph_sample and read twice by ph_sums; a launch of the bench
// writes 0.4 - 0.8 GB of it.  The probe writes S megabytes the way ph_sample does (a wave stores rows of 256 B), then reads them the way
// ph_sums does (a wave streams its own rows, 32 loads in flight), and compares with the same read after 2 GB of other traffic.
//   hipcc --offload-arch=gfx950 -O3 -o mall_wr mall_wr.hip && ./mall_wr
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

// `mall_wr mixed`: what HBM delivers when a writer (the stash stores of one trace group's ph_sample) and a reader (the other group's
// ph_sums) stream at the same time on two streams: 1.5 GB written beside 3 GB read, the proportions of a 164-trace step
static int mixed()
{
    const size_t wbytes = (size_t)3 << 29, rbytes = (size_t)3 << 30, rows = 1024;
    float *wb, *rb, *out;
    CK(hipMalloc(&wb, wbytes)); CK(hipMalloc(&rb, rbytes)); CK(hipMalloc(&out, 1 << 22));
    CK(hipMemset(rb, 0, rbytes));
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t a0, a1, b0, b1;
    CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    const unsigned wgrid = (unsigned)(wbytes / (rows * 256) / 4), rgrid = (unsigned)(rbytes / (rows * 256));
    printf("stores   writer alone [GB/s]   reader alone [GB/s]   together: writer / reader / sum [GB/s]\n");
    for (int nt = 0; nt < 2; nt++) {
        float best[5] = {0, 0, 0, 0, 0};
        for (int rep = 0; rep < 3; rep++) {
            float ms;
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a0, s0));
            if (nt) hipLaunchKernelGGL(wr<true>, dim3(wgrid), dim3(256), 0, s0, wb, rows); else hipLaunchKernelGGL(wr<false>, dim3(wgrid), dim3(256), 0, s0, wb, rows);
            CK(hipEventRecord(a1, s0)); CK(hipEventSynchronize(a1)); CK(hipEventElapsedTime(&ms, a0, a1));
            if (wbytes / ms / 1e6f > best[0]) best[0] = wbytes / ms / 1e6f;
            CK(hipEventRecord(b0, s1));
            hipLaunchKernelGGL(rd, dim3(rgrid), dim3(64), 0, s1, rb, rows, out);
            CK(hipEventRecord(b1, s1)); CK(hipEventSynchronize(b1)); CK(hipEventElapsedTime(&ms, b0, b1));
            if (rbytes / ms / 1e6f > best[1]) best[1] = rbytes / ms / 1e6f;
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a0, s0)); CK(hipEventRecord(b0, s1));
            if (nt) hipLaunchKernelGGL(wr<true>, dim3(wgrid), dim3(256), 0, s0, wb, rows); else hipLaunchKernelGGL(wr<false>, dim3(wgrid), dim3(256), 0, s0, wb, rows);
            hipLaunchKernelGGL(rd, dim3(rgrid), dim3(64), 0, s1, rb, rows, out);
            CK(hipEventRecord(a1, s0)); CK(hipEventRecord(b1, s1));
            CK(hipEventSynchronize(a1)); CK(hipEventSynchronize(b1));
            float mw, mr;
            CK(hipEventElapsedTime(&mw, a0, a1)); CK(hipEventElapsedTime(&mr, b0, b1));
            const float tot = (wbytes + rbytes) / (mw > mr ? mw : mr) / 1e6f;
            if (tot > best[4]) { best[4] = tot; best[2] = wbytes / mw / 1e6f; best[3] = rbytes / mr / 1e6f; }
        }
        printf("%-6s   %10.0f            %10.0f            %8.0f / %8.0f / %8.0f\n", nt ? "nt" : "plain", best[0], best[1], best[2], best[3], best[4]);
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 1 && argv[1][0] == 'm') return mixed();
    const size_t big = (size_t)2 << 30;
    float *buf, *other, *out;
    CK(hipMalloc(&buf, big)); CK(hipMalloc(&other, big)); CK(hipMalloc(&out, 1 << 20));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t rows_per_wave = 1024; // 256 KB per wave
    printf("MB written   stores   read back at once [GB/s]   read after 2 GB of other stores [GB/s]\n");
    for (int mb : {32, 64, 128, 192, 256, 384, 768, 1536}) {
        const size_t bytes = (size_t)mb << 20, waves = bytes / (rows_per_wave * 256);
        for (int nt = 0; nt < 2; nt++) {
            float best[2] = {0, 0};
            for (int mode = 0; mode < 2; mode++)
                for (int rep = 0; rep < 3; rep++) {
                    if (nt) hipLaunchKernelGGL(wr<true>, dim3((unsigned)(waves / 4)), dim3(256), 0, 0, buf, rows_per_wave);
                    else hipLaunchKernelGGL(wr<false>, dim3((unsigned)(waves / 4)), dim3(256), 0, 0, buf, rows_per_wave);
                    if (mode == 1) hipLaunchKernelGGL(wr<false>, dim3((unsigned)(big / (rows_per_wave * 256) / 4)), dim3(256), 0, 0, other, rows_per_wave);
                    CK(hipEventRecord(e0, 0));
                    hipLaunchKernelGGL(rd, dim3((unsigned)waves), dim3(64), 0, 0, buf, rows_per_wave, out);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    const float gbs = (float)bytes / ms / 1e6f;
                    if (gbs > best[mode]) best[mode] = gbs;
                }
            printf("%6d       %-6s   %10.0f                 %10.0f\n", mb, nt ? "nt" : "plain", best[0], best[1]);
        }
    }
    return 0;
}
