// Can a SLICE of the sample stash live in the 256 MB Infinity Cache for good while the rest streams past it?
// The stash of a trace group is rewritten at the same addresses every SMC step (it is indexed by list position): a slice that is written
// with plain stores (they allocate in the cache, profiles/r05_mall_window_probe.txt) and read back twice could stay on-die step after
// step -- never reaching HBM -- IF the streaming rest (non-temporal stores, and loads that do not allocate) does not evict it.
//   part 1: do non-temporal LOADS allocate?  region read once with nt loads, then with plain loads: a second pass at the HBM pace = no.
//   part 2: the product's pattern.  Per "step": the resident slice A (S_A MB) is written (plain), then read twice (plain); beside it, on a
//           second stream, B MB are written with nt stores and read twice with nt (or plain) loads -- the streaming 3/4 of the stash.
//           Reported: the read rate of A's two passes in steady state, against the same with nothing streaming and against B's own rate.
//   hipcc --offload-arch=gfx950 -O3 -o mall_resident mall_resident.hip && ./mall_resident
#include <hip/hip_runtime.h>
#include <cstdio>
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

template <bool NT>
__global__ __launch_bounds__(64) void rd(const float *p, size_t rows_per_wave, float *out)
{
    const float *q = p + (size_t)blockIdx.x * rows_per_wave * 64 + threadIdx.x;
    float acc = 0.f;
    for (size_t r = 0; r + 32 <= rows_per_wave; r += 32) {
        float v[32];
#pragma unroll
        for (int j = 0; j < 32; j++) v[j] = NT ? __builtin_nontemporal_load(q + (r + j) * 64) : q[(r + j) * 64];
#pragma unroll
        for (int j = 0; j < 32; j++) acc += v[j];
    }
    if (acc == -1.f) out[blockIdx.x] = acc;
}

int main()
{
    const size_t rows = 5632, MB = 1 << 20, cap = (size_t)2048 * MB;
    float *A, *B, *flush, *out;
    CK(hipMalloc(&A, 512 * MB)); CK(hipMalloc(&B, cap)); CK(hipMalloc(&flush, cap)); CK(hipMalloc(&out, 1 << 22));
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    auto gw = [&](size_t bytes) { return (unsigned)(bytes / (rows * 256) / 4); };
    auto gr = [&](size_t bytes) { return (unsigned)(bytes / (rows * 256)); };
    auto rnd = [&](size_t mb) { return mb * MB / (rows * 256 * 4) * (rows * 256 * 4); };
    hipEvent_t e[4];
    for (auto &x : e) CK(hipEventCreate(&x));

    printf("# part 1: region written with nt stores (not in the cache), read with nt / plain loads, then read again with plain loads\n");
    printf("S [MB]   first pass   [GB/s]   second pass (plain) [GB/s]\n");
    for (int ntl = 1; ntl >= 0; ntl--)
        for (size_t S : {64, 128, 192}) {
            const size_t sb = rnd(S);
            float b1 = 0, b2 = 0;
            for (int rep = 0; rep < 3; rep++) {
                hipLaunchKernelGGL(wr<false>, dim3(gw(cap)), dim3(256), 0, s0, flush, rows);
                hipLaunchKernelGGL(wr<true>, dim3(gw(sb)), dim3(256), 0, s0, A, rows);
                CK(hipEventRecord(e[0], s0));
                if (ntl) hipLaunchKernelGGL(rd<true>, dim3(gr(sb)), dim3(64), 0, s0, A, rows, out);
                else hipLaunchKernelGGL(rd<false>, dim3(gr(sb)), dim3(64), 0, s0, A, rows, out);
                CK(hipEventRecord(e[1], s0));
                hipLaunchKernelGGL(rd<false>, dim3(gr(sb)), dim3(64), 0, s0, A, rows, out);
                CK(hipEventRecord(e[2], s0));
                CK(hipDeviceSynchronize());
                float m1, m2;
                CK(hipEventElapsedTime(&m1, e[0], e[1])); CK(hipEventElapsedTime(&m2, e[1], e[2]));
                if (sb / m1 / 1e6f > b1) b1 = sb / m1 / 1e6f;
                if (sb / m2 / 1e6f > b2) b2 = sb / m2 / 1e6f;
            }
            printf("%5zu    %-5s %8.0f          %8.0f\n", S, ntl ? "nt" : "plain", b1, b2);
        }

    printf("\n# part 2: slice A rewritten (plain) and read twice (plain) every step; beside it B megabytes stream on a second stream (nt stores, then two read passes)\n");
    printf("S_A [MB]   B [MB]   B's loads   A: write / pass 1 / pass 2 [GB/s]   (alone: write / pass 1 / pass 2)   B: total bytes / time [GB/s]\n");
    for (size_t SA : {64, 128, 192})
        for (size_t SB : {512, 1024})
            for (int ntl = 1; ntl >= 0; ntl--) {
                const size_t sa = rnd(SA), sbb = rnd(SB);
                float alone[3] = {0, 0, 0}, with[3] = {0, 0, 0}, brate = 0;
                for (int mode = 0; mode < 2; mode++) { // 0: A alone, 1: with the stream beside it
                    hipLaunchKernelGGL(wr<false>, dim3(gw(cap)), dim3(256), 0, s0, flush, rows);
                    CK(hipDeviceSynchronize());
                    for (int step = 0; step < 6; step++) {
                        hipEvent_t b0, b1;
                        CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
                        if (mode) {
                            CK(hipEventRecord(b0, s1));
                            hipLaunchKernelGGL(wr<true>, dim3(gw(sbb)), dim3(256), 0, s1, B, rows);
                            for (int pass = 0; pass < 2; pass++) {
                                if (ntl) hipLaunchKernelGGL(rd<true>, dim3(gr(sbb)), dim3(64), 0, s1, B, rows, out);
                                else hipLaunchKernelGGL(rd<false>, dim3(gr(sbb)), dim3(64), 0, s1, B, rows, out);
                            }
                            CK(hipEventRecord(b1, s1));
                        }
                        // A's step is repeated while B streams, so that A's accesses are spread over B's time as in the product
                        float m[3] = {0, 0, 0};
                        const int reps = mode ? 4 : 1;
                        for (int k = 0; k < reps; k++) {
                            CK(hipEventRecord(e[0], s0));
                            hipLaunchKernelGGL(wr<false>, dim3(gw(sa)), dim3(256), 0, s0, A, rows);
                            CK(hipEventRecord(e[1], s0));
                            hipLaunchKernelGGL(rd<false>, dim3(gr(sa)), dim3(64), 0, s0, A, rows, out);
                            CK(hipEventRecord(e[2], s0));
                            hipLaunchKernelGGL(rd<false>, dim3(gr(sa)), dim3(64), 0, s0, A, rows, out);
                            CK(hipEventRecord(e[3], s0));
                            CK(hipStreamSynchronize(s0));
                            float t;
                            CK(hipEventElapsedTime(&t, e[0], e[1])); m[0] += t;
                            CK(hipEventElapsedTime(&t, e[1], e[2])); m[1] += t;
                            CK(hipEventElapsedTime(&t, e[2], e[3])); m[2] += t;
                        }
                        CK(hipDeviceSynchronize());
                        if (step >= 2) { // steady state
                            float *dst = mode ? with : alone;
                            for (int k = 0; k < 3; k++) { const float r = reps * sa / m[k] / 1e6f; if (r > dst[k]) dst[k] = r; }
                            if (mode) { float t; CK(hipEventElapsedTime(&t, b0, b1)); if (3 * sbb / t / 1e6f > brate) brate = 3 * sbb / t / 1e6f; }
                        }
                        CK(hipEventDestroy(b0)); CK(hipEventDestroy(b1));
                    }
                }
                printf("%6zu   %6zu   %-9s   %6.0f / %6.0f / %6.0f              (%6.0f / %6.0f / %6.0f)          %6.0f\n", SA, SB, ntl ? "nt" : "plain", with[0], with[1], with[2],
                       alone[0], alone[1], alone[2], brate);
            }
    return 0;
}
