// probe: the ordered sums of ph_sums (znccBBB's mean / corra / corrb chains, tracker.cpp:1940-1955) as a function of how many stash
// values a wave keeps in flight.  One wave owns 64 chains (lane = chain) and walks M samples twice; every value is a handful of
// DEPENDENT operations (8 cycles in the first pass, 29.5 in the second: scripts/probes/f64_chain.hip), so a wave is latency-bound
// unless enough loads are in flight to cover an HBM round trip (1000 - 2500 cycles under load).
//   V0  the kernel as shipped until round 3: [sample][lane] f32, dword loads, 32 values per lane "in flight" (the compiler folds the
//       two software-pipeline buffers into one, so every chunk waits for its own loads: dead end (m) of DESIGN.md)
//   Vn  [sample / 4][lane][4] f32 (a lane's four consecutive samples are one dwordx4), a ring of NB buffers of 32 values written by
//       hand (global_load_dwordx4 + counted s_waitcnt vmcnt): (NB - 1) x 32 values really in flight while 32 are summed
// All variants must give the same bits.  Times for T traces x 12 waves (4 x M = 845, 8 x M = 5625: three scales, four chain groups).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -o sums_ring sums_ring.hip && ./sums_ring
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdio>
#include <cstring>
#include <vector>

typedef long long i64;
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Job { i64 off; int M; int woff; }; // stash offset (floats), samples, offset of the template weights

// ---- V0: as shipped (smc_device.h zncc_from_stash<64, 32>) ------------------------------------------------------------------
__device__ __forceinline__ float zncc_v0(const float *__restrict__ stash_lane, int M, const float *__restrict__ wd, float corrc)
{
    constexpr int CH = 32, STRIDE = 64;
    const int nfull = M / CH, tail = M - nfull * CH;
    float cur[CH], nxt[CH];
    float ag = 0.f;
    if (nfull > 0) {
#pragma unroll
        for (int j = 0; j < CH; j++) cur[j] = stash_lane[j * STRIDE];
    }
    for (int c = 0; c < nfull; c++) {
        const float *nx = stash_lane + (i64)(c + 1) * CH * STRIDE;
        if (c + 1 < nfull) {
#pragma unroll
            for (int j = 0; j < CH; j++) nxt[j] = nx[j * STRIDE];
        }
#pragma unroll
        for (int j = 0; j < CH; j++) ag += cur[j];
#pragma unroll
        for (int j = 0; j < CH; j++) cur[j] = nxt[j];
    }
    {
        const float *tp = stash_lane + (i64)nfull * CH * STRIDE;
#pragma unroll
        for (int j = 0; j < CH; j++) cur[j] = (j < tail) ? tp[j * STRIDE] : 0.f;
#pragma unroll
        for (int j = 0; j < CH; j++)
            if (j < tail) ag += cur[j];
    }
    ag /= (float)M;
    float corra = 0.f, corrb = 0.f;
    if (nfull > 0) {
#pragma unroll
        for (int j = 0; j < CH; j++) cur[j] = stash_lane[j * STRIDE];
    }
    for (int c = 0; c < nfull; c++) {
        const float *nx = stash_lane + (i64)(c + 1) * CH * STRIDE;
        if (c + 1 < nfull) {
#pragma unroll
            for (int j = 0; j < CH; j++) nxt[j] = nx[j * STRIDE];
        }
        const float *wk = wd + c * CH;
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const float di = cur[j] - ag;
            corra += di * wk[j];
            corrb = (float)__builtin_fma((double)di, (double)di, (double)corrb);
        }
#pragma unroll
        for (int j = 0; j < CH; j++) cur[j] = nxt[j];
    }
    {
        const float *tp = stash_lane + (i64)nfull * CH * STRIDE;
        const float *wk = wd + nfull * CH;
#pragma unroll
        for (int j = 0; j < CH; j++) cur[j] = (j < tail) ? tp[j * STRIDE] : 0.f;
#pragma unroll
        for (int j = 0; j < CH; j++)
            if (j < tail) {
                const float di = cur[j] - ag;
                corra += di * wk[j];
                corrb = (float)__builtin_fma((double)di, (double)di, (double)corrb);
            }
    }
    const float prod = corrb * corrc;
    return (prod > FLT_MIN) ? corra / sqrtf(prod) : 0.f;
}

__global__ __launch_bounds__(64) void sums_v0(const float *__restrict__ stash, const Job *__restrict__ jobs, const float *__restrict__ wd, float *__restrict__ out)
{
    const Job jb = jobs[blockIdx.x];
    out[(i64)blockIdx.x * 64 + threadIdx.x] = zncc_v0(stash + jb.off + threadIdx.x, jb.M, wd + jb.woff, 0.37f);
}

// ---- Vn: x4 layout, hand-written ring -----------------------------------------------------------------------------------------
// chunk = 32 samples = 8 blocks of 4; block b of a chain group lies at 256 * b floats (64 lanes x 4), a lane's quad at + 4 * lane
struct Chunk { f32x4 q[8]; };

__device__ __forceinline__ void issue(Chunk &c, const float *lane_base, int chunk)
{
    const char *p = (const char *)(lane_base + (i64)chunk * 8 * 256), *p2 = p + 4096; // (immediate offsets reach 4095 bytes)
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(c.q[0]) : "v"(p));
    asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(c.q[1]) : "v"(p));
    asm volatile("global_load_dwordx4 %0, %1, off offset:2048" : "=v"(c.q[2]) : "v"(p));
    asm volatile("global_load_dwordx4 %0, %1, off offset:3072" : "=v"(c.q[3]) : "v"(p));
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(c.q[4]) : "v"(p2));
    asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(c.q[5]) : "v"(p2));
    asm volatile("global_load_dwordx4 %0, %1, off offset:2048" : "=v"(c.q[6]) : "v"(p2));
    asm volatile("global_load_dwordx4 %0, %1, off offset:3072" : "=v"(c.q[7]) : "v"(p2));
}

template <int N>
__device__ __forceinline__ void wait_for(Chunk &c) // at most N loads still outstanding: everything issued before them has landed
{
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(c.q[0]), "+v"(c.q[1]), "+v"(c.q[2]), "+v"(c.q[3]), "+v"(c.q[4]), "+v"(c.q[5]), "+v"(c.q[6]), "+v"(c.q[7])
                 : "n"(N));
}

template <int NB>
__device__ __forceinline__ float zncc_ring(const float *__restrict__ lane_base, int M, const float *__restrict__ wd, float corrc)
{
    static_assert(NB >= 2 && NB <= 6, "ring depth");
    constexpr int AHEAD = NB - 1;           // chunks in flight while one is summed
    const int nchunk = (M + 31) / 32;       // the last one may be partial (the stash is padded to whole chunks)
    const int nfull = M / 32, tail = M - nfull * 32;
    Chunk ring[NB];
    float ag = 0.f;
    // ---- pass 1: the mean, in sample order
#pragma unroll
    for (int b = 0; b < AHEAD; b++) issue(ring[b], lane_base, b < nchunk ? b : nchunk - 1);
    for (int c0 = 0; c0 < nfull; c0 += NB) {
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const int c = c0 + b;
            if (c >= nfull) break; // wave-uniform
            const int nx = c + AHEAD;
            issue(ring[(b + AHEAD) % NB], lane_base, nx < nchunk ? nx : nchunk - 1);
            wait_for<8 * AHEAD>(ring[b]);
#pragma unroll
            for (int i = 0; i < 8; i++) { ag += ring[b].q[i].x; ag += ring[b].q[i].y; ag += ring[b].q[i].z; ag += ring[b].q[i].w; }
        }
    }
    // the loop's last prefetches are never summed: tie every buffer to a full wait, or a load that is still in flight lands in a
    // register the compiler has already given to something else (this faulted in the library's first version)
#pragma unroll
    for (int b = 0; b < NB; b++) wait_for<0>(ring[b]);
    {
        Chunk &t = ring[0]; // (whatever is still in flight into the ring lands before the wait below returns)
        issue(t, lane_base, nchunk - 1);
        wait_for<0>(t);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (4 * i + 0 < tail) ag += t.q[i].x;
            if (4 * i + 1 < tail) ag += t.q[i].y;
            if (4 * i + 2 < tail) ag += t.q[i].z;
            if (4 * i + 3 < tail) ag += t.q[i].w;
        }
    }
    ag /= (float)M;
    // ---- pass 2: corra / corrb, in sample order
    float corra = 0.f, corrb = 0.f;
#pragma unroll
    for (int b = 0; b < AHEAD; b++) issue(ring[b], lane_base, b < nchunk ? b : nchunk - 1);
    for (int c0 = 0; c0 < nfull; c0 += NB) {
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const int c = c0 + b;
            if (c >= nfull) break;
            const int nx = c + AHEAD;
            issue(ring[(b + AHEAD) % NB], lane_base, nx < nchunk ? nx : nchunk - 1);
            wait_for<8 * AHEAD>(ring[b]);
            const float *wk = wd + c * 32;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float v4[4] = {ring[b].q[i].x, ring[b].q[i].y, ring[b].q[i].z, ring[b].q[i].w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float di = v4[e] - ag;
                    corra += di * wk[4 * i + e];
                    corrb = (float)__builtin_fma((double)di, (double)di, (double)corrb);
                }
            }
        }
    }
#pragma unroll
    for (int b = 0; b < NB; b++) wait_for<0>(ring[b]);
    {
        Chunk &t = ring[0];
        issue(t, lane_base, nchunk - 1);
        wait_for<0>(t);
        const float *wk = wd + nfull * 32;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float v4[4] = {t.q[i].x, t.q[i].y, t.q[i].z, t.q[i].w};
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (4 * i + e < tail) {
                    const float di = v4[e] - ag;
                    corra += di * wk[4 * i + e];
                    corrb = (float)__builtin_fma((double)di, (double)di, (double)corrb);
                }
        }
    }
    const float prod = corrb * corrc;
    return (prod > FLT_MIN) ? corra / sqrtf(prod) : 0.f;
}

template <int NB>
__global__ __launch_bounds__(64) void sums_ring(const float *__restrict__ stash4, const Job *__restrict__ jobs, const float *__restrict__ wd, float *__restrict__ out)
{
    const Job jb = jobs[blockIdx.x];
    out[(i64)blockIdx.x * 64 + threadIdx.x] = zncc_ring<NB>(stash4 + jb.off + 4 * threadIdx.x, jb.M, wd + jb.woff, 0.37f);
}

__global__ void fill(float *a, float *b, const Job *jobs, int njobs)
{
    // a: [sample][lane]; b: [sample / 4][lane][4]; same values (interpolated bytes look like f32 in [0, 255] with full mantissas)
    const int jn = blockIdx.x;
    const Job jb = jobs[jn];
    const int Mp = (jb.M + 31) / 32 * 32;
    for (i64 e = threadIdx.x; e < (i64)Mp * 64; e += blockDim.x) {
        const int k = (int)(e / 64), lane = (int)(e % 64);
        unsigned h = (unsigned)(jb.off + e) * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float v = (float)(h >> 8) * (255.0f / 16777216.0f);
        a[jb.off + (i64)k * 64 + lane] = v;
        b[jb.off + (i64)(k / 4) * 256 + lane * 4 + (k & 3)] = v;
    }
}

int main(int argc, char **argv)
{
    const int Ms[12] = {845, 845, 845, 845, 5625, 5625, 5625, 5625, 5625, 5625, 5625, 5625};
    const int Tmax = 256;
    std::vector<Job> jobs;
    i64 off = 0;
    int woff[12], wtot = 0;
    for (int j = 0; j < 12; j++) { woff[j] = wtot; wtot += (Ms[j] + 31) / 32 * 32; }
    for (int t = 0; t < Tmax; t++)
        for (int j = 0; j < 12; j++) { jobs.push_back({off, Ms[j], woff[j]}); off += (i64)((Ms[j] + 31) / 32 * 32) * 64; }
    float *a, *b, *wd, *out0, *out1;
    Job *dj;
    CK(hipMalloc(&a, off * 4)); CK(hipMalloc(&b, off * 4)); CK(hipMalloc(&wd, wtot * 4));
    CK(hipMalloc(&out0, jobs.size() * 64 * 4)); CK(hipMalloc(&out1, jobs.size() * 64 * 4));
    CK(hipMalloc(&dj, jobs.size() * sizeof(Job)));
    CK(hipMemcpy(dj, jobs.data(), jobs.size() * sizeof(Job), hipMemcpyHostToDevice));
    std::vector<float> hw(wtot);
    for (int i = 0; i < wtot; i++) hw[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    CK(hipMemcpy(wd, hw.data(), wtot * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fill, dim3((unsigned)jobs.size()), dim3(256), 0, 0, a, b, (const Job *)dj, (int)jobs.size());
    CK(hipDeviceSynchronize());
    printf("stash %.2f GB for %d traces (%.2f MB per trace)\n", off * 4 / 1e9, Tmax, off * 4 / 1e6 / Tmax);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> h0[5], h1(jobs.size() * 64);
    for (auto &h : h0) h.resize(jobs.size() * 64);
    const int Ts[5] = {8, 32, 61, 122, 244};
    for (int variant = 0; variant <= 5; variant++) {
        if (variant == 1) continue;
        for (int ti = 0; ti < 5; ti++) {
            const int T = Ts[ti];
            const unsigned grid = (unsigned)(T * 12);
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                // a different part of the stash every repetition: nothing is left in L2 / the Infinity Cache from the last launch
                const Job *jp = dj + (size_t)((rep * 61) % (Tmax - T + 1)) * 12;
                float *o = variant == 0 ? out0 : out1;
                CK(hipEventRecord(e0, 0));
                switch (variant) {
                case 0: hipLaunchKernelGGL(sums_v0, dim3(grid), dim3(64), 0, 0, a, jp, wd, o); break;
                case 2: hipLaunchKernelGGL(sums_ring<2>, dim3(grid), dim3(64), 0, 0, b, jp, wd, o); break;
                case 3: hipLaunchKernelGGL(sums_ring<3>, dim3(grid), dim3(64), 0, 0, b, jp, wd, o); break;
                case 4: hipLaunchKernelGGL(sums_ring<4>, dim3(grid), dim3(64), 0, 0, b, jp, wd, o); break;
                case 5: hipLaunchKernelGGL(sums_ring<6>, dim3(grid), dim3(64), 0, 0, b, jp, wd, o); break;
                }
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                if (rep == 4) { // the last repetition's jobs are the same for every variant: compare the bits
                    CK(hipMemcpy(variant == 0 ? h0[ti].data() : h1.data(), o, (size_t)grid * 64 * 4, hipMemcpyDeviceToHost));
                    if (variant != 0 && memcmp(h0[ti].data(), h1.data(), (size_t)grid * 64 * 4) != 0) { printf("variant %d: results DIFFER from V0\n", variant); return 2; }
                }
            }
            const double bytes = 0;
            (void)bytes;
            double gb = 0;
            for (int j = 0; j < 12; j++) gb += 2.0 * Ms[j] * 256;
            gb *= T / 1e9;
            printf("%s  T=%3d  %.3f ms  %.0f GB/s\n", variant == 0 ? "V0 shipped (dword, 32 folded)" : variant == 2 ? "ring NB=2 (x4, 32 ahead)   " : variant == 3 ? "ring NB=3 (x4, 64 ahead)   " : variant == 4 ? "ring NB=4 (x4, 96 ahead)   " : "ring NB=6 (x4, 160 ahead)  ", T, best, gb / (best * 1e-3));
        }
    }
    return 0;
}
