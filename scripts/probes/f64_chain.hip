// probe: what does one value of the ordered corrb chain of ph_sums cost on gfx950?
//   A  corrb = (float)fma((double)di, (double)di, (double)corrb)            -- the reference's "corrb += pow(f32, 2)" as built
//   B  t = fma(dd, dd, c); c = (t + M) - M   with M = 1.5 * 2^(e + 29)         -- f64 arithmetic only: round-to-f32-grid by a magic constant
// measured as cycles per value for ONE wave on a SIMD (the latency-bound regime of a small ph_sums launch) and for 2 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void chain(const float *__restrict__ in, float *__restrict__ out, int n, long long *cyc)
{
    const int lane = threadIdx.x & 63;
    float v[32];
    for (int j = 0; j < 32; j++) v[j] = in[j * 64 + lane];
    float corrb = 0.f, corra = 0.f;
    double c = 0.0;
    const float ag = in[lane] * 0.5f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int j = 0; j < 32; j++) asm volatile("" : "+v"(v[j])); // a fresh value every iteration: nothing of the chain is hoisted
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 32; j++) {
                const float di = v[j] - ag;
                corra += di * 0.37f;
                corrb = (float)__builtin_fma((double)di, (double)di, (double)corrb);
            }
        } else {
            // chunk-start exponent -> magic constant (monotone chain: checked again at the end of the chunk in the real kernel)
            const unsigned long long cb = (unsigned long long)__double_as_longlong(c);
            const double M = __longlong_as_double((long long)(((cb & 0x7FF0000000000000ull) + (29ull << 52)) | (1ull << 51)));
#pragma unroll
            for (int j = 0; j < 32; j++) {
                const float di = v[j] - ag;
                corra += di * 0.37f;
                const double dd = (double)di;
                const double t = __builtin_fma(dd, dd, c);
                c = (t + M) - M;
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (MODE == 1) corrb = (float)c;
    out[blockIdx.x * blockDim.x + threadIdx.x] = corrb + corra;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
    float *in, *out; long long *cyc;
    hipMalloc(&in, 32 * 64 * 4); hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
    std::vector<float> h(32 * 64);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) % 255) + 0.37f;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int n = 2000;
    for (int mode = 0; mode < 2; mode++)
        for (int waves : {1, 2, 4}) { // waves per SIMD: one block of 256 * waves threads on one CU
            long long hc = 0;
            for (int rep = 0; rep < 2; rep++) {
                if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(1), dim3(256 * waves), 0, 0, in, out, n, cyc);
                else hipLaunchKernelGGL(chain<1>, dim3(1), dim3(256 * waves), 0, 0, in, out, n, cyc);
                hipDeviceSynchronize();
            }
            hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
            printf("mode %c  %d wave(s)/SIMD: %.1f s_memtime ticks (100 MHz) per value -> x24 = %.0f shader cycles at 2.4 GHz\n", mode ? 'B' : 'A', waves, (double)hc / (n * 32.0),
                   (double)hc / (n * 32.0) * 24.0);
        }
    return 0;
}
