// probe: the issue rate of plain and PACKED f32 multiply / add on gfx950, for the register-tiled Gaussian passes (acc = acc + x * tap,
// separate multiply and add -- no contraction, the reference's rounding).  One block of 256 * W threads on one CU (W waves per SIMD),
// N_ACC independent accumulator chains per lane, shader-clock cycles per wave-instruction of the loop body.
//   mode 0: v_mul_f32 + v_add_f32        (NA chains)
//   mode 1: v_pk_mul_f32 + v_pk_add_f32  (NA / 2 packed chains: the same FLOPs in half the instructions)
//   mode 2: v_fma_f32 (one rounding: not usable for the Gaussian, the ceiling of the unit)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE, int NA>
__global__ void rate(const float *__restrict__ in, float *__restrict__ out, int n, long long *cyc)
{
    const int lane = threadIdx.x;
    float acc[NA], x[NA];
    for (int j = 0; j < NA; j++) { acc[j] = in[j * 64 + (lane & 63)]; x[j] = in[(j + NA) * 64 + (lane & 63)]; }
    const float tap = in[lane & 63] * 1e-3f + 0.999f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int rep = 0; rep < 8; rep++) {
#pragma unroll
            for (int j = 0; j < NA; j++) asm volatile("" : "+v"(x[j])); // a fresh operand every time: nothing is hoisted
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < NA; j++) { const float p = x[j] * tap; acc[j] = acc[j] + p; }
            } else if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < NA; j += 2) {
                    f32x2 a = {acc[j], acc[j + 1]}, xx = {x[j], x[j + 1]};
                    const f32x2 p = xx * (f32x2){tap, tap};
                    a = a + p;
                    acc[j] = a.x; acc[j + 1] = a.y;
                }
            } else {
#pragma unroll
                for (int j = 0; j < NA; j++) acc[j] = __builtin_fmaf(x[j], tap, acc[j]);
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < NA; j++) s += acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE, int NA>
void run(const float *in, float *out, long long *cyc, const char *name)
{
    const int n = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves : {1, 2, 4, 8}) { // waves per SIMD: 256 CUs x `waves` blocks of 256 threads
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL((rate<MODE, NA>), dim3(256 * waves), dim3(256), 0, 0, in, out, n, cyc);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double laneops = (double)n * 8 * NA * 2 * 64 * 4 * 256 * waves; // multiply + add, per lane
        printf("%-28s NA=%2d waves/SIMD=%d : %7.3f ms, %6.1f T lane-ops/s\n", name, NA, waves, ms, laneops / (ms * 1e-3) / 1e12);
    }
}
int main()
{
    float *in, *out; long long *cyc;
    hipMalloc(&in, 64 * 64 * 4); hipMalloc(&out, 64 << 20); hipMalloc(&cyc, 8);
    float h[64 * 64];
    for (int i = 0; i < 64 * 64; i++) h[i] = 1.0f + (float)(i % 97) * 1e-3f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0, 8>(in, out, cyc, "v_mul_f32 + v_add_f32");
    run<1, 8>(in, out, cyc, "v_pk_mul_f32 + v_pk_add_f32");
    run<0, 16>(in, out, cyc, "v_mul_f32 + v_add_f32");
    run<1, 16>(in, out, cyc, "v_pk_mul_f32 + v_pk_add_f32");
    run<2, 8>(in, out, cyc, "v_fma_f32");
    return 0;
}
