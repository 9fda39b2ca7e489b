// frangi.hip -- multi-scale 3-D Frangi vesselness on gfx950 (wave64), hand-written HIP.
//
// Replaces Frangi::frangi3d / hessian3d / imgaussian / eigen_decomposition
// (frangi.cpp:152-289, :291-390, :647-784, :1269-1493) and the J -> J8 rule
// (Advantra_plugin.cpp:2499-2512).
//
// Per scale sigma:
//   K1 gauss_x_u8      u8  -> f32   row staged in LDS, taps ascending, mul then add (no FMA)
//   K2 gauss_axis (y)  f32 -> f32   64(x) x 32(y) tile + halo in LDS, x stays the coalesced axis
//   K3 gauss_axis (z)  f32 -> f32   same kernel, axis stride w*h, sigma/zdist taps
//   K4 hessian_eigen   radius-2 stencil of first differences applied twice (border rules of
//                      the reference), fp64 Householder+QL eigen-solver for the 3x3 symmetric
//                      Hessian, vesselness, max over scales into J / Vx,Vy,Vz, block-reduced
//                      min/max -> 2 atomics per block
//   K5 j8              J -> u8
// Build: hipcc --offload-arch=gfx950 -ffp-contract=off : every f32/f64 operation is the IEEE
// operation the reference's scalar loop performs, in the same order, so Gaussian / Hessian /
// eigenvectors are bit-identical; only exp() (fp64, ocml vs glibc, <1 ulp each) can differ.
#include "ctx.h"
#include "smc_device.h" // expf_libm: the device expf that equals the host libm value
#include <cfloat>
#include <cmath>
#include <type_traits>
#include <utility>
#include <vector>

namespace {

typedef long long i64;

// ----------------------------------------------------------------------------------------
// K1: Gaussian along x, u8 -> f32 (frangi.cpp:683-714, clamp-to-edge)
// ----------------------------------------------------------------------------------------
constexpr int GX_BLOCK = 256;
constexpr int MAX_L = 64; // ceil(3*sigma) <= 64  (sigma <= 21)
// the taps of one pass on the device: a +0, the 2L + 1 taps, a +0 (gauss_sums_packed reads the neighbours of a tap pairwise), in a
// slot of TAPS_SLOT floats; a scale owns two slots (x / y taps, z taps)
constexpr int TAPS_SLOT = 2 * MAX_L + 4;

__global__ __launch_bounds__(GX_BLOCK) void gauss_x_u8(const uint8_t *__restrict__ img, float *__restrict__ out, int w,
                                                        i64 rows, int tiles_x, const float *__restrict__ taps, int L)
{
    __shared__ float s_in[GX_BLOCK + 2 * MAX_L];
    __shared__ float s_tap[2 * MAX_L + 1];
    const i64 b = blockIdx.x;
    const i64 row = b / tiles_x;
    const int x0 = (int)(b % tiles_x) * GX_BLOCK;
    const uint8_t *src = img + row * w;
    const int span = GX_BLOCK + 2 * L;
    for (int t = threadIdx.x; t < span; t += GX_BLOCK) {
        int x = x0 - L + t;
        x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
        s_in[t] = (float)src[x];
    }
    for (int t = threadIdx.x; t < 2 * L + 1; t += GX_BLOCK) s_tap[t] = taps[t];
    __syncthreads();
    const int x = x0 + threadIdx.x;
    if (x >= w) return;
    float acc = 0.f;
    for (int k = 0; k <= 2 * L; ++k) acc = acc + s_in[threadIdx.x + k] * s_tap[k];
    out[row * w + x] = acc;
}

// work-group id -> tile id such that the work-groups of one XCD (id mod 8) own a contiguous range of tiles (a bijection for any n)
__device__ __forceinline__ unsigned int xcd_contiguous(unsigned int b, unsigned int n)
{
    const unsigned int x = b & 7u, idx = b >> 3, q = n >> 3, r = n & 7u;
    return x * q + (x < r ? x : r) + idx;
}

constexpr int GR = 8; // consecutive outputs per thread in the register-tiled Gaussian passes
typedef float gf32x2 __attribute__((ext_vector_type(2)));

// The GR ascending-tap sums of a register-tiled pass, two outputs per instruction: outputs 2m + 1 and 2m live in the halves (.x,
// .y) of one 64-bit register pair, input t is broadcast into both halves (op_sel) and meets the tap pair (tap[t - 2m - 1],
// tap[t - 2m]) -- two neighbours of the tap array, one scalar 64-bit load -- in a v_pk_mul_f32, the products join the sums in a
// v_pk_add_f32: the two IEEE operations per output and tap of the scalar loop, in its order, each rounded like the scalar one.
// Where one half has no tap (the first input of an odd output, the last of an even one) the tap array is padded with +0: the sum
// takes a +0 (0 x a finite input), which leaves it bit for bit what it was.  `taps` points at tap 0 of an array with one zero in
// front and one behind (upload_taps).
// Why (round 4, scripts/probes/valu_rate.hip + PMC): the passes with the long radii are VALU-bound -- gauss_axis_t issued 4.29e9
// wave-instructions in 7.0 ms, exactly four cycles each on the 1024 SIMDs -- and on gfx950 a packed f32 instruction issues in the
// same four cycles as a plain one (38.7 against 72.6 T lane-operations per second measured).  The instructions are written by hand:
// left to the compiler the four sums of a thread come out one after the other, every v_pk_add_f32 right behind the v_pk_mul_f32 it
// depends on with an s_nop in between, and the pass gets slower (2.47 -> 2.99 ms at L = 18; round 3 saw the same and called packed
// f32 an anti-lever).  Here the multiplies of one input come first, then the adds: no dependent pair is adjacent.
#define PNR_PK_LO " op_sel:[0,0] op_sel_hi:[0,1]\n"
#define PNR_PK_HI " op_sel:[1,0] op_sel_hi:[1,1]\n"
template <int N, bool HI>
__device__ __forceinline__ void pk_step(gf32x2 *a, const gf32x2 in2, const gf32x2 *t)
{
    static_assert(N >= 1 && N <= 4, "");
    gf32x2 p0, p1, p2, p3;
    if constexpr (N == 4) {
        if (HI)
            asm("v_pk_mul_f32 %0, %8, %9" PNR_PK_HI "v_pk_mul_f32 %1, %8, %10" PNR_PK_HI "v_pk_mul_f32 %2, %8, %11" PNR_PK_HI "v_pk_mul_f32 %3, %8, %12" PNR_PK_HI
                "v_pk_add_f32 %4, %4, %0\nv_pk_add_f32 %5, %5, %1\nv_pk_add_f32 %6, %6, %2\nv_pk_add_f32 %7, %7, %3"
                : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])
                : "v"(in2), "s"(t[0]), "s"(t[1]), "s"(t[2]), "s"(t[3]));
        else
            asm("v_pk_mul_f32 %0, %8, %9" PNR_PK_LO "v_pk_mul_f32 %1, %8, %10" PNR_PK_LO "v_pk_mul_f32 %2, %8, %11" PNR_PK_LO "v_pk_mul_f32 %3, %8, %12" PNR_PK_LO
                "v_pk_add_f32 %4, %4, %0\nv_pk_add_f32 %5, %5, %1\nv_pk_add_f32 %6, %6, %2\nv_pk_add_f32 %7, %7, %3"
                : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])
                : "v"(in2), "s"(t[0]), "s"(t[1]), "s"(t[2]), "s"(t[3]));
    } else if constexpr (N == 3) {
        if (HI)
            asm("v_pk_mul_f32 %0, %6, %7" PNR_PK_HI "v_pk_mul_f32 %1, %6, %8" PNR_PK_HI "v_pk_mul_f32 %2, %6, %9" PNR_PK_HI
                "v_pk_add_f32 %3, %3, %0\nv_pk_add_f32 %4, %4, %1\nv_pk_add_f32 %5, %5, %2"
                : "=&v"(p0), "=&v"(p1), "=&v"(p2), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]) : "v"(in2), "s"(t[0]), "s"(t[1]), "s"(t[2]));
        else
            asm("v_pk_mul_f32 %0, %6, %7" PNR_PK_LO "v_pk_mul_f32 %1, %6, %8" PNR_PK_LO "v_pk_mul_f32 %2, %6, %9" PNR_PK_LO
                "v_pk_add_f32 %3, %3, %0\nv_pk_add_f32 %4, %4, %1\nv_pk_add_f32 %5, %5, %2"
                : "=&v"(p0), "=&v"(p1), "=&v"(p2), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]) : "v"(in2), "s"(t[0]), "s"(t[1]), "s"(t[2]));
    } else if constexpr (N == 2) {
        if (HI)
            asm("v_pk_mul_f32 %0, %4, %5" PNR_PK_HI "v_pk_mul_f32 %1, %4, %6" PNR_PK_HI "v_pk_add_f32 %2, %2, %0\nv_pk_add_f32 %3, %3, %1"
                : "=&v"(p0), "=&v"(p1), "+v"(a[0]), "+v"(a[1]) : "v"(in2), "s"(t[0]), "s"(t[1]));
        else
            asm("v_pk_mul_f32 %0, %4, %5" PNR_PK_LO "v_pk_mul_f32 %1, %4, %6" PNR_PK_LO "v_pk_add_f32 %2, %2, %0\nv_pk_add_f32 %3, %3, %1"
                : "=&v"(p0), "=&v"(p1), "+v"(a[0]), "+v"(a[1]) : "v"(in2), "s"(t[0]), "s"(t[1]));
    } else { // a lone pair: the add would sit right behind its multiply (one wait state)
        if (HI) asm("v_pk_mul_f32 %0, %2, %3" PNR_PK_HI "s_nop 0\nv_pk_add_f32 %1, %1, %0" : "=&v"(p0), "+v"(a[0]) : "v"(in2), "s"(t[0]));
        else asm("v_pk_mul_f32 %0, %2, %3" PNR_PK_LO "s_nop 0\nv_pk_add_f32 %1, %1, %0" : "=&v"(p0), "+v"(a[0]) : "v"(in2), "s"(t[0]));
    }
}
#undef PNR_PK_LO
#undef PNR_PK_HI

// acc[m] = (output 2m + 1, output 2m); ld2(q) = the inputs (2q, 2q + 1) of this thread's window; taps: see above
template <int L, class LD2>
__device__ __forceinline__ void gauss_sums_packed(gf32x2 (&acc)[GR / 2], const float *__restrict__ taps, LD2 &&ld2)
{
    static_assert(GR == 8, "four register pairs of sums");
#pragma unroll
    for (int q = 0; q < (2 * L + GR) / 2; q++) {
        const gf32x2 in2 = ld2(q);
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int t = 2 * q + half;
            // pair m takes the taps (k0 - 1, k0) with k0 = t - 2m, for 0 <= k0 <= 2L + 1
            const int m_hi = t / 2 < GR / 2 - 1 ? t / 2 : GR / 2 - 1;
            const int m_lo = t - 2 * L - 1 > 0 ? (t - 2 * L - 1 + 1) / 2 : 0;
            const int n = m_hi - m_lo + 1;
            gf32x2 tp2[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int k0 = t - 2 * (m_lo + (i < n ? i : 0));
                __builtin_memcpy(&tp2[i], taps + k0 - 1, 8); // wave-uniform address, static offset: a scalar load
            }
            if (n == 4) { if (half) pk_step<4, true>(acc + m_lo, in2, tp2); else pk_step<4, false>(acc + m_lo, in2, tp2); }
            else if (n == 3) { if (half) pk_step<3, true>(acc + m_lo, in2, tp2); else pk_step<3, false>(acc + m_lo, in2, tp2); }
            else if (n == 2) { if (half) pk_step<2, true>(acc + m_lo, in2, tp2); else pk_step<2, false>(acc + m_lo, in2, tp2); }
            else if (n == 1) { if (half) pk_step<1, true>(acc + m_lo, in2, tp2); else pk_step<1, false>(acc + m_lo, in2, tp2); }
        }
    }
}

// The x pass with the radius a compile-time constant (the radii the usual parameters give; gauss_x_u8 otherwise): a work-group
// converts a tile of 64 rows x (32 + 2L) bytes to f32 in LDS once; lane = row, and every thread computes GR consecutive outputs
// of its row from 2L + GR LDS reads (row pitch odd: the 64 rows of a wave-read fall into different banks), taps in scalar
// registers, fully unrolled; the 64 x 32 results go back through LDS so that the stores are whole 128-byte rows.  Same sums as
// gauss_x_u8: ascending taps, separate multiply and add.
constexpr int GXT_W = 32, GXT_H = 64;
template <int L>
__global__ __launch_bounds__(256) void gauss_x_u8_t(const uint8_t *__restrict__ img, float *__restrict__ out, int w, i64 rows, int tiles_x,
                                                     const float *__restrict__ taps)
{
    constexpr int SPAN = GXT_W + 2 * L, PITCH = SPAN | 1; // odd
    __shared__ float s_in[GXT_H * PITCH];
    __shared__ float s_out[GXT_H * (GXT_W + 1)];
    // consecutive work-groups go round-robin to the 8 XCDs: with contiguous tile ranges per XCD the 128-byte lines that x-neighbours
    // share (a tile reads 32 + 2L bytes of a row) are fetched into one L2 once instead of once per XCD
    const i64 b = xcd_contiguous(blockIdx.x, gridDim.x);
    const int x0 = (int)(b % tiles_x) * GXT_W;
    const i64 r0 = (b / tiles_x) * GXT_H;
    const int tid = threadIdx.x;
    // interior in x: whole dwords, no clamping (the block-uniform common case).  The last dword read ends at byte x0 + L + 35 - mis
    // of the row at most (NDW dwords from the aligned-down start), so x0 + GXT_W + L + 4 <= w keeps every read inside the row --
    // the last row of a caller-owned buffer (pnr_set_volume_device) must not be read past its end
    if (x0 - L >= 0 && x0 + GXT_W + L + 4 <= w) {
        constexpr int NDW = (SPAN + 6) / 4;      // dwords that cover SPAN bytes at any misalignment
        const int a = (x0 - L) & ~3, mis = (x0 - L) - a;
        typedef unsigned __attribute__((aligned(1))) u32u; // (rows are only dword-aligned when w and the base pointer are)
        for (int e = tid; e < GXT_H * NDW; e += 256) {
            const int r = e / NDW, d = e - r * NDW;
            i64 row = r0 + r;
            if (row > rows - 1) row = rows - 1;
            const unsigned q = *(const u32u *)(img + row * w + a + 4 * d);
#pragma unroll
            for (int b8 = 0; b8 < 4; b8++) {
                const int cidx = 4 * d + b8 - mis;
                if (cidx >= 0 && cidx < SPAN) s_in[r * PITCH + cidx] = (float)((q >> (8 * b8)) & 0xffu);
            }
        }
    } else {
        for (int e = tid; e < GXT_H * SPAN; e += 256) { // clamp-to-edge (frangi.cpp:690)
            const int r = e / SPAN, cidx = e - r * SPAN;
            i64 row = r0 + r;
            if (row > rows - 1) row = rows - 1;
            int x = x0 - L + cidx;
            x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
            s_in[r * PITCH + cidx] = (float)img[row * w + x];
        }
    }
    __syncthreads();
    const int lane = tid & 63, wv = tid >> 6; // row, chunk of GR outputs
    gf32x2 acc[GR / 2];
#pragma unroll
    for (int m = 0; m < GR / 2; m++) acc[m] = (gf32x2){0.f, 0.f};
    const float *src = s_in + lane * PITCH + wv * GR;
    gauss_sums_packed<L>(acc, taps, [&](int q) { return (gf32x2){src[2 * q], src[2 * q + 1]}; });
#pragma unroll
    for (int j = 0; j < GR; j++) s_out[lane * (GXT_W + 1) + wv * GR + j] = (j & 1) ? acc[j / 2].x : acc[j / 2].y;
    __syncthreads();
    for (int e = tid; e < GXT_H * GXT_W; e += 256) { // 32 consecutive floats of a row per half wave
        const int r = e / GXT_W, cidx = e - r * GXT_W;
        const i64 row = r0 + r;
        const int x = x0 + cidx;
        if (row < rows && x < w) out[row * w + x] = s_out[r * (GXT_W + 1) + cidx];
    }
}

static bool launch_gauss_x_t(hipStream_t st, const uint8_t *src, float *dst, int w, i64 rows, const float *d_taps, int L)
{
    static_assert(GXT_W == 4 * GR, "four waves of GR outputs");
    const int tiles_x = (w + GXT_W - 1) / GXT_W;
    const dim3 grid((unsigned)(tiles_x * ((rows + GXT_H - 1) / GXT_H)));
#define PNR_GX(LL) case LL: hipLaunchKernelGGL(gauss_x_u8_t<LL>, grid, dim3(256), 0, st, src, dst, w, rows, tiles_x, d_taps); return true;
    switch (L) {
        PNR_GX(6) PNR_GX(12) PNR_GX(18) PNR_GX(24)
    default: return false;
    }
#undef PNR_GX
}

// ----------------------------------------------------------------------------------------
// K2/K3: Gaussian along a strided axis (y or z), f32 -> f32 (frangi.cpp:717-782)
// tile: 64 consecutive x (one wave-row, 256 B coalesced) x TA outputs along the axis
// ----------------------------------------------------------------------------------------
constexpr int TA = 32;

__global__ __launch_bounds__(256) void gauss_axis(const float *__restrict__ in, float *__restrict__ out, int w, int n_axis,
                                                   i64 axis_stride, int n_other, i64 other_stride, int tiles_x,
                                                   int tiles_a, const float *__restrict__ taps, int L)
{
    extern __shared__ float smem[]; // [(TA + 2L)][64] then taps
    float *s_in = smem;
    float *s_tap = smem + (TA + 2 * L) * 64;
    i64 b = blockIdx.x;
    const int tx_tile = (int)(b % tiles_x);
    b /= tiles_x;
    const int ta_tile = (int)(b % tiles_a);
    const int o = (int)(b / tiles_a);
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6; // 4 row groups
    const int x = tx_tile * 64 + lane;
    const int a0 = ta_tile * TA;
    const float *base = in + (i64)o * other_stride;
    const int rows = TA + 2 * L;
    if (x < w) {
        for (int r = grp; r < rows; r += 4) {
            int a = a0 - L + r;
            a = a < 0 ? 0 : (a > n_axis - 1 ? n_axis - 1 : a);
            s_in[r * 64 + lane] = base[(i64)a * axis_stride + x];
        }
    }
    for (int t = threadIdx.x; t < 2 * L + 1; t += 256) s_tap[t] = taps[t];
    __syncthreads();
    if (x >= w) return;
    float *dst = out + (i64)o * other_stride;
#pragma unroll 1
    for (int j = 0; j < TA / 4; ++j) {
        const int ra = grp + 4 * j;
        const int a = a0 + ra;
        if (a >= n_axis) break;
        float acc = 0.f;
        for (int k = 0; k <= 2 * L; ++k) acc = acc + s_in[(ra + k) * 64 + lane] * s_tap[k];
        dst[(i64)a * axis_stride + x] = acc;
    }
}

// The same pass with the radius L a compile-time constant: every thread computes GR CONSECUTIVE outputs along the axis, so an
// input row read from LDS feeds up to GR accumulators (2L + GR LDS reads for GR outputs instead of GR (2L + 1)), the taps are
// scalar registers, and the loop over the input rows is fully unrolled so that all indices are static.  For every output the
// products are still added in ascending tap order with separate multiply and add -- the reference's sum, bit for bit.
constexpr int TAT = 64, GT = TAT / GR; // outputs per tile along the axis, row groups (waves) per block (GR outputs per thread)

template <int L>
__global__ __launch_bounds__(64 * GT) void gauss_axis_t(const float *__restrict__ in, float *__restrict__ out, int w, int n_axis, i64 axis_stride, int n_other,
                                                     i64 other_stride, int tiles_x, int tiles_a, const float *__restrict__ taps)
{
    __shared__ float s_in[(TAT + 2 * L) * 64];
    i64 b = blockIdx.x;
    const int tx_tile = (int)(b % tiles_x);
    b /= tiles_x;
    const int ta_tile = (int)(b % tiles_a);
    const int o = (int)(b / tiles_a);
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6; // GT groups of GR consecutive rows
    const int x = tx_tile * 64 + lane;
    const int a0 = ta_tile * TAT;
    const float *base = in + (i64)o * other_stride;
    constexpr int rows = TAT + 2 * L;
    if (x < w) {
        // all of a thread's rows requested before the first is stored (one memory latency per work-group)
        constexpr int NITA = (rows + GT - 1) / GT;
        float rv[NITA];
#pragma unroll
        for (int i = 0; i < NITA; i++) {
            const int r = grp + GT * i;
            int a = a0 - L + (r < rows ? r : rows - 1);
            a = a < 0 ? 0 : (a > n_axis - 1 ? n_axis - 1 : a);
            rv[i] = base[(i64)a * axis_stride + x];
        }
#pragma unroll
        for (int i = 0; i < NITA; i++) {
            const int r = grp + GT * i;
            if (i + 1 < NITA || r < rows) s_in[r * 64 + lane] = rv[i];
        }
    }
    __syncthreads();
    if (x >= w) return;
    gf32x2 acc[GR / 2];
#pragma unroll
    for (int m = 0; m < GR / 2; m++) acc[m] = (gf32x2){0.f, 0.f};
    const float *col = s_in + grp * GR * 64 + lane;
    gauss_sums_packed<L>(acc, taps, [&](int q) { return (gf32x2){col[2 * q * 64], col[(2 * q + 1) * 64]}; }); // input row t of this thread's window feeds output j with tap k = t - j
    float *dst = out + (i64)o * other_stride;
#pragma unroll
    for (int j = 0; j < GR; j++) {
        const int a = a0 + grp * GR + j;
        if (a < n_axis) dst[(i64)a * axis_stride + x] = (j & 1) ? acc[j / 2].x : acc[j / 2].y;
    }
}

// ----------------------------------------------------------------------------------------
// K1+K2 fused: Gaussian along x, then along y, of a u8 stack in ONE kernel (frangi.cpp:683-748): the x pass of a 64 x TY tile and
// of the L halo rows above and below it goes to LDS instead of HBM, the y pass reads it from there.  Saves the f32 round trip of the
// x result (8 B per voxel and scale: 8.6 of the 30 GB a scale moves at 1024^3) for (TY + 2L) / TY times the x arithmetic.  Every
// output is still the ascending-tap sum with separate multiply and add of the two unfused passes: the same bits.
// ----------------------------------------------------------------------------------------
constexpr int GXY_TY = 64;
template <int L>
__global__ __launch_bounds__(256) void gauss_xy_u8_t(const uint8_t *__restrict__ img, float *__restrict__ out, int w, int h, int tiles_x, int tiles_y,
                                                      const float *__restrict__ taps)
{
    constexpr int NR = GXY_TY + 2 * L;            // rows of the tile with their halo
    constexpr int SPAN = 64 + 2 * L;              // bytes of a row the x pass reads
    constexpr int PB = ((SPAN + 3 + 3) / 4 * 4) | 4; // byte pitch: whole dwords at any misalignment, an odd number of dwords (lane = row: banks)
    constexpr int PX = 65;                        // float pitch of the x-pass result (written lane = row, read lane = x)
    __shared__ unsigned char s_u8[NR * PB];
    __shared__ float s_x[NR * PX];
    i64 b = xcd_contiguous(blockIdx.x, gridDim.x);
    const int x0 = (int)(b % tiles_x) * 64;
    b /= tiles_x;
    const int y0 = (int)(b % tiles_y) * GXY_TY;
    const i64 z = b / tiles_y;
    const uint8_t *plane = img + z * (i64)w * h;
    const int tid = threadIdx.x;
    int mis = 0;
    if (x0 - L >= 0 && x0 + 64 + L + 4 <= w) { // interior in x: whole (unaligned) dwords, rows clamped in y (frangi.cpp:725)
        constexpr int NDW = (SPAN + 6) / 4;
        const int a = (x0 - L) & ~3;
        mis = (x0 - L) - a;
        typedef unsigned __attribute__((aligned(1))) u32u;
        // every dword of the tile is requested before the first one is stored: a thread's NIT loads are in flight together (one
        // memory latency per work-group instead of NIT in a row -- the loop form waited for each load with vmcnt(0), ten times at
        // L = 18, which was a third of the kernel's time)
        constexpr int NE = NR * NDW, NIT = (NE + 255) / 256;
        unsigned qv[NIT];
        int so[NIT];
#pragma unroll
        for (int i = 0; i < NIT; i++) {
            const int e = tid + 256 * i;
            const bool on = i + 1 < NIT || e < NE;
            const int r = (on ? e : 0) / NDW, d = (on ? e : 0) - r * NDW;
            int y = y0 - L + r;
            y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
            so[i] = on ? r * PB + 4 * d : -1;
            qv[i] = on ? *(const u32u *)(plane + (i64)y * w + a + 4 * d) : 0u;
        }
#pragma unroll
        for (int i = 0; i < NIT; i++)
            if (i + 1 < NIT || so[i] >= 0) *(unsigned *)(s_u8 + so[i]) = qv[i];
    } else {
        for (int e = tid; e < NR * SPAN; e += 256) { // clamp-to-edge in x and y (frangi.cpp:690, :725)
            const int r = e / SPAN, cidx = e - r * SPAN;
            int y = y0 - L + r;
            y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
            int x = x0 - L + cidx;
            x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
            s_u8[r * PB + cidx] = plane[(i64)y * w + x];
        }
    }
    __syncthreads();
    // ---- x pass: task = (row, chunk of GR outputs); consecutive lanes take consecutive rows
    for (int task = tid; task < NR * (64 / GR); task += 256) {
        const int c = task / NR, r = task - c * NR;
        const unsigned char *src = s_u8 + r * PB + mis + c * GR;
        gf32x2 acc[GR / 2];
#pragma unroll
        for (int m = 0; m < GR / 2; m++) acc[m] = (gf32x2){0.f, 0.f};
        gauss_sums_packed<L>(acc, taps, [&](int q) { return (gf32x2){(float)src[2 * q], (float)src[2 * q + 1]}; });
#pragma unroll
        for (int j = 0; j < GR; j++) s_x[r * PX + c * GR + j] = (j & 1) ? acc[j / 2].x : acc[j / 2].y;
    }
    __syncthreads();
    // ---- y pass: task = (column, group of GR consecutive output rows)
    const int lane = tid & 63;
    const int x = x0 + lane;
    for (int gy = tid >> 6; gy < GXY_TY / GR; gy += 4) {
        gf32x2 acc[GR / 2];
#pragma unroll
        for (int m = 0; m < GR / 2; m++) acc[m] = (gf32x2){0.f, 0.f};
        const float *col = s_x + gy * GR * PX + lane;
        gauss_sums_packed<L>(acc, taps, [&](int q) { return (gf32x2){col[2 * q * PX], col[(2 * q + 1) * PX]}; });
        if (x < w) {
#pragma unroll
            for (int j = 0; j < GR; j++) {
                const int y = y0 + gy * GR + j;
                if (y < h) out[(z * h + y) * (i64)w + x] = (j & 1) ? acc[j / 2].x : acc[j / 2].y;
            }
        }
    }
}

// The fused x-y pass MARCHING down a strip (round 5): a work-group owns 64 columns x GXM_SEG rows of one slice and walks them in
// chunks of 64 output rows.  The x pass of a row is computed once and stays in LDS for the 2L + 1 output rows that need it (the tile
// kernel above recomputes the 2L halo rows of every 64-row tile: (64 + 2L) / 64 of the x arithmetic, 1.56 x at L = 18), a chunk is
// exactly two rounds of x tasks and two of y tasks for the 256 threads (the tile kernel: 3.1 rounds of x tasks), and the bytes of
// the next chunk are requested before the current one is computed, so no wave ever waits for the image.  Same sums, same order:
// bit-identical.  Buffer: rows [0, 2L) hold the x pass of the 2L rows above the chunk's own (carried over from the previous chunk
// by an LDS copy), rows [2L, 2L + 64) the chunk's new rows.
constexpr int GXM_SEG = 512;
template <int L>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void gauss_xy_u8_m(const uint8_t *__restrict__ img, float *__restrict__ out, int w, int h, int tiles_x, int segs,
                                                      const float *__restrict__ taps)
{
    static_assert(2 * L <= 64, "the carried rows fit one chunk");
    constexpr int CH = 64;
    constexpr int NRB = CH + 2 * L;
    constexpr int SPAN = 64 + 2 * L;
    constexpr int NDW = (SPAN + 6) / 4;
    constexpr int PB = ((SPAN + 3 + 3) / 4 * 4) | 4;
    constexpr int PX = 65;
    constexpr int NE = CH * NDW, NIT = (NE + 255) / 256; // dwords of a chunk's new rows, per thread
    __shared__ unsigned char s_u8[CH * PB];
    __shared__ float s_x[NRB * PX];
    unsigned b = xcd_contiguous(blockIdx.x, gridDim.x);
    const int x0 = (int)(b % (unsigned)tiles_x) * 64;
    b /= (unsigned)tiles_x;
    const int seg = (int)(b % (unsigned)segs);
    const i64 z = b / (unsigned)segs;
    const int ys = seg * GXM_SEG, ye = ys + GXM_SEG < h ? ys + GXM_SEG : h;
    const uint8_t *plane = img + z * (i64)w * h;
    const int tid = threadIdx.x;
    const bool interior = x0 - L >= 0 && x0 + 64 + L + 4 <= w; // whole (unaligned) dwords of a row, no clamping in x
    const int a = interior ? (x0 - L) & ~3 : 0, mis = interior ? (x0 - L) - a : 0;
    typedef unsigned __attribute__((aligned(1))) u32u;
    auto clampy = [&](int y) { return y < 0 ? 0 : (y > h - 1 ? h - 1 : y); };
    // the NR rows from ybase on: requested into registers (interior) ...
    auto request = [&](int ybase, unsigned (&qv)[NIT]) {
#pragma unroll
        for (int i = 0; i < NIT; i++) {
            const int e = tid + 256 * i;
            const bool on = i + 1 < NIT || e < NE;
            const int r = (on ? e : 0) / NDW, d = (on ? e : 0) - r * NDW;
            qv[i] = *(const u32u *)(plane + (i64)clampy(ybase + r) * w + a + 4 * d);
        }
    };
    auto deposit = [&](const unsigned (&qv)[NIT]) {
#pragma unroll
        for (int i = 0; i < NIT; i++) {
            const int e = tid + 256 * i;
            if (i + 1 < NIT || e < NE) {
                const int r = e / NDW, d = e - r * NDW;
                *(unsigned *)(s_u8 + r * PB + 4 * d) = qv[i];
            }
        }
    };
    // ... or byte by byte with clamp-to-edge in x and y (frangi.cpp:690, :725): the tiles at the left and right border
    auto stage_slow = [&](int ybase, int nrows) {
        for (int e = tid; e < nrows * SPAN; e += 256) {
            const int r = e / SPAN, cidx = e - r * SPAN;
            int x = x0 - L + cidx;
            x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
            s_u8[r * PB + cidx] = plane[(i64)clampy(ybase + r) * w + x];
        }
    };
    // x pass of rows [0, nrows) of s_u8 into rows [dst0, dst0 + nrows) of s_x; task = (row, chunk of GR outputs), lane = row
    auto xpass = [&](int nrows, int dst0) {
        for (int task = tid; task < nrows * (64 / GR); task += 256) {
            const int c = task / nrows, r = task - c * nrows;
            const unsigned char *src = s_u8 + r * PB + mis + c * GR;
            gf32x2 acc[GR / 2];
#pragma unroll
            for (int m = 0; m < GR / 2; m++) acc[m] = (gf32x2){0.f, 0.f};
            gauss_sums_packed<L>(acc, taps, [&](int q) { return (gf32x2){(float)src[2 * q], (float)src[2 * q + 1]}; });
#pragma unroll
            for (int j = 0; j < GR; j++) s_x[(dst0 + r) * PX + c * GR + j] = (j & 1) ? acc[j / 2].x : acc[j / 2].y;
        }
    };
    unsigned qv[NIT];
    // the 2L rows above the first chunk's own: rows ys - L .. ys + L - 1 (clamped at the top of the slice)
    if (interior) { request(ys - L, qv); deposit(qv); }
    else stage_slow(ys - L, 2 * L);
    __syncthreads();
    xpass(2 * L, 0);
    if (interior) request(ys + L, qv);
    const int lane = tid & 63;
    const int x = x0 + lane;
    for (int y0 = ys; y0 < ye; y0 += CH) {
        __syncthreads(); // the x pass has read s_u8, the y pass of the previous chunk has read s_x
        if (y0 > ys) { // carry the last 2L rows of the buffer to its top (source and destination are disjoint: 2L <= 64)
            constexpr int NC = 2 * L * 64, NCI = (NC + 255) / 256;
            float cv[NCI];
#pragma unroll
            for (int i = 0; i < NCI; i++) {
                const int e = tid + 256 * i;
                cv[i] = (i + 1 < NCI || e < NC) ? s_x[(CH + (e >> 6)) * PX + (e & 63)] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < NCI; i++) {
                const int e = tid + 256 * i;
                if (i + 1 < NCI || e < NC) s_x[(e >> 6) * PX + (e & 63)] = cv[i];
            }
        }
        if (interior) deposit(qv);
        else stage_slow(y0 + L, CH);
        __syncthreads();
        if (interior && y0 + CH < ye) request(y0 + CH + L, qv); // lands while this chunk is computed
        xpass(CH, 2 * L);
        __syncthreads();
        for (int gy = tid >> 6; gy < CH / GR; gy += 4) {
            gf32x2 acc[GR / 2];
#pragma unroll
            for (int m = 0; m < GR / 2; m++) acc[m] = (gf32x2){0.f, 0.f};
            const float *col = s_x + gy * GR * PX + lane;
            gauss_sums_packed<L>(acc, taps, [&](int q) { return (gf32x2){col[2 * q * PX], col[(2 * q + 1) * PX]}; });
            if (x < w) {
#pragma unroll
                for (int j = 0; j < GR; j++) {
                    const int y = y0 + gy * GR + j;
                    if (y < ye) out[(z * h + y) * (i64)w + x] = (j & 1) ? acc[j / 2].x : acc[j / 2].y;
                }
            }
        }
    }
}

static bool launch_gauss_xy_m(hipStream_t st, const uint8_t *src, float *dst, int w, int h, i64 l, const float *d_taps, int L)
{
    const int tiles_x = (w + 63) / 64, segs = (h + GXM_SEG - 1) / GXM_SEG;
    const i64 nblk = (i64)tiles_x * segs * l;
    if (nblk >= 2147483647LL) return false;
    const dim3 grid((unsigned)nblk);
#define PNR_GXM(LL) case LL: hipLaunchKernelGGL(gauss_xy_u8_m<LL>, grid, dim3(256), 0, st, src, dst, w, h, tiles_x, segs, d_taps); return true;
    switch (L) {
        PNR_GXM(6) PNR_GXM(12) PNR_GXM(18)
    default: return false;
    }
#undef PNR_GXM
}

static bool launch_gauss_xy_t(hipStream_t st, const uint8_t *src, float *dst, int w, int h, i64 l, const float *d_taps, int L)
{
    const int tiles_x = (w + 63) / 64, tiles_y = (h + GXY_TY - 1) / GXY_TY;
    const i64 nblk = (i64)tiles_x * tiles_y * l;
    if (nblk >= 2147483647LL) return false;
    const dim3 grid((unsigned)nblk);
#define PNR_GXY(LL) case LL: hipLaunchKernelGGL(gauss_xy_u8_t<LL>, grid, dim3(256), 0, st, src, dst, w, h, tiles_x, tiles_y, d_taps); return true;
    // Measured per radius at 1024^3 (profiles/r03_frangi_fused_xy.txt): L = 6 2.52 ms against 1.89 + 1.82 for the two passes, L = 12 4.42
    // against 2.26 + 2.18 (a tie in time, 8.6 GB less HBM traffic), L = 18 8.2 against 2.63 + 2.47 -- the x pass of the 2L halo rows
    // ((64 + 36) / 64 of the arithmetic at L = 18) costs more than the round trip saves.  Fused up to L = 12 only.
    switch (L) {
        PNR_GXY(6) PNR_GXY(12) PNR_GXY(18)
    default: return false;
    }
#undef PNR_GXY
}

// launch the strided-axis pass: the templated kernel for the radii the default parameters produce, the generic one otherwise
static void launch_gauss_axis(hipStream_t st, const float *in, float *out, int w, int n_axis, i64 axis_stride, int n_other, i64 other_stride,
                              const float *d_taps, int L)
{
    const int tiles_x = (w + 63) / 64;
    int tiles_a = (n_axis + TAT - 1) / TAT;
    dim3 grid((unsigned)((i64)tiles_x * tiles_a * n_other));
#define PNR_GA(LL) case LL: hipLaunchKernelGGL(gauss_axis_t<LL>, grid, dim3(64 * GT), 0, st, in, out, w, n_axis, axis_stride, n_other, other_stride, tiles_x, tiles_a, d_taps); return;
    switch (L) {
        PNR_GA(2) PNR_GA(3) PNR_GA(5) PNR_GA(6) PNR_GA(9) PNR_GA(12) PNR_GA(18) PNR_GA(24)
    default: break;
    }
#undef PNR_GA
    tiles_a = (n_axis + TA - 1) / TA;
    grid = dim3((unsigned)((i64)tiles_x * tiles_a * n_other));
    const size_t sm = ((size_t)(TA + 2 * L) * 64 + 2 * L + 1) * 4;
    hipLaunchKernelGGL(gauss_axis, grid, dim3(256), sm, st, in, out, w, n_axis, axis_stride, n_other, other_stride, tiles_x, tiles_a, d_taps, L);
}

// ----------------------------------------------------------------------------------------
// K4: Hessian (frangi.cpp:305-381) + eigen (:1269-1493) + vesselness (:190-273)
// ----------------------------------------------------------------------------------------
// first difference along an axis (stride s, coordinate c of n): one-sided at the borders
__device__ __forceinline__ float d1(const float *__restrict__ F, i64 i, i64 s, int c, int n)
{
    if (c == 0) return F[i + s] - F[i];
    if (c < n - 1) return 0.5f * (F[i + s] - F[i - s]); // .5*(a-b): exact scaling of the f32 difference
    return F[i] - F[i - s];
}
// difference along `so` of the first difference along `si`; same != 0 when both are one axis
__device__ __forceinline__ float d2(const float *__restrict__ F, i64 i, i64 si, int ci, int ni, i64 so, int co, int no,
                                    int same)
{
    if (co == 0) return d1(F, i + so, si, ci + same, ni) - d1(F, i, si, ci, ni);
    if (co < no - 1) return 0.5f * (d1(F, i + so, si, ci + same, ni) - d1(F, i - so, si, ci - same, ni));
    return d1(F, i, si, ci, ni) - d1(F, i - so, si, ci - same, ni);
}

struct Sym3 {
    double V[3][3];
    double d[3];
};

// Householder tridiagonalisation (EISPACK/JAMA tred2), n = 3, fully unrolled: all indices static
__device__ __forceinline__ void tridiag3(double (&V)[3][3], double (&d)[3], double (&e)[3])
{
#pragma unroll
    for (int j = 0; j < 3; j++) d[j] = V[2][j];
#pragma unroll
    for (int i = 2; i > 0; i--) {
        double scale = 0.0, h = 0.0;
#pragma unroll
        for (int k = 0; k < i; k++) scale = scale + fabs(d[k]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
#pragma unroll
            for (int j = 0; j < i; j++) {
                d[j] = V[i - 1][j];
                V[i][j] = 0.0;
                V[j][i] = 0.0;
            }
        } else {
#pragma unroll
            for (int k = 0; k < i; k++) {
                d[k] /= scale;
                h += d[k] * d[k];
            }
            double f = d[i - 1];
            double g = sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g;
            h = h - f * g;
            d[i - 1] = f - g;
#pragma unroll
            for (int j = 0; j < i; j++) e[j] = 0.0;
#pragma unroll
            for (int j = 0; j < i; j++) {
                f = d[j];
                V[j][i] = f;
                g = e[j] + V[j][j] * f;
#pragma unroll
                for (int k = j + 1; k <= i - 1; k++) {
                    g += V[k][j] * d[k];
                    e[k] += V[k][j] * f;
                }
                e[j] = g;
            }
            f = 0.0;
#pragma unroll
            for (int j = 0; j < i; j++) {
                e[j] /= h;
                f += e[j] * d[j];
            }
            const double hh = f / (h + h);
#pragma unroll
            for (int j = 0; j < i; j++) e[j] -= hh * d[j];
#pragma unroll
            for (int j = 0; j < i; j++) {
                f = d[j];
                g = e[j];
#pragma unroll
                for (int k = j; k <= i - 1; k++) V[k][j] -= (f * e[k] + g * d[k]);
                d[j] = V[i - 1][j];
                V[i][j] = 0.0;
            }
        }
        d[i] = h;
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        V[2][i] = V[i][i];
        V[i][i] = 1.0;
        const double h = d[i + 1];
        if (h != 0.0) {
#pragma unroll
            for (int k = 0; k <= i; k++) d[k] = V[k][i + 1] / h;
#pragma unroll
            for (int j = 0; j <= i; j++) {
                double g = 0.0;
#pragma unroll
                for (int k = 0; k <= i; k++) g += V[k][i + 1] * V[k][j];
#pragma unroll
                for (int k = 0; k <= i; k++) V[k][j] -= g * d[k];
            }
        }
#pragma unroll
        for (int k = 0; k <= i; k++) V[k][i + 1] = 0.0;
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
        d[j] = V[2][j];
        V[2][j] = 0.0;
    }
    V[2][2] = 1.0;
    e[0] = 0.0;
}

__device__ __forceinline__ double hyp2(double a, double b) { return sqrt(a * a + b * b); }

// one Givens step of the implicit QL sweep at static position I (rotates columns I, I+1)
template <int I>
__device__ __forceinline__ void ql_rotate(double (&V)[3][3], double (&d)[3], double (&e)[3], double &p, double &c,
                                          double &c2, double &c3, double &s, double &s2)
{
    c3 = c2;
    c2 = c;
    s2 = s;
    double g = c * e[I];
    double h = c * p;
    const double r = hyp2(p, e[I]);
    e[I + 1] = s * r;
    s = e[I] / r;
    c = p / r;
    p = c * d[I] - s * g;
    d[I + 1] = h + s * (c * g + s * d[I]);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        h = V[k][I + 1];
        V[k][I + 1] = s * V[k][I] + c * h;
        V[k][I] = c * V[k][I] - s * h;
    }
}

// implicit-shift QL for row L of the tridiagonal matrix (JAMA tql2 body), static L
template <int L>
__device__ __forceinline__ void ql_row(double (&V)[3][3], double (&d)[3], double (&e)[3], double &f, double &tst1)
{
    const double eps = 2.220446049250313e-16; // 2^-52
    const double t = fabs(d[L]) + fabs(e[L]);
    tst1 = (tst1 > t) ? tst1 : t;
    int m = L;
#pragma unroll
    for (int q = L; q < 3; q++) { // while (m < n) { if (|e[m]| <= eps*tst1) break; m++; }
        if (m == q && !(fabs(e[q]) <= eps * tst1)) m = q + 1;
    }
    if (L < 2 && m > L) {
        constexpr int L1 = (L < 2) ? L + 1 : 2; // keeps indices in range for the never-taken L == 2 case
        do {
            double g = d[L];
            double p = (d[L1] - g) / (2.0 * e[L]);
            double r = hyp2(p, 1.0);
            if (p < 0) r = -r;
            d[L] = e[L] / (p + r);
            d[L1] = e[L] * (p + r);
            const double dl1 = d[L1];
            double h = g - d[L];
            if (L == 0) d[2] -= h; // for (i = l+2; i < n; i++) d[i] -= h
            f = f + h;
            // m == 3 only if e[2] failed the test, impossible (e[2] == 0): d[m] is d[1] or d[2]
            p = (m == 2) ? d[2] : d[1];
            double c = 1.0, c2 = 1.0, c3 = 1.0;
            const double el1 = e[L1];
            double s = 0.0, s2 = 0.0;
            if (m - 1 >= 1 && 1 >= L) ql_rotate<1>(V, d, e, p, c, c2, c3, s, s2);
            if (L == 0) ql_rotate<0>(V, d, e, p, c, c2, c3, s, s2); // m-1 >= 0 always here
            p = -s * s2 * c3 * el1 * e[L] / dl1;
            e[L] = s * p;
            d[L] = c * p;
        } while (fabs(e[L]) > eps * tst1);
    }
    d[L] = d[L] + f;
    e[L] = 0.0;
}

__device__ __forceinline__ void swap_col(double (&V)[3][3], double (&d)[3], int a, int b)
{
    // only called with literal a,b after inlining
    double t = d[a]; d[a] = d[b]; d[b] = t;
#pragma unroll
    for (int r = 0; r < 3; r++) { t = V[r][a]; V[r][a] = V[r][b]; V[r][b] = t; }
}

// Frangi::eigen_decomposition: eigenvalues sorted by |lambda| ascending, columns = eigenvectors
__device__ __forceinline__ void eigen3(double (&V)[3][3], double (&d)[3])
{
    double e[3];
    tridiag3(V, d, e);
    e[0] = e[1]; // for (i = 1; i < n; i++) e[i-1] = e[i]; e[n-1] = 0
    e[1] = e[2];
    e[2] = 0.0;
    double f = 0.0, tst1 = 0.0;
    ql_row<0>(V, d, e, f, tst1);
    ql_row<1>(V, d, e, f, tst1);
    ql_row<2>(V, d, e, f, tst1);
    // ascending selection sort (tql2 tail), static indices
    {
        int k = 0;
        double p = d[0];
        if (d[1] < p) { k = 1; p = d[1]; }
        if (d[2] < p) { k = 2; p = d[2]; }
        if (k == 1) swap_col(V, d, 0, 1);
        else if (k == 2) swap_col(V, d, 0, 2);
        if (d[2] < d[1]) swap_col(V, d, 1, 2);
    }
    // re-sort by |lambda| (frangi.cpp:1286-1304)
    const double a0 = fabs(d[0]), a1 = fabs(d[1]), a2 = fabs(d[2]);
    double b0 = a0, b1 = a1;
    if ((a0 >= a1) && (a0 > a2)) { swap_col(V, d, 0, 2); b0 = a2; }
    else if ((a1 >= a0) && (a1 > a2)) { swap_col(V, d, 1, 2); b1 = a2; }
    if (b0 > b1) swap_col(V, d, 0, 1);
}

__device__ __forceinline__ unsigned int f2ord(float f)
{
    const unsigned int b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ __forceinline__ float ord2f(unsigned int u)
{
    const unsigned int b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    union { unsigned int i; float f; } cv;
    cv.i = b;
    return cv.f;
}

__device__ __forceinline__ unsigned char quant_dir(double v)
{
    const double r = ((v + 1) / 2) * 255;
    int val = (int)((r > 0.0) ? floor(r + 0.5) : ceil(r - 0.5)); // round(), Advantra_plugin.cpp:120-123
    val = (val < 0) ? 0 : (val > 255) ? 255 : val;
    return (unsigned char)val;
}

struct HessOut {
    float *Dzz, *Dyy, *Dyz, *Dxx, *Dxy, *Dxz;
};

// Frangi vesselness of one voxel from its eigenvalues sorted by |lambda| (frangi.cpp:218-231)
__device__ __forceinline__ double vesselness(const double (&d)[3], float two_a2, float two_b2, float two_c2)
{
    const double L2 = d[1], L3 = d[2];
    const double a1 = fabs(d[0]), a2 = fabs(L2), a3 = fabs(L3);
    const double Ra = a2 / a3;
    const double Rb = a1 / sqrt(a2 * a3);
    const double S = sqrt(a1 * a1 + a2 * a2 + a3 * a3);
    const double expRa = (1 - exp(-((Ra * Ra) / (double)two_a2)));
    const double expRb = exp(-((Rb * Rb) / (double)two_b2));
    const double expS = (1 - exp(-(S * S) / (double)two_c2));
    double vox = expRa * expRb * expS;
    vox = (L2 > 0) ? 0 : vox;
    vox = (L3 > 0) ? 0 : vox;
    vox = (vox != vox) ? 0 : vox; // NaN -> 0
    return vox;
}

// ---- K4a: the Hessian stencil, LDS-tiled ------------------------------------------------------------------------------
// A work-group owns a column of HT_X x HT_Y voxels and marches HT_Z planes along z with a ring of six planes (tile + halo 2)
// in LDS: every plane is fetched from HBM once per work-group (1.6x the tile for the x / y halo, 1.8x with the z halo of a
// 32-plane march) instead of once per stencil point.  Each voxel's six second derivatives go through the test that proves the
// response zero (below); the survivors are appended to the work-group's own region of a queue in HBM (no global atomics) for
// the eigen-solver kernel, so that kernel runs with full wavefronts whatever the survival rate.
// (HT_VY voxels per thread, rows ty, ty + 8, ... of the tile.  Two per thread -- a 64 x 16 tile, the per-plane scalar work shared by
// twice the voxels, the x / y halo 1.33 instead of 1.59 x the tile -- was measured in round 4: 93 VGPRs instead of 62, two instead of
// four work-groups per CU, 16.2 instead of 13.3 ms per stack.  The kernel lives on its occupancy: one voxel per thread.)
#ifndef PNR_HT_TY
#define PNR_HT_TY 8
#define PNR_HT_YB 3
#endif
#ifndef PNR_HT_Z
#define PNR_HT_Z 32
#endif
constexpr int HT_X = 64, HT_VY = 1, HT_TY = PNR_HT_TY, HT_Y = HT_TY * HT_VY, HT_Z = PNR_HT_Z, HT_PX = HT_X + 4, HT_PY = HT_Y + 4, HT_PLANE = HT_PX * HT_PY, HT_THREADS = HT_X * HT_TY;
constexpr int HT_YBITS = PNR_HT_YB, HT_POSBITS = 6 + HT_YBITS; // queue entry: (z - z0) << HT_POSBITS | y in tile << 6 | x in tile
static_assert(HT_X == 64 && (1 << HT_YBITS) == HT_Y, "the position code of a queue entry");
constexpr int HT_LD = (HT_PLANE + HT_THREADS - 1) / HT_THREADS; // halo'd plane elements a thread fetches
constexpr int HT_REGION = HT_X * HT_Y * HT_Z; // queue entries a work-group can produce
constexpr int HT_RING = 6;

struct Tile {
    const float *pl[5]; // planes z-2 .. z+2 of the ring
    int o;              // this thread's voxel inside a plane
};
__device__ __forceinline__ float t_at(const Tile &T, int dx, int dy, int dz) { return T.pl[dz + 2][T.o + dy * HT_PX + dx]; }
// first difference along axis A at the voxel displaced by (dx,dy,dz), whose coordinate along A is c of n: frangi.cpp:305-381
template <int A>
__device__ __forceinline__ float td1(const Tile &T, int dx, int dy, int dz, int c, int n)
{
    constexpr int ax = A == 0, ay = A == 1, az = A == 2;
    if (c == 0) return t_at(T, dx + ax, dy + ay, dz + az) - t_at(T, dx, dy, dz);
    if (c < n - 1) return 0.5f * (t_at(T, dx + ax, dy + ay, dz + az) - t_at(T, dx - ax, dy - ay, dz - az));
    return t_at(T, dx, dy, dz) - t_at(T, dx - ax, dy - ay, dz - az);
}
// difference along AO of the first difference along AI (d2 above, on the tile)
template <int AI, int AO>
__device__ __forceinline__ float td2(const Tile &T, int ci, int ni, int co, int no)
{
    constexpr int same = AI == AO, ox = AO == 0, oy = AO == 1, oz = AO == 2;
    if (co == 0) return td1<AI>(T, ox, oy, oz, ci + same, ni) - td1<AI>(T, 0, 0, 0, ci, ni);
    if (co < no - 1) return 0.5f * (td1<AI>(T, ox, oy, oz, ci + same, ni) - td1<AI>(T, -ox, -oy, -oz, ci - same, ni));
    return td1<AI>(T, 0, 0, 0, ci, ni) - td1<AI>(T, -ox, -oy, -oz, ci - same, ni);
}

// The same second derivative where every rule is the centred one (no voxel of the stencil within 2 of a border), times sigma^2:
// td2 computes 0.5 (0.5 a - 0.5 b) with a, b two f32 differences, then x s2.  Scaling by a power of two is exact and commutes with
// rounding (no operand here can underflow: a difference of two Gaussian sums of a u8 image is 0 or above 1e-12), so that value is
// fl(fl(a - b) x (0.25 s2)) -- three subtractions and one multiplication instead of three, three and one: the same bits (the golden
// Hessians decide, tests/test_gpu_frangi.py) for 18 vector instructions per voxel less in a kernel that issues ~90.
template <int AI, int AO>
__device__ __forceinline__ float td2c(const Tile &T, float q4)
{
    constexpr int ix = AI == 0, iy = AI == 1, iz = AI == 2, ox = AO == 0, oy = AO == 1, oz = AO == 2;
    const float a = t_at(T, ox + ix, oy + iy, oz + iz) - t_at(T, ox - ix, oy - iy, oz - iz);
    const float b = t_at(T, -ox + ix, -oy + iy, -oz + iz) - t_at(T, -ox - ix, -oy - iy, -oz - iz);
    return (a - b) * q4;
}

struct HessQueue {
    float *h;            // [region][6][HT_REGION]
    unsigned int *idx;   // [region][HT_REGION]: (z - z0) << HT_POSBITS | y in tile << 6 | x in tile
    unsigned int *count; // [region]
};

#ifndef PNR_HT_EU
#define PNR_HT_EU 8
#endif
#ifdef PNR_HT_STAMPS
// diagnostic build only (make variant NAME=hts DEFS=-DPNR_HT_STAMPS; scripts/ht_stamps.py): shader-clock sums per wave over the INNER planes
// of hessian_tile: [0] stencil (LDS reads + the six derivatives), [1] tests + queue, [2] the wait for the plane requested two planes
// ago + its LDS store, [3] the barrier, [4] wave-planes counted
__device__ unsigned long long g_ht_stamps[8];
#define HT_STAMP_V(t, vdep) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(vdep) : "memory")
#define HT_STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory")
#endif
template <bool DUMP>
__global__ __launch_bounds__(HT_THREADS) __attribute__((amdgpu_waves_per_eu(PNR_HT_EU, 8))) void hessian_tile(const float *__restrict__ F, int w, int h, int l, int tiles_x, int tiles_y, int zc0, int zc1,
                                                           float s2, HessQueue Q, unsigned int *__restrict__ minmax, int first, int zs0, int zs1, HessOut dump,
                                                           float two_c2, int prune)
{
    __shared__ float ring[HT_RING][HT_PLANE];
    __shared__ unsigned int s_cnt, s_zero;
    __shared__ double s_s2max;
    const int tid = threadIdx.x, tx = tid & (HT_X - 1), ty = tid >> 6;
    // consecutive work-groups go round-robin to the 8 XCDs, each with its own L2: give every XCD a contiguous range of tiles, so
    // that the halo a tile shares with its x / y neighbours is found in the L2 of the same XCD
    const unsigned int region = xcd_contiguous(blockIdx.x, gridDim.x);
    unsigned int b = region;
    const int bx = (int)(b % (unsigned)tiles_x); b /= (unsigned)tiles_x;
    const int by = (int)(b % (unsigned)tiles_y);
    const int bz = (int)(b / (unsigned)tiles_y);
    const int x0 = bx * HT_X, y0 = by * HT_Y, z0 = zc0 + bz * HT_Z;
    const int z1 = z0 + HT_Z < zc1 ? z0 + HT_Z : zc1;
    const i64 wh = (i64)w * h;
    if (tid == 0) {
        s_cnt = 0; s_zero = 0;
        // Fourth way past the solver (option frangi_prune): a response that cannot reach the first non-zero J8 level.  The response
        // is at most its factor 1 - exp(-S^2 / 2c^2), and S^2 = l1^2 + l2^2 + l3^2 is the squared Frobenius norm of the matrix -- no
        // eigenvalues needed.  J8 = round(255 (J - Jmin) / (Jmax - Jmin)) is 0 below (Jmax - Jmin) / 510 (Advantra_plugin.cpp:2499-2512);
        // once an exact 0 has been written in the planes this context owns, Jmin is 0 for good (a response is never negative), and
        // the largest response written SO FAR is a lower bound of the final Jmax (of the global one when the stack is sharded).  A
        // voxel whose bound lies below that / 510 -- with margins far above the rounding of the solver, of 1 - exp() and of the
        // quantisation -- ends as J8 = 0 whatever its exact response: it is skipped like a proven zero, except that it is none (it
        // does not establish Jmin = 0).  J8, Jmin, Jmax and every voxel with J8 > 0 (all seeds) keep their exact values and winning
        // scale; the f32 J and the scale of J8 = 0 voxels do not, so pnr_get_frangi recomputes without this test when J or the
        // direction volumes are asked for.  Which voxels are skipped depends on the order the work-groups run in; the results above do not.
        double s2max = 0.0;
        if (!DUMP && prune) {
            const unsigned int mn = __hip_atomic_load(&minmax[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int mx = __hip_atomic_load(&minmax[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (mn == f2ord(0.f) && mx != 0u) {
                const double th = (double)ord2f(mx) / 510.0 * (1.0 - 1e-5) - 1e-15;
                if (th > 0.0 && th < 0.5) s2max = -(double)two_c2 * log1p(-th) * (1.0 - 1e-9);
            }
        }
        s_s2max = s2max;
    }
    // a plane of the ring: HT_PLANE floats, HT_LD per thread; out-of-volume halo cells repeat the border (never read: the
    // one-sided border rules of the reference do not look past the border)
    auto clampi = [](int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); };
    i64 gofs[HT_LD];
    bool hasv[HT_LD];
#pragma unroll
    for (int q = 0; q < HT_LD; q++) {
        const int e = tid + q * HT_THREADS;
        hasv[q] = e < HT_PLANE;
        const int r = (hasv[q] ? e : 0) / HT_PX, cc = (hasv[q] ? e : 0) - r * HT_PX;
        gofs[q] = (i64)clampi(y0 - 2 + r, h - 1) * w + clampi(x0 - 2 + cc, w - 1);
    }
    struct PlaneVals { float v[HT_LD]; };
    // (every thread loads HT_LD elements -- a thread without a last element re-reads element 0 and drops it in put(): a load behind an
    // exec-mask branch makes the number of loads in flight depend on the path, and the compiler then waits for ALL of them, vmcnt(0),
    // where the ring needs the oldest plane only)
    auto fetch = [&](int zp, PlaneVals &pv) {
        const i64 zo = (i64)clampi(zp, l - 1) * wh;
#pragma unroll
        for (int q = 0; q < HT_LD; q++) pv.v[q] = F[zo + gofs[q]];
    };
    auto put = [&](int slot, const PlaneVals &pv) {
#pragma unroll
        for (int q = 0; q < HT_LD; q++)
            if (q + 1 < HT_LD || hasv[q]) ring[slot][tid + q * HT_THREADS] = pv.v[q];
    };
    // slots: plane zp lives in slot (zp - (z0 - 2)) mod 6
    // Two planes are in flight per work-group: plane z + 3 was requested during plane z - 1 and is stored behind plane z, plane
    // z + 4 is requested at the start of plane z.  (With one plane in flight the kernel sat on its memory latency: four work-groups
    // x 3.3 KB per CU are 3.3 MB in flight on the chip, 1.3 - 1.6 TB/s at 2 - 2.5 us -- what it ran at, with half its VALU slots idle.)
    PlaneVals pend; // plane z + 3 of the plane z about to be computed
    {
        PlaneVals pv[5];
#pragma unroll
        for (int k = 0; k < 5; k++) fetch(z0 - 2 + k, pv[k]);
        if (z0 + 1 < z1) fetch(z0 + 3, pend);
#pragma unroll
        for (int k = 0; k < 5; k++) put(k, pv[k]);
    }
    __syncthreads();
    const int x = x0 + tx;
    const bool interior_xy = x0 >= 2 && x0 + HT_X + 2 <= w && y0 >= 2 && y0 + HT_Y + 2 <= h; // block-uniform
    float *const qh = DUMP ? nullptr : Q.h + (size_t)region * 6 * HT_REGION;
    unsigned int *const qi = DUMP ? nullptr : Q.idx + (size_t)region * HT_REGION;
#ifdef PNR_HT_STAMPS
    unsigned long long hs0 = 0, hs1 = 0, hs2 = 0, hs3 = 0, hsn = 0;
#endif
    bool zero_here = false;
    const double s2max = s_s2max; // (written before the barrier above)
    const float s2max_f = (float)(s2max * 0.99999);
    const float q4 = 0.25f * s2; // (exact)
    // One plane of the march.  `s0` = ring slot of plane z - 2.  FAST: no voxel of this plane of the tile is within 2 of a border (the
    // centred differences everywhere) AND s0 is a compile-time constant: the 19 stencil reads are ds_read_b32 with immediate offsets
    // from one address register and the slot arithmetic is gone.  Why it matters (PMC, profiles/r04_frangi_pmc_baseline.txt): the
    // kernel issued 117 scalar instructions per plane and wave against 92 vector ones -- ring-slot selects, 64-bit plane offsets, the
    // exec-mask bookkeeping of the border rules -- and the scalar unit is shared by the four SIMDs of a CU.
    // INNER: a FAST plane with at least two more planes of the march behind it -- the request for plane z + 4 and the store of plane
    // z + 3 are unconditional, so no scalar branch stands between the loads and the wait in front of the store and the compiler can
    // count: it waits for the older plane (vmcnt(HT_LD)) and leaves the newer one in flight.  With the conditions in place the ISA had
    // s_waitcnt vmcnt(0) there: ONE plane in flight, not two, and the kernel sat on its memory latency.
    auto plane = [&](const int z, const int s0, auto fast_tag, auto inner_tag) {
        constexpr bool FAST = decltype(fast_tag)::value, INNER = decltype(inner_tag)::value;
        static_assert(FAST || !INNER, "");
        PlaneVals nx;
        const bool more = INNER || z + 1 < z1;
#ifdef PNR_HT_STAMPS
        unsigned long long ht0 = 0, ht1 = 0, ht2 = 0, ht3 = 0, ht4 = 0;
        if (INNER) HT_STAMP(ht0);
#endif
        if (INNER || z + 2 < z1) fetch(z + 4, nx); // lands while this plane and the next are computed
        Tile T;
#pragma unroll
        for (int k = 0; k < 5; k++) { const int sl = s0 + k; T.pl[k] = ring[sl >= HT_RING ? sl - HT_RING : sl]; }
#pragma unroll
        for (int v = 0; v < HT_VY; v++) {
        const int yt = ty + v * HT_TY, y = y0 + yt; // row of the tile / of the volume
        const bool inside = x < w && y < h;
        T.o = (yt + 2) * HT_PX + tx + 2;
        bool surv = false, skipped = false;
        float Dzz = 0, Dyy = 0, Dyz = 0, Dxx = 0, Dxy = 0, Dxz = 0;
        if (FAST || inside) { // (a tile clear of the borders lies inside the volume)
            // six second derivatives, each x sigma^2 (frangi.cpp:319,339,345,368,374,380)
            if (FAST || (interior_xy && z >= 2 && z + 2 < l)) {
                // the same operations as the general path below takes for such a voxel (coordinate 2 of 5 stands for "interior"), but
                // with the border rules resolved at compile time: 19 LDS reads instead of every variant's
                Dzz = td2c<2, 2>(T, q4);
                Dyy = td2c<1, 1>(T, q4);
                Dyz = td2c<1, 2>(T, q4);
                Dxx = td2c<0, 0>(T, q4);
                Dxy = td2c<0, 1>(T, q4);
                Dxz = td2c<0, 2>(T, q4);
            } else {
                Dzz = td2<2, 2>(T, z, l, z, l) * s2;
                Dyy = td2<1, 1>(T, y, h, y, h) * s2;
                Dyz = td2<1, 2>(T, y, h, z, l) * s2;
                Dxx = td2<0, 0>(T, x, w, x, w) * s2;
                Dxy = td2<0, 1>(T, x, w, y, h) * s2;
                Dxz = td2<0, 2>(T, x, w, z, l) * s2;
            }
#ifdef PNR_HT_STAMPS
            if (INNER) HT_STAMP_V(ht1, ((Dzz + Dyy) + (Dyz + Dxx)) + (Dxy + Dxz));
#endif
            if (DUMP) {
                const i64 i = (i64)z * wh + (i64)y * w + x;
                dump.Dzz[i] = Dzz; dump.Dyy[i] = Dyy; dump.Dyz[i] = Dyz;
                dump.Dxx[i] = Dxx; dump.Dxy[i] = Dxy; dump.Dxz[i] = Dxz;
            } else {
                // A response > 0 needs lambda2 <= 0 and lambda3 <= 0 (the two largest-magnitude eigenvalues).  Then
                // trace = l1+l2+l3 <= |l2| + l2 + l3 = l3 <= 0.  So a trace that is positive by a margin far above the
                // solver's rounding error (1e-9 of the matrix 1-norm vs ~1e-15) proves the response is exactly 0: at the first
                // scale J stays the 0 it was cleared to, at later scales the voxel cannot beat J >= 0 -- no eigen-solver either
                // way.  (NaN compares false: no skip.)
                // First in f32, conservatively -- most voxels are settled here and the f64 tests below run for the waves that
                // still hold an unsettled voxel only (in a pruned run, one wave in a few).  (a) The trace: the f32 sum is within
                // 2 * 2^-24 of the norm of the exact one, so a trace above 1e-5 of the norm proves what the f64 test asks for
                // (> 1e-9 of the norm).  (b) The J8 bound: six non-negative squares, relative error below 1e-6 in f32, against the
                // bound lowered by 1e-5.  NaN / inf compare false: such a voxel stays unsettled and takes the f64 path as before.
                const float nrmf = fabsf(Dxx) + fabsf(Dyy) + fabsf(Dzz) + 2.f * (fabsf(Dxy) + fabsf(Dxz) + fabsf(Dyz));
                const float trf = (Dxx + Dyy) + Dzz;
                const float S2f = (Dxx * Dxx + Dyy * Dyy + Dzz * Dzz) + 2.f * (Dxy * Dxy + Dxz * Dxz + Dyz * Dyz);
                const bool zero_f = nrmf > 1e-30f && trf > 1e-5f * nrmf;
                skipped = !zero_f && S2f < s2max_f; // (s2max_f = 0 without the shortcut)
                const bool open = !(zero_f || skipped);
                if (__builtin_amdgcn_ballot_w64(open) != 0ull) { // wave-uniform
                    // A response > 0 needs lambda2 <= 0 and lambda3 <= 0 (the two largest-magnitude eigenvalues).  Then
                    // trace = l1+l2+l3 <= |l2| + l2 + l3 = l3 <= 0.  So a trace that is positive by a margin far above the
                    // solver's rounding error (1e-9 of the matrix 1-norm vs ~1e-15) proves the response is exactly 0: at the first
                    // scale J stays the 0 it was cleared to, at later scales the voxel cannot beat J >= 0 -- no eigen-solver either
                    // way.  (NaN compares false: no skip.)
                    const double tr = (double)Dxx + (double)Dyy + (double)Dzz;
                    const double nrm = fabs((double)Dxx) + fabs((double)Dyy) + fabs((double)Dzz) + 2.0 * (fabs((double)Dxy) + fabs((double)Dxz) + fabs((double)Dyz));
                    surv = open && !(tr > 1e-9 * nrm);
                    const double xx = Dxx, yy = Dyy, zz = Dzz, xy = Dxy, xz = Dxz, yz = Dyz;
                    // cannot reach J8 = 1 (see the top of the kernel): skipped, but not a proven zero
                    const double S2 = (xx * xx + yy * yy + zz * zz) + 2.0 * (xy * xy + xz * xz + yz * yz);
                    const bool skip_d = surv && s2max > 0.0 && S2 < s2max;
                    skipped = skipped || skip_d;
                    surv = surv && !skip_d;
                    if (__builtin_amdgcn_ballot_w64(surv) != 0ull) {
                        // Second proof of a zero response: TWO positive eigenvalues.  With all roots real, Descartes' rule on the
                        // characteristic polynomial l^3 - c2 l^2 + c1 l - c0 (c2 = trace <= 0 here) gives exactly two positive roots iff
                        // c1 < 0 and c0 < 0; then (p1 + p2) |n| > -c1 puts the larger positive root above 5e-10 of the norm -- far above
                        // the solver's rounding -- and at least one of the two largest-magnitude eigenvalues is that positive one.
                        const double c1 = (xx * yy - xy * xy) + (xx * zz - xz * xz) + (yy * zz - yz * yz);
                        const double c0 = xx * (yy * zz - yz * yz) - xy * (xy * zz - yz * xz) + xz * (xy * yz - yy * xz);
                        const double n2 = nrm * nrm, n3 = n2 * nrm;
                        // Third: ONE positive eigenvalue p (c0 > 0: n1 <= n2 < 0 < p) that is not the smallest in magnitude.  The
                        // product of the pairwise sums (n1 + n2)(n1 + p)(n2 + p) = c2 c1 - c0 is positive only if n2 + p > 0 (and
                        // n1 + p < 0); above the margin that puts p at least 2.5e-10 of the norm beyond |n2|, so p is one of the two
                        // largest-magnitude eigenvalues and the response is 0.
                        const bool two_pos = c1 < -1e-9 * n2 && c0 < -1e-9 * n3;
                        const bool one_pos_big = c0 > 1e-9 * n3 && (tr * c1 - c0) > 1e-9 * n3;
                        surv = surv && !(two_pos || one_pos_big);
                    }
                }
                zero_here = zero_here || (!surv && !skipped && z >= zs0 && z < zs1);
            }
        }
        if (!DUMP) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(surv);
            if (m) {
                unsigned int base = 0;
                if ((tid & 63) == 0) base = atomicAdd(&s_cnt, (unsigned int)__builtin_popcountll(m));
                base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
                if (surv) {
                    const unsigned int p = base + (unsigned int)__builtin_popcountll(m & ((1ull << (tid & 63)) - 1ull));
                    qh[0 * HT_REGION + p] = Dxx; qh[1 * HT_REGION + p] = Dxy; qh[2 * HT_REGION + p] = Dxz;
                    qh[3 * HT_REGION + p] = Dyy; qh[4 * HT_REGION + p] = Dyz; qh[5 * HT_REGION + p] = Dzz;
                    qi[p] = ((unsigned int)(z - z0) << HT_POSBITS) | (unsigned int)(yt << 6) | (unsigned int)tx;
                }
            }
        }
        } // v
#ifdef PNR_HT_STAMPS
        if (INNER) HT_STAMP(ht2);
#endif
        if (more) {
            const int sl = s0 + 5; // plane z + 3 replaces plane z - 3, which nobody reads any more
            put(sl >= HT_RING ? sl - HT_RING : sl, pend);
        }
#ifdef PNR_HT_STAMPS
        if (INNER) HT_STAMP(ht3);
#endif
        pend = nx;
        __syncthreads();
#ifdef PNR_HT_STAMPS
        if (INNER) { HT_STAMP(ht4); hs0 += ht1 - ht0; hs1 += ht2 - ht1; hs2 += ht3 - ht2; hs3 += ht4 - ht3; hsn++; }
#endif
    };
    // the march: runs of HT_RING planes with compile-time ring slots wherever the tile and the planes are clear of every border,
    // single planes with the general rules otherwise (the first / last two planes of the stack, tiles at an x / y border, the
    // remainder of a march)
    int z = z0;
    int s0 = 0; // slot of plane z - 2 = (z - z0) mod HT_RING
#pragma unroll 1
    while (z < z1) {
        if (interior_xy && s0 == 0 && z >= 2 && z + HT_RING + 2 <= l && z + HT_RING + 2 <= z1) {
#pragma unroll
            for (int k = 0; k < HT_RING; k++) plane(z + k, k, std::true_type{}, std::true_type{});
            z += HT_RING;
        } else if (interior_xy && s0 == 0 && z >= 2 && z + HT_RING + 2 <= l && z + HT_RING <= z1) {
#pragma unroll
            for (int k = 0; k < HT_RING; k++) plane(z + k, k, std::true_type{}, std::false_type{});
            z += HT_RING;
        } else {
            plane(z, s0, std::false_type{}, std::false_type{});
            z++;
            s0 = s0 + 1 >= HT_RING ? 0 : s0 + 1;
        }
    }
#ifdef PNR_HT_STAMPS
    if ((tid & 63) == 0 && hsn) {
        atomicAdd(&g_ht_stamps[0], hs0); atomicAdd(&g_ht_stamps[1], hs1); atomicAdd(&g_ht_stamps[2], hs2); atomicAdd(&g_ht_stamps[3], hs3);
        atomicAdd(&g_ht_stamps[4], hsn);
    }
#endif
    if (DUMP) return;
    if (zero_here) s_zero = 1;
    __syncthreads();
    if (tid == 0) {
        Q.count[region] = s_cnt;
        // the first scale writes every voxel (frangi.cpp:237-250): a voxel whose response is proven 0 takes part in Jmin / Jmax
        // with that 0 (conditional atomics: min only falls, max only rises, a stale read costs one redundant atomic at most)
        if (first && s_zero) {
            const unsigned int z0o = f2ord(0.f);
            if (z0o < __hip_atomic_load(&minmax[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&minmax[0], z0o);
            if (z0o > __hip_atomic_load(&minmax[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&minmax[1], z0o);
        }
    }
}

// ---- K4b: eigenvalues + vesselness of the queued voxels ------------------------------------------------------------------
// SUB work-groups share a region and take its entries in turns of 256, so every wavefront but a region's last is full.  Only the
// eigenVALUES are needed here (JAMA's d / e recurrences do not depend on the accumulated eigenvectors, so leaving V unused lets
// the compiler drop that half of the arithmetic; the values are bit-identical): the direction bytes are produced where they are
// consumed -- at the seeds (seed_dirs) or, on request, for the whole volume (v_fill) -- from the winning scale kept per voxel.
constexpr int EQ_SUB = 8, EQ_BLOCK = 256;

__global__ __launch_bounds__(EQ_BLOCK, 8) void eigen_queue(HessQueue Q, float *__restrict__ J, unsigned char *__restrict__ Sc, int w, int h, int tiles_x,
                                                           int tiles_y, int zc0, float two_a2, float two_b2, float two_c2, int first, int scale,
                                                           unsigned int *__restrict__ minmax, int zs0, int zs1)
{
    const unsigned int region = blockIdx.x / EQ_SUB, sub = blockIdx.x % EQ_SUB;
    const unsigned int cnt = Q.count[region];
    if (sub * EQ_BLOCK >= cnt) return;
    unsigned int b = region;
    const int bx = (int)(b % (unsigned)tiles_x); b /= (unsigned)tiles_x;
    const int by = (int)(b % (unsigned)tiles_y);
    const int bz = (int)(b / (unsigned)tiles_y);
    const i64 wh = (i64)w * h;
    const float *qh = Q.h + (size_t)region * 6 * HT_REGION;
    const unsigned int *qi = Q.idx + (size_t)region * HT_REGION;
    unsigned int omin = 0xffffffffu, omax = 0u;
    for (unsigned int e = sub * EQ_BLOCK + threadIdx.x; e < cnt; e += EQ_SUB * EQ_BLOCK) {
        const unsigned int code = qi[e];
        const int z = zc0 + bz * HT_Z + (int)(code >> HT_POSBITS), y = by * HT_Y + (int)((code >> 6) & (unsigned)(HT_Y - 1)), x = bx * HT_X + (int)(code & 63u);
        const i64 i = (i64)z * wh + (i64)y * w + x;
        double V[3][3], d[3];
        const float Dxx = qh[0 * HT_REGION + e], Dxy = qh[1 * HT_REGION + e], Dxz = qh[2 * HT_REGION + e];
        const float Dyy = qh[3 * HT_REGION + e], Dyz = qh[4 * HT_REGION + e], Dzz = qh[5 * HT_REGION + e];
        V[0][0] = Dxx; V[0][1] = Dxy; V[0][2] = Dxz;
        V[1][0] = Dxy; V[1][1] = Dyy; V[1][2] = Dyz;
        V[2][0] = Dxz; V[2][1] = Dyz; V[2][2] = Dzz;
        eigen3(V, d);
        const double vox = vesselness(d, two_a2, two_b2, two_c2);
        bool wr = first != 0;
        if (!wr) wr = vox > (double)J[i];
        if (wr) {
            const float jf = (float)vox;
            J[i] = jf;
            Sc[i] = (unsigned char)scale;
            if (z >= zs0 && z < zs1) { // Jmin / Jmax over the planes this context owns (z-slab sharding), on writes only (frangi.cpp:237,257)
                const unsigned int oo = f2ord(jf);
                omin = oo < omin ? oo : omin;
                omax = oo > omax ? oo : omax;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned int m1 = __shfl_xor(omin, o), m2 = __shfl_xor(omax, o);
        omin = m1 < omin ? m1 : omin;
        omax = m2 > omax ? m2 : omax;
    }
    if ((threadIdx.x & 63) == 0) {
        // millions of same-address atomics serialise (~30 ns each): look at the current extremes first (relaxed device-scope
        // load; a stale value can only cause a redundant atomic, never a missed one) and touch them only to improve them
        const unsigned int cur_min = __hip_atomic_load(&minmax[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int cur_max = __hip_atomic_load(&minmax[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (omin < cur_min) atomicMin(&minmax[0], omin);
        if (omax > cur_max) atomicMax(&minmax[1], omax);
    }
}

// ---- K4c: the direction bytes Vx, Vy, Vz (frangi.cpp:240-250) of single voxels: the axis eigenvector of the Hessian at the scale
// that wrote J there (the first scale where no later one did), solved with the full JAMA routine -- the sign of the vector is
// solver-defined.  `list` == nullptr: every voxel of the volume (pnr_get_frangi asking for V).
struct ScaleVols {
    const float *F[PNR_MAX_SIGMAS];
    float s2[PNR_MAX_SIGMAS];
};
__global__ __launch_bounds__(256) void vdir_points(ScaleVols SV, const unsigned char *__restrict__ Sc, const i64 *__restrict__ list, i64 n, int w, int h, int l,
                                                   unsigned char *__restrict__ ox, unsigned char *__restrict__ oy, unsigned char *__restrict__ oz, int packed)
{
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const i64 i = list ? list[t] : t;
    const i64 wh = (i64)w * h;
    const int z = (int)(i / wh), y = (int)((i - (i64)z * wh) / w), x = (int)(i % w);
    const int sc = Sc[i];
    const float *F = SV.F[sc];
    const float s2 = SV.s2[sc];
    const float Dzz = d2(F, i, wh, z, l, wh, z, l, 1) * s2;
    const float Dyy = d2(F, i, w, y, h, w, y, h, 1) * s2;
    const float Dyz = d2(F, i, w, y, h, wh, z, l, 0) * s2;
    const float Dxx = d2(F, i, 1, x, w, 1, x, w, 1) * s2;
    const float Dxy = d2(F, i, 1, x, w, w, y, h, 0) * s2;
    const float Dxz = d2(F, i, 1, x, w, wh, z, l, 0) * s2;
    double V[3][3], d[3];
    V[0][0] = Dxx; V[0][1] = Dxy; V[0][2] = Dxz;
    V[1][0] = Dxy; V[1][1] = Dyy; V[1][2] = Dyz;
    V[2][0] = Dxz; V[2][1] = Dyz; V[2][2] = Dzz;
    eigen3(V, d);
    if (packed) { // 3 bytes per list entry
        ox[3 * t] = quant_dir(V[0][0]); ox[3 * t + 1] = quant_dir(V[1][0]); ox[3 * t + 2] = quant_dir(V[2][0]);
    } else {
        ox[i] = quant_dir(V[0][0]); oy[i] = quant_dir(V[1][0]); oz[i] = quant_dir(V[2][0]);
    }
}

// ---- test tap (include/pnr_hip_test.h, pnr_eigen_batch): eigen3 on caller-supplied matrices, in the two forms the pipeline compiles --
// eigenvalues only (eigen_queue: V is dead, the compiler drops that half of the arithmetic) and the full solver (vdir_points) -- so that
// the reference's known-answer matrices (tests/golden/eigen_kat.npz: zero, diagonal, repeated eigenvalues ...) reach the device code.
template <bool VECTORS>
__global__ __launch_bounds__(256) void eigen_kat(const double *__restrict__ A, i64 n, double *__restrict__ Vo, double *__restrict__ dout)
{
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    double V[3][3], d[3];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int cc = 0; cc < 3; cc++) V[r][cc] = A[t * 9 + r * 3 + cc];
    eigen3(V, d);
    dout[t * 3 + 0] = d[0]; dout[t * 3 + 1] = d[1]; dout[t * 3 + 2] = d[2];
    if (VECTORS) {
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int cc = 0; cc < 3; cc++) Vo[t * 9 + r * 3 + cc] = V[r][cc];
    }
}

// ----------------------------------------------------------------------------------------
// K4': single-slice stacks (P == 1): hessian2d (frangi.cpp:508-574) + the closed-form 2x2 eigen-analysis and the
// Rb / S2 vesselness of frangi2d (:392-506).  One thread per pixel; mixed f32 / f64 exactly as the reference
// (pow(float, 2) and the sqrt over such sums are double, exp / abs / the final sqrt take the float overloads).
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned char quant_dir2(float v, float n)
{
    const double r = (double)(((v / n) + 1) / 2) * 255.0; // round((((v/n)+1)/2)*255.0), half away from zero
    if (!(r == r)) return 0;                              // zero vector: (int)NaN is INT_MIN on the reference's platform -> clamped to 0
    int val = (int)((r > 0.0) ? floor(r + 0.5) : ceil(r - 0.5));
    val = (val < 0) ? 0 : (val > 255) ? 255 : val;
    return (unsigned char)val;
}

__global__ __launch_bounds__(256) void frangi2d_pixel(const float *__restrict__ F, float *__restrict__ J, unsigned char *__restrict__ Vx,
                                                      unsigned char *__restrict__ Vy, unsigned char *__restrict__ Vz, int w, int h, float sig2,
                                                      float beta, float cc, int first, unsigned int *__restrict__ minmax)
{
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    unsigned int omin = 0xffffffffu, omax = 0u;
    if (i < (i64)w * h) {
        const int x = (int)(i % w), y = (int)(i / w);
        const float Dyy = d2(F, i, w, y, h, w, y, h, 1) * sig2;
        const float Dxx = d2(F, i, 1, x, w, 1, x, w, 1) * sig2;
        const float Dxy = d2(F, i, 1, x, w, w, y, h, 0) * sig2;
        const float dd = Dxx - Dyy;
        const float tmp = (float)sqrt((double)dd * (double)dd + 4 * ((double)Dxy * (double)Dxy));
        float v2x = 2 * Dxy;
        float v2y = Dyy - Dxx + tmp;
        const float mag = (float)sqrt((double)v2x * (double)v2x + (double)v2y * (double)v2y);
        if (mag > 0) { v2x /= mag; v2y /= mag; }
        const float v1x = -v2y, v1y = v2x;
        const float mu1 = (float)(0.5 * (double)(Dxx + Dyy + tmp));
        const float mu2 = (float)(0.5 * (double)(Dxx + Dyy - tmp));
        const bool check = fabsf(mu1) < fabsf(mu2);
        float L1 = check ? mu2 : mu1;
        const float L2 = check ? mu1 : mu2;
        const float Vecx = check ? v2x : v1x, Vecy = check ? v2y : v1y;
        L1 = (L1 == 0) ? FLT_MIN : L1;
        const float q = L2 / L1;
        const float Rb = (float)((double)q * (double)q);
        const float S2 = (float)((double)L1 * (double)L1 + (double)L2 * (double)L2);
        float If = expf_libm(-Rb / beta) * (1 - expf_libm(-S2 / cc));
        If = (L1 > 0) ? 0 : If; // blackwhite == false
        if (first || If > J[i]) {
            J[i] = If;
            const float Vecn = sqrtf(Vecx * Vecx + Vecy * Vecy);
            Vx[i] = quant_dir2(Vecx, Vecn);
            Vy[i] = quant_dir2(Vecy, Vecn);
            Vz[i] = 0;
            omin = omax = f2ord(If);
        }
    }
    // Jmin / Jmax are updated only where a scale wrote (frangi.cpp:449-451, :475-477)
    for (int o = 32; o > 0; o >>= 1) {
        omin = min(omin, (unsigned int)__shfl_xor((int)omin, o));
        omax = max(omax, (unsigned int)__shfl_xor((int)omax, o));
    }
    if ((threadIdx.x & 63) == 0) {
        if (omin != 0xffffffffu) atomicMin(&minmax[0], omin);
        if (omax != 0u) atomicMax(&minmax[1], omax);
    }
}

// ----------------------------------------------------------------------------------------
// K5: J -> J8 (Advantra_plugin.cpp:2499-2512)
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void j8_kernel(const float *__restrict__ J, unsigned char *__restrict__ J8, i64 n,
                                                  float jmin, float jmax, int flat)
{
    const i64 stride = (i64)gridDim.x * 256 * 4;
    auto level = [&](float v) {
        const double r = (double)(((v - jmin) / (jmax - jmin)) * 255);
        int val = (int)((r > 0.0) ? floor(r + 0.5) : ceil(r - 0.5));
        return (val < 0) ? 0 : (val > 255) ? 255 : val;
    };
    // most of J is the exact 0 it was cleared to (no response, or one that cannot reach level 1): its level is computed once, by the same
    // operations, and a wave whose voxels are all zeros skips the division and the f64 rounding -- the kernel is then a plain stream
    const int level0 = flat ? 0 : level(0.f);
    for (i64 i0 = ((i64)blockIdx.x * 256 + threadIdx.x) * 4; i0 < n; i0 += stride) {
        unsigned char o[4];
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = (i0 + k < n) ? J[i0 + k] : 0.f;
        const bool zeros = v[0] == 0.f && v[1] == 0.f && v[2] == 0.f && v[3] == 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int val = (i0 + k < n) ? level0 : 0;
            if (!flat && !zeros && i0 + k < n) val = level(v[k]);
            o[k] = (unsigned char)val;
        }
        if (i0 + 3 < n && ((uintptr_t)(J8 + i0) & 3) == 0)
            *(uchar4 *)(J8 + i0) = make_uchar4(o[0], o[1], o[2], o[3]);
        else
            for (int k = 0; k < 4 && i0 + k < n; k++) J8[i0 + k] = o[k];
    }
}

} // namespace

// ========================================================================================
// host side
// ========================================================================================
#ifdef PNR_HT_STAMPS
extern "C" int pnr_debug_ht_stamps(unsigned long long *out8, int reset)
{
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_ht_stamps), 64) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_ht_stamps), z, 64) != hipSuccess) return -1; }
    return 0;
}
#endif
int pnr_ensure_frangi_buffers(pnr_ctx *c)
{
    if (c->frangi_cap >= c->N && c->d_J) return PNR_OK;
    hipFree(c->d_tmpA); hipFree(c->d_tmpB); hipFree(c->d_J);
    hipFree(c->d_Vx); hipFree(c->d_Vy); hipFree(c->d_Vz); hipFree(c->d_J8); hipFree(c->d_scale);
    c->d_tmpA = c->d_tmpB = c->d_J = nullptr;
    c->d_Vx = c->d_Vy = c->d_Vz = c->d_J8 = c->d_scale = nullptr;
    for (int s = 0; s < PNR_MAX_SIGMAS; s++) { hipFree(c->d_F[s]); c->d_F[s] = nullptr; }
    c->frangi_cap = 0;
    c->have_v = c->have_scale = false;
    // tmpA and the three direction volumes (7.5 GB at 1024^3) are only allocated where they are used -- pnr_ensure_tmpA / pnr_ensure_v:
    // the default pipeline never touches them, and a first hipMalloc of tens of GB costs a one-shot process about 25 ms per GB
    // (scripts/probes/malloc_probe.py)
    const size_t n = (size_t)c->N;
    PNR_HIP(hipMalloc(&c->d_tmpB, n * 4));
    PNR_HIP(hipMalloc(&c->d_J, n * 4));
    PNR_HIP(hipMalloc(&c->d_J8, n));
    PNR_HIP(hipMalloc(&c->d_scale, n));
    c->frangi_cap = c->N;
    return PNR_OK;
}

int pnr_ensure_tmpA(pnr_ctx *c)
{
    if (c->d_tmpA) return PNR_OK;
    PNR_REQUIRE(c->frangi_cap >= c->N, PNR_E_STATE, "the Frangi buffers are not allocated");
    PNR_HIP(hipMalloc(&c->d_tmpA, (size_t)c->frangi_cap * 4));
    return PNR_OK;
}

int pnr_ensure_v(pnr_ctx *c)
{
    if (c->d_Vx && c->d_Vy && c->d_Vz) return PNR_OK;
    PNR_REQUIRE(c->frangi_cap >= c->N, PNR_E_STATE, "the Frangi buffers are not allocated");
    if (!c->d_Vx) PNR_HIP(hipMalloc(&c->d_Vx, (size_t)c->frangi_cap));
    if (!c->d_Vy) PNR_HIP(hipMalloc(&c->d_Vy, (size_t)c->frangi_cap));
    if (!c->d_Vz) PNR_HIP(hipMalloc(&c->d_Vz, (size_t)c->frangi_cap));
    return PNR_OK;
}

// grow-only scratch: the per-scale Gaussian taps
static int ensure_taps(pnr_ctx *c)
{
    if (!c->d_taps) {
        PNR_HIP(hipMalloc(&c->d_taps, (size_t)PNR_MAX_SIGMAS * 2 * TAPS_SLOT * 4));
        c->taps_stage.assign((size_t)PNR_MAX_SIGMAS * 2 * TAPS_SLOT, 0.f);
    }
    return PNR_OK;
}

// the smoothed volume F of every scale stays resident (4 B/voxel and scale of the 288 GB): the direction bytes of a voxel are
// computed from the F of the scale that won there, when and where they are needed
static int ensure_scale_volume(pnr_ctx *c, int s)
{
    if (c->d_F[s]) return PNR_OK;
    PNR_HIP(hipMalloc(&c->d_F[s], (size_t)c->frangi_cap * 4));
    return PNR_OK;
}

// z-chunks of the Hessian stage and their survivor queue: a chunk holds at most 2^27 voxels (3.5 GB of queue at worst)
static int hess_chunk_planes(const pnr_ctx *c)
{
    const i64 wh = c->w * c->h;
    i64 cz = ((i64)1 << 27) / wh / HT_Z * HT_Z;
    if (cz < HT_Z) cz = HT_Z;
    return (int)std::min<i64>(cz, (c->l + HT_Z - 1) / HT_Z * HT_Z);
}
static int ensure_queue(pnr_ctx *c, size_t regions)
{
    if (c->q_regions >= regions) return PNR_OK;
    PNR_HIP(hipDeviceSynchronize());
    hipFree(c->d_qh); hipFree(c->d_qidx); hipFree(c->d_qcount);
    c->d_qh = nullptr; c->d_qidx = nullptr; c->d_qcount = nullptr; c->q_regions = 0;
    PNR_HIP(hipMalloc(&c->d_qh, regions * 6 * HT_REGION * 4));
    PNR_HIP(hipMalloc(&c->d_qidx, regions * HT_REGION * 4));
    PNR_HIP(hipMalloc(&c->d_qcount, regions * 4));
    c->q_regions = regions;
    return PNR_OK;
}

// `d_slot`: a slot of c->d_taps; the kernels get d_slot + 1 (tap 0).  Staged in the context (the copy is asynchronous)
static int upload_taps(pnr_ctx *c, const std::vector<float> &g, float *d_slot)
{
    float *stage = c->taps_stage.data() + (d_slot - c->d_taps);
    std::fill(stage, stage + TAPS_SLOT, 0.f);
    std::copy(g.begin(), g.end(), stage + 1);
    PNR_HIP(hipMemcpyAsync(d_slot, stage, (size_t)TAPS_SLOT * 4, hipMemcpyHostToDevice, c->stream));
    return PNR_OK;
}

// F(sigma) -> d_out (device).  uses tmpA/tmpB as ping-pong; d_out may be tmpA.
static int gaussian3d(pnr_ctx *c, const std::vector<float> &gxy, const std::vector<float> &gz, float *d_taps,
                      float *d_out)
{
    const int w = (int)c->w, h = (int)c->h, l = (int)c->l;
    const int Lxy = ((int)gxy.size() - 1) / 2, Lz = ((int)gz.size() - 1) / 2;
    PNR_REQUIRE(Lxy <= MAX_L && Lz <= MAX_L, PNR_E_ARG, "sigma too large: Gaussian radius %d > %d", Lxy, MAX_L);
    int rc = upload_taps(c, gxy, d_taps);
    if (rc) return rc;
    rc = upload_taps(c, gz, d_taps + TAPS_SLOT);
    if (rc) return rc;
    const float *d_txy = d_taps + 1, *d_tz = d_taps + TAPS_SLOT + 1; // tap 0 of each
    // The passes alternate buffers so that the last one lands in d_out: the (fused x-) y pass writes a scratch volume Y != d_out, the
    // z pass reads it into d_out; pass by pass, the x result may sit in d_out itself (it is consumed before d_out is written).  A
    // single-slice stack has no z pass: y writes d_out, x a scratch volume.  The scratch is tmpB (tmpA only when d_out is tmpB).
    const bool two_d = (l == 1); // the 2-D imgaussian (frangi.cpp:576-645)
    if (d_out == c->d_tmpB) { rc = pnr_ensure_tmpA(c); if (rc) return rc; }
    float *const scratch = d_out == c->d_tmpB ? c->d_tmpA : c->d_tmpB;
    float *bufY = two_d ? d_out : scratch;
    float *bufX = two_d ? scratch : d_out;
    float *bufZ = d_out;
    c->tic();
    int nlaunch = two_d ? 1 : 2;
    if (!((c->opt.gauss_march && launch_gauss_xy_m(c->stream, c->d_img, bufY, w, h, l, d_txy, Lxy)) ||
          launch_gauss_xy_t(c->stream, c->d_img, bufY, w, h, l, d_txy, Lxy))) { // x and y in one kernel for the usual radii; else pass by pass
        nlaunch++;
        if (!launch_gauss_x_t(c->stream, c->d_img, bufX, w, (i64)h * l, d_txy, Lxy)) {
            const int tiles_x = (w + GX_BLOCK - 1) / GX_BLOCK;
            const i64 rows = (i64)h * l;
            hipLaunchKernelGGL(gauss_x_u8, dim3((unsigned)(rows * tiles_x)), dim3(GX_BLOCK), 0, c->stream, c->d_img, bufX, w,
                               rows, tiles_x, d_txy, Lxy);
        }
        launch_gauss_axis(c->stream, bufX, bufY, w, h, (i64)w, l, (i64)w * h, d_txy, Lxy);
    }
    if (!two_d) launch_gauss_axis(c->stream, bufY, bufZ, w, l, (i64)w * h, h, (i64)w, d_tz, Lz);
    c->toc("gauss", nlaunch);
    PNR_HIP(hipGetLastError());
    return PNR_OK;
}

// x pass of the separable Gaussian on a u8 stack (also the first pass of the soma path's xy blur, frangi.cpp:806-836)
int pnr_gauss_x_u8_launch(pnr_ctx *c, const uint8_t *src, float *dst, const float *d_taps, int L)
{
    const int w = (int)c->w;
    const i64 rows = c->h * c->l;
    PNR_REQUIRE(L <= MAX_L, PNR_E_ARG, "Gaussian radius %d > %d", L, MAX_L);
    if (launch_gauss_x_t(c->stream, src, dst, w, rows, d_taps, L)) return PNR_OK;
    const int tiles_x = (w + GX_BLOCK - 1) / GX_BLOCK;
    hipLaunchKernelGGL(gauss_x_u8, dim3((unsigned)(rows * tiles_x)), dim3(GX_BLOCK), 0, c->stream, src, dst, w, rows, tiles_x, d_taps, L);
    return PNR_OK;
}

static int check_grid(pnr_ctx *c)
{
    const i64 rows = c->h * c->l;
    const i64 blocks = rows * ((c->w + 63) / 64);
    PNR_REQUIRE(blocks < 2147483647LL, PNR_E_ARG, "volume too large for a single launch grid");
    return PNR_OK;
}

int pnr_gaussian_run(pnr_ctx *c, float sig, float *d_out)
{
    int rc = check_grid(c);
    if (rc) return rc;
    rc = ensure_taps(c);
    if (rc) return rc;
    std::vector<float> gxy, gz;
    pnr::gaussian_taps(sig, gxy);
    pnr::gaussian_taps(sig / c->prm.zdist, gz);
    rc = gaussian3d(c, gxy, gz, c->d_taps, d_out);
    hipStreamSynchronize(c->stream);
    return rc;
}

// launches of the Hessian stencil over the z-chunk [zc0, zc1)
static void tile_grid(const pnr_ctx *c, int zc0, int zc1, int &tiles_x, int &tiles_y, unsigned &blocks)
{
    tiles_x = (int)((c->w + HT_X - 1) / HT_X);
    tiles_y = (int)((c->h + HT_Y - 1) / HT_Y);
    blocks = (unsigned)((i64)tiles_x * tiles_y * ((zc1 - zc0 + HT_Z - 1) / HT_Z));
}

int pnr_hessian_run(pnr_ctx *c, float sig, float *const d_out[6])
{
    int rc = pnr_ensure_tmpA(c);
    if (!rc) rc = pnr_gaussian_run(c, sig, c->d_tmpA);
    if (rc) return rc;
    const int w = (int)c->w, h = (int)c->h, l = (int)c->l;
    HessOut dump{d_out[0], d_out[1], d_out[2], d_out[3], d_out[4], d_out[5]};
    int tiles_x, tiles_y;
    unsigned blocks;
    tile_grid(c, 0, l, tiles_x, tiles_y, blocks);
    hipLaunchKernelGGL(hessian_tile<true>, dim3(blocks), dim3(HT_THREADS), 0, c->stream, (const float *)c->d_tmpA, w, h, l, tiles_x, tiles_y, 0, l,
                       sig * sig, HessQueue{}, (unsigned int *)nullptr, 1, 0, l, dump, 0.f, 0);
    PNR_HIP(hipGetLastError());
    PNR_HIP(hipStreamSynchronize(c->stream));
    return PNR_OK;
}

// Vx / Vy / Vz of the whole volume (pnr_get_frangi asking for them): not needed by the pipeline, which takes the directions at
// the seeds only (pnr_seed_dirs)
int pnr_frangi_materialise_v(pnr_ctx *c)
{
    if (c->have_v) return PNR_OK;
    PNR_REQUIRE(c->have_scale, PNR_E_STATE, "no Frangi response: the direction volumes cannot be produced");
    { const int rcv = pnr_ensure_v(c); if (rcv) return rcv; }
    ScaleVols SV{};
    for (int s = 0; s < c->prm.nsig; s++) { SV.F[s] = c->d_F[s]; SV.s2[s] = c->prm.sig[s] * c->prm.sig[s]; }
    hipLaunchKernelGGL(vdir_points, dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, c->stream, SV, (const unsigned char *)c->d_scale, (const i64 *)nullptr,
                       (i64)c->N, (int)c->w, (int)c->h, (int)c->l, c->d_Vx, c->d_Vy, c->d_Vz, 0);
    PNR_HIP(hipGetLastError());
    PNR_HIP(hipStreamSynchronize(c->stream));
    c->have_v = true;
    return PNR_OK;
}

// direction bytes at `n` voxels (device list) -> d_dirs[3n]: from the direction volumes when they exist (pnr_set_j8_v, 2-D
// stacks, after pnr_get_frangi), otherwise solved on the spot from the winning scale's smoothed volume
int pnr_seed_dirs(pnr_ctx *c, const long long *d_idx, int n, unsigned char *d_dirs)
{
    if (n == 0) return PNR_OK;
    if (!c->have_v) {
        PNR_REQUIRE(c->have_scale, PNR_E_STATE, "no direction field: run pnr_frangi (or pnr_set_j8_v) first");
        ScaleVols SV{};
        for (int s = 0; s < c->prm.nsig; s++) { SV.F[s] = c->d_F[s]; SV.s2[s] = c->prm.sig[s] * c->prm.sig[s]; }
        hipLaunchKernelGGL(vdir_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, SV, (const unsigned char *)c->d_scale, (const i64 *)d_idx,
                           (i64)n, (int)c->w, (int)c->h, (int)c->l, d_dirs, (unsigned char *)nullptr, (unsigned char *)nullptr, 1);
        PNR_HIP(hipGetLastError());
        return PNR_OK;
    }
    return 1; // the caller gathers from the volumes
}

// test tap: Frangi::eigen_decomposition (frangi.cpp:1269-1306) of n symmetric 3 x 3 matrices (host, row-major doubles) through the
// device solver; V == nullptr runs the eigenvalues-only form eigen_queue compiles
int pnr_eigen_run(pnr_ctx *c, const double *A, int64_t n, double *V, double *d)
{
    if (n == 0) return PNR_OK;
    double *dA = nullptr, *dV = nullptr, *dd = nullptr;
    PNR_HIP(hipMalloc(&dA, (size_t)n * 72));
    hipError_t e = hipMalloc(&dd, (size_t)n * 24);
    if (e == hipSuccess && V) e = hipMalloc(&dV, (size_t)n * 72);
    if (e == hipSuccess) e = hipMemcpyAsync(dA, A, (size_t)n * 72, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        if (V) hipLaunchKernelGGL(eigen_kat<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const double *)dA, (i64)n, dV, dd);
        else hipLaunchKernelGGL(eigen_kat<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const double *)dA, (i64)n, dV, dd);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(d, dd, (size_t)n * 24, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && V) e = hipMemcpyAsync(V, dV, (size_t)n * 72, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dA); (void)hipFree(dV); (void)hipFree(dd);
    PNR_HIP(e);
    return PNR_OK;
}

// J -> J8 with the given extremes (Advantra_plugin.cpp:2499-2512)
int pnr_j8_run(pnr_ctx *c, float jmin, float jmax)
{
    PNR_REQUIRE(c->d_J && c->frangi_cap >= c->N, PNR_E_STATE, "no Frangi response to quantise");
    c->Jmin = jmin;
    c->Jmax = jmax;
    const int flat = std::fabs(c->Jmax - c->Jmin) <= FLT_MIN;
    c->tic();
    {
        i64 blocks = (c->N + 1023) / 1024;
        if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL(j8_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, c->d_J, c->d_J8, c->N, c->Jmin,
                           c->Jmax, flat);
    }
    c->toc("j8");
    PNR_HIP(hipGetLastError());
    PNR_HIP(hipStreamSynchronize(c->stream));
    c->have_j8 = true;
    return PNR_OK;
}

int pnr_frangi_run(pnr_ctx *c, float *Jmin, float *Jmax) { return pnr_frangi_run_range(c, 0, c->l, true, Jmin, Jmax); }

// Frangi with Jmin / Jmax taken over the planes [zs0, zs1) only; finish = false leaves J unquantised (z-slab sharding: the
// ranks first agree on the global extremes, then call pnr_j8_run)
int pnr_frangi_run_range(pnr_ctx *c, int64_t zs0, int64_t zs1, bool finish, float *Jmin, float *Jmax)
{
    int rc = check_grid(c);
    if (rc) return rc;
    rc = pnr_ensure_frangi_buffers(c);
    if (rc) return rc;
    rc = ensure_taps(c);
    if (rc) return rc;
    const int w = (int)c->w, h = (int)c->h, l = (int)c->l;
    const pnr_params &P = c->prm;
    const unsigned int init[2] = {0xffffffffu, 0u};
    PNR_HIP(hipMemcpyAsync(c->d_minmax, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    c->have_v = c->have_scale = false;
    // f32 products, as "2*alpha*alpha" etc. in frangi.cpp:214-216
    const float two_a2 = 2 * P.alpha * P.alpha, two_b2 = 2 * P.beta * P.beta, two_c2 = 2 * P.C * P.C;
    const int cz = hess_chunk_planes(c);
    const int prune = (c->opt.frangi_prune && !c->frangi_exact_once) ? 1 : 0;
    c->frangi_exact_once = false;
    c->fr_zs0 = zs0; c->fr_zs1 = zs1;
    c->frangi_pruned = prune && l > 1;
    if (l > 1) {
        int tx, ty;
        unsigned blocks;
        tile_grid(c, 0, std::min(cz, l), tx, ty, blocks);
        rc = ensure_queue(c, blocks);
        if (rc) return rc;
        // the first scale writes every voxel; those whose response is proven 0 without the solver keep this 0 (and scale 0)
        PNR_HIP(hipMemsetAsync(c->d_J, 0, (size_t)c->N * 4, c->stream));
        PNR_HIP(hipMemsetAsync(c->d_scale, 0, (size_t)c->N, c->stream));
    }
    for (int s = 0; s < P.nsig; s++) {
        float *Fs = nullptr;
        if (l == 1) { // (a single slice: the smoothed image of the scale in tmpA, the direction bytes written by the pixel kernel)
            rc = pnr_ensure_tmpA(c);
            if (!rc) rc = pnr_ensure_v(c);
            if (rc) return rc;
            Fs = c->d_tmpA;
        }
        if (l > 1) {
            rc = ensure_scale_volume(c, s);
            if (rc) return rc;
            Fs = c->d_F[s];
        }
        rc = gaussian3d(c, c->tab.gxy[s], c->tab.gz[s], c->d_taps + (size_t)s * 2 * TAPS_SLOT, Fs);
        if (rc) return rc;
        if (l == 1) { // P == 1: frangi2d (Advantra_plugin.cpp:2496-2497) with frangi_betaone = .5, frangi_betatwo = 15 (:69-70)
            c->tic();
            const float beta2d = (float)(2 * std::pow((double).5f, 2)), c2d = (float)(2 * std::pow((double)15.f, 2));
            hipLaunchKernelGGL(frangi2d_pixel, dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, c->stream, (const float *)Fs, c->d_J,
                               c->d_Vx, c->d_Vy, c->d_Vz, w, h, P.sig[s] * P.sig[s], beta2d, c2d, s == 0 ? 1 : 0, c->d_minmax);
            c->toc("hessian_eigen");
            continue;
        }
        // z-chunks of at most cz planes.  With the J8 shortcut the first scale starts with one march (HT_Z planes) from the middle of
        // the stack: the responses found there give the other chunks a lower bound of Jmax to prune with from their first voxel
        std::vector<std::pair<int, int>> chunks;
        if (prune && s == 0 && l > 2 * HT_Z) {
            const int m0 = (l / 2) / HT_Z * HT_Z, m1 = m0 + HT_Z;
            chunks.push_back({m0, m1});
            for (int z = 0; z < m0; z += cz) chunks.push_back({z, std::min(m0, z + cz)});
            for (int z = m1; z < l; z += cz) chunks.push_back({z, std::min(l, z + cz)});
        } else {
            for (int z = 0; z < l; z += cz) chunks.push_back({z, std::min(l, z + cz)});
        }
        for (const auto &ch : chunks) {
            const int zc0 = ch.first, zc1 = ch.second;
            int tiles_x, tiles_y;
            unsigned blocks;
            tile_grid(c, zc0, zc1, tiles_x, tiles_y, blocks);
            const HessQueue Q{c->d_qh, c->d_qidx, c->d_qcount};
            c->tic();
            hipLaunchKernelGGL(hessian_tile<false>, dim3(blocks), dim3(HT_THREADS), 0, c->stream, (const float *)Fs, w, h, l, tiles_x, tiles_y, zc0, zc1,
                               P.sig[s] * P.sig[s], Q, c->d_minmax, s == 0 ? 1 : 0, (int)zs0, (int)zs1, HessOut{}, two_c2, prune);
            c->toc("hessian_tile");
            c->tic();
            hipLaunchKernelGGL(eigen_queue, dim3(blocks * EQ_SUB), dim3(EQ_BLOCK), 0, c->stream, Q, c->d_J, c->d_scale, w, h, tiles_x, tiles_y, zc0,
                               two_a2, two_b2, two_c2, s == 0 ? 1 : 0, s, c->d_minmax, (int)zs0, (int)zs1);
            c->toc("hessian_eigen");
        }
    }
    unsigned int mm[2];
    PNR_HIP(hipMemcpyAsync(mm, c->d_minmax, sizeof(mm), hipMemcpyDeviceToHost, c->stream));
    PNR_HIP(hipStreamSynchronize(c->stream));
    PNR_HIP(hipGetLastError());
    c->Jmin = ord2f(mm[0]);
    c->Jmax = ord2f(mm[1]);
    c->Jmax_run = c->Jmax;
    if (c->frangi_pruned && c->Jmin != 0.f) c->frangi_pruned = false; // (no exact zero was ever written: nothing was skipped)
    c->have_j8 = false;
    c->have_scale = l > 1;
    c->have_v = l == 1; // the 2-D kernel writes the direction bytes itself
    if (finish) {
        rc = pnr_j8_run(c, c->Jmin, c->Jmax);
        if (rc) return rc;
    }
    if (Jmin) *Jmin = c->Jmin;
    if (Jmax) *Jmax = c->Jmax;
    return PNR_OK;
}
