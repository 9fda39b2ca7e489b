// smc_device.h -- device-side building blocks shared by the two SMC drivers (smc.hip: one persistent
// work-group per trace; smc_phased.hip: one launch per phase, a trace spread over several work-groups):
// Tracker::interp / znccBBB pieces (tracker.cpp:1891-1964, :2138-2215), the libm-identical expf, the LDS
// cube gathers, the sampling work item and the ordered stash sums.  Internal header, not part of the C ABI.
#pragma once
#include "ctx.h"
#include <cfloat>
#include <cmath>

namespace {
typedef long long i64;
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Vol {
    const unsigned char *img;
    int w, h, l;
    i64 wh;
    float xmax, ymax, zmax; // (float)(dim - 1.001), tracker.cpp:2140,2145,2178
};

struct Tab {
    const float *p, *u, *w0, *w0cws, *v, *w, *wcws;
    const float4 *tmpl; // (v_off, u_off, w_off, wgt - avg)
    const int *M, *moff;
    const float *corrc, *sig;
    const unsigned int *rng;
    int sz, ndir, nsig, Mtot;
};

// glibc 2.35 expf (sysdeps/ieee754/flt-32/e_expf.c, the exp2f_data N = 32 scheme): the
// reference's likelihood is exp(Kc*corr) through std::exp(float) = libm expf; this restates the
// published algorithm so the device value equals the host libm value (checked over 3e8 inputs in
// [-25,25]; tests/test_gpu_smc.py re-checks on the GPU).  Table: round(2^(i/32)) - (i << 47).
__device__ const unsigned long long EXP2F_T[32] = {
    0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL, 0x3fef72b83c7d517bULL,
    0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL, 0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL,
    0x3feedea64c123422ULL, 0x3feece086061892dULL, 0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL,
    0x3feea47eb03a5585ULL, 0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
    0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL, 0x3feee89f995ad3adULL,
    0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL, 0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL,
    0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL};

__device__ __forceinline__ float expf_libm(float x)
{
    if (!(fabsf(x) < 80.0f)) return (float)exp((double)x); // outside the likelihood's range (|Kc*corr| <= ~20)
    const double InvLn2N = 0x1.71547652b82fep+0 * 32, SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, C2 = 0x1.62e42ff0c52d6p-1 / 32;
    const double xd = (double)x;
    double z = InvLn2N * xd;
    double kd = z + SHIFT;
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd -= SHIFT;
    const double r = z - kd;
    unsigned long long t = EXP2F_T[ki % 32];
    t += ki << (52 - 5);
    const double s = __longlong_as_double((long long)t);
    z = C0 * r + C1;
    const double r2 = r * r;
    double y = C2 * r + 1;
    y = z * r2 + y;
    y = y * s;
    return (float)y;
}

__device__ __forceinline__ float clampf(float x, float lo, float hi)
{
    const float c = (x < lo) ? lo : x;
    return (c > hi) ? hi : c;
}


// Tracker::interp, 3-D branch (tracker.cpp:2178-2213)
__device__ __forceinline__ float interp(const Vol &V, float x, float y, float z)
{
    const float xc = clampf(x, 0.f, V.xmax);
    const int x1 = (int)xc;
    const float xf = xc - (float)x1;
    const float yc = clampf(y, 0.f, V.ymax);
    const int y1 = (int)yc;
    const float yf = yc - (float)y1;
    if (V.l == 1) { // single-slice stack: bilinear, z is not used (tracker.cpp:2152-2175)
        const unsigned char *a = V.img + (i64)y1 * V.w + x1;
        const float a00 = a[0], a01 = a[1], a10 = a[V.w], a11 = a[V.w + 1];
        return (1 - yf) * ((1 - xf) * a00 + xf * a01) + (yf) * ((1 - xf) * a10 + xf * a11);
    }
    const float zc = clampf(z, 0.f, V.zmax);
    const int z1 = (int)zc;
    const float zf = zc - (float)z1;
    const unsigned char *a = V.img + (i64)z1 * V.wh + (i64)y1 * V.w + x1;
    const unsigned char *b = a + V.wh;
    const float a00 = a[0], a01 = a[1], a10 = a[V.w], a11 = a[V.w + 1];
    const float b00 = b[0], b01 = b[1], b10 = b[V.w], b11 = b[V.w + 1];
    return (1 - zf) * ((1 - yf) * ((1 - xf) * a00 + xf * a01) + (yf) * ((1 - xf) * a10 + xf * a11)) +
           (zf) * ((1 - yf) * ((1 - xf) * b00 + xf * b01) + (yf) * ((1 - xf) * b10 + xf * b11));
}

struct Frame {
    float px, py, pz, nvx, nvy, nvz, ux, uy, uz, wx, wy, wz;
};

// local frame of znccBBB (tracker.cpp:1893-1917)
__device__ __forceinline__ Frame make_frame(float _x, float _y, float _z, float _vx, float _vy, float _vz)
{
    Frame f;
    const float nrm = (float)sqrt((double)_vx * (double)_vx + (double)_vy * (double)_vy); // pow(f32,2): f64
    if (nrm > 0.0001) {
        const int sg = (_vy < 0) ? -1 : 1;
        f.ux = (float)sg * (_vy / nrm);
        f.uy = (float)(-sg) * (_vx / nrm);
        f.uz = 0;
    } else {
        f.ux = 1; f.uy = 0; f.uz = 0;
    }
    f.wx = f.uy * _vz - f.uz * _vy;
    f.wy = -f.ux * _vz + f.uz * _vx;
    f.wz = f.ux * _vy - f.uy * _vx;
    f.px = _x; f.py = _y; f.pz = _z;
    f.nvx = -_vx; f.nvy = -_vy; f.nvz = -_vz;
    return f;
}

__device__ __forceinline__ float sample(const Vol &V, const Frame &f, const float4 t)
{
    const float x = f.px + t.x * f.nvx + t.y * f.ux + t.z * f.wx; // tracker.cpp:1931-1933
    const float y = f.py + t.x * f.nvy + t.y * f.uy + t.z * f.wy;
    const float z = f.pz + t.x * f.nvz + t.y * f.uz + t.z * f.wz;
    return interp(V, x, y, z);
}

// one (pose, sigma) chain: two sequential passes, sums in sample order
__device__ __forceinline__ float zncc_chain(const Vol &V, const Frame &f, const float4 *__restrict__ tm, int M, float corrc)
{
    // the samples are read straight from HBM / L2 (seed scoring: one evaluation per seed, no cube to share): ZB samples -- 8 ZB
    // corner bytes per lane -- are in flight before the first is used; the sums stay in sample order
    constexpr int ZB = 8;
    float ag = 0.f;
    int k = 0;
    for (; k + ZB <= M; k += ZB) {
        float v[ZB];
#pragma unroll
        for (int j = 0; j < ZB; j++) v[j] = sample(V, f, tm[k + j]);
#pragma unroll
        for (int j = 0; j < ZB; j++) ag += v[j];
    }
    for (; k < M; ++k) ag += sample(V, f, tm[k]);
    ag /= (float)M;
    float corra = 0.f, corrb = 0.f;
    for (k = 0; k + ZB <= M; k += ZB) {
        float v[ZB], w[ZB];
#pragma unroll
        for (int j = 0; j < ZB; j++) {
            const float4 t = tm[k + j];
            v[j] = sample(V, f, t);
            w[j] = t.w;
        }
#pragma unroll
        for (int j = 0; j < ZB; j++) {
            const float di = v[j] - ag;
            corra += di * w[j];
            corrb = (float)((double)corrb + (double)di * (double)di); // corrb += pow(f32,2)
        }
    }
    for (; k < M; ++k) {
        const float4 t = tm[k];
        const float di = sample(V, f, t) - ag;
        corra += di * t.w;
        corrb = (float)((double)corrb + (double)di * (double)di);
    }
    const float prod = corrb * corrc;
    return (prod > FLT_MIN) ? corra / sqrtf(prod) : 0.f; // tracker.cpp:1955
}


enum { PX, PY, PZ, PVX, PVY, PVZ, PW, PCORR, PSIG, PSTRIDE }; // struct X (tracker.h:13-17)

struct TraceOut {
    int *T, *stop;
    float *xc; // ntr x ni x 8 : x,y,z,vx,vy,vz,sig,corr (struct X_est)
    int dbg_iters;
    float *xfilt;
    int *idxres;
    float *neff;
};

// template sample grid of one sigma: nested loops vv (outer) / uu / ww (inner), tracker.cpp:219-221
struct Grid {
    int nv, nu, nw, off; // off: first sample in tmpl / wd
};

// first index s with !(u > cws[s]), clamped to n-1: the monotone walk of tracker.cpp:1013,1120
__device__ __forceinline__ int cdf_search(const float *__restrict__ cws, int n, float u)
{
    int lo = 0, hi = n - 1; // invariant: answer in [lo, hi]
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (u > cws[mid]) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// The same index as cdf_search (the array is a running sum of non-negative terms: monotone, so "first s with !(u > cws[s])" does
// not depend on the order of the probes), found with seven probes in flight per round instead of one: 2 + 1 memory latencies for
// the 256 prediction offsets instead of 8.
__device__ __forceinline__ int cdf_search_wide(const float *__restrict__ cws, int n, float u)
{
    int lo = 0, hi = n - 1; // invariant: answer in [lo, hi]
    while (hi - lo > 8) {
        const int w = hi - lo;
        int pos[7];
        float c[7];
#pragma unroll
        for (int j = 0; j < 7; j++) pos[j] = lo + ((w * (j + 1)) >> 3); // lo < pos[0] < ... < pos[6] < hi (w > 8)
#pragma unroll
        for (int j = 0; j < 7; j++) c[j] = cws[pos[j]];
        int nlo = lo, nhi = hi;
        bool found = false;
#pragma unroll
        for (int j = 0; j < 7; j++)
            if (!found) {
                if (!(u > c[j])) { nhi = pos[j]; found = true; }
                else nlo = pos[j] + 1;
            }
        lo = nlo;
        hi = nhi;
    }
    float c[8];
#pragma unroll
    for (int j = 0; j < 8; j++) c[j] = cws[lo + j < hi ? lo + j : hi];
    int ans = hi;
#pragma unroll
    for (int j = 7; j >= 0; j--)
        if (lo + j < hi && !(u > c[j])) ans = lo + j;
    return ans;
}

// The u8 neighbourhood of the current particle cloud, staged in LDS once per SMC iteration: a
// CS^3 byte cube (CS compile-time: corner offsets become immediates) centred on the bounding box of
// every particle's template.  The union of 200 differently oriented 13x37x37 templates does not fit
// 160 KB in general (~65^3): samples whose corners fall outside the cube are fetched from HBM/L2.
// The sample stash is written once and read back (twice) by another kernel after hundreds of megabytes of other traffic: its stores
// carry the non-temporal hint (global_store_dword ... nt), so that they do not push the cube rows, the template tables and the other
// trace group's lines out of L2.  Measured on the bench step: tracing 1033 -> 978 ms; the same hint on the loads changes nothing
// (EXPERIMENTS.md, round 3).
#ifndef STASH_ST // (an experiment build may have chosen its own: scripts/probes/experiments/ph_sample_hooks.h)
#define STASH_ST(ptr, v) __builtin_nontemporal_store((v), (ptr))
#endif
#define STASH_LD(ptr) (*(ptr))
typedef __attribute__((address_space(3))) const unsigned char lds_cu8; // LDS-qualified: ds_read_u8, never flat_load
struct Box {
    lds_cu8 *lds;
    int ox, oy, oz;
    unsigned org; // (oz * CS + oy) * PITCH + ox: the origin's offset in cube addressing (FAST path: one subtraction instead of three)
};

__device__ __forceinline__ float bcast(float v, int lane) // wave-uniform lane index -> SGPR broadcast
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// same value as clampf for every non-NaN x (lo <= hi); the sign of a zero result may differ, which
// cannot change an interpolated value (it only multiplies/adds into non-negative image samples)
__device__ __forceinline__ float clamp3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }

// Corner bytes are fetched with ds_read_u8 only.  Measured on gfx950 (scripts/probes/lds_gather.hip): an
// unaligned ds_read_u16 costs ~333 cycles per wave-instruction (misaligned lanes are replayed), a
// ds_read_u8 ~3 -- and hipcc merges the adjacent loads p[0], p[1] into one ds_read_u16.  The x+1
// bytes are therefore read through a second base pointer whose relation to the first is hidden
// from the optimiser (one v_add per sample, no volatile: the 8*G loads of a group stay in flight).
__device__ __forceinline__ lds_cu8 *lds_plus1_opaque(lds_cu8 *a)
{
    unsigned v = (unsigned)(unsigned long long)(a + 1);
    asm volatile("" : "+v"(v));
    return (lds_cu8 *)(unsigned long long)v;
}

// Trilinear samples of G consecutive template points (Tracker::interp, tracker.cpp:2178-2213),
// corners fetched as four (x1, x1+1) byte pairs from the LDS box.  A corner group outside the box
// (the bounding box of the whole particle cloud does not always fit the 160 KB LDS) is fetched from
// HBM/L2 instead; all G*4 loads of a group are issued before the first use, so the memory latency
// is paid once per group and the G interpolations overlap.
template <int G>
struct Samples {
    float v[G];
};

// FAST: the caller guarantees that every sample of the group lies inside the volume (the clamp is the identity) and that its
// corner pairs lie inside the cube (ph_predict proved it for all templates of this sigma): no clamp, no range test, no fallback.
// PITCH: bytes between two rows of the cube in LDS (CS, or CS rounded up to a multiple of 4 where the cube is filled by LDS-DMA)
// INVOL: the caller guarantees that every sample lies inside the volume (no clamp) but not that its corners lie inside the cube.
template <int G, int CS, bool IS2D = false, bool FAST = false, int PITCH = CS, bool INVOL = false>
__device__ __forceinline__ Samples<G> interp_group(const Vol &V, const Box &B, const float (&x)[G], const float (&y)[G],
                                                   const float (&z)[G])
{
    constexpr bool NOCLAMP = FAST || INVOL;
    float xf[G], yf[G], zf[G];
    unsigned c[G][8]; // corner bytes: a00 a01 a10 a11 b00 b01 b10 b11
    unsigned loff[G];
    bool in[G];
    bool all_in = true;
#pragma unroll
    for (int j = 0; j < G; j++) {
        const float xc = NOCLAMP ? x[j] : clamp3(x[j], 0.f, V.xmax), yc = NOCLAMP ? y[j] : clamp3(y[j], 0.f, V.ymax);
        const float zc = IS2D ? 0.f : (NOCLAMP ? z[j] : clamp3(z[j], 0.f, V.zmax));
        // xc - (float)(int)xc of the reference == xc - floor(xc) for xc >= 0, an exact subtraction: v_fract_f32
        xf[j] = __builtin_amdgcn_fractf(xc);
        yf[j] = __builtin_amdgcn_fractf(yc);
        zf[j] = __builtin_amdgcn_fractf(zc);
        if (FAST) { // inside the cube for sure: the voxel's cube offset straight from its coordinates (two multiply-adds, one subtraction)
            in[j] = true;
            loff[j] = __umul24((unsigned)(int)zc, CS * PITCH) + (__umul24((unsigned)(int)yc, PITCH) + (unsigned)(int)xc) - B.org;
            continue;
        }
        const unsigned rx = (unsigned)((int)xc - B.ox), ry = (unsigned)((int)yc - B.oy), rz = (unsigned)((int)zc - B.oz);
        const unsigned m = max(max(rx, ry), rz);
        in[j] = m < (unsigned)(CS - 1);
        all_in = all_in && in[j];
        const unsigned l = __umul24(rz, CS * PITCH) + __umul24(ry, PITCH) + rx;
        loff[j] = in[j] ? l : 0u;
    }
#pragma unroll
    for (int j = 0; j < G; j++) {
        lds_cu8 *a = B.lds + loff[j];
        lds_cu8 *a1 = lds_plus1_opaque(a);
        c[j][0] = a[0];            c[j][1] = a1[0];
        c[j][2] = a[PITCH];        c[j][3] = a1[PITCH];
        if (!IS2D) {
            c[j][4] = a[CS * PITCH];         c[j][5] = a1[CS * PITCH];
            c[j][6] = a[CS * PITCH + PITCH]; c[j][7] = a1[CS * PITCH + PITCH];
        }
    }
    if (!FAST && __builtin_amdgcn_ballot_w64(!all_in) != 0ull) { // wave-uniform: some lane has a corner group outside the cube
#pragma unroll
        for (int j = 0; j < G; j++) {
            if (!in[j]) { // rare: recompute the voxel index instead of keeping it live for every sample
                const int x1 = (int)clamp3(x[j], 0.f, V.xmax), y1 = (int)clamp3(y[j], 0.f, V.ymax), z1 = IS2D ? 0 : (int)clamp3(z[j], 0.f, V.zmax);
                const unsigned char *a = V.img + ((i64)z1 * V.wh + (i64)y1 * V.w + x1);
                c[j][0] = a[0];        c[j][1] = a[1];
                c[j][2] = a[V.w];      c[j][3] = a[V.w + 1];
                if (!IS2D) {
                    c[j][4] = a[V.wh];     c[j][5] = a[V.wh + 1];
                    c[j][6] = a[V.wh + V.w]; c[j][7] = a[V.wh + V.w + 1];
                }
            }
        }
    }
    Samples<G> r;
    if (IS2D) { // single-slice stack: (1-fy)*((1-fx)*I11 + fx*I12) + fy*((1-fx)*I21 + fx*I22) (tracker.cpp:2175)
#pragma unroll
        for (int j = 0; j < G; j++) {
            const float fx = xf[j], fy = yf[j];
            r.v[j] = (1 - fy) * ((1 - fx) * (float)c[j][0] + fx * (float)c[j][1]) + (fy) * ((1 - fx) * (float)c[j][2] + fx * (float)c[j][3]);
        }
        return r;
    }
#pragma unroll
    for (int j = 0; j < G; j++) {
        // (1-fz)*((1-fy)*((1-fx)*a00 + fx*a01) + fy*((1-fx)*a10 + fx*a11)) + fz*((1-fy)*((1-fx)*b00 + fx*b01) + fy*(...b10, b11)),
        // the z = z1 (a) and z = z1+1 (b) planes side by side in the two halves of packed f32 operations (v_pk_mul_f32 /
        // v_pk_add_f32: two IEEE single operations per instruction, each rounded exactly like the scalar one)
        const f32x2 c00 = {(float)c[j][0], (float)c[j][4]}, c01 = {(float)c[j][1], (float)c[j][5]};
        const f32x2 c10 = {(float)c[j][2], (float)c[j][6]}, c11 = {(float)c[j][3], (float)c[j][7]};
        const float fx = xf[j], fy = yf[j], fz = zf[j];
        const f32x2 om = (f32x2){1.f, 1.f} - (f32x2){fx, fy};
        const float omz = 1 - fz;
        const f32x2 u0 = (f32x2){om.x, om.x} * c00 + (f32x2){fx, fx} * c01;
        const f32x2 u1 = (f32x2){om.x, om.x} * c10 + (f32x2){fx, fx} * c11;
        const f32x2 yv = (f32x2){om.y, om.y} * u0 + (f32x2){fy, fy} * u1;
        const f32x2 zv = (f32x2){omz, fz} * yv;
        r.v[j] = zv.x + zv.y;
    }
    return r;
}

// one (pose, sigma) chain on the LDS box.  Same operations in the same order as znccBBB: the
// position is ((p + vv*(-v)) + uu*u) + ww*w; the two inner partial sums only change in the outer
// loops, so they are hoisted (bit-identical, 6 instead of 18 f32 ops per sample).  `ax`: the
// three axis value lists (vv | uu | ww) of this sigma, `wd` = wgt - avg per sample.
// The template values are wave-uniform: each is fetched ONCE per wave by a coalesced vector load
// (lane i holds element i of the row) and broadcast with v_readlane, so the inner loop has no
// memory access other than the corner-pair reads.  Samples are interpolated G at a time (their
// loads in flight together) and then added in sample order.  MUST be called with all 64 lanes of
// the wave active (callers give idle lanes a dummy pose).
constexpr int CHAIN_G = 5; // nw is 25 or 13: groups of 5 leave no / little tail; 40 corner bytes in flight per lane

// `stash` (may be null): this wave's scratch region in HBM, [sample][lane] f32.  Pass 1 writes
// every interpolated sample there (coalesced 256 B per wave-store); pass 2 reads them back in
// order instead of re-sampling -- same values, same order, so the sums are unchanged, while the
// second pass drops from ~65 VALU + 8 LDS gathers per sample to one coalesced load and the
// (di, di*wd, di^2) updates.  19.4 MB of streamed scratch per SMC iteration per work-group.
template <int CS, bool STASH>
__device__ __forceinline__ float zncc_chain_box(const Vol &V, const Box &B, const Frame &f, int nv, int nu, int nw,
                                                const float *__restrict__ ax, const float *__restrict__ wd, float corrc,
                                                float *__restrict__ stash)
{
    constexpr int G = CHAIN_G;
    const int lane = threadIdx.x & 63;
    const float r_av = ax[lane < nv ? lane : 0];
    const float r_au = ax[nv + (lane < nu ? lane : 0)];
    const float r_aw = ax[nv + nu + (lane < nw ? lane : 0)];
    float ag = 0.f;
    float *sp = stash + lane;
    for (int iv = 0; iv < nv; ++iv) {
        const float vv = bcast(r_av, iv);
        const float x0 = f.px + vv * f.nvx, y0 = f.py + vv * f.nvy, z0 = f.pz + vv * f.nvz;
        for (int iu = 0; iu < nu; ++iu) {
            const float uu = bcast(r_au, iu);
            const float x1 = x0 + uu * f.ux, y1 = y0 + uu * f.uy, z1 = z0 + uu * f.uz;
            for (int iw0 = 0; iw0 < nw; iw0 += G) {
                float xs[G], ys[G], zs[G];
#pragma unroll
                for (int j = 0; j < G; j++) {
                    const int iw = (iw0 + j < nw) ? iw0 + j : nw - 1; // wave-uniform
                    const float ww = bcast(r_aw, iw);
                    const f32x2 xy = (f32x2){x1, y1} + (f32x2){ww, ww} * (f32x2){f.wx, f.wy}; // packed: same two roundings each
                    xs[j] = xy.x;
                    ys[j] = xy.y;
                    zs[j] = z1 + ww * f.wz;
                }
                const Samples<G> sm = interp_group<G, CS>(V, B, xs, ys, zs);
#pragma unroll
                for (int j = 0; j < G; j++)
                    if (iw0 + j < nw) {
                        ag += sm.v[j];
                        if (STASH) STASH_ST(&sp[(iw0 + j) * 64], sm.v[j]);
                    }
            }
            if (STASH) sp += nw * 64;
        }
    }
    ag /= (float)(nv * nu * nw);
    float corra = 0.f, corrb = 0.f;
    if (STASH) {
        // pass 2 from the stash: a flat, software-pipelined stream over the M samples.  Chunk c+1 (32
        // values per lane + the 32 template weights of the wave) is in flight while chunk c is summed.
        constexpr int CH = 32;
        const int M = nv * nu * nw;
        sp = stash + lane;
        float cur[CH], nxt[CH];
        float w_cur, w_nxt;
#pragma unroll
        for (int j = 0; j < CH; j++) cur[j] = sp[(j < M ? j : M - 1) * 64];
        w_cur = wd[(lane < CH && lane < M) ? lane : 0];
        for (int k0 = 0; k0 < M; k0 += CH) {
            const int k1 = k0 + CH;
            if (k1 < M) {
#pragma unroll
                for (int j = 0; j < CH; j++) nxt[j] = sp[(k1 + j < M ? k1 + j : M - 1) * 64];
                w_nxt = wd[(lane < CH && k1 + lane < M) ? k1 + lane : 0];
            }
#pragma unroll
            for (int j = 0; j < CH; j++)
                if (k0 + j < M) { // wave-uniform
                    const float di = cur[j] - ag;
                    corra += di * bcast(w_cur, j);
                    corrb = (float)((double)corrb + (double)di * (double)di); // corrb += pow(f32,2)
                }
#pragma unroll
            for (int j = 0; j < CH; j++) cur[j] = nxt[j];
            w_cur = w_nxt;
        }
    } else {
    const float *wk = wd;
    float r_wd = wk[lane < nw ? lane : 0]; // row 0 of (wgt - avg); next rows are prefetched one row ahead
    for (int iv = 0; iv < nv; ++iv) {
        const float vv = bcast(r_av, iv);
        const float x0 = f.px + vv * f.nvx, y0 = f.py + vv * f.nvy, z0 = f.pz + vv * f.nvz;
        for (int iu = 0; iu < nu; ++iu) {
            const float uu = bcast(r_au, iu);
            const float x1 = x0 + uu * f.ux, y1 = y0 + uu * f.uy, z1 = z0 + uu * f.uz;
            const float r_cur = r_wd;
            wk += nw;
            const bool more = (iv * nu + iu + 1) < nv * nu;
            const float *nxt = more ? wk : wk - nw; // after the last row: re-read it (stays in range)
            r_wd = nxt[lane < nw ? lane : 0];
            for (int iw0 = 0; iw0 < nw; iw0 += G) {
                float xs[G], ys[G], zs[G];
#pragma unroll
                for (int j = 0; j < G; j++) {
                    const int iw = (iw0 + j < nw) ? iw0 + j : nw - 1;
                    const float ww = bcast(r_aw, iw);
                    const f32x2 xy = (f32x2){x1, y1} + (f32x2){ww, ww} * (f32x2){f.wx, f.wy}; // packed: same two roundings each
                    xs[j] = xy.x;
                    ys[j] = xy.y;
                    zs[j] = z1 + ww * f.wz;
                }
                const Samples<G> sm = interp_group<G, CS>(V, B, xs, ys, zs);
#pragma unroll
                for (int j = 0; j < G; j++)
                    if (iw0 + j < nw) {
                        const float di = sm.v[j] - ag;
                        corra += di * bcast(r_cur, iw0 + j);
                        corrb = (float)((double)corrb + (double)di * (double)di); // corrb += pow(f32,2)
                    }
            }
        }
    }
    }
    const float prod = corrb * corrc;
    return (prod > FLT_MIN) ? corra / sqrtf(prod) : 0.f; // tracker.cpp:1955
}

// ---- balanced two-phase form of the chains (used when a stash slot is held) ----------------------
// Phase A: sampling is order-free, so it is cut into work items (sigma, group of 64 particles,
// v-slice) that the 12 waves pull from a shared counter: the long (sigma >= 4) and short chains
// no longer pin the iteration time to the longest chain.  Values go to the stash at [sample][lane].
template <int CS, bool IS2D = false, bool FAST = false, int PITCH = CS, bool INVOL = false>
__device__ __forceinline__ void sample_slice(const Vol &V, const Box &B, const Frame &f, int nv, int nu, int nw,
                                             const float *__restrict__ ax, int iv, float *__restrict__ stash_wave, int iu0 = 0,
                                             int iu1 = 1 << 30)
{
    constexpr int G = CHAIN_G;
    const int lane = threadIdx.x & 63;
    const float r_au = ax[nv + (lane < nu ? lane : 0)];
    const float r_aw = ax[nv + nu + (lane < nw ? lane : 0)];
    const float vv = ax[iv]; // wave-uniform
    const float x0 = f.px + vv * f.nvx, y0 = f.py + vv * f.nvy, z0 = f.pz + vv * f.nvz;
    if (iu1 > nu) iu1 = nu; // rows [iu0, iu1) of the v-slice (default: all)
    // one 64-bit scalar base for the whole row chunk (told to the compiler with readfirstlane), the lane as a 32-bit offset and
    // the sample within a group as the store's immediate: one address instruction per group of five stores, not one per store
    typedef __attribute__((address_space(1))) float gfloat; // (an integer round trip would make the pointer generic: flat stores)
    gfloat *sp;
    {
        const unsigned long long a = (unsigned long long)(stash_wave + ((i64)iv * nu + iu0) * nw * 64);
        sp = (gfloat *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)a));
    }
    const unsigned ulane = (unsigned)lane;
    for (int iu = iu0; iu < iu1; ++iu) {
        const float uu = bcast(r_au, iu);
        const float x1 = x0 + uu * f.ux, y1 = y0 + uu * f.uy, z1 = z0 + uu * f.uz;
        // full groups first, free of per-sample tests (nw = 25: all of them; the five interpolations of a group are then one
        // straight piece of code for the scheduler); a last, partly filled group (nw = 13) afterwards
        int iw0 = 0;
        for (; iw0 + G <= nw; iw0 += G) {
            float xs[G], ys[G], zs[G];
#pragma unroll
            for (int j = 0; j < G; j++) {
                const float ww = bcast(r_aw, iw0 + j);
                const f32x2 xy = (f32x2){x1, y1} + (f32x2){ww, ww} * (f32x2){f.wx, f.wy}; // packed: same two roundings each
                xs[j] = xy.x;
                ys[j] = xy.y;
                zs[j] = z1 + ww * f.wz;
            }
            const Samples<G> sm = interp_group<G, CS, IS2D, FAST, PITCH, INVOL>(V, B, xs, ys, zs);
            gfloat *const pl = (sp + iw0 * 64) + ulane;
#pragma unroll
            for (int j = 0; j < G; j++) STASH_ST(&pl[j * 64], sm.v[j]);
        }
        if (iw0 < nw) {
            float xs[G], ys[G], zs[G];
#pragma unroll
            for (int j = 0; j < G; j++) {
                const int iw = (iw0 + j < nw) ? iw0 + j : nw - 1; // wave-uniform
                const float ww = bcast(r_aw, iw);
                const f32x2 xy = (f32x2){x1, y1} + (f32x2){ww, ww} * (f32x2){f.wx, f.wy};
                xs[j] = xy.x;
                ys[j] = xy.y;
                zs[j] = z1 + ww * f.wz;
            }
            const Samples<G> sm = interp_group<G, CS, IS2D, FAST, PITCH, INVOL>(V, B, xs, ys, zs);
#pragma unroll
            for (int j = 0; j < G; j++)
                if (iw0 + j < nw) STASH_ST(&((sp + iw0 * 64) + ulane)[j * 64], sm.v[j]);
        }
        sp += nw * 64;
    }
}

// The last, partly filled group of chains (np + 1 is rarely a multiple of 64): its `cnt` chains are spread over the
// wave `parts` = 64 / cnt times, copy p of chain j taking the template rows iu = p, p + parts, ... of the v-slice.
// Sampling is order-free, so the values are the same as in sample_slice; they go to a narrow [sample][stride] region
// (stride = cnt rounded up to 16 floats).  The per-lane uu comes from the row register by ds_bpermute.
template <int CS, bool IS2D = false, bool FAST = false, int PITCH = CS, bool INVOL = false>
__device__ __forceinline__ void sample_slice_packed(const Vol &V, const Box &B, const Frame &f, int nv, int nu, int nw,
                                                    const float *__restrict__ ax, int iv, int parts, int p, bool active,
                                                    float *__restrict__ stash_col, int stride, int r0 = 0, int r1 = 1 << 30)
{
    constexpr int G = CHAIN_G;
    const int lane = threadIdx.x & 63;
    const float r_au = ax[nv + (lane < nu ? lane : 0)];
    const float r_aw = ax[nv + nu + (lane < nw ? lane : 0)];
    const float vv = ax[iv]; // wave-uniform
    const float x0 = f.px + vv * f.nvx, y0 = f.py + vv * f.nvy, z0 = f.pz + vv * f.nvz;
    int rows = (nu + parts - 1) / parts;
    if (r1 < rows) rows = r1; // rounds [r0, r1) of the v-slice (default: all)
    for (int r = r0; r < rows; ++r) {
        const int iu = r * parts + p;
        const bool ok = active && iu < nu;
        const int iuc = iu < nu ? iu : nu - 1;
        const float uu = __shfl(r_au, iuc);
        const float x1 = x0 + uu * f.ux, y1 = y0 + uu * f.uy, z1 = z0 + uu * f.uz;
        float *sp = stash_col + ((i64)(iv * nu + iuc) * nw) * stride;
        int iw0 = 0;
        for (; iw0 + G <= nw; iw0 += G) { // full groups, free of per-sample tests (see sample_slice)
            float xs[G], ys[G], zs[G];
#pragma unroll
            for (int j = 0; j < G; j++) {
                const float ww = bcast(r_aw, iw0 + j);
                const f32x2 xy = (f32x2){x1, y1} + (f32x2){ww, ww} * (f32x2){f.wx, f.wy}; // packed: same two roundings each
                xs[j] = xy.x;
                ys[j] = xy.y;
                zs[j] = z1 + ww * f.wz;
            }
            const Samples<G> sm = interp_group<G, CS, IS2D, FAST, PITCH, INVOL>(V, B, xs, ys, zs);
            if (ok) {
#pragma unroll
                for (int j = 0; j < G; j++) STASH_ST(&sp[(iw0 + j) * stride], sm.v[j]);
            }
        }
        if (iw0 < nw) {
            float xs[G], ys[G], zs[G];
#pragma unroll
            for (int j = 0; j < G; j++) {
                const int iw = (iw0 + j < nw) ? iw0 + j : nw - 1; // wave-uniform
                const float ww = bcast(r_aw, iw);
                const f32x2 xy = (f32x2){x1, y1} + (f32x2){ww, ww} * (f32x2){f.wx, f.wy};
                xs[j] = xy.x;
                ys[j] = xy.y;
                zs[j] = z1 + ww * f.wz;
            }
            const Samples<G> sm = interp_group<G, CS, IS2D, FAST, PITCH, INVOL>(V, B, xs, ys, zs);
#pragma unroll
            for (int j = 0; j < G; j++)
                if (ok && iw0 + j < nw) STASH_ST(&sp[(iw0 + j) * stride], sm.v[j]);
        }
    }
}

// Phase B: the ordered sums of znccBBB (tracker.cpp:1940-1955) for one chain, streamed from the
// stash: mean in sample order, then corra / corrb in sample order.  Software-pipelined: 32 values
// per lane in flight.  All 64 lanes of the wave must call it (template weights are broadcast).
template <int STRIDE = 64, int CH = 32>
__device__ __forceinline__ float zncc_from_stash(const float *__restrict__ stash_lane, int M, const float *__restrict__ wd,
                                                 float corrc
#ifdef PNR_SMC_STAMPS
                                                 , unsigned long long *stamp_pass1 = nullptr // diagnostic build: shader clock at the end of pass 1
#endif
)
{
    // M = nfull full chunks of CH values + a tail.  The full chunks run without any per-value test (each value is a handful of
    // dependent VALU operations: a branch per value doubled the serial time of a chain); the next chunk is in flight meanwhile.
    const int nfull = M / CH, tail = M - nfull * CH; // wave-uniform
    float cur[CH], nxt[CH];
    float ag = 0.f;
    if (nfull > 0) {
#pragma unroll
        for (int j = 0; j < CH; j++) cur[j] = STASH_LD(&stash_lane[j * STRIDE]);
    }
    for (int c = 0; c < nfull; c++) {
        const float *nx = stash_lane + (i64)(c + 1) * CH * STRIDE;
        if (c + 1 < nfull) {
#pragma unroll
            for (int j = 0; j < CH; j++) nxt[j] = STASH_LD(&nx[j * STRIDE]);
        }
#pragma unroll
        for (int j = 0; j < CH; j++) ag += cur[j];
#pragma unroll
        for (int j = 0; j < CH; j++) cur[j] = nxt[j];
    }
    // the tail (M is not a multiple of CH): all its loads in flight together, then the ordered adds -- one memory latency, not `tail`
    {
        const float *tp = stash_lane + (i64)nfull * CH * STRIDE;
#pragma unroll
        for (int j = 0; j < CH; j++) cur[j] = (j < tail) ? STASH_LD(&tp[j * STRIDE]) : 0.f;
#pragma unroll
        for (int j = 0; j < CH; j++)
            if (j < tail) ag += cur[j];
    }
    ag /= (float)M;
#ifdef PNR_SMC_STAMPS
    if (stamp_pass1) *stamp_pass1 = __builtin_amdgcn_s_memtime();
#endif
    float corra = 0.f, corrb = 0.f;
    if (nfull > 0) {
#pragma unroll
        for (int j = 0; j < CH; j++) cur[j] = STASH_LD(&stash_lane[j * STRIDE]);
    }
    for (int c = 0; c < nfull; c++) {
        const float *nx = stash_lane + (i64)(c + 1) * CH * STRIDE;
        if (c + 1 < nfull) {
#pragma unroll
            for (int j = 0; j < CH; j++) nxt[j] = STASH_LD(&nx[j * STRIDE]);
        }
        const float *wk = wd + c * CH; // wave-uniform address: scalar loads
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const float di = cur[j] - ag;
            corra += di * wk[j];
            // corrb += pow(f32,2): the f64 product of two f32 values is exact (48 bits), so one fused multiply-add rounds exactly
            // like the multiply followed by the add -- one f64-rate instruction less per value
            corrb = (float)__builtin_fma((double)di, (double)di, (double)corrb);
        }
#pragma unroll
        for (int j = 0; j < CH; j++) cur[j] = nxt[j];
    }
    {
        const float *tp = stash_lane + (i64)nfull * CH * STRIDE;
        const float *wk = wd + nfull * CH;
#pragma unroll
        for (int j = 0; j < CH; j++) cur[j] = (j < tail) ? STASH_LD(&tp[j * STRIDE]) : 0.f;
#pragma unroll
        for (int j = 0; j < CH; j++)
            if (j < tail) {
                const float di = cur[j] - ag;
                corra += di * wk[j];
                corrb = (float)__builtin_fma((double)di, (double)di, (double)corrb);
            }
    }
    const float prod = corrb * corrc;
    return (prod > FLT_MIN) ? corra / sqrtf(prod) : 0.f; // tracker.cpp:1955
}

// The same sums with NB chunk buffers that take turns (the loop runs over NB chunks at a time, no buffer is ever copied): chunk
// c + NB - 1 is requested before chunk c is added up, so a lane waits for loads issued NB - 1 chunks ago instead of for its own
// (above, the compiler folds `cur` / `nxt` into one buffer).  More registers (NB x CH values), so fewer waves per SIMD: this is the
// form for launches of a few dozen traces, whose duration is the latency of one chain and not the bandwidth of the stash -- and for
// launches that have the GPU to themselves.  Same values in the same order: bit-identical.
template <int STRIDE, int CH, int NB, class F>
__device__ __forceinline__ void stash_chunks(const float *__restrict__ stash_lane, int nfull, F &&consume)
{
    static_assert((NB - 1) * CH <= 48, "vmcnt counts at most 63 loads in flight");
    float buf[NB][CH];
#pragma unroll
    for (int b = 0; b < NB - 1; b++)
        if (b < nfull) {
#pragma unroll
            for (int j = 0; j < CH; j++) buf[b][j] = STASH_LD(&stash_lane[((i64)b * CH + j) * STRIDE]);
        }
    int c0 = 0;
    // steady state: every chunk requested here exists, so the body is free of branches and the compiler's wait before chunk c is
    // "all but the (NB - 1) x CH newest loads" (behind a branch that may skip loads it has to assume the fewest and waits for more)
    for (; c0 + 2 * NB - 1 <= nfull; c0 += NB) {
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const int c = c0 + b, cn = c + NB - 1;
#pragma unroll
            for (int j = 0; j < CH; j++) buf[(b + NB - 1) % NB][j] = STASH_LD(&stash_lane[((i64)cn * CH + j) * STRIDE]);
            consume(buf[b], c);
        }
    }
    for (; c0 < nfull; c0 += NB) { // the last one or two rounds
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const int c = c0 + b, cn = c + NB - 1; // wave-uniform
            if (cn < nfull) {
#pragma unroll
                for (int j = 0; j < CH; j++) buf[(b + NB - 1) % NB][j] = STASH_LD(&stash_lane[((i64)cn * CH + j) * STRIDE]);
            }
            if (c < nfull) consume(buf[b], c);
        }
    }
}

template <int STRIDE = 64, int CH = 16, int NB = 4>
__device__ __forceinline__ float zncc_from_stash_deep(const float *__restrict__ stash_lane, int M, const float *__restrict__ wd, float corrc
#ifdef PNR_SMC_STAMPS
                                                      , unsigned long long *stamp_pass1 = nullptr
#endif
)
{
    const int nfull = M / CH, tail = M - nfull * CH; // wave-uniform
    float ag = 0.f;
    stash_chunks<STRIDE, CH, NB>(stash_lane, nfull, [&](const float (&v)[CH], int) {
#pragma unroll
        for (int j = 0; j < CH; j++) ag += v[j];
    });
    float t[CH];
    {
        const float *tp = stash_lane + (i64)nfull * CH * STRIDE;
#pragma unroll
        for (int j = 0; j < CH; j++) t[j] = (j < tail) ? STASH_LD(&tp[j * STRIDE]) : 0.f;
#pragma unroll
        for (int j = 0; j < CH; j++)
            if (j < tail) ag += t[j];
    }
    ag /= (float)M;
#ifdef PNR_SMC_STAMPS
    if (stamp_pass1) *stamp_pass1 = __builtin_amdgcn_s_memtime();
#endif
    float corra = 0.f, corrb = 0.f;
    stash_chunks<STRIDE, CH, NB>(stash_lane, nfull, [&](const float (&v)[CH], int c) {
        const float *wk = wd + c * CH; // wave-uniform address: scalar loads
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const float di = v[j] - ag;
            corra += di * wk[j];
            corrb = (float)__builtin_fma((double)di, (double)di, (double)corrb); // (see zncc_from_stash)
        }
    });
    {
        const float *wk = wd + nfull * CH;
#pragma unroll
        for (int j = 0; j < CH; j++)
            if (j < tail) { // (the tail's values are still in t[])
                const float di = t[j] - ag;
                corra += di * wk[j];
                corrb = (float)__builtin_fma((double)di, (double)di, (double)corrb);
            }
    }
    const float prod = corrb * corrc;
    return (prod > FLT_MIN) ? corra / sqrtf(prod) : 0.f; // tracker.cpp:1955
}

struct TabX { // extra template tables for the box kernel
    const Grid *grid;   // per sigma
    const float *axes;  // per sigma: vv[nv] | uu[nu] | ww[nw], at axes_off[s]
    const int *axes_off;
    const float *wd;    // sum(M): wgt - avg
    float ext_v, ext_uw; // largest template half-extents (voxels) along v and along u / w
    float ext_vs[8], ext_uws[8]; // ... per sigma (PNR_MAX_SIGMAS = 8)
    float *stash;        // nslots x waves x Mmax x 64 f32 of HBM scratch (null: re-sample in pass 2)
    int *slot_busy;      // nslots flags, 0 = free
    int nslots;
    long long slot_floats, wave_floats;
};


size_t trace_fixed_lds_bytes(int np, int np_pad, int S)
{
    return (size_t)(2 * np * PSTRIDE + S * np_pad + 3 * np + np + 16 + 4 + 16 + 2) * 4;
}

int make_vol(pnr_ctx *c, Vol &V)
{
    PNR_REQUIRE(c->d_img, PNR_E_STATE, "no volume set (pnr_set_volume)");
    V.img = c->d_img;
    V.w = (int)c->w; V.h = (int)c->h; V.l = (int)c->l;
    V.wh = c->w * c->h;
    V.xmax = (float)(V.w - 1.001);
    V.ymax = (float)(V.h - 1.001);
    V.zmax = (float)(V.l - 1.001);
    return PNR_OK;
}

void make_tab(pnr_ctx *c, Tab &T)
{
    T.p = c->d_p; T.u = c->d_u; T.w0 = c->d_w0; T.w0cws = c->d_w0cws; T.v = c->d_v; T.w = c->d_w; T.wcws = c->d_wcws;
    T.tmpl = (const float4 *)c->d_tmpl;
    T.M = c->d_M; T.moff = c->d_moff; T.corrc = c->d_corrc; T.sig = c->d_sig; T.rng = c->d_rng;
    T.sz = c->tab.sz; T.ndir = c->tab.ndir; T.nsig = c->tab.nsig;
    T.Mtot = c->tab.moff.back() + c->tab.M.back();
}


} // namespace
