// interp_group_w7.h -- NOT part of the product and not compiled by any target: the source of round 3's dead end (o), kept for
// the record.  The cube rows as 7 overlapping 8-byte windows, corner pairs by ds_read_b64 + v_perm_b32: bit-identical, LDS
// bank-conflict cycles halved, and the sampling kernel 8.5 % SLOWER (VALU-bound then: EXPERIMENTS.md (o),
// profiles/r03_pmc_phased_w7_512_s600.txt).  It was wired into smc_device.h / smc_phased.hip through a template parameter
// W7PLANE and the switch -DPNR_PH_W7=1 up to commit 81a7bbd (`git show 81a7bbd:pnr_amd/csrc/smc_device.h`); to run it again,
// take the wiring from there.  Caveat found by the round-3 advisor: the "=v" outputs of the hand-written ds_read_b64 must be
// early-clobber ("=&v") and every use tied to the s_waitcnt asm, or the compiler may reuse a destination while the load is in
// flight.
#pragma once
//
// W7 (the phased sampling kernel): every cube row is stored as 7 overlapping 8-byte WINDOWS, window k holding the voxels
// x = 7k .. 7k + 7, so that the pair (x, x + 1) of any x <= 48 lies inside ONE aligned window: a sample's eight corner bytes
// come from four ds_read_b64 instead of eight ds_read_u8.  Why: the gather is bound by LDS bank conflicts (64 lanes = 64
// particles at unrelated addresses; scripts/sim_lds_banks.py reproduces the measured 5.3 LDS cycles per ds_read_u8 from the bank
// rules: 5.5) -- a ds_read_b64 conflicts just as often (5.3 cycles) but carries both x bytes of a row, so the LDS cycles per
// sample halve (44 -> 21).  The two bytes are picked out of the 64-bit window by one v_perm_b32 whose selector depends on
// x mod 7 only.  Same bytes, same interpolation: bit-identical.  The cube then spans CSX = 50 voxels in x (7 windows of 7).
constexpr int W7_CSX = 50, W7_NWIN = 7;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const u32x2 lds_cu64;

template <int G, int CS, bool IS2D, bool FAST, int PITCH, int PLANE>
__device__ __forceinline__ Samples<G> interp_group_w7(const Vol &V, const Box &B, const float (&x)[G], const float (&y)[G],
                                                      const float (&z)[G])
{
    static_assert(PITCH == 8 * W7_NWIN && PLANE % 8 == 0 && PLANE >= CS * PITCH, "rows of 7 windows, 8-byte aligned planes");
    float xf[G], yf[G], zf[G];
    unsigned pr[G][4]; // per (z, y) row of the corner group: byte 0 = voxel x, byte 1 = voxel x + 1
    unsigned loff[G], sel[G];
    bool in[G];
    bool all_in = true;
#pragma unroll
    for (int j = 0; j < G; j++) {
        const float xc = FAST ? x[j] : clamp3(x[j], 0.f, V.xmax), yc = FAST ? y[j] : clamp3(y[j], 0.f, V.ymax);
        const float zc = IS2D ? 0.f : (FAST ? z[j] : clamp3(z[j], 0.f, V.zmax));
        xf[j] = __builtin_amdgcn_fractf(xc);
        yf[j] = __builtin_amdgcn_fractf(yc);
        zf[j] = __builtin_amdgcn_fractf(zc);
        const unsigned rx = (unsigned)((int)xc - B.ox), ry = (unsigned)((int)yc - B.oy), rz = (unsigned)((int)zc - B.oz);
        in[j] = FAST || (rx < (unsigned)(W7_CSX - 1) && max(ry, rz) < (unsigned)(CS - 1));
        all_in = all_in && in[j];
        const unsigned rxs = in[j] ? rx : 0u;
        const unsigned xw = __umul24(rxs, 37u) >> 8; // rx / 7 for rx < 56
        const unsigned r = rxs - 7u * xw;
        sel[j] = __umul24(r, 0x0101u) + 0x0c0c0100u; // v_perm_b32: byte 0 <- window byte r, byte 1 <- window byte r + 1, bytes 2, 3 <- 0
        const unsigned l = __umul24(rz, PLANE) + __umul24(ry, PITCH) + 8u * xw;
        loff[j] = in[j] ? l : 0u;
    }
    // The window loads are written as ds_read_b64 by hand: left to itself hipcc merges the two rows of a plane into one
    // ds_read2_b64 (and, where it cannot prove 8-byte alignment, splits a window into a ds_read2_b32) -- both run at half the
    // rate of ds_read_b64 and bank modulo 32 instead of 64 (MI355X_MICROARCH.md, LDS table), which is the whole gain.  All
    // loads of the group are issued first; each sample then waits for its own four (LDS returns in order, so "at most n
    // operations outstanding" can only over-wait, whatever else the compiler has in flight on the same counter).
    constexpr int NL = IS2D ? 2 : 4; // loads per sample
    unsigned long long wn[G][4];
#pragma unroll
    for (int j = 0; j < G; j++) {
        const unsigned a = (unsigned)(unsigned long long)(B.lds + loff[j]);
        asm volatile("ds_read_b64 %0, %1" : "=v"(wn[j][0]) : "v"(a));
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(wn[j][1]) : "v"(a), "n"(PITCH));
        if (!IS2D) {
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(wn[j][2]) : "v"(a), "n"(PLANE));
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(wn[j][3]) : "v"(a), "n"(PLANE + PITCH));
        }
    }
#pragma unroll
    for (int j = 0; j < G; j++) {
        constexpr int LGKM_MAX = 15; // the counter field has four bits
        const int left = NL * (G - 1 - j);
        if (IS2D) {
            if (left >= LGKM_MAX) asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(wn[j][0]), "+v"(wn[j][1]));
            else if (left == 8) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(wn[j][0]), "+v"(wn[j][1]));
            else if (left == 6) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(wn[j][0]), "+v"(wn[j][1]));
            else if (left == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(wn[j][0]), "+v"(wn[j][1]));
            else if (left == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(wn[j][0]), "+v"(wn[j][1]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wn[j][0]), "+v"(wn[j][1]));
        } else {
            if (left >= LGKM_MAX) asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(wn[j][0]), "+v"(wn[j][1]), "+v"(wn[j][2]), "+v"(wn[j][3]));
            else if (left == 12) asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(wn[j][0]), "+v"(wn[j][1]), "+v"(wn[j][2]), "+v"(wn[j][3]));
            else if (left == 8) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(wn[j][0]), "+v"(wn[j][1]), "+v"(wn[j][2]), "+v"(wn[j][3]));
            else if (left == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(wn[j][0]), "+v"(wn[j][1]), "+v"(wn[j][2]), "+v"(wn[j][3]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wn[j][0]), "+v"(wn[j][1]), "+v"(wn[j][2]), "+v"(wn[j][3]));
        }
#pragma unroll
        for (int q = 0; q < NL; q++) pr[j][q] = __builtin_amdgcn_perm((unsigned)(wn[j][q] >> 32), (unsigned)wn[j][q], sel[j]);
    }
    if (!FAST && __builtin_amdgcn_ballot_w64(!all_in) != 0ull) { // wave-uniform: some lane has a corner group outside the cube
#pragma unroll
        for (int j = 0; j < G; j++) {
            if (!in[j]) { // rare: straight from HBM / L2
                const int x1 = (int)clamp3(x[j], 0.f, V.xmax), y1 = (int)clamp3(y[j], 0.f, V.ymax), z1 = IS2D ? 0 : (int)clamp3(z[j], 0.f, V.zmax);
                const unsigned char *a = V.img + ((i64)z1 * V.wh + (i64)y1 * V.w + x1);
                pr[j][0] = (unsigned)a[0] | ((unsigned)a[1] << 8);
                pr[j][1] = (unsigned)a[V.w] | ((unsigned)a[V.w + 1] << 8);
                if (!IS2D) {
                    pr[j][2] = (unsigned)a[V.wh] | ((unsigned)a[V.wh + 1] << 8);
                    pr[j][3] = (unsigned)a[V.wh + V.w] | ((unsigned)a[V.wh + V.w + 1] << 8);
                }
            }
        }
    }
    Samples<G> r;
    if (IS2D) {
#pragma unroll
        for (int j = 0; j < G; j++) {
            const float fx = xf[j], fy = yf[j];
            const float a00 = (float)(pr[j][0] & 0xffu), a01 = (float)((pr[j][0] >> 8) & 0xffu);
            const float a10 = (float)(pr[j][1] & 0xffu), a11 = (float)((pr[j][1] >> 8) & 0xffu);
            r.v[j] = (1 - fy) * ((1 - fx) * a00 + fx * a01) + (fy) * ((1 - fx) * a10 + fx * a11);
        }
        return r;
    }
#pragma unroll
    for (int j = 0; j < G; j++) { // the same packed blend as interp_group below (z and z + 1 planes in the two halves)
        const f32x2 c00 = {(float)(pr[j][0] & 0xffu), (float)(pr[j][2] & 0xffu)}, c01 = {(float)((pr[j][0] >> 8) & 0xffu), (float)((pr[j][2] >> 8) & 0xffu)};
        const f32x2 c10 = {(float)(pr[j][1] & 0xffu), (float)(pr[j][3] & 0xffu)}, c11 = {(float)((pr[j][1] >> 8) & 0xffu), (float)((pr[j][3] >> 8) & 0xffu)};
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

