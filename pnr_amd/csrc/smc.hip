// smc.hip -- ZNCC tube-template likelihood and the batched SMC particle filter on gfx950.
//
// Replaces Tracker::znccBBB / interp (tracker.cpp:1891-1964, :2138-2215), getdirection
// (:751-768) and iter0New / iterINew (:1001-1198).  No MFMA: there is no dense contraction.
//
// Work decomposition (wave64), details in DESIGN.md section 4:
//   * a "chain" = one (particle, sigma) ZNCC; its sums run in sample order inside ONE lane (the
//     reference sums sequentially in f32; any other order changes weights in the last bit and,
//     sooner or later, a resampling index and the rest of the trace).
//   * one work-group = one trace (seed x direction) iterating to its own stop; the image
//     neighbourhood lives in LDS (CS^3 byte cube, ds_read_u8 gathers), particle state in LDS.
//   * sampling is order-free: it is cut into (sigma, 64-particle group, v-slice) work items pulled
//     by the 12 waves, values go to an HBM stash [sample][lane]; the ordered sums stream them back.
//   * template offsets / weights are wave-uniform: one coalesced load per row + v_readlane.
//   * the centroid's own ZNCC rides along as S extra chains of the next iteration.
//   * early DENSITY stop against the node-density map of earlier seed batches (pnr_trace_replay).
#include "ctx.h"
#include "replay.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>

#include "smc_device.h"

namespace {

// ----------------------------------------------------------------------------------------
// K8: batched znccBBB (seed scoring, tests)
// ----------------------------------------------------------------------------------------
// K8', seed scoring in two phases (what pnr_zncc_run uses): one lane per chain leaves the GPU nearly empty -- 19 117 seeds x 3 scales
// are 900 wavefronts, each a serial walk over up to 5625 samples with eight scattered byte loads per sample (17 ms at 1024^3).
// Sampling is order-free, so phase 1 spreads it: a work-group takes 64 poses x 64 CONSECUTIVE template samples of one scale, a
// wavefront evaluates 64 neighbouring samples of one pose per step (the eight corner loads of a wave fall into a few cache lines)
// and the 64 x 64 tile goes through LDS into the stash layout of the tracker's ordered sums, [sample][pose] rows of 256 bytes.
// Phase 2 is zncc_from_stash, one wavefront per (64 poses, scale): the same values in the same order as zncc_chain.
__global__ __launch_bounds__(256) void zncc_sample(Vol V, Tab T, const float *__restrict__ pos_dir, int n, int ngroups, float *__restrict__ stash)
{
    __shared__ float tile[64][65];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // block -> (scale, group of 64 poses, chunk of 64 samples)
    i64 b = blockIdx.x;
    int s = 0;
    i64 soff = 0; // floats of the scales before s
    for (;; s++) {
        const i64 nb = (i64)((T.M[s] + 63) / 64) * ngroups;
        if (b < nb || s == T.nsig - 1) break;
        b -= nb;
        soff += (i64)T.M[s] * 64 * ngroups;
    }
    const int M = T.M[s], nchunk = (M + 63) / 64;
    const int g = (int)(b / nchunk), c = (int)(b - (i64)g * nchunk);
    if (g >= ngroups) return;
    const float4 *tm = T.tmpl + T.moff[s];
    const int k = c * 64 + lane;
    const float4 t = tm[k < M ? k : M - 1];
    for (int q = 0; q < 16; q++) {
        const int i = g * 64 + wv * 16 + q; // wave-uniform
        float v = 0.f;
        if (i < n) {
            const float *p6 = pos_dir + (i64)i * 6;
            const Frame f = make_frame(p6[0], p6[1], p6[2], p6[3], p6[4], p6[5]);
            v = sample(V, f, t);
        }
        tile[wv * 16 + q][lane] = v;
    }
    __syncthreads();
    float *out = stash + soff + (i64)g * M * 64 + (i64)c * 64 * 64;
    for (int e = tid; e < 64 * 64; e += 256) {
        const int j = e >> 6, col = e & 63;
        if (c * 64 + j < M) STASH_ST(&out[j * 64 + col], tile[col][j]);
    }
}

__global__ __launch_bounds__(64) void zncc_sums(Tab T, const float *__restrict__ wd, const float *__restrict__ stash, int ngroups, int n_pad, int i0,
                                                float *__restrict__ corr_s)
{
    const int s = blockIdx.x / ngroups, g = blockIdx.x - s * ngroups, lane = threadIdx.x;
    i64 soff = 0;
    for (int q = 0; q < s; q++) soff += (i64)T.M[q] * 64 * ngroups;
    const int M = T.M[s];
    const float cv = zncc_from_stash<64, 32>(stash + soff + (i64)g * M * 64 + lane, M, wd + T.moff[s], T.corrc[s]);
    corr_s[(i64)s * n_pad + i0 + g * 64 + lane] = cv; // (n_pad is a multiple of 64: the lanes past n write padding)
}

__global__ void zncc_pick(Tab T, const float *__restrict__ corr_s, int n, int n_pad, float *__restrict__ corr,
                          float *__restrict__ sig)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float best = -FLT_MAX, bs = 0.f; // max over sigma, ties keep the first (tracker.cpp:1957-1960)
    for (int s = 0; s < T.nsig; s++) {
        const float cv = corr_s[(i64)s * n_pad + i];
        if (cv > best) { best = cv; bs = T.sig[s]; }
    }
    corr[i] = best;
    sig[i] = bs;
}

__global__ void expf_kernel(const float *__restrict__ x, i64 n, float *__restrict__ y)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = expf_libm(x[i]);
}

// Diagnostic build only (make STAMPS=1 -> libpnr_hip_stamps.so): per-phase shader-clock sums of
// wave 0, written to a buffer of their own.  No stamp executes in the product library.
#ifdef PNR_SMC_STAMPS
#define STAMP(i)                                                       \
    do {                                                               \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();    \
        if (threadIdx.x == 0) st_acc[i] += t_ - st_prev;               \
        st_prev = t_;                                                  \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// ----------------------------------------------------------------------------------------
// K9: SMC trace kernel -- one work-group per trace, image neighbourhood resident in LDS
// ----------------------------------------------------------------------------------------
template <int CS>
__global__ __launch_bounds__(768) void smc_trace(Vol V, Tab T, TabX X, const float *__restrict__ seeds6, int np, int np_pad,
                                                  int ni, float Kc, float znccth, float neff_ratio, const unsigned char *__restrict__ den, int nodepervol,
                                                  TraceOut O)
{
    extern __shared__ float lds[];
    const int tr = blockIdx.x, tid = threadIdx.x, B = blockDim.x, S = T.nsig;
    float *part = lds;                         // [2][np][9]
    float *corr_ks = part + 2 * np * PSTRIDE;  // [S][np_pad]  (slot k == np: centroid of the previous iteration)
    float *prior = corr_ks + S * np_pad;       // [np]
    float *lhood = prior + np;                 // [np]
    float *csw = lhood + np;                   // [np]
    int *idxres = (int *)(csw + np);           // [np]
    float *sxc = (float *)(idxres + np);       // [2][8] centroid of iteration it (cur) and it-1 (pending)
    int *sflag = (int *)(sxc + 16);            // [0]=resampled(prev) [1]=stop code [2]=T [3]=box clipped
    int *sbox = sflag + 4;                     // [0..2]=lo xyz, [3..5]=hi xyz (atomics), [6..8] = cube origin, [12] = stash slot
    float *sneff = (float *)(sbox + 16);       // [2]
    unsigned char *cube = (unsigned char *)(sneff + 2);

    const float *sd = seeds6 + (i64)tr * 6;
    const float x0 = sd[0], y0 = sd[1], z0 = sd[2], vx0 = sd[3], vy0 = sd[4], vz0 = sd[5];
    if (tid == 0) {
        sflag[0] = 0; sflag[1] = 0; sflag[2] = ni; sflag[3] = 0;
        // scratch slot for the pass-1 sample stash: bounded scan of the free flags (one work-group per
        // CU is resident, the pool is larger than the CU count); no slot -> this trace re-samples in pass 2
        int slot = -1;
        if (X.stash) {
            for (int tries = 0; tries < 2 * X.nslots && slot < 0; tries++) {
                const int cand = (int)((blockIdx.x + (unsigned)tries) % (unsigned)X.nslots);
                if (atomicCAS(&X.slot_busy[cand], 0, 1) == 0) slot = cand;
            }
        }
        sbox[12] = slot;
    }
    int pending = -1; // iteration whose centroid ZNCC has not been evaluated yet (uniform)
    __syncthreads();
    const int slot = sbox[12];
#ifdef PNR_SMC_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_prev = __builtin_amdgcn_s_memtime();
#endif

    for (int it = 0; it <= ni; ++it) {
        // it == ni, or a stop already decided by geometry: tail pass that only finishes the pending centroid
        const bool tail = (it == ni) || (sflag[1] != 0);
        float *cur = part + (it & 1) * np * PSTRIDE;
        const float *prv = part + ((it & 1) ^ 1) * np * PSTRIDE;
        float *xc_cur = sxc + (it & 1) * 8;
        const float *xc_pen = sxc + ((it & 1) ^ 1) * 8;
        const int resampled_prev = sflag[0];
        if (tid < 3) { sbox[tid] = 0x7fffffff; sbox[3 + tid] = -0x7fffffff; }
        if (tid == 3) { sflag[3] = 0; sbox[13] = 0; } // [13]: phase-A work-item counter
        __syncthreads();
        STAMP(0);

        // ---- P1: prediction (tracker.cpp:1006-1024 / :1104-1132) + bounding box of all templates ----
        for (int k = tid; k <= np; k += B) {
            float qx, qy, qz, qvx, qvy, qvz;
            if (k == np) { // the pending centroid is evaluated with this iteration's chains
                if (pending < 0) continue;
                qx = xc_pen[0]; qy = xc_pen[1]; qz = xc_pen[2]; qvx = xc_pen[3]; qvy = xc_pen[4]; qvz = xc_pen[5];
            } else {
                if (tail) continue;
                float *q = cur + k * PSTRIDE;
                if (it == 0) {
                    const float stepw = T.w0cws[T.sz - 1] / np;
                    const float u1 = stepw * ((float)T.rng[0] / (float)2147483647);
                    const float ui = u1 + k * stepw;
                    const int s = cdf_search(T.w0cws, T.sz, ui);
                    q[PX] = x0 + T.p[3 * s + 0];
                    q[PY] = y0 + T.p[3 * s + 1];
                    q[PZ] = z0 + T.p[3 * s + 2];
                    q[PVX] = (vx0 != vx0) ? T.u[3 * s + 0] : vx0;
                    q[PVY] = (vy0 != vy0) ? T.u[3 * s + 1] : vy0;
                    q[PVZ] = (vz0 != vz0) ? T.u[3 * s + 2] : vz0;
                    prior[k] = T.w0[s];
                } else {
                    const int k1 = resampled_prev ? idxres[k] : k;
                    const float *par = prv + k1 * PSTRIDE;
                    int vi = -1; // getdirection: first maximum of the 50 dot products
                    float best = -FLT_MAX;
                    for (int a = 0; a < T.ndir; a++) {
                        const float dp = par[PVX] * T.v[3 * a] + par[PVY] * T.v[3 * a + 1] + par[PVZ] * T.v[3 * a + 2];
                        if (dp > best) { best = dp; vi = a; }
                    }
                    if (vi < 0) vi = 0; // NaN direction: the reference would index v[-1]; keep in range
                    const float *cws = T.wcws + (i64)vi * T.sz;
                    const float u1 = cws[T.sz - 1] * ((float)T.rng[k] / (float)2147483647);
                    const int s = cdf_search(cws, T.sz, u1);
                    q[PX] = par[PX] + T.p[3 * s + 0];
                    q[PY] = par[PY] + T.p[3 * s + 1];
                    q[PZ] = par[PZ] + T.p[3 * s + 2];
                    q[PVX] = T.u[3 * s + 0];
                    q[PVY] = T.u[3 * s + 1];
                    q[PVZ] = T.u[3 * s + 2];
                    prior[k] = T.w[(i64)vi * T.sz + s];
                }
                qx = q[PX]; qy = q[PY]; qz = q[PZ]; qvx = q[PVX]; qvy = q[PVY]; qvz = q[PVZ];
            }
            // conservative extent of this pose's largest template along each axis (+ margin)
            const Frame f = make_frame(qx, qy, qz, qvx, qvy, qvz);
            const float ex = X.ext_v * fabsf(qvx) + X.ext_uw * (fabsf(f.ux) + fabsf(f.wx)) + 1.5f;
            const float ey = X.ext_v * fabsf(qvy) + X.ext_uw * (fabsf(f.uy) + fabsf(f.wy)) + 1.5f;
            const float ez = X.ext_v * fabsf(qvz) + X.ext_uw * (fabsf(f.uz) + fabsf(f.wz)) + 1.5f;
            if (qx == qx && qy == qy && qz == qz && ex == ex && ey == ey && ez == ez) {
                const float big = 1e6f;
                atomicMin(&sbox[0], (int)floorf(fmaxf(qx - ex, -big)));
                atomicMin(&sbox[1], (int)floorf(fmaxf(qy - ey, -big)));
                atomicMin(&sbox[2], (int)floorf(fmaxf(qz - ez, -big)));
                atomicMax(&sbox[3], (int)floorf(fminf(qx + ex, big)) + 2);
                atomicMax(&sbox[4], (int)floorf(fminf(qy + ey, big)) + 2);
                atomicMax(&sbox[5], (int)floorf(fminf(qz + ez, big)) + 2);
            } else {
                atomicOr(&sflag[3], 1); // NaN/inf pose: its samples are range-checked and fetched from HBM
            }
        }
        __syncthreads();
        STAMP(1); // P1 prediction
        if (tid == 0) { // cube origin: centred on the bounding box of all templates, kept inside the volume
            const int dim[3] = {V.w, V.h, V.l};
            for (int a = 0; a < 3; a++) {
                const int lo = sbox[a] < 0 ? 0 : (sbox[a] > dim[a] - 1 ? dim[a] - 1 : sbox[a]);
                const int hi = sbox[3 + a] > dim[a] - 1 ? dim[a] - 1 : (sbox[3 + a] < lo ? lo : sbox[3 + a]);
                int o = (lo + hi + 1) / 2 - CS / 2;
                if (o > dim[a] - CS) o = dim[a] - CS;
                if (o < 0) o = 0;
                sbox[6 + a] = o;
            }
        }
        __syncthreads();
        Box Bx;
        Bx.lds = (lds_cu8 *)cube;
        Bx.ox = sbox[6]; Bx.oy = sbox[7]; Bx.oz = sbox[8];
        Bx.org = 0; // (only the FAST variants of the phased sampling kernel address the cube through it)
        { // stage the cube: one wave per (z,y) row, lanes along x (coalesced bytes), 4 rows in flight
            const int lane = tid & 63, wv = tid >> 6, nwv = B >> 6;
            const int xg = Bx.ox + lane < V.w ? Bx.ox + lane : V.w - 1; // beyond the volume: never addressed, load anything valid
            for (int r0 = wv; r0 < CS * CS; r0 += 4 * nwv) {
                unsigned char v[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int r = r0 + j * nwv < CS * CS ? r0 + j * nwv : CS * CS - 1;
                    const int zz = r / CS, yy = r - zz * CS;
                    const int zg = Bx.oz + zz < V.l ? Bx.oz + zz : V.l - 1, yg = Bx.oy + yy < V.h ? Bx.oy + yy : V.h - 1;
                    v[j] = V.img[(i64)zg * V.wh + (i64)yg * V.w + xg];
                }
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (r0 + j * nwv < CS * CS && lane < CS) cube[(r0 + j * nwv) * CS + lane] = v[j];
            }
        }
        __syncthreads();
        STAMP(2); // box staging

        // ---- P2: likelihood chains, sigma-major so a wave shares sigma and sample index ----
        if (slot >= 0) {
            float *const slot_base = X.stash + (i64)slot * X.slot_floats;
            const int ngroups = np_pad >> 6;
            // phase A: balanced sampling.  Items are enumerated largest sigma first: (s, iv, g).
            for (;;) {
                int item = 0;
                if ((tid & 63) == 0) item = atomicAdd(&sbox[13], 1);
                item = __builtin_amdgcn_readfirstlane(item);
                int sI = S - 1, rem = item;
                while (sI >= 0) {
                    const int cnt = __builtin_amdgcn_readfirstlane(X.grid[sI].nv) * ngroups;
                    if (rem < cnt) break;
                    rem -= cnt;
                    sI--;
                }
                if (sI < 0) break; // uniform: all items handed out
                const int iv = rem / ngroups, g = rem - iv * ngroups;
                const int k = g * 64 + (tid & 63);
                const bool is_cen = (k == np) && (pending >= 0);
                const bool valid = (k < np && !tail) || is_cen;
                if (__builtin_amdgcn_ballot_w64(valid) == 0ull) continue;
                const float *q = is_cen ? xc_pen : (valid ? cur + k * PSTRIDE : (tail ? xc_pen : cur));
                const Frame f = make_frame(q[0], q[1], q[2], q[3], q[4], q[5]);
                const Grid gr = X.grid[sI];
                const int nv = __builtin_amdgcn_readfirstlane(gr.nv), nu = __builtin_amdgcn_readfirstlane(gr.nu);
                const int nw = __builtin_amdgcn_readfirstlane(gr.nw);
                const float *ax = X.axes + __builtin_amdgcn_readfirstlane(X.axes_off[sI]);
                sample_slice<CS>(V, Bx, f, nv, nu, nw, ax, iv, slot_base + (i64)(sI * ngroups + g) * X.wave_floats);
            }
            __syncthreads(); // stash written by other waves of this work-group: same CU, same L1
            STAMP(6); // phase A (sampling)
            // phase B: ordered sums, one lane per chain
            for (int c = tid; c < S * np_pad; c += B) {
                const int sI = __builtin_amdgcn_readfirstlane(c / np_pad);
                const int k = c - sI * np_pad;
                const bool is_cen = (k == np) && (pending >= 0);
                const bool valid = (k < np && !tail) || is_cen;
                if (__builtin_amdgcn_ballot_w64(valid) == 0ull) continue;
                const Grid gr = X.grid[sI];
                const int M = __builtin_amdgcn_readfirstlane(gr.nv) * __builtin_amdgcn_readfirstlane(gr.nu) * __builtin_amdgcn_readfirstlane(gr.nw);
                const int goff = __builtin_amdgcn_readfirstlane(gr.off);
                const float cv = zncc_from_stash(slot_base + (i64)(sI * ngroups + (k >> 6)) * X.wave_floats + (k & 63), M,
                                                 X.wd + goff, T.corrc[sI]);
                if (valid) corr_ks[sI * np_pad + k] = cv;
            }
        } else
        for (int c = tid; c < S * np_pad; c += B) {
            const int s = __builtin_amdgcn_readfirstlane(c / np_pad);
            const int k = c - s * np_pad;
            const bool is_cen = (k == np) && (pending >= 0);
            const bool valid = (k < np && !tail) || is_cen;
            if (__builtin_amdgcn_ballot_w64(valid) == 0ull) continue; // wave-uniform: nothing to do in this wave
            // every lane of the wave runs the chain (the template broadcasts need a full wave);
            // idle lanes recompute a pose that is known to lie inside the box and discard the result
            const float *q = is_cen ? xc_pen : (valid ? cur + k * PSTRIDE : (tail ? xc_pen : cur));
            const Frame f = make_frame(q[0], q[1], q[2], q[3], q[4], q[5]);
            const Grid g = X.grid[s];
            const int nv = __builtin_amdgcn_readfirstlane(g.nv), nu = __builtin_amdgcn_readfirstlane(g.nu);
            const int nw = __builtin_amdgcn_readfirstlane(g.nw), goff = __builtin_amdgcn_readfirstlane(g.off);
            const float *ax = X.axes + __builtin_amdgcn_readfirstlane(X.axes_off[s]);
            const float cv = zncc_chain_box<CS, false>(V, Bx, f, nv, nu, nw, ax, X.wd + goff, T.corrc[s], nullptr);
            if (valid) corr_ks[s * np_pad + k] = cv;
        }
        __syncthreads();
        STAMP(3); // chains

        // ---- P3a: finish the pending centroid: corr, stop tests of that iteration (:1072-1079) ----
        if (pending >= 0) {
            if (tid == 0) {
                float best = -FLT_MAX, bs = xc_pen[6];
                for (int s = 0; s < S; s++) {
                    const float cv = corr_ks[s * np_pad + np];
                    if (cv > best) { best = cv; bs = T.sig[s]; }
                }
                float *xo = O.xc + ((i64)tr * ni + pending) * 8;
                xo[0] = xc_pen[0]; xo[1] = xc_pen[1]; xo[2] = xc_pen[2]; xo[3] = xc_pen[3]; xo[4] = xc_pen[4]; xo[5] = xc_pen[5];
                xo[6] = bs; xo[7] = best;
                if ((sflag[1] == 0 || sflag[1] == 3) && best < znccth) { sflag[1] = 2; sflag[2] = pending; }
            }
            __syncthreads();
        }
        if (tail || sflag[1] != 0) break; // uniform (flags written before the barrier above / at loop top)

        // ---- P3b: max over sigma, likelihood exp(Kc*corr) (tracker.cpp:1028-1029) ----
        for (int k = tid; k < np; k += B) {
            float best = -FLT_MAX, bs = 0.f;
            for (int s = 0; s < S; s++) {
                const float cv = corr_ks[s * np_pad + k];
                if (cv > best) { best = cv; bs = T.sig[s]; }
            }
            cur[k * PSTRIDE + PCORR] = best;
            cur[k * PSTRIDE + PSIG] = bs;
            lhood[k] = expf_libm(Kc * best);
        }
        __syncthreads();
        STAMP(4); // P3

        // ---- P4: weights, N_eff, CDF, centroid (:1035-1071), out-of-volume test, systematic resampling
        //      (:1075-1090).  Every SUM is a sequential f32 (or f64-add) chain in particle order, as in the
        //      reference; the element-wise parts run on all lanes and the nine independent chains run side
        //      by side on nine lanes of wave 0. ----
        const bool carry = (it > 0) && !resampled_prev;
        if (tid == 0) {
            float a = 0.f;
            for (int k = 0; k < np; k++) a += prior[k];
            sneff[1] = a; // wnorm_prior
        }
        __syncthreads();
        {
            const float wnorm_prior = sneff[1];
            for (int k = tid; k < np; k += B) {
                const double base = carry ? (double)prv[k * PSTRIDE + PW] : (1.0 / np);
                cur[k * PSTRIDE + PW] = (float)(base * (double)(prior[k] / wnorm_prior) * (double)lhood[k]);
            }
        }
        __syncthreads();
        if (tid == 0) {
            float a = 0.f;
            for (int k = 0; k < np; k++) a += cur[k * PSTRIDE + PW];
            sneff[1] = a; // wnorm_posterior
        }
        __syncthreads();
        {
            const float wsum = sneff[1];
            for (int k = tid; k < np; k += B) cur[k * PSTRIDE + PW] = cur[k * PSTRIDE + PW] / wsum;
        }
        __syncthreads();
        if (tid < 7) { // centroid components x,y,z,vx,vy,vz,sig
            const int comp = (tid < 6) ? tid : PSIG;
            float a = 0.f;
            for (int k = 0; k < np; k++) a += cur[k * PSTRIDE + PW] * cur[k * PSTRIDE + comp];
            xc_cur[tid] = a;
        } else if (tid == 7) { // N_eff: neff += pow(w,2) is an f64 add stored to f32
            float neff = 0.f;
            for (int k = 0; k < np; k++) {
                const float wk = cur[k * PSTRIDE + PW];
                neff = (float)((double)neff + (double)wk * (double)wk);
            }
            sneff[0] = (float)(1.0 / (double)neff);
        } else if (tid == 8) { // cumulative sum of weights
            float acc = 0.f;
            for (int k = 0; k < np; k++) {
                acc = cur[k * PSTRIDE + PW] + ((k > 0) ? acc : 0.f);
                csw[k] = acc;
            }
        }
        __syncthreads();
        if (tid == 0) {
            const float cx = xc_cur[0], cy = xc_cur[1], cz = xc_cur[2], cvx = xc_cur[3], cvy = xc_cur[4], cvz = xc_cur[5];
            const float neff = sneff[0];
            const float vnorm = (float)sqrt((double)cvx * (double)cvx + (double)cvy * (double)cvy + (double)cvz * (double)cvz);
            xc_cur[3] = cvx / vnorm; xc_cur[4] = cvy / vnorm; xc_cur[5] = cvz / vnorm;
            if (it < O.dbg_iters && O.neff) O.neff[(i64)tr * O.dbg_iters + it] = neff;
            const int x1 = (int)roundf(cx), y1 = (int)roundf(cy), z1 = (int)roundf(cz);
            int res = 0;
            if (x1 < 0 || x1 >= V.w || y1 < 0 || y1 >= V.h || z1 < 0 || z1 >= V.l) {
                sflag[1] = 1; // left the volume: the centroid's corr is still evaluated (tail pass) for the record
                sflag[2] = it;
            } else if (den && (int)den[(i64)z1 * V.wh + (i64)y1 * V.w + x1] >= nodepervol) {
                // the centroid sits on a voxel that EARLIER batches already saturated: the replay will end this
                // trace here (DENSITY stop, tracker.cpp:855) at the latest.  The iteration itself still has to
                // pass its corr test for the link to be made, so its centroid ZNCC is evaluated (tail pass).
                sflag[1] = 3;
                sflag[2] = it + 1;
            } else if (neff / np < neff_ratio) {
                res = 1; // resampling does not depend on the centroid's corr; if that later fails znccth the
                         // trace ends at this iteration and these indices are never used
            }
            sflag[0] = res;
        }
        __syncthreads();
        if (sflag[0]) {
            // systematic resampling: the reference walks s upward while ui > csw[s] (unbounded, :1087,:1192;
            // clamped here to np-1).  csw is non-decreasing, so each k finds the same s by bisection.
            const float u1 = (float)((1.0 / np) * (double)((float)T.rng[(it == 0) ? 1 : np] / (float)2147483647));
            for (int k = tid; k < np; k += B) {
                const float ui = (float)((double)u1 + k * (1.0 / np));
                int lo = 0, hi = np - 1;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (ui > csw[mid]) lo = mid + 1; else hi = mid;
                }
                idxres[k] = lo;
            }
        }
        __syncthreads();
        STAMP(5); // P4 serial
        if (it < O.dbg_iters) { // debug taps (tests): particle set and resampling indices of this iteration
            if (O.xfilt) {
                float *dst = O.xfilt + ((i64)tr * O.dbg_iters + it) * np * PSTRIDE;
                for (int e = tid; e < np * PSTRIDE; e += B) dst[e] = cur[e];
            }
            if (O.idxres && sflag[0]) {
                int *dst = O.idxres + ((i64)tr * O.dbg_iters + it) * np;
                for (int k = tid; k < np; k += B) dst[k] = idxres[k];
            }
        }
        pending = it;
    }
    if (tid == 0) {
        O.T[tr] = sflag[2];
        O.stop[tr] = sflag[1];
        if (slot >= 0) atomicExch(&X.slot_busy[slot], 0); // all waves are past their last stash access (barrier at loop exit)
#ifdef PNR_SMC_STAMPS
        if (O.neff) // diagnostic build: phase cycle sums replace the neff tap (8 x u64 per trace needs dbg_iters >= 16)
            for (int i = 0; i < 8; i++) ((unsigned long long *)(O.neff + (i64)tr * O.dbg_iters))[i] = st_acc[i];
#endif
    }
}

} // namespace

int pnr_zncc_run(pnr_ctx *c, const float *h_pos_dir, int64_t n, float *h_corr, float *h_sig)
{
    if (n == 0) return PNR_OK;
    Vol V;
    int rc = make_vol(c, V);
    if (rc) return rc;
    Tab T;
    make_tab(c, T);
    PNR_REQUIRE(n < (1LL << 28), PNR_E_ARG, "too many poses");
    const int n_pad = (int)((n + 63) / 64 * 64);
    float *d_pd = nullptr, *d_cs = nullptr, *d_corr = nullptr, *d_sig = nullptr, *d_stash = nullptr;
    rc = c->scratch_get("zncc_pd", (size_t)n * 6, &d_pd);
    if (!rc) rc = c->scratch_get("zncc_cs", (size_t)T.nsig * n_pad, &d_cs);
    if (!rc) rc = c->scratch_get("zncc_corr", (size_t)n, &d_corr);
    if (!rc) rc = c->scratch_get("zncc_sig", (size_t)n, &d_sig);
    if (rc) return rc;
    PNR_REQUIRE(c->d_wd, PNR_E_STATE, "tracker tables not loaded");
    // poses in batches of at most 32 768: the stash of a batch is sum(M) x 4 B per pose (1.6 GB at the usual three scales)
    i64 Mtot = 0;
    for (int s = 0; s < T.nsig; s++) Mtot += c->tab.M[s];
    const int batch = (int)std::min<i64>(n_pad, 32768);
    rc = c->scratch_get("zncc_stash", (size_t)Mtot * batch, &d_stash);
    if (rc) return rc;
    PNR_HIP(hipMemcpyAsync(d_pd, h_pos_dir, (size_t)n * 24, hipMemcpyHostToDevice, c->stream));
    c->tic();
    int launches = 1;
    for (i64 i0 = 0; i0 < n; i0 += batch) {
        const int nb = (int)std::min<i64>(batch, n - i0), ngroups = (nb + 63) / 64;
        i64 blocks = 0;
        for (int s = 0; s < T.nsig; s++) blocks += (i64)((c->tab.M[s] + 63) / 64) * ngroups;
        hipLaunchKernelGGL(zncc_sample, dim3((unsigned)blocks), dim3(256), 0, c->stream, V, T, (const float *)(d_pd + i0 * 6), nb, ngroups, d_stash);
        hipLaunchKernelGGL(zncc_sums, dim3((unsigned)(T.nsig * ngroups)), dim3(64), 0, c->stream, T, (const float *)c->d_wd, (const float *)d_stash, ngroups,
                           n_pad, (int)i0, d_cs);
        launches += 2;
    }
    hipLaunchKernelGGL(zncc_pick, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, T, d_cs, (int)n, n_pad, d_corr,
                       d_sig);
    c->toc("zncc", launches);
    PNR_HIP(hipGetLastError());
    PNR_HIP(hipMemcpyAsync(h_corr, d_corr, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    if (h_sig) PNR_HIP(hipMemcpyAsync(h_sig, d_sig, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    PNR_HIP(hipStreamSynchronize(c->stream));
    return PNR_OK;
}

// ---- asynchronous trace jobs: launch on the job's stream, collect later -------------------------------
struct pnr_trace_job {
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int64_t n = 0;      // seeds of the launch in flight (0: idle)
    int dbg_iters = 0;
    float *d_s6 = nullptr;
    TraceOut O{};
    size_t cap_tr = 0, cap_dbg = 0; // capacities (traces; traces*dbg_iters) the device buffers were sized for
    std::vector<float> s6;
    // launch-per-phase driver: the host loop runs in pnr_job_finish
    bool phased = false;
    std::vector<pnr_seed> seeds;
    bool want_xfilt = false, want_idxres = false, want_neff = false;
    int use_density = 0;
};

pnr_trace_job *pnr_job_create(pnr_ctx *c, bool own_stream)
{
    pnr_trace_job *j = new pnr_trace_job();
    if (own_stream) {
        if (hipStreamCreateWithFlags(&j->stream, hipStreamNonBlocking) != hipSuccess) { delete j; return nullptr; }
        j->own_stream = true;
    } else {
        j->stream = c->stream;
    }
    return j;
}

static void job_free_buffers(pnr_trace_job *j)
{
    hipFree(j->d_s6); hipFree(j->O.T); hipFree(j->O.stop); hipFree(j->O.xc); hipFree(j->O.xfilt); hipFree(j->O.idxres); hipFree(j->O.neff);
    j->d_s6 = nullptr;
    j->O = TraceOut{};
    j->cap_tr = j->cap_dbg = 0;
}

void pnr_job_destroy(pnr_trace_job *j)
{
    if (!j) return;
    if (j->stream) (void)hipStreamSynchronize(j->stream);
    job_free_buffers(j);
    if (j->own_stream) (void)hipStreamDestroy(j->stream);
    delete j;
}

int pnr_job_launch(pnr_ctx *c, pnr_trace_job *j, const pnr_seed *seeds, int64_t n, int dbg_iters, bool want_xfilt,
                   bool want_idxres, bool want_neff, int use_density)
{
    j->n = 0;
    if (n == 0) return PNR_OK;
    if (!j->own_stream) j->stream = c->stream;
    j->phased = (c->smc_driver == 0);
    PNR_REQUIRE(j->phased || c->l > 1, PNR_E_ARG, "single-slice (2-D) stacks are traced by the phased SMC driver only");
    if (j->phased) {
        j->seeds.assign(seeds, seeds + n);
        j->n = n; j->dbg_iters = dbg_iters; j->use_density = use_density;
        j->want_xfilt = want_xfilt; j->want_idxres = want_idxres; j->want_neff = want_neff;
        return PNR_OK;
    }
    Vol V;
    int rc = make_vol(c, V);
    if (rc) return rc;
    Tab T;
    make_tab(c, T);
    const int np = c->prm.np, ni = c->prm.ni, S = T.nsig;
    const int np_pad = (np + 1 + 63) / 64 * 64; // slot np = the pending centroid
    const i64 ntr = 2 * n;
    PNR_REQUIRE(ntr < (1LL << 30), PNR_E_ARG, "too many traces in one batch");
    if (dbg_iters > ni) dbg_iters = ni;
    if (dbg_iters < 0) dbg_iters = 0;
    const size_t fixed = trace_fixed_lds_bytes(np, np_pad, S);
    const size_t lds_total = 160 * 1024;
    // largest cube that still fits beside the particle state
    static const int cube_sides[] = {52, 48, 44, 40, 36, 32};
    int CS = 0;
    for (int cs : cube_sides)
        if (fixed + (size_t)cs * cs * cs + 64 <= lds_total) { CS = cs; break; }
    PNR_REQUIRE(CS > 0, PNR_E_ARG, "np=%d needs %zu B of LDS state: no room for the image cube", np, fixed);
    const size_t lds = fixed + (size_t)CS * CS * CS;
    int block = S * np_pad;
    if (block > 768) block = 768; // 12 waves = 3 per SIMD: up to 168 VGPRs for the sample groups
    for (int s = 0; s < S; s++)
        PNR_REQUIRE(c->tab.grid[4 * s] <= 64 && c->tab.grid[4 * s + 1] <= 64 && c->tab.grid[4 * s + 2] <= 64, PNR_E_ARG,
                    "template grid axis longer than a wavefront");
    TabX X;
    X.grid = (const Grid *)c->d_grid; X.axes = c->d_axes; X.axes_off = c->d_axes_off; X.wd = c->d_wd;
    X.ext_v = c->tab.ext_v; X.ext_uw = c->tab.ext_uw;
    for (int s2 = 0; s2 < 8; s2++) { X.ext_vs[s2] = 0.f; X.ext_uws[s2] = 0.f; }
    { // pass-1 sample stash: one region per resident work-group (<= 1 per CU: each takes all 160 KB of LDS)
        int Mmax = 0;
        for (int s = 0; s < S; s++) Mmax = std::max(Mmax, c->tab.M[s]);
        hipDeviceProp_t prop;
        PNR_HIP(hipGetDeviceProperties(&prop, c->device));
        const int nslots = prop.multiProcessorCount + 64;
        const long long wave_floats = (long long)Mmax * 64, slot_floats = wave_floats * S * (np_pad / 64);
        const size_t need = (size_t)nslots * slot_floats * 4;
        const bool want_stash = !c->opt.no_stash; // option "no_stash": the in-lane two-pass form
        if ((want_stash && (c->stash_bytes < need || c->stash_slots != nslots)) || (!want_stash && c->d_stash)) {
            PNR_HIP(hipDeviceSynchronize()); // no trace kernel may still hold a slot
            hipFree(c->d_stash); hipFree(c->d_slot_busy);
            c->d_stash = nullptr; c->d_slot_busy = nullptr; c->stash_bytes = 0;
            if (want_stash && hipMalloc(&c->d_stash, need) == hipSuccess && hipMalloc(&c->d_slot_busy, nslots * 4) == hipSuccess) {
                c->stash_bytes = need;
                c->stash_slots = nslots;
                // flags are cleared once: every work-group releases its slot, also with several launches in flight
                PNR_HIP(hipMemset(c->d_slot_busy, 0, nslots * 4));
            } else {
                (void)hipGetLastError(); // not enough HBM for the stash: the kernel re-samples in pass 2
                hipFree(c->d_stash); c->d_stash = nullptr;
            }
        }
        X.stash = c->d_stash; X.slot_busy = c->d_slot_busy; X.nslots = nslots;
        X.slot_floats = slot_floats; X.wave_floats = wave_floats;
    }

    j->s6.resize((size_t)ntr * 6);
    for (i64 i = 0; i < n; i++) {
        float *a = &j->s6[(size_t)(2 * i) * 6], *b = a + 6;
        a[0] = b[0] = seeds[i].x; a[1] = b[1] = seeds[i].y; a[2] = b[2] = seeds[i].z;
        a[3] = seeds[i].vx; a[4] = seeds[i].vy; a[5] = seeds[i].vz;
        b[3] = -seeds[i].vx; b[4] = -seeds[i].vy; b[5] = -seeds[i].vz; // trackNeg (tracker.cpp:819-823)
    }
    const size_t need_dbg = (size_t)ntr * dbg_iters;
    if (j->cap_tr < (size_t)ntr || j->cap_dbg < need_dbg || (want_xfilt && !j->O.xfilt && dbg_iters) ||
        (want_idxres && !j->O.idxres && dbg_iters) || (want_neff && !j->O.neff && dbg_iters)) {
        PNR_HIP(hipStreamSynchronize(j->stream));
        job_free_buffers(j);
        PNR_HIP(hipMalloc(&j->d_s6, (size_t)ntr * 24));
        PNR_HIP(hipMalloc(&j->O.T, (size_t)ntr * 4));
        PNR_HIP(hipMalloc(&j->O.stop, (size_t)ntr * 4));
        PNR_HIP(hipMalloc(&j->O.xc, (size_t)ntr * ni * 32));
        if (dbg_iters > 0 && want_xfilt) PNR_HIP(hipMalloc(&j->O.xfilt, need_dbg * np * PSTRIDE * 4));
        if (dbg_iters > 0 && want_idxres) PNR_HIP(hipMalloc(&j->O.idxres, need_dbg * np * 4));
        if (dbg_iters > 0 && want_neff) PNR_HIP(hipMalloc(&j->O.neff, need_dbg * 4));
        j->cap_tr = (size_t)ntr;
        j->cap_dbg = need_dbg;
    }
    TraceOut O = j->O;
    O.dbg_iters = dbg_iters;
    if (!want_xfilt || !dbg_iters) O.xfilt = nullptr;
    if (!want_idxres || !dbg_iters) O.idxres = nullptr;
    if (!want_neff || !dbg_iters) O.neff = nullptr;
    PNR_HIP(hipMemsetAsync(O.xc, 0, (size_t)ntr * ni * 32, j->stream));
    if (O.idxres) PNR_HIP(hipMemsetAsync(O.idxres, 0xff, need_dbg * np * 4, j->stream));
    PNR_HIP(hipMemcpyAsync(j->d_s6, j->s6.data(), j->s6.size() * 4, hipMemcpyHostToDevice, j->stream));
    c->tic(j->stream);
#define PNR_LAUNCH_TRACE(cs)                                                                                                   \
    case cs:                                                                                                                   \
        PNR_HIP(hipFuncSetAttribute((const void *)smc_trace<cs>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      \
        hipLaunchKernelGGL(smc_trace<cs>, dim3((unsigned)ntr), dim3(block), lds, j->stream, V, T, X, j->d_s6, np, np_pad, ni, \
                           c->prm.Kc, c->prm.znccth, c->prm.neff_ratio, use_density ? c->d_den : nullptr, c->prm.nodepervol, O); \
        break;
    switch (CS) {
        PNR_LAUNCH_TRACE(52)
        PNR_LAUNCH_TRACE(48)
        PNR_LAUNCH_TRACE(44)
        PNR_LAUNCH_TRACE(40)
        PNR_LAUNCH_TRACE(36)
        PNR_LAUNCH_TRACE(32)
    }
#undef PNR_LAUNCH_TRACE
    c->toc("smc", 1, j->stream);
    PNR_HIP(hipGetLastError());
    j->n = n;
    j->dbg_iters = dbg_iters;
    return PNR_OK;
}

int pnr_job_finish(pnr_ctx *c, pnr_trace_job *j, int32_t *T_out, int32_t *stop_out, pnr_xest *xc, float *xfilt, int32_t *idxres,
                   float *neff)
{
    if (j->n == 0) return PNR_OK;
    if (j->phased) {
        const int64_t n = j->n;
        j->n = 0;
        return pnr_trace_run_phased(c, j->seeds.data(), n, T_out, stop_out, xc, j->dbg_iters, j->want_xfilt ? xfilt : nullptr,
                                    j->want_idxres ? idxres : nullptr, j->want_neff ? neff : nullptr, j->use_density);
    }
    const i64 ntr = 2 * j->n;
    const int np = c->prm.np, ni = c->prm.ni, dbg = j->dbg_iters;
    PNR_HIP(hipMemcpyAsync(T_out, j->O.T, (size_t)ntr * 4, hipMemcpyDeviceToHost, j->stream));
    PNR_HIP(hipMemcpyAsync(stop_out, j->O.stop, (size_t)ntr * 4, hipMemcpyDeviceToHost, j->stream));
    PNR_HIP(hipMemcpyAsync(xc, j->O.xc, (size_t)ntr * ni * 32, hipMemcpyDeviceToHost, j->stream));
    if (dbg && xfilt && j->O.xfilt) PNR_HIP(hipMemcpyAsync(xfilt, j->O.xfilt, (size_t)ntr * dbg * np * PSTRIDE * 4, hipMemcpyDeviceToHost, j->stream));
    if (dbg && idxres && j->O.idxres) PNR_HIP(hipMemcpyAsync(idxres, j->O.idxres, (size_t)ntr * dbg * np * 4, hipMemcpyDeviceToHost, j->stream));
    if (dbg && neff && j->O.neff) PNR_HIP(hipMemcpyAsync(neff, j->O.neff, (size_t)ntr * dbg * 4, hipMemcpyDeviceToHost, j->stream));
    PNR_HIP(hipStreamSynchronize(j->stream));
    j->n = 0;
    return PNR_OK;
}

int pnr_trace_run(pnr_ctx *c, const pnr_seed *seeds, int64_t n, int32_t *T_out, int32_t *stop_out, pnr_xest *xc, int dbg_iters,
                  float *xfilt, int32_t *idxres, float *neff, int use_density)
{
    if (n == 0) return PNR_OK;
    pnr_trace_job *&sj = c->job;
    if (!sj) sj = pnr_job_create(c, false);
    PNR_REQUIRE(sj, PNR_E_HIP, "could not create a trace job");
    int rc = pnr_job_launch(c, sj, seeds, n, dbg_iters, xfilt != nullptr, idxres != nullptr, neff != nullptr, use_density);
    if (rc) return rc;
    return pnr_job_finish(c, sj, T_out, stop_out, xc, xfilt, idxres, neff);
}

namespace {
// Monotone: a voxel's density only ever grows (the replay adds nodes; a soma voxel is saturated from the start), and updates of
// different trace groups arrive on different streams in no particular order -- so a value is only written over a smaller one (a
// compare-and-swap on the byte's dword), and a late older update can never lower what a newer one wrote.
__global__ void den_scatter(unsigned char *den, const i64 *idx, const unsigned char *val, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const i64 at = idx[i];
    unsigned *w = (unsigned *)(den + (at & ~(i64)3)); // (the map is allocated in whole dwords)
    const int sh = (int)(at & 3) * 8;
    const unsigned v = val[i];
    unsigned old = __atomic_load_n(w, __ATOMIC_RELAXED);
    while (((old >> sh) & 0xffu) < v) {
        const unsigned got = atomicCAS(w, old, (old & ~(0xffu << sh)) | (v << sh));
        if (got == old) break;
        old = got;
    }
}
} // namespace

int pnr_density_reset(pnr_ctx *c)
{
    if (c->den_cap < c->N) {
        hipFree(c->d_den);
        c->d_den = nullptr;
        c->den_cap = 0;
        PNR_HIP(hipMalloc(&c->d_den, ((size_t)c->N + 3) / 4 * 4)); // whole dwords: den_scatter updates a byte through its dword
        c->den_cap = c->N;
    }
    PNR_HIP(hipMemsetAsync(c->d_den, 0, ((size_t)c->N + 3) / 4 * 4, c->stream));
    PNR_HIP(hipStreamSynchronize(c->stream)); // trace jobs run on their own streams
    if (!c->soma_vox.empty()) {
        // a trace that reaches a soma voxel stops there in the replay (tracker.cpp:858-869): for the kernels' early stop
        // the soma is simply saturated density
        const size_t n = c->soma_vox.size();
        long long *d_idx = nullptr;
        unsigned char *d_val = nullptr;
        int rc = c->scratch_get("soma_den_idx", n, &d_idx);
        if (!rc) rc = c->scratch_get("soma_den_val", n, &d_val);
        if (rc) return rc;
        PNR_HIP(hipMemcpyAsync(d_idx, c->soma_vox.data(), n * 8, hipMemcpyHostToDevice, c->stream));
        PNR_HIP(hipMemsetAsync(d_val, 0xff, n, c->stream));
        hipLaunchKernelGGL(den_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->d_den, (const i64 *)d_idx, (const unsigned char *)d_val, (int)n);
        PNR_HIP(hipStreamSynchronize(c->stream));
    }
    return PNR_OK;
}

int pnr_density_update(pnr_ctx *c, const pnr::Replayer &r, hipStream_t on)
{
    hipStream_t st = on ? on : c->stream;
    const size_t n = r.touched.size();
    if (n == 0) return PNR_OK;
    std::vector<unsigned char> val(n);
    for (size_t i = 0; i < n; i++) val[i] = (unsigned char)r.den_at(r.touched[i]); // final value: duplicates agree
    // grow-only staging buffers (hipFree synchronises the whole device)
    if (c->den_stage_cap < n) {
        PNR_HIP(hipDeviceSynchronize());
        hipFree(c->d_den_idx);
        hipFree(c->d_den_val);
        c->d_den_idx = nullptr;
        c->d_den_val = nullptr;
        c->den_stage_cap = 0;
        const size_t cap = std::max<size_t>(2 * n, 1 << 16);
        PNR_HIP(hipMalloc(&c->d_den_idx, cap * 8));
        PNR_HIP(hipMalloc(&c->d_den_val, cap));
        c->den_stage_cap = cap;
    }
    PNR_HIP(hipMemcpyAsync(c->d_den_idx, r.touched.data(), n * 8, hipMemcpyHostToDevice, st));
    PNR_HIP(hipMemcpyAsync(c->d_den_val, val.data(), n, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(den_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c->d_den, (const i64 *)c->d_den_idx,
                       (const unsigned char *)c->d_den_val, (int)n);
    PNR_HIP(hipGetLastError());
    PNR_HIP(hipStreamSynchronize(st));
    return PNR_OK;
}

// the scatter alone, on a stream of the caller's, from device staging the caller owns: nothing here waits for anything
int pnr_density_scatter_async(pnr_ctx *c, const long long *d_idx, const unsigned char *d_val, size_t n, hipStream_t st)
{
    if (n == 0) return PNR_OK;
    hipLaunchKernelGGL(den_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c->d_den, (const i64 *)d_idx, d_val, (int)n);
    PNR_HIP(hipGetLastError());
    return PNR_OK;
}

int pnr_expf_run(pnr_ctx *c, const float *x, int64_t n, float *y)
{
    if (n == 0) return PNR_OK;
    float *dx = nullptr, *dy = nullptr;
    PNR_HIP(hipMalloc(&dx, (size_t)n * 4));
    PNR_HIP(hipMalloc(&dy, (size_t)n * 4));
    PNR_HIP(hipMemcpyAsync(dx, x, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(expf_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, dx, (i64)n, dy);
    PNR_HIP(hipGetLastError());
    PNR_HIP(hipMemcpyAsync(y, dy, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    PNR_HIP(hipStreamSynchronize(c->stream));
    hipFree(dx); hipFree(dy);
    return PNR_OK;
}
