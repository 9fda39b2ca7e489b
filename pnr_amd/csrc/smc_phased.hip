// smc_phased.hip -- the SMC particle filter with ONE LAUNCH PER PHASE, so that a trace is not tied to one CU.
//
// smc.hip runs a trace as one persistent work-group: simple, but a batch then lasts as long as its longest trace
// (200 iterations x ~2.2 ms) while the CUs of finished traces idle, and batches must stay small for the early
// DENSITY stops to bite.  Here every SMC iteration of a batch is four launches over all still-active traces:
//
//   ph_predict  1 WG / trace      P1: parents -> particles, priors, cube origin            (tracker.cpp:1104-1132)
//   ph_sample   nsplit WG / trace phase A: cube -> LDS, (sigma, group, v-slice) items -> HBM stash  (:1929-1938)
//   ph_sums     3+ WG / trace     phase B: ordered mean / corra / corrb per chain            (:1940-1955)
//   ph_update   1 WG / trace      pending centroid + stop tests, weights, N_eff, CDF, centroid, resampling
//                                                                                            (:1035-1090, :1140-1195)
// The host picks nsplit from the number of active traces (about two sampling work-groups per CU), so the last
// stragglers of a batch are sampled by dozens of CUs each instead of one.  Particle state lives in HBM between
// launches (14 KB per trace).  Every device function is shared with smc.hip (smc_device.h): same operations, same
// order, bit-identical traces (tests run both drivers against the oracle and against each other).
#include "ctx.h"
#include "replay.h"
#include "stream_sched.h"
#ifdef PNR_EXPERIMENT_HOOKS // variant libraries only (make variant): scripts/probes/experiments/ph_sample_hooks.h
#include "../../scripts/probes/experiments/ph_sample_hooks.h"
#else
#define PNR_HOOK_AFTER_FULL_ITEM(seg_lane, cnt)
#endif
#include "smc_device.h"
#include <algorithm>
#include <cstdlib>

namespace {

enum { FL_RES = 0, FL_STOP = 1, FL_T = 2, FL_IT = 3, FL_DONE = 4, FL_OX = 5, FL_OY = 6, FL_OZ = 7, FL_Y0 = 8, FL_Y1 = 9, FL_Z0 = 10, FL_Z1 = 11, FL_FAST = 12, FL_NCH = 13, FL_PAUSE = 14, FL_N = 16 };
constexpr int PH_THREADS = 1024; // sampling work-group: 16 waves (<= 128 VGPRs each)
constexpr int PH_CS = 54; // the sampling kernel holds nothing but the cube in LDS: 54 x 54 rows of 56 bytes = 163 296 B of the 160 KB (53 costs 1.5 %)
constexpr int PH_PITCH = 56; // row pitch: a multiple of 4, so that every staged dword lands with one aligned ds_write_b32
constexpr int PH_PLANE = PH_CS * PH_PITCH; // bytes between two planes of the cube
static_assert((size_t)PH_CS * PH_PLANE <= 160 * 1024, "the cube must fit the LDS of a CU");

struct PhState {
    float *part;   // [NT][2][np][9]
    float *prior;  // [NT][np]
    int *idxres;   // [NT][np]
    float *corr;   // [NT][S][np_pad]
    float *xcs;    // [NT][2][8]
    int *flags;    // [NT][FL_N]
    // The sample stash lives from ph_sample to ph_sums of ONE step, so it is indexed by a trace's POSITION in the step's list, not by its
    // slot: [stash_base + position][trace_floats] (per sigma, ngf regions [M][64] of the full chain groups, then [M][R] of the last one).
    // What it has to hold is the largest launch of a trace group, not the window of slots (paused traces keep slot and state, no stash).
    float *stash;
    long long trace_floats;
    int stash_base; // first stash row of this trace group (the groups step concurrently: each has a region of its own)
    unsigned char *cubes; // [stash_base + position][PH_CS * PH_PLANE] the traces' cubes in LDS layout, fetched from the image ONCE per step by ph_cube
                          // (nullptr: every sampling work-group stages its cube from the image itself)
    int *list;     // [2][cap]: traces still running in iteration it: list[it & 1][0 .. cnt[it & 1])
    int *cnt;      // [2]
    int *ctr;      // [NT] sampling work-item counter of the current iteration
    int *uidx;     // [NT][np_pad] chain -> particle (exact duplicates among the particles are evaluated once); the last chain is the centroid
    int *cmap;     // [NT][np_pad] particle -> chain
    int cap;
    int np_pad;
    int W;         // floats per sample row of a trace's stash region at most (64 per full group + the last group's stride)
    int dedup;     // 0: every particle is its own chain
};

__device__ __forceinline__ unsigned int pose_hash(float x, float y, float z, float vx, float vy, float vz)
{
    // multiply-and-fold over the six words (poses of one trace differ by small lattice offsets and share directions: an XOR of
    // rotated words put 1.2 % of the particles behind a hash leader with another pose -- each of them a scan over everything in
    // front of it)
    unsigned int h = 0x811c9dc5u;
    const unsigned int w[6] = {__float_as_uint(x), __float_as_uint(y), __float_as_uint(z), __float_as_uint(vx), __float_as_uint(vy), __float_as_uint(vz)};
#pragma unroll
    for (int i = 0; i < 6; i++) {
        h = (h ^ w[i]) * 0x9e3779b1u;
        h ^= h >> 15;
    }
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    return h;
}

// chains of a trace in this iteration: FL_NCH = unique particles + the centroid (1 in the tail pass); they are laid out as
// ngf = nch / 64 full groups of 64 lanes and a last group of rem = nch % 64 chains whose stash rows are R floats wide
__device__ __forceinline__ int last_group_stride(int rem) { return rem > 32 ? 64 : (rem > 16 ? 32 : 16); }


// Stage the rows of the cube the templates can reach: a wave-load fetches four (z,y) rows, 16 lanes x one unaligned dword each, NR of
// them in flight per wave; wave `wv` of `nwv` takes the rows wv * 4 + sub, + nwv * 4, ...; `cube32` = the cube in LDS layout (LDS in
// ph_sample, the compact global copy in ph_cube).  What made the first version slow (45 k cycles per work-group, 18 % of the kernel;
// scripts/ph_stamps.py) was neither the memory latency (8, 12 or 23 loads in flight: the same), nor the unaligned addresses (aligned
// dwords + DPP shift + v_alignbyte: the same), nor the path into LDS (LDS-DMA: global_load_lds_dword lands one 256-byte block per ~96
// cycles and CU, 57 k cycles for a cube) but the address arithmetic: a division by the run-time row count per load and four byte
// writes per dword.  The (plane, row) pair advances incrementally, and the rows are PH_PITCH = 56 bytes apart so that every dword
// lands with one aligned 32-bit store (the last dword of a row carries two pad bytes).
struct CubeRows { int ox, oy, oz, y0, y1, z0, z1; }; // the cube's origin in the volume and the rows / planes of it the templates can reach
template <class DST>
__device__ __forceinline__ void stage_cube_rows(const Vol &V, const CubeRows &R, int wv, int nwv, int lane, DST *cube32)
{
    static_assert(PH_PITCH % 4 == 0 && PH_PITCH >= PH_CS && PH_PITCH <= 64, "one dword per lane, 16 lanes per row");
    constexpr int NR = 16;
    typedef unsigned __attribute__((aligned(1))) u32u;
    // dword d of a row holds the voxels x = 4 d .. 4 d + 3
    const int sub = lane >> 4, l4 = (lane & 15) * 4, sx = l4;
    const int y0 = R.y0, ny = R.y1 - y0, z0 = R.z0, nrows = (R.z1 - z0) * ny;
    const i64 nvox = V.wh * V.l;
    const int stride = nwv * 4;                       // rows between two loads of a lane
    const int sq = stride / ny, sr = stride - sq * ny; // wave-uniform: one division per work-group
    int r = wv * 4 + sub;
    int zq = r / ny, yr = r - zq * ny;                 // plane / row of r, advanced without dividing again
    for (; r - sub < nrows; ) { // wave-uniform trip count (r - sub is the wave's first row of this round)
        unsigned v[NR];
        int at[NR];
        bool ok[NR];
#pragma unroll
        for (int j = 0; j < NR; j++) {
            ok[j] = r < nrows;
            const int zz = z0 + (ok[j] ? zq : 0), yy = y0 + (ok[j] ? yr : 0);
            const int zg = R.oz + zz < V.l ? R.oz + zz : V.l - 1, yg = R.oy + yy < V.h ? R.oy + yy : V.h - 1;
            const i64 idx = (i64)zg * V.wh + (i64)yg * V.w + R.ox + sx;
            if (idx + 3 < nvox) {
                v[j] = *(const u32u *)(V.img + idx);
            } else { // the last bytes of the volume: byte loads, clamped (values past the row end are never addressed)
                v[j] = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) v[j] |= (unsigned)V.img[idx + q < nvox ? idx + q : nvox - 1] << (8 * q);
            }
            at[j] = (zz * PH_PLANE + yy * PH_PITCH + l4) >> 2;
            r += stride; zq += sq; yr += sr;
            if (yr >= ny) { yr -= ny; zq++; }
        }
#pragma unroll
        for (int j = 0; j < NR; j++)
            if (ok[j] && l4 < PH_PITCH) cube32[at[j]] = v[j];
    }
}

#ifdef PNR_SMC_STAMPS
// diagnostic build only: shader-clock sums over the phases of ph_predict ([0..6], [7] = work-groups) and ph_update ([8..14], [15])
__device__ unsigned long long g_pu_stamps[16];
__device__ unsigned long long g_pu_fine[8]; // ph_predict, thread 0: [0] until the parent pose is there, [1] direction loop, [2] CDF search, [3] count [4] particles whose hash leader has another pose (fall-back scans), [5] particles looked up (racy adds: a diagnostic)
#define PU_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define PU_STAMP(var)
#endif

__global__ __launch_bounds__(256) void ph_predict(Tab T, TabX X, PhState P, const float *__restrict__ seeds6, Vol V, int np, int ni, int it_arg,
                                                   int lp, int CS, int tbl)
{
    const int tid = threadIdx.x, B = blockDim.x;
    PU_STAMP(pt0);
    if (blockIdx.x == 0 && tid == 0) P.cnt[lp ^ 1] = 0; // filled by ph_update of this step
    if ((int)blockIdx.x >= P.cnt[lp]) return;
    const int tr = P.list[lp * P.cap + blockIdx.x];
    int *fl = P.flags + (i64)tr * FL_N;
    if (fl[FL_PAUSE]) return; // paused by the scheduler (stream_sched.h, tentative replay): not stepped, state kept
    const int it = it_arg >= 0 ? it_arg : fl[FL_IT]; // streaming mode: every trace has its own iteration count
    __shared__ int sbox[8];
    __shared__ float s_v[3 * 64]; // the ndir (50 / 30) unit directions of the predictor: read np x ndir times
    extern __shared__ unsigned int dsm[]; // [np] pose hash, [np] representative, [6 np] pose words, [tbl] + [tbl] table (the duplicate search)
    const bool tail = (it == ni) || (fl[FL_STOP] != 0);
    const int pending = it - 1; // the previous iteration's centroid is evaluated with this iteration's chains
    float *part = P.part + (i64)tr * 2 * np * PSTRIDE;
    float *cur = part + (it & 1) * np * PSTRIDE;
    const float *prv = part + ((it & 1) ^ 1) * np * PSTRIDE;
    const float *xc_pen = P.xcs + (i64)tr * 16 + ((it & 1) ^ 1) * 8;
    float *prior = P.prior + (i64)tr * np;
    const int *idxres = P.idxres + (i64)tr * np;
    const int resampled_prev = fl[FL_RES];
    const float *sd = seeds6 + (i64)tr * 6;
    const float x0 = sd[0], y0 = sd[1], z0 = sd[2], vx0 = sd[3], vy0 = sd[4], vz0 = sd[5];
    if (tid < 3) { sbox[tid] = 0x7fffffff; sbox[3 + tid] = -0x7fffffff; }
    if (tid == 3) { P.ctr[tr] = 0; sbox[6] = 0; }
    if (tid < 3 * T.ndir && tid < 3 * 64) s_v[tid] = T.v[tid];
    unsigned int *spose = dsm + 2 * np;
    // first-occurrence table of the duplicate search: hash -> smallest particle index with that hash (open addressing, tbl slots)
    unsigned int *tkey = spose + 6 * np;
    int *tval = (int *)(tkey + tbl);
    if (P.dedup)
        for (int i = tid; i < tbl; i += B) { tkey[i] = 0xffffffffu; tval[i] = 0x7fffffff; }
    int bmin[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, bmax[3] = {-0x7fffffff, -0x7fffffff, -0x7fffffff}; // this thread's share of the box
    bool badpose = false;
    __syncthreads();
    PU_STAMP(pt1);
    for (int k = tid; k <= np; k += B) {
        float qx, qy, qz, qvx, qvy, qvz;
        if (k == np) {
            if (pending < 0) continue;
            qx = xc_pen[0]; qy = xc_pen[1]; qz = xc_pen[2]; qvx = xc_pen[3]; qvy = xc_pen[4]; qvz = xc_pen[5];
        } else {
            if (tail) continue;
            float *q = cur + k * PSTRIDE;
            if (it == 0) { // iter0New: systematic sample of the isotropic prior (tracker.cpp:1006-1024)
                const float stepw = T.w0cws[T.sz - 1] / np;
                const float u1 = stepw * ((float)T.rng[0] / (float)2147483647);
                const float ui = u1 + k * stepw;
                const int s = cdf_search_wide(T.w0cws, T.sz, ui);
                q[PX] = x0 + T.p[3 * s + 0];
                q[PY] = y0 + T.p[3 * s + 1];
                q[PZ] = z0 + T.p[3 * s + 2];
                q[PVX] = (vx0 != vx0) ? T.u[3 * s + 0] : vx0;
                q[PVY] = (vy0 != vy0) ? T.u[3 * s + 1] : vy0;
                q[PVZ] = (vz0 != vz0) ? T.u[3 * s + 2] : vz0;
                prior[k] = T.w0[s];
            } else { // iterINew (:1104-1132)
                const int k1 = resampled_prev ? idxres[k] : k;
                const float *par = prv + k1 * PSTRIDE;
                int vi = -1;
                float best = -FLT_MAX;
#ifdef PNR_SMC_STAMPS
                if (tid == 0) { const float keep = par[PVX]; asm volatile("" ::"v"(keep)); g_pu_fine[0] += __builtin_amdgcn_s_memtime() - pt1; }
                const unsigned long long fa = __builtin_amdgcn_s_memtime();
#endif
                const float pvx = par[PVX], pvy = par[PVY], pvz = par[PVZ];
#pragma unroll 10
                for (int a = 0; a < T.ndir; a++) {
                    const float dp = pvx * s_v[3 * a] + pvy * s_v[3 * a + 1] + pvz * s_v[3 * a + 2];
                    if (dp > best) { best = dp; vi = a; }
                }
                if (vi < 0) vi = 0;
#ifdef PNR_SMC_STAMPS
                asm volatile("" ::"v"(vi));
                const unsigned long long fb = __builtin_amdgcn_s_memtime();
#endif
                const float *cws = T.wcws + (i64)vi * T.sz;
                const float u1 = cws[T.sz - 1] * ((float)T.rng[k] / (float)2147483647);
                const int s = cdf_search_wide(cws, T.sz, u1);
#ifdef PNR_SMC_STAMPS
                asm volatile("" ::"v"(s));
                if (tid == 0) { const unsigned long long fc = __builtin_amdgcn_s_memtime(); g_pu_fine[1] += fb - fa; g_pu_fine[2] += fc - fb; g_pu_fine[3] += 1; }
#endif
                q[PX] = par[PX] + T.p[3 * s + 0];
                q[PY] = par[PY] + T.p[3 * s + 1];
                q[PZ] = par[PZ] + T.p[3 * s + 2];
                q[PVX] = T.u[3 * s + 0];
                q[PVY] = T.u[3 * s + 1];
                q[PVZ] = T.u[3 * s + 2];
                prior[k] = T.w[(i64)vi * T.sz + s];
            }
            qx = q[PX]; qy = q[PY]; qz = q[PZ]; qvx = q[PVX]; qvy = q[PVY]; qvz = q[PVZ];
            {
                unsigned int h = pose_hash(qx, qy, qz, qvx, qvy, qvz);
                if (h == 0xffffffffu) h = 0xfffffffeu; // (the table's "empty")
                if (P.dedup) dsm[k] = h;
                if (P.dedup) {
                    for (unsigned int slot = h & (tbl - 1);; slot = (slot + 1) & (tbl - 1)) { // tbl >= 2 np: an empty slot exists
                        const unsigned int old = atomicCAS(&tkey[slot], 0xffffffffu, h);
                        if (old == 0xffffffffu || old == h) { atomicMin(&tval[slot], k); break; }
                    }
                }
            }
            if (P.dedup) { // (without the search the kernel is launched without room for the poses and the table)
                spose[6 * k + 0] = __float_as_uint(qx); spose[6 * k + 1] = __float_as_uint(qy); spose[6 * k + 2] = __float_as_uint(qz);
                spose[6 * k + 3] = __float_as_uint(qvx); spose[6 * k + 4] = __float_as_uint(qvy); spose[6 * k + 5] = __float_as_uint(qvz);
            }
        }
        const Frame f = make_frame(qx, qy, qz, qvx, qvy, qvz);
        const float ex = X.ext_v * fabsf(qvx) + X.ext_uw * (fabsf(f.ux) + fabsf(f.wx)) + 1.5f;
        const float ey = X.ext_v * fabsf(qvy) + X.ext_uw * (fabsf(f.uy) + fabsf(f.wy)) + 1.5f;
        const float ez = X.ext_v * fabsf(qvz) + X.ext_uw * (fabsf(f.uz) + fabsf(f.wz)) + 1.5f;
        if (qx == qx && qy == qy && qz == qz && ex == ex && ey == ey && ez == ez) {
            const float big = 1e6f;
            bmin[0] = min(bmin[0], (int)floorf(fmaxf(qx - ex, -big)));
            bmin[1] = min(bmin[1], (int)floorf(fmaxf(qy - ey, -big)));
            bmin[2] = min(bmin[2], (int)floorf(fmaxf(qz - ez, -big)));
            bmax[0] = max(bmax[0], (int)floorf(fminf(qx + ex, big)) + 2);
            bmax[1] = max(bmax[1], (int)floorf(fminf(qy + ey, big)) + 2);
            bmax[2] = max(bmax[2], (int)floorf(fminf(qz + ez, big)) + 2);
        } else {
            badpose = true; // NaN / inf pose: its clamped samples may land anywhere in the cube
        }
    }
    // the box of the wave first (six values through the lanes), then one LDS atomic each: 200 atomics on one address are served
    // one lane after the other
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            bmin[a] = min(bmin[a], __shfl_xor(bmin[a], o));
            bmax[a] = max(bmax[a], __shfl_xor(bmax[a], o));
        }
    }
    if ((tid & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) { atomicMin(&sbox[a], bmin[a]); atomicMax(&sbox[3 + a], bmax[a]); }
    }
    if (badpose) sbox[6] = 1;
    __syncthreads();
    PU_STAMP(pt2);
    if (tid == 0) { // cube origin: centred on the bounding box of all templates, kept inside the volume
        const int dim[3] = {V.w, V.h, V.l};
        for (int a = 0; a < 3; a++) {
            // samples beyond a border are clamped onto it (interp: [0, dim - 1.001]) and then read the two outermost
            // rows, so the box always keeps those when it touches or lies past the border
            int lo = sbox[a] < 0 ? 0 : (sbox[a] > dim[a] - 1 ? dim[a] - 1 : sbox[a]);
            int hi = sbox[3 + a] > dim[a] - 1 ? dim[a] - 1 : (sbox[3 + a] < lo ? lo : sbox[3 + a]);
            if (lo > dim[a] - 2) lo = dim[a] - 2 < 0 ? 0 : dim[a] - 2;
            if (hi < 1) hi = dim[a] - 1 < 1 ? dim[a] - 1 : 1;
            int o = (lo + hi + 1) / 2 - CS / 2;
            if (o > dim[a] - CS) o = dim[a] - CS;
            if (o < 0) o = 0;
            fl[FL_OX + a] = o;
            if (a > 0) { // rows of the cube that can be sampled at all: only those are staged
                int r0 = lo - o, r1 = hi - o + 1;
                if (r0 < 0) r0 = 0;
                if (r1 > CS) r1 = CS;
                if (sbox[6] || r1 <= r0) { r0 = 0; r1 = CS; }
                fl[FL_Y0 + 2 * (a - 1)] = r0;
                fl[FL_Y0 + 2 * (a - 1) + 1] = r1;
            }
            sbox[a] = o; // origin for the per-sigma test below
        }
        sbox[7] = ((1 << T.nsig) - 1) * 0x101; // bits 0..7: inside cube and volume, bits 8..15: inside the volume
    }
    __syncthreads();
    PU_STAMP(pt3);
    // Per sigma: do ALL templates of this trace lie inside the cube and inside the volume?  Then the sampling kernel takes
    // the variant without clamps, range tests and fallback for that sigma (the common case for the smaller scales).
    {
        const int dim[3] = {V.w, V.h, V.l};
        const float hi_lim[3] = {V.xmax, V.ymax, V.zmax};
        unsigned ok = (1u << T.nsig) - 1u, okv = ok; // inside cube and volume | inside the volume
        for (int k = tid; k <= np; k += B) {
            const float *q;
            if (k == np) { if (pending < 0) continue; q = xc_pen; }
            else { if (tail) continue; q = cur + k * PSTRIDE; }
            const Frame f = make_frame(q[0], q[1], q[2], q[3], q[4], q[5]);
            const float av[3] = {fabsf(q[3]), fabsf(q[4]), fabsf(q[5])};
            const float auw[3] = {fabsf(f.ux) + fabsf(f.wx), fabsf(f.uy) + fabsf(f.wy), fabsf(f.uz) + fabsf(f.wz)};
            for (int s = 0; s < T.nsig; s++) {
                bool fit = true, fitv = true;
                for (int a = 0; a < (V.l == 1 ? 2 : 3); a++) {
                    const float e = X.ext_vs[s] * av[a] + X.ext_uws[s] * auw[a] + 0.5f;
                    const float lo = q[a] - e, hi = q[a] + e;
                    // inside the volume: the clamp to [0, dim - 1.001] is the identity; inside the cube: (int)coord - origin <= CS - 2
                    const bool inv = lo >= 0.f && hi <= hi_lim[a]; // (false for NaN poses)
                    fitv = fitv && inv;
                    fit = fit && inv && floorf(lo) >= (float)sbox[a] && floorf(hi) + 1.f <= (float)(sbox[a] + CS - 1);
                }
                (void)dim;
                if (!fit) ok &= ~(1u << s);
                if (!fitv) okv &= ~(1u << s);
            }
        }
        if (ok != (1u << T.nsig) - 1u || okv != (1u << T.nsig) - 1u) atomicAnd(&sbox[7], (int)(ok | (okv << 8)));
    }
    __syncthreads();
    PU_STAMP(pt4);
    if (tid == 0) fl[FL_FAST] = sbox[7];
    // ---- exact duplicates among the particles (after a resampling several children of one parent draw the same prediction
    // offset: identical pose, identical likelihood -- 17 % of the evaluations of the bench workload): a particle whose six pose
    // words equal those of an earlier particle shares that particle's chain.  Bit-identical results by construction.
    int *uidx = P.uidx + (i64)tr * P.np_pad, *cmap = P.cmap + (i64)tr * P.np_pad;
    if (tail) { // only the pending centroid is evaluated
        if (tid == 0) { uidx[0] = np; fl[FL_NCH] = 1; }
        return;
    }
    unsigned int *hs = dsm; // filled where the particles were made; later the chain number of a representative
    int *rep = (int *)(dsm + np);
    constexpr int PSTRIDE6 = 6; // (the six pose words of a particle, kept in LDS where the particles were made)
    const unsigned int *curw = spose;
    for (int k = tid; k < np; k += B) {
        int r = k;
        if (P.dedup) {
            // the earliest particle with this hash, from the table; an identical pose has an identical hash, so if that particle is
            // k itself nobody in front of k has k's pose.  Otherwise compare the poses; if they differ (two poses, one hash: about
            // np^2 / 2^33 of the steps) fall back to the scan over everything in front of k.
            const unsigned int hk = hs[k];
            const unsigned int *qk = curw + k * PSTRIDE6;
            int r0 = k;
#ifdef PNR_SMC_STAMPS
            if (tid == 0) atomicAdd(&g_pu_fine[5], (unsigned long long)np);
#endif
            for (unsigned int slot = hk & (tbl - 1);; slot = (slot + 1) & (tbl - 1))
                if (tkey[slot] == hk) { r0 = tval[slot]; break; }
            if (r0 < k) {
                const unsigned int *q0 = curw + r0 * PSTRIDE6;
                if (q0[PX] == qk[PX] && q0[PY] == qk[PY] && q0[PZ] == qk[PZ] && q0[PVX] == qk[PVX] && q0[PVY] == qk[PVY] && q0[PVZ] == qk[PVZ]) {
                    r = r0;
                } else {
#ifdef PNR_SMC_STAMPS
                    atomicAdd(&g_pu_fine[4], 1ull);
#endif
                    for (int j = 0; j < k && r == k; j++) {
                        if (hs[j] != hk) continue;
                        const unsigned int *qj = curw + j * PSTRIDE6;
                        if (qj[PX] == qk[PX] && qj[PY] == qk[PY] && qj[PZ] == qk[PZ] && qj[PVX] == qk[PVX] && qj[PVY] == qk[PVY] && qj[PVZ] == qk[PVZ]) r = j;
                    }
                }
            }
        }
        rep[k] = r;
    }
    __syncthreads();
    PU_STAMP(pt5);
    if (tid < 64) { // chain numbers in particle order: ballots over 64 particles at a time
        int cbase = 0;
        for (int base = 0; base < np; base += 64) {
            const int k = base + tid;
            const bool is = k < np && rep[k] == k;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(is);
            if (is) {
                const int c = cbase + (int)__builtin_popcountll(m & ((1ull << tid) - 1ull));
                uidx[c] = k;
                hs[k] = (unsigned int)c;
            }
            cbase += (int)__builtin_popcountll(m);
        }
        if (tid == 0) { uidx[cbase] = np; fl[FL_NCH] = cbase + 1; } // the centroid closes the list
    }
    __syncthreads();
    for (int k = tid; k < np; k += B) cmap[k] = (int)hs[rep[k]];
#ifdef PNR_SMC_STAMPS
    if (tid == 0) {
        const unsigned long long pt6 = __builtin_amdgcn_s_memtime();
        atomicAdd(&g_pu_stamps[0], pt1 - pt0); atomicAdd(&g_pu_stamps[1], pt2 - pt1); atomicAdd(&g_pu_stamps[2], pt3 - pt2);
        atomicAdd(&g_pu_stamps[3], pt4 - pt3); atomicAdd(&g_pu_stamps[4], pt5 - pt4); atomicAdd(&g_pu_stamps[5], pt6 - pt5);
        atomicAdd(&g_pu_stamps[7], 1ull);
    }
#endif
}

// The cube of every trace of the step, fetched from the image ONCE (PH_CUBE_SPLIT work-groups per trace) into the compact copy the
// sampling work-groups load (PhState::cubes).  Runs between ph_predict (which places the cube) and ph_sample.
// Small work-groups (four waves, one per SIMD): they find room beside the other trace group's sampling work-groups, which a
// 1024-thread work-group only does once one of those has drained (24 us per launch instead of the memory latency it needs).
constexpr int PH_CUBE_SPLIT = 16, PH_CUBE_THREADS = 256;
__global__ __launch_bounds__(PH_CUBE_THREADS) void ph_cube(Vol V, PhState P, int lp, int nslots)
{
    const int slot = blockIdx.x % nslots, part = blockIdx.x / nslots, tid = threadIdx.x;
    if (slot >= P.cnt[lp]) return;
    const int tr = P.list[lp * P.cap + slot];
    const int *fl = P.flags + (i64)tr * FL_N;
    if (fl[FL_PAUSE]) return;
    const CubeRows R = {fl[FL_OX], fl[FL_OY], fl[FL_OZ], fl[FL_Y0], fl[FL_Y1], fl[FL_Z0], fl[FL_Z1]};
    unsigned *dst = (unsigned *)(P.cubes + (i64)(P.stash_base + slot) * ((i64)PH_CS * PH_PLANE));
    stage_cube_rows(V, R, part * (PH_CUBE_THREADS >> 6) + (tid >> 6), PH_CUBE_SPLIT * (PH_CUBE_THREADS >> 6), tid & 63, dst);
}

#ifdef PNR_SMC_STAMPS
// diagnostic build only (make stamps): shader-clock sums over the sampling work-groups: [0] cube staging, [1] item loop of wave 0,
// [2] work-groups, [3] items taken by wave 0 of each work-group
__device__ unsigned long long g_ph_stamps[8];
#endif

template <int CS, bool IS2D, bool GCUBE>
__global__ __launch_bounds__(PH_THREADS) __attribute__((amdgpu_waves_per_eu(5, 5))) void ph_sample(Vol V, Tab T, TabX X, PhState P, int np, int ni, int it_arg, int lp, int nslots)
{
    extern __shared__ unsigned char cube[];
    // part-major: the first nslots work-groups are one per trace, the later ones join whatever is left of their trace
    const int slot = blockIdx.x % nslots, tid = threadIdx.x, B = blockDim.x, S = T.nsig;
    if (slot >= P.cnt[lp]) return;
    const int tr = P.list[lp * P.cap + slot];
    const int *fl = P.flags + (i64)tr * FL_N;
    if (fl[FL_PAUSE]) return;
    const int it = it_arg >= 0 ? it_arg : fl[FL_IT];
    const float *cur = P.part + (i64)tr * 2 * np * PSTRIDE + (it & 1) * np * PSTRIDE;
    const float *xc_pen = P.xcs + (i64)tr * 16 + ((it & 1) ^ 1) * 8;
    Box Bx;
    Bx.lds = (lds_cu8 *)cube;
    Bx.ox = fl[FL_OX]; Bx.oy = fl[FL_OY]; Bx.oz = fl[FL_OZ];
    Bx.org = (unsigned)(Bx.oz * PH_PLANE + Bx.oy * PH_PITCH + Bx.ox);
#ifdef PNR_SMC_STAMPS
    const unsigned long long st0 = __builtin_amdgcn_s_memtime();
    unsigned long long st_items = 0;
    if (Bx.ox + Bx.oy + Bx.oz + fl[FL_Y0] + fl[FL_Y1] + fl[FL_Z0] + fl[FL_Z1] == -12345) return; // the flags have landed
    const unsigned long long st0b = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (GCUBE) {
        // The cube was fetched from the image once, by ph_cube, into a compact copy in LDS layout: every work-group of the trace copies
        // the planes the templates can reach with 16-byte loads and stores (the planes are whole multiples of 16 bytes).  Fetched here by
        // every work-group, the 54-byte rows cost whole 128-byte lines of the image each -- 0.42 GB of HBM reads per 106-trace step,
        // 13 % of the evaluation's traffic -- and a dword load with its address arithmetic per four voxels.
        static_assert(PH_PLANE % 16 == 0, "whole uint4 per plane");
        const int z0 = fl[FL_Z0], z1 = fl[FL_Z1];
        const uint4 *src = (const uint4 *)(P.cubes + (i64)(P.stash_base + slot) * ((i64)PH_CS * PH_PLANE) + (i64)z0 * PH_PLANE);
        uint4 *dst = (uint4 *)(cube + (size_t)z0 * PH_PLANE);
        const int n16 = (z1 - z0) * (PH_PLANE / 16);
        int i = tid;
        for (; i + 3 * B < n16; i += 4 * B) { // four loads in flight per thread
            const uint4 a0 = src[i], a1 = src[i + B], a2 = src[i + 2 * B], a3 = src[i + 3 * B];
            dst[i] = a0; dst[i + B] = a1; dst[i + 2 * B] = a2; dst[i + 3 * B] = a3;
        }
        for (; i < n16; i += B) dst[i] = src[i];
    } else {
        const CubeRows R = {Bx.ox, Bx.oy, Bx.oz, fl[FL_Y0], fl[FL_Y1], fl[FL_Z0], fl[FL_Z1]};
        stage_cube_rows(V, R, tid >> 6, B >> 6, tid & 63, (unsigned *)cube);
    }
    __syncthreads();
#ifdef PNR_SMC_STAMPS
    const unsigned long long st1 = __builtin_amdgcn_s_memtime();
#endif
    const int nch = __builtin_amdgcn_readfirstlane(fl[FL_NCH]);
    const int ngf = nch >> 6, rem = nch & 63, Rt = last_group_stride(rem);
    const int *uidx = P.uidx + (i64)tr * P.np_pad;
    const int fastword = __builtin_amdgcn_readfirstlane(fl[FL_FAST]);
    const int fastmask = fastword & 0xff;        // sigmas whose templates all lie inside the cube and the volume
    const int volmask = (fastword >> 8) & 0xff;  // sigmas whose templates all lie inside the volume (no clamp), whatever the cube holds
    // work items: a full group's item is ROWS template rows (iu) of one v-slice; the last group's chains are spread over the wave
    // `parts` times (sample_slice_packed) and its item is ROWS rounds of `parts` rows each
    constexpr int ROWS = 5;
    const int parts = rem > 0 ? 64 / rem : 1;
    int nchsum = 0, npksum = 0;
    for (int s = 0; s < S; s++) {
        const int nv_s = __builtin_amdgcn_readfirstlane(X.grid[s].nv), nu_s = __builtin_amdgcn_readfirstlane(X.grid[s].nu);
        nchsum += nv_s * ((nu_s + ROWS - 1) / ROWS);
        npksum += nv_s * (((nu_s + parts - 1) / parts + ROWS - 1) / ROWS);
    }
    const int nfull = nchsum * ngf, nitems = nfull + (rem > 0 ? npksum : 0);
    float *const tbase = P.stash + (i64)(P.stash_base + slot) * P.trace_floats;
    const int lane = tid & 63;
    (void)ni;
    // Items in descending cost: (sigma descending, v-slice, row chunk, full group), then the last group's per (sigma, v-slice,
    // round chunk); every wave of the work-groups that share this trace pulls the next one from the trace's counter, so the waves
    // finish within one item of each other.  A tail pass has one chain, the centroid: only last-group items.
    for (;;) {
        int item = 0;
        if (lane == 0) item = atomicAdd(&P.ctr[tr], 1);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= nitems) break;
#ifdef PNR_SMC_STAMPS
        st_items++;
#endif
        const bool packed = item >= nfull;
        int sI = S - 1, r = packed ? item - nfull : item;
        while (sI > 0) {
            const int nv_s = __builtin_amdgcn_readfirstlane(X.grid[sI].nv), nu_s = __builtin_amdgcn_readfirstlane(X.grid[sI].nu);
            const int c = packed ? nv_s * (((nu_s + parts - 1) / parts + ROWS - 1) / ROWS) : nv_s * ((nu_s + ROWS - 1) / ROWS) * ngf;
            if (r < c) break;
            r -= c;
            sI--;
        }
        const Grid gr = X.grid[sI];
        const int nv = __builtin_amdgcn_readfirstlane(gr.nv), nu = __builtin_amdgcn_readfirstlane(gr.nu);
        const int nw = __builtin_amdgcn_readfirstlane(gr.nw), goff = __builtin_amdgcn_readfirstlane(gr.off);
        const int Ms = nv * nu * nw;
        const float *ax = X.axes + __builtin_amdgcn_readfirstlane(X.axes_off[sI]);
        float *const sbase = tbase + (i64)goff * P.W;
        const int nch_rows = packed ? ((nu + parts - 1) / parts + ROWS - 1) / ROWS : (nu + ROWS - 1) / ROWS;
        const int per = packed ? nch_rows : nch_rows * ngf;
        const int iv = r / per, r2 = r - iv * per;
        if (!packed) {
            const int ch = r2 / ngf, g = r2 - ch * ngf;
            const int c = g * 64 + lane; // < nch: the group is full
            const int k = uidx[c];
            const float *q = (k >= np) ? xc_pen : cur + k * PSTRIDE; // (the centroid before the first one exists: zeros, discarded)
            const Frame f = make_frame(q[0], q[1], q[2], q[3], q[4], q[5]);
            // Where ph_predict could not promise the fast form for ALL templates of this sigma (a slab of 37 x 37 samples at an oblique
            // angle pokes out of the 54^3 cube somewhere), this item alone may still lie inside: its five rows of this v-slice for its
            // 64 chains.  The positions are affine in (uu, ww), so the extremes sit at the four corners of the item's (uu, ww) range,
            // evaluated with the very operations sample_slice uses; a margin covers the rounding of the samples in between.
            bool item_fast = (fastmask >> sI & 1) != 0;
            if (!item_fast) {
                const int iu0 = ch * ROWS, iu1 = min(ch * ROWS + ROWS, nu) - 1;
                const float vv = ax[iv], uA = ax[nv + iu0], uB = ax[nv + iu1], wA = ax[nv + nu], wB = ax[nv + nu + nw - 1];
                const float bx = f.px + vv * f.nvx, by = f.py + vv * f.nvy, bz = f.pz + vv * f.nvz;
                bool ok = true;
                const float base[3] = {bx, by, bz}, ud[3] = {f.ux, f.uy, f.uz}, wd3[3] = {f.wx, f.wy, f.wz}, lim[3] = {V.xmax, V.ymax, V.zmax};
                const int org[3] = {Bx.ox, Bx.oy, Bx.oz};
#pragma unroll
                for (int a = 0; a < (IS2D ? 2 : 3); a++) {
                    const float c0 = (base[a] + uA * ud[a]) + wA * wd3[a], c1 = (base[a] + uA * ud[a]) + wB * wd3[a];
                    const float c2 = (base[a] + uB * ud[a]) + wA * wd3[a], c3 = (base[a] + uB * ud[a]) + wB * wd3[a];
                    const float lo = fminf(fminf(c0, c1), fminf(c2, c3)) - 1e-3f, hi = fmaxf(fmaxf(c0, c1), fmaxf(c2, c3)) + 1e-3f;
                    // inside the volume (the clamp is the identity) and the corner pairs inside the cube; false for NaN
                    ok = ok && lo >= 0.f && hi <= lim[a] && floorf(lo) >= (float)org[a] && floorf(hi) + 1.f <= (float)(org[a] + CS - 1);
                }
                item_fast = __builtin_amdgcn_ballot_w64(!ok) == 0ull;
            }
#ifdef PNR_SMC_STAMPS
            if (lane == 0) atomicAdd(&g_pu_fine[item_fast ? 6 : 7], 1ull);
#endif
            if (item_fast)
                sample_slice<CS, IS2D, true, PH_PITCH>(V, Bx, f, nv, nu, nw, ax, iv, sbase + (i64)g * Ms * 64, ch * ROWS, ch * ROWS + ROWS);
            else if (volmask >> sI & 1)
                sample_slice<CS, IS2D, false, PH_PITCH, true>(V, Bx, f, nv, nu, nw, ax, iv, sbase + (i64)g * Ms * 64, ch * ROWS, ch * ROWS + ROWS);
            else
                sample_slice<CS, IS2D, false, PH_PITCH>(V, Bx, f, nv, nu, nw, ax, iv, sbase + (i64)g * Ms * 64, ch * ROWS, ch * ROWS + ROWS);
            PNR_HOOK_AFTER_FULL_ITEM(sbase + (i64)g * Ms * 64 + lane + ((i64)iv * nu + ch * ROWS) * nw * 64, (min(ch * ROWS + ROWS, nu) - ch * ROWS) * nw);
        } else {
            const bool act = lane < parts * rem;
            const int pp = act ? lane / rem : 0, j = act ? lane - pp * rem : 0;
            const int k = uidx[ngf * 64 + j];
            const float *q = (k >= np) ? xc_pen : cur + k * PSTRIDE;
            const Frame f = make_frame(q[0], q[1], q[2], q[3], q[4], q[5]);
            sample_slice_packed<CS, IS2D, false, PH_PITCH>(V, Bx, f, nv, nu, nw, ax, iv, parts, pp, act, sbase + (i64)ngf * Ms * 64 + j, Rt, r2 * ROWS, r2 * ROWS + ROWS);
        }
    }
#ifdef PNR_SMC_STAMPS
    if (tid == 0) {
        const unsigned long long st2 = __builtin_amdgcn_s_memtime();
        atomicAdd(&g_ph_stamps[0], st1 - st0);
        atomicAdd(&g_ph_stamps[4], st0b - st0);
        atomicAdd(&g_ph_stamps[1], st2 - st1);
        atomicAdd(&g_ph_stamps[2], 1ull);
        atomicAdd(&g_ph_stamps[3], st_items);
    }
#endif
}

// one wave per (trace, sigma, chain group): the ordered sums of the chains from the stash
constexpr int PH_CH = 32; // stash values in flight per lane (64: 256 VGPRs, slower with many traces, no faster with few)
// DEEP: zncc_from_stash_deep (four chunk buffers in turn) for launches bound by the latency of a chain, not by the stash's bandwidth
template <bool DEEP>
__global__ __launch_bounds__(64) void ph_sums(Tab T, TabX X, PhState P, int np, int np_pad, int ni, int it_arg, int lp, int ng_max)
{
    const int S = T.nsig, lane = threadIdx.x;
    const int slot = blockIdx.x / (S * ng_max);
    if (slot >= P.cnt[lp]) return;
    const int tr = P.list[lp * P.cap + slot];
    const int *fl = P.flags + (i64)tr * FL_N;
    if (fl[FL_PAUSE]) return;
    const int r = blockIdx.x - slot * (S * ng_max);
    const int sI = r / ng_max, g = r - sI * ng_max;
    const int nch = fl[FL_NCH], ngf = nch >> 6, rem = nch & 63;
    if (g > ngf || (g == ngf && rem == 0)) return; // this trace has fewer chain groups in this iteration
    (void)np; (void)ni; (void)it_arg;
    const Grid gr = X.grid[sI];
    const int M = gr.nv * gr.nu * gr.nw;
    const float *sbase = P.stash + (i64)(P.stash_base + slot) * P.trace_floats + (i64)gr.off * P.W;
    const float *wd = X.wd + gr.off;
    float cv;
    bool valid = true;
#ifdef PNR_SMC_STAMPS
    const unsigned long long sst0 = __builtin_amdgcn_s_memtime();
    unsigned long long sst1 = sst0;
#endif
    if (g < ngf) {
#ifdef PNR_SMC_STAMPS
        if constexpr (DEEP) cv = zncc_from_stash_deep<64>(sbase + (i64)g * M * 64 + lane, M, wd, T.corrc[sI], &sst1);
        else cv = zncc_from_stash<64, PH_CH>(sbase + (i64)g * M * 64 + lane, M, wd, T.corrc[sI], &sst1);
#else
        if constexpr (DEEP) cv = zncc_from_stash_deep<64>(sbase + (i64)g * M * 64 + lane, M, wd, T.corrc[sI]);
        else cv = zncc_from_stash<64, PH_CH>(sbase + (i64)g * M * 64 + lane, M, wd, T.corrc[sI]);
#endif
    } else { // last group: narrow rows; lanes without a chain re-read a valid column (same 64 B granules)
        valid = lane < rem;
        const int j = valid ? lane : rem - 1;
        const float *col = sbase + (i64)ngf * M * 64 + j;
        if constexpr (DEEP) {
            switch (last_group_stride(rem)) {
            case 16: cv = zncc_from_stash_deep<16>(col, M, wd, T.corrc[sI]); break;
            case 32: cv = zncc_from_stash_deep<32>(col, M, wd, T.corrc[sI]); break;
            default: cv = zncc_from_stash_deep<64>(col, M, wd, T.corrc[sI]); break;
            }
        } else {
            switch (last_group_stride(rem)) {
            case 16: cv = zncc_from_stash<16, PH_CH>(col, M, wd, T.corrc[sI]); break;
            case 32: cv = zncc_from_stash<32, PH_CH>(col, M, wd, T.corrc[sI]); break;
            default: cv = zncc_from_stash<64, PH_CH>(col, M, wd, T.corrc[sI]); break;
            }
        }
    }
    if (valid) P.corr[((i64)tr * S + sI) * np_pad + g * 64 + lane] = cv; // indexed by chain
#ifdef PNR_SMC_STAMPS
    if (lane == 0 && g < ngf && sI == S - 1) { // full groups of the longest template: [5] pass 1, [6] pass 2, [7] waves
        const unsigned long long sst2 = __builtin_amdgcn_s_memtime();
        atomicAdd(&g_ph_stamps[5], sst1 - sst0);
        atomicAdd(&g_ph_stamps[6], sst2 - sst1);
        atomicAdd(&g_ph_stamps[7], 1ull);
    }
#endif
}

// a[0] + a[1] + ... in this order (one f32 add after the other, as the reference's loops), read from LDS eight terms at a time so
// that the chain of adds does not wait for one load after the other.  Every thread that calls it gets the sum: the address is
// the same in all lanes (a broadcast read), so a value that all threads need is simply added up by all of them.
__device__ __forceinline__ float lds_seq_sum(const float *a, int n)
{
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= n; k += 8) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; j++) t[j] = a[k + j];
#pragma unroll
        for (int j = 0; j < 8; j++) s += t[j];
    }
    for (; k < n; k++) s += a[k];
    return s;
}

__global__ __launch_bounds__(256) void ph_update(Vol V, Tab T, PhState P, int np, int np_pad, int ni, int it_arg, int lp, float Kc, float znccth,
                                                  float neff_ratio, const unsigned char *__restrict__ den, int nodepervol, TraceOut O)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x, B = blockDim.x, S = T.nsig;
    PU_STAMP(ut0);
    if ((int)blockIdx.x >= P.cnt[lp]) return;
    const int tr = P.list[lp * P.cap + blockIdx.x];
    int *fl = P.flags + (i64)tr * FL_N;
    if (fl[FL_PAUSE]) return; // (not appended to the next step's list: it comes back through ph_control)
    const int it = it_arg >= 0 ? it_arg : fl[FL_IT];
    float *cur = lds;                 // [np][9]
    float *prvw = cur + np * PSTRIDE; // [np] weights of the previous iteration
    float *prior = prvw + np;         // [np]
    float *lhood = prior + np;        // [np]
    float *csw = lhood + np;          // [np]
    float *corr_ks = csw + np;        // [S][np_pad]
    float *sneff = corr_ks + S * np_pad; // [2]
    int *sres = (int *)(sneff + 2);   // [0] resample, [1] stop code after the pending centroid
    float *sxc = (float *)(sres + 2); // [8] centroid of this iteration
    const bool tail = (it == ni) || (fl[FL_STOP] != 0);
    const int pending = it - 1;
    const int resampled_prev = fl[FL_RES];
    float *gpart = P.part + (i64)tr * 2 * np * PSTRIDE;
    float *gcur = gpart + (it & 1) * np * PSTRIDE;
    const float *gprv = gpart + ((it & 1) ^ 1) * np * PSTRIDE;
    float *xc_cur = P.xcs + (i64)tr * 16 + (it & 1) * 8;
    const float *xc_pen = P.xcs + (i64)tr * 16 + ((it & 1) ^ 1) * 8;
    int *gidx = P.idxres + (i64)tr * np;
    for (int e = tid; e < S * np_pad; e += B) corr_ks[e] = P.corr[(i64)tr * S * np_pad + e]; // by chain (ph_predict's uidx / cmap)
    const int cen = fl[FL_NCH] - 1; // the centroid is this iteration's last chain
    const int *cmap = P.cmap + (i64)tr * P.np_pad;
    if (!tail) {
        for (int e = tid; e < np * PSTRIDE; e += B) cur[e] = gcur[e];
        for (int k = tid; k < np; k += B) { prvw[k] = gprv[k * PSTRIDE + PW]; prior[k] = P.prior[(i64)tr * np + k]; }
    }
    __syncthreads();
    PU_STAMP(ut1);

    // ---- finish the pending centroid: corr, stop tests of that iteration (tracker.cpp:1072-1079) ----
    if (pending >= 0 && tid == 0) {
        float best = -FLT_MAX, bs = xc_pen[6];
        for (int s = 0; s < S; s++) {
            const float cv = corr_ks[s * np_pad + cen];
            if (cv > best) { best = cv; bs = T.sig[s]; }
        }
        float *xo = O.xc + ((i64)tr * ni + pending) * 8;
        xo[0] = xc_pen[0]; xo[1] = xc_pen[1]; xo[2] = xc_pen[2]; xo[3] = xc_pen[3]; xo[4] = xc_pen[4]; xo[5] = xc_pen[5];
        xo[6] = bs; xo[7] = best;
        if ((fl[FL_STOP] == 0 || fl[FL_STOP] == 3) && best < znccth) { fl[FL_STOP] = 2; fl[FL_T] = pending; }
    }
    if (tid == 0) sres[1] = fl[FL_STOP];
    __syncthreads();
    if (tail || sres[1] != 0) {
        if (tid == 0) {
            O.T[tr] = fl[FL_T];
            O.stop[tr] = fl[FL_STOP];
            fl[FL_DONE] = 1;
        }
        return;
    }

    PU_STAMP(ut2);
    // ---- max over sigma, likelihood exp(Kc*corr) (:1028-1029) ----
    for (int k = tid; k < np; k += B) {
        float best = -FLT_MAX, bs = 0.f;
        const int ck = cmap[k];
        for (int s = 0; s < S; s++) {
            const float cv = corr_ks[s * np_pad + ck];
            if (cv > best) { best = cv; bs = T.sig[s]; }
        }
        cur[k * PSTRIDE + PCORR] = best;
        cur[k * PSTRIDE + PSIG] = bs;
        lhood[k] = expf_libm(Kc * best);
    }
    __syncthreads();

    PU_STAMP(ut3);
    // ---- weights, N_eff, CDF, centroid: sequential sums in particle order (:1035-1071) ----
    // The sums run in the reference's order, one add after the other.  What all threads need (the two normalisations) every
    // thread adds up for itself from contiguous LDS arrays (broadcast reads, eight terms in flight): nothing is handed over and
    // waited for.  The nine chains of the last step run side by side: the seven centroid sums in seven lanes of wave 0, N_eff in
    // wave 1, the CDF in wave 2 (in one wave they would run one after the other).
    const bool carry = (it > 0) && !resampled_prev;
    const int lane = tid & 63, wv = tid >> 6;
    {
        const float wnorm_prior = lds_seq_sum(prior, np);
        for (int k = tid; k < np; k += B) {
            const double base = carry ? (double)prvw[k] : (1.0 / np);
            const float w = (float)(base * (double)(prior[k] / wnorm_prior) * (double)lhood[k]);
            cur[k * PSTRIDE + PW] = w;
            lhood[k] = w; // from here on: the unnormalised weights, contiguous
        }
    }
    __syncthreads();
    {
        const float wsum = lds_seq_sum(lhood, np);
        for (int k = tid; k < np; k += B) {
            const float wn = lhood[k] / wsum;
            cur[k * PSTRIDE + PW] = wn;
            prvw[k] = wn; // from here on: the normalised weights, contiguous (nobody reads prvw or writes lhood any more)
        }
    }
    __syncthreads();
    PU_STAMP(ut4);
    const float *wn = prvw;
    if (wv == 0) {
        if (lane < 7) {
            const int comp = (lane < 6) ? lane : PSIG;
            float a = 0.f;
            int k = 0;
            for (; k + 8 <= np; k += 8) {
                float w8[8], c8[8];
#pragma unroll
                for (int j = 0; j < 8; j++) { w8[j] = wn[k + j]; c8[j] = cur[(k + j) * PSTRIDE + comp]; }
#pragma unroll
                for (int j = 0; j < 8; j++) a += w8[j] * c8[j];
            }
            for (; k < np; k++) a += wn[k] * cur[k * PSTRIDE + comp];
            sxc[lane] = a;
        }
    } else if (wv == 1) {
        if (lane == 0) { // N_eff: f64 add of the f64 square, rounded to f32 at every step
            float neff = 0.f;
            int k = 0;
            for (; k + 8 <= np; k += 8) {
                float w8[8];
#pragma unroll
                for (int j = 0; j < 8; j++) w8[j] = wn[k + j];
#pragma unroll
                for (int j = 0; j < 8; j++) neff = (float)((double)neff + (double)w8[j] * (double)w8[j]);
            }
            for (; k < np; k++) {
                const float wk = wn[k];
                neff = (float)((double)neff + (double)wk * (double)wk);
            }
            sneff[0] = (float)(1.0 / (double)neff);
        }
    } else if (wv == 2) {
        if (lane == 0) { // CDF (:1062: the first term is added to 0.f, as there)
            float acc = 0.f;
            int k = 0;
            for (; k + 8 <= np; k += 8) {
                float w8[8];
#pragma unroll
                for (int j = 0; j < 8; j++) w8[j] = wn[k + j];
#pragma unroll
                for (int j = 0; j < 8; j++) { acc = w8[j] + acc; csw[k + j] = acc; }
            }
            for (; k < np; k++) { acc = wn[k] + acc; csw[k] = acc; }
        }
    }
    __syncthreads();
    PU_STAMP(ut5);
    if (tid == 0) {
        const float cx = sxc[0], cy = sxc[1], cz = sxc[2], cvx = sxc[3], cvy = sxc[4], cvz = sxc[5];
        const float neff = sneff[0];
        const float vnorm = (float)sqrt((double)cvx * (double)cvx + (double)cvy * (double)cvy + (double)cvz * (double)cvz);
        sxc[3] = cvx / vnorm; sxc[4] = cvy / vnorm; sxc[5] = cvz / vnorm;
        if (it < O.dbg_iters && O.neff) O.neff[(i64)tr * O.dbg_iters + it] = neff;
        const int x1 = (int)roundf(cx), y1 = (int)roundf(cy), z1 = (int)roundf(cz);
        int res = 0;
        if (x1 < 0 || x1 >= V.w || y1 < 0 || y1 >= V.h || z1 < 0 || z1 >= V.l) {
            fl[FL_STOP] = 1;
            fl[FL_T] = it;
        } else if (den && (int)den[(i64)z1 * V.wh + (i64)y1 * V.w + x1] >= nodepervol) {
            fl[FL_STOP] = 3; // saturated by earlier batches: DENSITY stop at the latest here (tracker.cpp:855)
            fl[FL_T] = it + 1;
        } else if (neff / np < neff_ratio) {
            res = 1;
        }
        fl[FL_RES] = res;
        sres[0] = res;
    }
    __syncthreads();
    PU_STAMP(ut6);
    if (tid < 7) xc_cur[tid] = sxc[tid];
    if (sres[0]) { // systematic resampling by bisection on the monotone CDF (clamped; :1082-1090)
        const float u1 = (float)((1.0 / np) * (double)((float)T.rng[(it == 0) ? 1 : np] / (float)2147483647));
        for (int k = tid; k < np; k += B) {
            const float ui = (float)((double)u1 + k * (1.0 / np));
            int lo = 0, hi = np - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (ui > csw[mid]) lo = mid + 1; else hi = mid;
            }
            gidx[k] = lo;
            if (it < O.dbg_iters && O.idxres) O.idxres[((i64)tr * O.dbg_iters + it) * np + k] = lo;
        }
    }
    if (tid == 0) { // still running
        P.list[(lp ^ 1) * P.cap + atomicAdd(&P.cnt[lp ^ 1], 1)] = tr;
        fl[FL_IT] = it + 1;
    }
    for (int e = tid; e < np * PSTRIDE; e += B) gcur[e] = cur[e];
    if (it < O.dbg_iters && O.xfilt) {
        float *dst = O.xfilt + ((i64)tr * O.dbg_iters + it) * np * PSTRIDE;
        for (int e = tid; e < np * PSTRIDE; e += B) dst[e] = cur[e];
    }
#ifdef PNR_SMC_STAMPS
    if (tid == 0) {
        const unsigned long long ut7 = __builtin_amdgcn_s_memtime();
        atomicAdd(&g_pu_stamps[8], ut1 - ut0); atomicAdd(&g_pu_stamps[9], ut2 - ut1); atomicAdd(&g_pu_stamps[10], ut3 - ut2);
        atomicAdd(&g_pu_stamps[11], ut4 - ut3); atomicAdd(&g_pu_stamps[12], ut5 - ut4); atomicAdd(&g_pu_stamps[13], ut6 - ut5);
        atomicAdd(&g_pu_stamps[14], ut7 - ut6); atomicAdd(&g_pu_stamps[15], 1ull);
    }
#endif
}

// the state the host reads after a poll (the step lists' counters and the flags of all slots: contiguous in P.cnt) into its pinned
// snapshot
__global__ __launch_bounds__(256) void ph_snapshot(const int4 *__restrict__ src, int4 *__restrict__ dst, int n4)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n4) dst[i] = src[i];
}

// streaming mode: hand `m` free slots to new traces and append them to the list of this step (one work-group, runs
// alone in stream order between two steps)
__global__ __launch_bounds__(256) void ph_admit(PhState P, float *__restrict__ s6, const int *__restrict__ new_slots,
                                                 const float *__restrict__ new_s6, int m, int lp, int ni)
{
    __shared__ int base;
    if (threadIdx.x == 0) base = P.cnt[lp];
    __syncthreads();
    for (int j = threadIdx.x; j < m; j += blockDim.x) {
        const int slot = new_slots[j];
        for (int a = 0; a < 6; a++) s6[(i64)slot * 6 + a] = new_s6[(i64)j * 6 + a];
        int *fl = P.flags + (i64)slot * FL_N;
        for (int a = 0; a < FL_N; a++) fl[a] = 0;
        fl[FL_T] = ni;
        for (int a = 0; a < 16; a++) P.xcs[(i64)slot * 16 + a] = 0.f;
        P.list[lp * P.cap + base + j] = slot;
    }
    __syncthreads();
    if (threadIdx.x == 0) P.cnt[lp] = base + m;
}

// streaming mode, tentative replay: take some traces off the list of this step (they keep slot and state; FL_PAUSE marks them) and
// put paused ones back on it (one work-group, runs alone in stream order between two steps).  The list is compacted here, so a
// pause takes effect at once and the slot of a trace the host has ended can be handed to a new trace in the same turn.
__global__ __launch_bounds__(256) void ph_control(PhState P, const int *__restrict__ pause, int np_, const int *__restrict__ resume, int nr, int lp)
{
    __shared__ int kept;
    const int n = P.cnt[lp];
    int *list = P.list + (size_t)lp * P.cap, *tmp = P.list + (size_t)(lp ^ 1) * P.cap; // (the other list is only filled by the next ph_update)
    if (threadIdx.x == 0) kept = 0;
    for (int j = threadIdx.x; j < np_; j += blockDim.x) P.flags[(i64)pause[j] * FL_N + FL_PAUSE] = 1;
    __syncthreads();
    if (np_ > 0) {
        for (int j = threadIdx.x; j < n; j += blockDim.x) {
            const int tr = list[j];
            if (!P.flags[(i64)tr * FL_N + FL_PAUSE]) tmp[atomicAdd(&kept, 1)] = tr;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < kept; j += blockDim.x) list[j] = tmp[j];
        __syncthreads();
    } else if (threadIdx.x == 0) {
        kept = n;
    }
    __syncthreads();
    const int base = kept;
    for (int j = threadIdx.x; j < nr; j += blockDim.x) {
        P.flags[(i64)resume[j] * FL_N + FL_PAUSE] = 0;
        list[base + j] = resume[j];
    }
    __syncthreads();
    if (threadIdx.x == 0) P.cnt[lp] = base + nr;
}

// What a poll asks of a trace group, in ONE dispatch (each dispatch in front of the group's next step costs ~14 us of its chain, and
// there were three): work-group 0 runs ph_control's and then ph_admit's work (same order, same code); the work-groups behind it
// raise the density map (den_scatter's monotone byte-wise maximum).  The lists are read from pinned host staging by the kernel.
__global__ __launch_bounds__(256) void ph_poll(PhState P, float *__restrict__ s6, int do_ctl, const int *__restrict__ pause, int np_,
                                                const int *__restrict__ resume, int nr, const int *__restrict__ new_slots,
                                                const float *__restrict__ new_s6, int m, int lp, int ni, unsigned char *__restrict__ den,
                                                const i64 *__restrict__ didx, const unsigned char *__restrict__ dval, int nt)
{
    if (blockIdx.x > 0) {
        const int i = ((int)blockIdx.x - 1) * (int)blockDim.x + (int)threadIdx.x;
        if (i >= nt) return;
        const i64 at = didx[i];
        unsigned *w = (unsigned *)(den + (at & ~(i64)3)); // (the map is allocated in whole dwords)
        const int sh = (int)(at & 3) * 8;
        const unsigned v = dval[i];
        unsigned old = __atomic_load_n(w, __ATOMIC_RELAXED);
        while (((old >> sh) & 0xffu) < v) {
            const unsigned got = atomicCAS(w, old, (old & ~(0xffu << sh)) | (v << sh));
            if (got == old) break;
            old = got;
        }
        return;
    }
    __shared__ int kept, cntv;
    if (threadIdx.x == 0) { cntv = P.cnt[lp]; kept = 0; }
    __syncthreads();
    int *list = P.list + (size_t)lp * P.cap, *tmp = P.list + (size_t)(lp ^ 1) * P.cap; // (the other list is only filled by the next ph_update)
    if (do_ctl) { // ---- ph_control
        const int n = cntv;
        for (int j = threadIdx.x; j < np_; j += blockDim.x) P.flags[(i64)pause[j] * FL_N + FL_PAUSE] = 1;
        __syncthreads();
        if (np_ > 0) {
            for (int j = threadIdx.x; j < n; j += blockDim.x) {
                const int tr = list[j];
                if (!P.flags[(i64)tr * FL_N + FL_PAUSE]) tmp[atomicAdd(&kept, 1)] = tr;
            }
            __syncthreads();
            for (int j = threadIdx.x; j < kept; j += blockDim.x) list[j] = tmp[j];
            __syncthreads();
        } else if (threadIdx.x == 0) {
            kept = n;
        }
        __syncthreads();
        const int base = kept;
        for (int j = threadIdx.x; j < nr; j += blockDim.x) {
            P.flags[(i64)resume[j] * FL_N + FL_PAUSE] = 0;
            list[base + j] = resume[j];
        }
        __syncthreads();
        if (threadIdx.x == 0) cntv = base + nr;
        __syncthreads();
    }
    if (m > 0) { // ---- ph_admit
        const int base = cntv;
        for (int j = threadIdx.x; j < m; j += blockDim.x) {
            const int slot = new_slots[j];
            for (int a = 0; a < 6; a++) s6[(i64)slot * 6 + a] = new_s6[(i64)j * 6 + a];
            int *fl = P.flags + (i64)slot * FL_N;
            for (int a = 0; a < FL_N; a++) fl[a] = 0;
            fl[FL_T] = ni;
            for (int a = 0; a < 16; a++) P.xcs[(i64)slot * 16 + a] = 0.f;
            list[base + j] = slot;
        }
        __syncthreads();
        if (threadIdx.x == 0) cntv = base + m;
        __syncthreads();
    }
    if (threadIdx.x == 0) P.cnt[lp] = cntv;
}

} // namespace

// ---------------------------------------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------------------------------------
struct pnr_phased {
    static constexpr int RING = 8;
    int64_t cap_traces = 0, cap_dbg = 0;
    int64_t cap_stash = 0; // traces the sample stash holds (list positions of all trace groups together)
    int np = 0, np_pad = 0, S = 0, ni = 0;
    long long trace_floats = 0;
    PhState P{};
    float *d_s6 = nullptr;
    TraceOut O{};
    int *h_cnt = nullptr;       // pinned [RING]: active-trace counters copied back by the stream
    hipEvent_t ev[RING] = {};   // ... and the events that say so
    // streaming trace + replay: pinned records written by the kernels / read back at every poll, admission staging
    pnr_xest *h_xc = nullptr; int *h_flags = nullptr; int *h_new = nullptr; float *h_new_s6 = nullptr;
    int *d_new = nullptr; float *d_new_s6 = nullptr;
    int *h_ctl = nullptr, *d_ctl = nullptr; // pause / resume lists of the tentative replay: [group][2][stream_cap]
    // density updates of the streaming tracer: pinned and device staging per trace group (grow-only), so that an update is queued on the
    // group's own stream and nothing waits for it
    long long *h_den_idx[4] = {}, *d_den_idx[4] = {};
    unsigned char *h_den_val[4] = {}, *d_den_val[4] = {};
    size_t den_cap[4] = {};
    int64_t stream_cap = 0;
    int stream_ni = 0;
    static constexpr int MAXG = 4;
    static_assert(MAXG == 4, "the density staging above is declared with four entries");
    hipStream_t stg[MAXG] = {}, st_den = nullptr; // [1..]: the further trace groups of the streaming tracer; density uploads
    hipEvent_t ev_start = nullptr;
    hipEvent_t ev_state[4] = {}; // per trace group: the state copies of its last launch have landed (PhasedEngine::wait)
    // per trace group: the last upload from its pinned admission / control / density staging has been consumed.  A group whose traces are
    // all paused is not launched, so no wait() lies between two of its turns: the staging is only rewritten behind these events.
    hipEvent_t ev_adm[4] = {}, ev_ctl[4] = {}, ev_den[4] = {};
};

static void phased_free(pnr_phased *h)
{
    hipFree(h->P.part); hipFree(h->P.prior); hipFree(h->P.idxres); hipFree(h->P.corr); hipFree(h->P.xcs);
    hipFree(h->P.stash); hipFree(h->P.cubes); hipFree(h->P.list); hipFree(h->P.cnt) /* (and the flags behind the counters) */; hipFree(h->P.ctr); hipFree(h->P.uidx); hipFree(h->P.cmap); hipFree(h->d_s6);
    hipFree(h->O.T); hipFree(h->O.stop); hipFree(h->O.xc); hipFree(h->O.xfilt); hipFree(h->O.idxres); hipFree(h->O.neff);
    h->P = PhState{};
    h->O = TraceOut{};
    h->d_s6 = nullptr;
    h->cap_traces = h->cap_dbg = 0;
    h->cap_stash = 0;
}

#ifdef PNR_SMC_STAMPS
extern "C" int pnr_debug_pu_fine(unsigned long long *out4, int reset)
{
    if (out4 && hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_pu_fine), 64) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_pu_fine), z, 64) != hipSuccess) return -1; }
    return 0;
}

extern "C" int pnr_debug_pu_stamps(unsigned long long *out16, int reset)
{
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_pu_stamps), 128) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_pu_stamps), z, 128) != hipSuccess) return -1; }
    return 0;
}

extern "C" int pnr_debug_ph_stamps(unsigned long long *out8, int reset)
{
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_ph_stamps), 64) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_ph_stamps), z, 64) != hipSuccess) return -1; }
    return 0;
}
#endif

void pnr_phased_destroy(pnr_phased *h)
{
    if (!h) return;
    phased_free(h);
    if (h->h_cnt) hipHostFree(h->h_cnt);
    if (h->h_xc) hipHostFree(h->h_xc);
    if (h->h_flags) hipHostFree(h->h_flags);
    if (h->h_new) hipHostFree(h->h_new);
    if (h->h_new_s6) hipHostFree(h->h_new_s6);
    hipFree(h->d_new); hipFree(h->d_new_s6);
    if (h->h_ctl) (void)hipHostFree(h->h_ctl);
    (void)hipFree(h->d_ctl);
    for (int g = 0; g < pnr_phased::MAXG; g++) {
        if (h->h_den_idx[g]) (void)hipHostFree(h->h_den_idx[g]);
        if (h->h_den_val[g]) (void)hipHostFree(h->h_den_val[g]);
        (void)hipFree(h->d_den_idx[g]);
        (void)hipFree(h->d_den_val[g]);
    }
    for (int r = 0; r < pnr_phased::RING; r++)
        if (h->ev[r]) (void)hipEventDestroy(h->ev[r]);
    for (int g = 1; g < pnr_phased::MAXG; g++)
        if (h->stg[g]) (void)hipStreamDestroy(h->stg[g]);
    if (h->st_den) (void)hipStreamDestroy(h->st_den);
    if (h->ev_start) (void)hipEventDestroy(h->ev_start);
    for (hipEvent_t &e : h->ev_state) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t &e : h->ev_adm) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t &e : h->ev_ctl) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t &e : h->ev_den) if (e) (void)hipEventDestroy(e);
    delete h;
}

// Work-groups per trace for the sampling launch.  One work-group per CU is resident (the cube fills the LDS); all
// work-groups of a trace pull items from its counter, so what matters is that there are a few times more work-groups
// than CUs (the dispatcher keeps every CU busy until the items run out) without paying the cube staging too often.
// Which form of the ordered sums a launch of `active` traces takes (option "sums_deep": -1 automatic, 0 / 1 forced; the results
// are the same): the four-buffer form when the launch has the GPU to itself, or is so small that a chain's latency is all it costs
static bool sums_deep(const pnr_ctx *c, int active, int ngroups)
{
    if (c->opt.sums_deep >= 0) return c->opt.sums_deep != 0;
    return ngroups <= 1 || active <= c->opt.sums_deep_max;
}

// slots of ph_predict's first-occurrence table: a power of two >= 2 np (its LDS: hashes, representatives, poses, the table)
static int ph_tbl(int np)
{
    int t = 64;
    while (t < 2 * np) t <<= 1;
    return t;
}

// dynamic LDS of ph_predict: representatives (+ hashes, poses and the table when it searches for duplicates)
static size_t ph_predict_lds(int np, int dedup) { return dedup ? (size_t)(np * 8 + 2 * ph_tbl(np)) * 4 : (size_t)np * 8; }

static int pick_nsplit(int active, int ncu, int max_split, int x10 /* work-groups per CU x 10 */)
{
    if (x10 <= 0) x10 = 40;
    int ns = (int)(((long long)x10 * ncu / 10 + active - 1) / active);
    return ns < 1 ? 1 : (ns > max_split ? max_split : ns);
}

// the sampling launch of a step: the traces' cubes fetched once into their compact copies (ph_cube), then the sampling work-groups
static void launch_cube(hipStream_t st, const Vol &V, const PhState &P, int active, int lp)
{
    hipLaunchKernelGGL(ph_cube, dim3((unsigned)(active * PH_CUBE_SPLIT)), dim3(PH_CUBE_THREADS), 0, st, V, P, lp, active);
}
static void launch_sample(hipStream_t st, const Vol &V, const Tab &T, const TabX &X, const PhState &P, int np, int ni, int it, int lp, int active, int nsplit, size_t cube_bytes)
{
    const dim3 grid((unsigned)(active * nsplit)), blk(PH_THREADS);
    if (P.cubes) {
        if (V.l == 1) hipLaunchKernelGGL((ph_sample<PH_CS, true, true>), grid, blk, cube_bytes, st, V, T, X, P, np, ni, it, lp, active);
        else hipLaunchKernelGGL((ph_sample<PH_CS, false, true>), grid, blk, cube_bytes, st, V, T, X, P, np, ni, it, lp, active);
    } else {
        if (V.l == 1) hipLaunchKernelGGL((ph_sample<PH_CS, true, false>), grid, blk, cube_bytes, st, V, T, X, P, np, ni, it, lp, active);
        else hipLaunchKernelGGL((ph_sample<PH_CS, false, false>), grid, blk, cube_bytes, st, V, T, X, P, np, ni, it, lp, active);
    }
}

struct PhEnv {
    Vol V; Tab T; TabX X; PhState P;
    int np, ni, S, np_pad, ng, ncu, max_split, dbg_iters;
    size_t cube_bytes, upd_lds;
    long long trace_floats;
    int64_t NT; // trace slots available (<= the number asked for: bounded by the stash budget)
    pnr_phased *h;
};

// device state for up to `want` concurrent traces (fewer if their sample stash exceeds the budget: PNR_STASH_GB,
// default 64 GB, or half of the free HBM) and everything the four kernels take as arguments
// the sample stash for `traces` list positions (all trace groups together); grows only
static int ensure_stash(pnr_ctx *c, pnr_phased *h, int64_t traces, long long trace_floats)
{
    if (h->cap_stash >= traces && h->P.stash) return PNR_OK;
    PNR_HIP(hipDeviceSynchronize());
    (void)hipFree(h->P.stash);
    h->P.stash = nullptr; h->cap_stash = 0;
    PNR_HIP(hipMalloc(&h->P.stash, (size_t)traces * trace_floats * 4));
    (void)hipFree(h->P.cubes);
    h->P.cubes = nullptr;
    PNR_HIP(hipMalloc(&h->P.cubes, (size_t)traces * PH_CS * PH_PLANE)); // the compact cubes (ph_cube), one per list position as well
    h->cap_stash = traces;
    // a stale stash value is only ever read for a chain whose result is discarded, but keep it finite
    PNR_HIP(hipMemsetAsync(h->P.stash, 0, (size_t)traces * trace_floats * 4, c->stream));
    PNR_HIP(hipStreamSynchronize(c->stream)); // (the trace groups' streams do not wait for this one)
    return PNR_OK;
}

// stash_want: list positions the stash must hold at once (<= 0: one per trace slot -- a one-shot batch starts all its traces together)
static int phased_env(pnr_ctx *c, int64_t want, int dbg_iters, bool xfilt, bool idxres, bool neff, PhEnv &E, int64_t stash_want = 0)
{
    int rc = make_vol(c, E.V);
    if (rc) return rc;
    make_tab(c, E.T);
    const int np = c->prm.np, ni = c->prm.ni, S = E.T.nsig;
    const int np_pad = (np + 1 + 63) / 64 * 64; // slot np = the pending centroid
    if (dbg_iters > ni) dbg_iters = ni;
    if (dbg_iters < 0) dbg_iters = 0;
    PNR_REQUIRE(want >= 1 && want < (1LL << 30), PNR_E_ARG, "bad number of traces");
    for (int s = 0; s < S; s++)
        PNR_REQUIRE(c->tab.grid[4 * s] <= 64 && c->tab.grid[4 * s + 1] <= 64 && c->tab.grid[4 * s + 2] <= 64, PNR_E_ARG,
                    "template grid axis longer than a wavefront");
    if (!c->phased) c->phased = new pnr_phased();
    pnr_phased *h = c->phased;
    if (!h->h_cnt) {
        PNR_HIP(hipHostMalloc(&h->h_cnt, sizeof(int) * pnr_phased::RING));
        for (int r = 0; r < pnr_phased::RING; r++) PNR_HIP(hipEventCreateWithFlags(&h->ev[r], hipEventDisableTiming));
        for (int g = 1; g < pnr_phased::MAXG; g++) PNR_HIP(hipStreamCreateWithFlags(&h->stg[g], hipStreamNonBlocking));
        PNR_HIP(hipStreamCreateWithFlags(&h->st_den, hipStreamNonBlocking));
        PNR_HIP(hipEventCreateWithFlags(&h->ev_start, hipEventDisableTiming));
        for (hipEvent_t &e : h->ev_state) PNR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (hipEvent_t &e : h->ev_adm) PNR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (hipEvent_t &e : h->ev_ctl) PNR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (hipEvent_t &e : h->ev_den) PNR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // stash rows of a trace hold at most np + 1 chains: full groups of 64 + the last group's stride (16 / 32 / 64)
    const int ngf = (np + 1) / 64, rem = (np + 1) - 64 * ngf, R = rem == 0 ? 0 : (rem > 32 ? 64 : (rem > 16 ? 32 : 16)), W = 64 * ngf + R;
    const long long Mtot = E.T.Mtot, trace_floats = Mtot * W;
    size_t free_b = 0, total_b = 0;
    PNR_HIP(hipMemGetInfo(&free_b, &total_b));
    size_t budget = (size_t)std::max<int64_t>(1, c->opt.stash_mb) << 20; // option "stash_mb" (tests: force several waves / a narrow window)
    const size_t have = (size_t)h->cap_stash * (size_t)h->trace_floats * 4; // our own stash counts as free
    budget = std::min(budget, (free_b + have) / 2);
    const int64_t nt_max = (int64_t)(budget / ((size_t)trace_floats * 4));
    PNR_REQUIRE(nt_max >= 1, PNR_E_HIP, "not enough device memory for one trace's sample stash (%lld B)", trace_floats * 4);
    const int64_t nwaves = (want + nt_max - 1) / nt_max;
    const int64_t NT = (want + nwaves - 1) / nwaves;
    const int64_t need_dbg = NT * dbg_iters;
    if (h->cap_traces < NT || h->np != np || h->np_pad != np_pad || h->S != S || h->ni != ni || h->trace_floats != trace_floats ||
        h->cap_dbg < need_dbg || (xfilt && dbg_iters && !h->O.xfilt) || (idxres && dbg_iters && !h->O.idxres) ||
        (neff && dbg_iters && !h->O.neff)) {
        PNR_HIP(hipDeviceSynchronize());
        phased_free(h);
        const int64_t cap = NT;
        PNR_HIP(hipMalloc(&h->P.part, (size_t)cap * 2 * np * PSTRIDE * 4));
        PNR_HIP(hipMalloc(&h->P.prior, (size_t)cap * np * 4));
        PNR_HIP(hipMalloc(&h->P.idxres, (size_t)cap * np * 4));
        PNR_HIP(hipMalloc(&h->P.corr, (size_t)cap * S * np_pad * 4));
        PNR_HIP(hipMalloc(&h->P.xcs, (size_t)cap * 16 * 4));
        // the counters of the step lists sit in front of the flags in ONE buffer: the streaming tracer's poll copies both with one copy
        PNR_HIP(hipMalloc(&h->P.cnt, ((size_t)cap * FL_N + 2 * pnr_phased::MAXG) * 4));
        h->P.flags = h->P.cnt + 2 * pnr_phased::MAXG;
        PNR_HIP(hipMalloc(&h->P.list, (size_t)cap * 2 * 4 * pnr_phased::MAXG)); // one pair of lists per trace group of the streaming tracer
        PNR_HIP(hipMalloc(&h->P.ctr, (size_t)cap * 4));
        PNR_HIP(hipMalloc(&h->P.uidx, (size_t)cap * np_pad * 4));
        PNR_HIP(hipMalloc(&h->P.cmap, (size_t)cap * np_pad * 4));
        PNR_HIP(hipMalloc(&h->d_s6, (size_t)cap * 24));
        PNR_HIP(hipMalloc(&h->O.T, (size_t)cap * 4));
        PNR_HIP(hipMalloc(&h->O.stop, (size_t)cap * 4));
        PNR_HIP(hipMalloc(&h->O.xc, (size_t)cap * ni * 32));
        const size_t dbg_cap = (size_t)cap * dbg_iters;
        if (dbg_iters && xfilt) PNR_HIP(hipMalloc(&h->O.xfilt, dbg_cap * np * PSTRIDE * 4));
        if (dbg_iters && idxres) PNR_HIP(hipMalloc(&h->O.idxres, dbg_cap * np * 4));
        if (dbg_iters && neff) PNR_HIP(hipMalloc(&h->O.neff, dbg_cap * 4));
        h->cap_traces = cap; h->cap_dbg = (int64_t)dbg_cap;
        h->np = np; h->np_pad = np_pad; h->S = S; h->ni = ni; h->trace_floats = trace_floats;
        PNR_HIP(hipMemsetAsync(h->P.part, 0, (size_t)cap * 2 * np * PSTRIDE * 4, c->stream));
    }
    { const int rcs = ensure_stash(c, h, stash_want > 0 ? std::min<int64_t>(stash_want, NT) : NT, trace_floats); if (rcs) return rcs; }
    E.P = h->P;
    E.P.stash_base = 0;
    E.P.trace_floats = trace_floats;
    E.P.cap = (int)h->cap_traces;
    E.P.W = W;
    E.P.np_pad = np_pad;
    E.P.dedup = np <= 1024 ? 1 : 0; // (beyond that the poses and the table of the duplicate search no longer fit 64 KB of LDS)
    E.X.grid = (const Grid *)c->d_grid; E.X.axes = c->d_axes; E.X.axes_off = c->d_axes_off; E.X.wd = c->d_wd;
    E.X.ext_v = c->tab.ext_v; E.X.ext_uw = c->tab.ext_uw;
    for (int s2 = 0; s2 < 8; s2++) { E.X.ext_vs[s2] = s2 < S ? c->tab.ext_vs[s2] : 0.f; E.X.ext_uws[s2] = s2 < S ? c->tab.ext_uws[s2] : 0.f; }
    E.X.stash = nullptr; E.X.slot_busy = nullptr; E.X.nslots = 0; E.X.slot_floats = 0; E.X.wave_floats = 0;
    hipDeviceProp_t prop;
    PNR_HIP(hipGetDeviceProperties(&prop, c->device));
    E.ncu = prop.multiProcessorCount;
    E.np = np; E.ni = ni; E.S = S; E.np_pad = np_pad; E.ng = ngf + (rem > 0 ? 1 : 0); E.dbg_iters = dbg_iters;
    E.cube_bytes = (size_t)PH_CS * PH_PLANE;
    E.upd_lds = ((size_t)np * PSTRIDE + 4 * (size_t)np + (size_t)S * np_pad + 2 + 2 + 8) * 4;
    E.trace_floats = trace_floats;
    E.NT = NT;
    E.h = h;
    E.max_split = std::max(1, c->opt.max_split);
    PNR_REQUIRE(E.upd_lds <= 160 * 1024, PNR_E_ARG, "np=%d with %d scales needs %zu B of LDS in the update step (limit 160 KB)", np, S, E.upd_lds);
    PNR_HIP(hipFuncSetAttribute((const void *)ph_update, hipFuncAttributeMaxDynamicSharedMemorySize, (int)E.upd_lds));
    PNR_HIP(hipFuncSetAttribute((const void *)ph_sample<PH_CS, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)E.cube_bytes));
    PNR_HIP(hipFuncSetAttribute((const void *)ph_sample<PH_CS, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)E.cube_bytes));
    PNR_HIP(hipFuncSetAttribute((const void *)ph_sample<PH_CS, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)E.cube_bytes));
    PNR_HIP(hipFuncSetAttribute((const void *)ph_sample<PH_CS, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)E.cube_bytes));
    if (!c->opt.cube_copy) E.P.cubes = nullptr; // option "cube_copy" = 0: every sampling work-group stages its cube from the image itself
    return PNR_OK;
}

int pnr_trace_run_phased(pnr_ctx *c, const pnr_seed *seeds, int64_t n, int32_t *T_out, int32_t *stop_out, pnr_xest *xc, int dbg_iters,
                         float *xfilt, int32_t *idxres, float *neff, int use_density)
{
    if (n == 0) return PNR_OK;
    PhEnv E;
    const int64_t ntr_all = 2 * n;
    int rc = phased_env(c, ntr_all, dbg_iters, xfilt != nullptr, idxres != nullptr, neff != nullptr, E);
    if (rc) return rc;
    const Vol &V = E.V; const Tab &T = E.T; const TabX &X = E.X; const PhState &P = E.P;
    pnr_phased *h = E.h;
    const int np = E.np, ni = E.ni, S = E.S, np_pad = E.np_pad, ng = E.ng, ncu = E.ncu, max_split = E.max_split;
    const int64_t NT = E.NT;
    const size_t cube_bytes = E.cube_bytes, upd_lds = E.upd_lds;
    dbg_iters = E.dbg_iters;
    constexpr int CS = PH_CS;
    hipStream_t st = c->stream;

    constexpr int LAG = 3, RING = pnr_phased::RING; // the host runs at most LAG iterations ahead of the active-trace counter
    std::vector<float> s6;
    std::vector<int> flags, list0;
    for (int64_t t0 = 0; t0 < ntr_all; t0 += NT) {
        const int nt = (int)std::min<int64_t>(NT, ntr_all - t0);
        s6.resize((size_t)nt * 6);
        for (int j = 0; j < nt; j++) {
            const pnr_seed &sd = seeds[(t0 + j) / 2];
            const bool neg = ((t0 + j) & 1) != 0; // odd trace: trackNeg (tracker.cpp:819-823)
            float *a = &s6[(size_t)j * 6];
            a[0] = sd.x; a[1] = sd.y; a[2] = sd.z;
            a[3] = neg ? -sd.vx : sd.vx; a[4] = neg ? -sd.vy : sd.vy; a[5] = neg ? -sd.vz : sd.vz;
        }
        flags.assign((size_t)nt * FL_N, 0);
        list0.resize((size_t)nt);
        for (int j = 0; j < nt; j++) { flags[(size_t)j * FL_N + FL_T] = ni; list0[(size_t)j] = j; }
        const int cnt0[2] = {nt, 0};
        TraceOut O = h->O;
        O.dbg_iters = dbg_iters;
        if (!xfilt || !dbg_iters) O.xfilt = nullptr;
        if (!idxres || !dbg_iters) O.idxres = nullptr;
        if (!neff || !dbg_iters) O.neff = nullptr;
        PNR_HIP(hipMemcpyAsync(h->d_s6, s6.data(), s6.size() * 4, hipMemcpyHostToDevice, st));
        PNR_HIP(hipMemcpyAsync(P.flags, flags.data(), flags.size() * 4, hipMemcpyHostToDevice, st));
        PNR_HIP(hipMemcpyAsync(P.list, list0.data(), list0.size() * 4, hipMemcpyHostToDevice, st));
        PNR_HIP(hipMemcpyAsync(P.cnt, cnt0, sizeof(cnt0), hipMemcpyHostToDevice, st));
        PNR_HIP(hipMemsetAsync(h->O.xc, 0, (size_t)nt * ni * 32, st));
        PNR_HIP(hipMemsetAsync(P.xcs, 0, (size_t)nt * 16 * 4, st));
        if (O.idxres) PNR_HIP(hipMemsetAsync(O.idxres, 0xff, (size_t)nt * dbg_iters * np * 4, st));
        PNR_HIP(hipStreamSynchronize(st)); // the uploads above come from pageable host vectors reused below
        int active = nt; // upper bound of the traces still running
        for (int it = 0; it <= ni; it++) {
            if (it >= LAG) { // the counter of iteration it - LAG has landed (no pipeline drain: the host stays LAG steps ahead)
                PNR_HIP(hipEventSynchronize(h->ev[(it - LAG) % RING]));
                active = h->h_cnt[(it - LAG) % RING];
                if (active <= 0) break;
            }
            const int nsplit = pick_nsplit(active, ncu, max_split, c->opt.split_x10);
            c->tic(st);
            hipLaunchKernelGGL(ph_predict, dim3(active), dim3(256), ph_predict_lds(np, P.dedup), st, T, X, P, (const float *)h->d_s6, V, np, ni, it, it & 1, CS, ph_tbl(np));
            c->toc("smc_predict", 1, st);
            if (P.cubes) {
                c->tic(st);
                launch_cube(st, V, P, active, it & 1);
                c->toc("smc_cube", 1, st);
            }
            c->tic(st);
            launch_sample(st, V, T, X, P, np, ni, it, it & 1, active, nsplit, cube_bytes);
            c->toc("smc", 1, st);
            c->tic(st);
            if (sums_deep(c, active, 1))
                hipLaunchKernelGGL(ph_sums<true>, dim3((unsigned)(active * S * ng)), dim3(64), 0, st, T, X, P, np, np_pad, ni, it, it & 1, ng);
            else
                hipLaunchKernelGGL(ph_sums<false>, dim3((unsigned)(active * S * ng)), dim3(64), 0, st, T, X, P, np, np_pad, ni, it, it & 1, ng);
            c->toc("smc_sums", 1, st);
            c->tic(st);
            hipLaunchKernelGGL(ph_update, dim3(active), dim3(256), upd_lds, st, V, T, P, np, np_pad, ni, it, it & 1, c->prm.Kc, c->prm.znccth,
                               c->prm.neff_ratio, use_density ? c->d_den : nullptr, c->prm.nodepervol, O);
            c->toc("smc_update", 1, st);
            PNR_HIP(hipMemcpyAsync(&h->h_cnt[it % RING], P.cnt + ((it + 1) & 1), 4, hipMemcpyDeviceToHost, st));
            PNR_HIP(hipEventRecord(h->ev[it % RING], st));
        }
        PNR_HIP(hipGetLastError());
        PNR_HIP(hipMemcpyAsync(T_out + t0, h->O.T, (size_t)nt * 4, hipMemcpyDeviceToHost, st));
        PNR_HIP(hipMemcpyAsync(stop_out + t0, h->O.stop, (size_t)nt * 4, hipMemcpyDeviceToHost, st));
        PNR_HIP(hipMemcpyAsync(xc + t0 * ni, h->O.xc, (size_t)nt * ni * 32, hipMemcpyDeviceToHost, st));
        if (O.xfilt) PNR_HIP(hipMemcpyAsync(xfilt + t0 * dbg_iters * np * PSTRIDE, O.xfilt, (size_t)nt * dbg_iters * np * PSTRIDE * 4, hipMemcpyDeviceToHost, st));
        if (O.idxres) PNR_HIP(hipMemcpyAsync(idxres + t0 * dbg_iters * np, O.idxres, (size_t)nt * dbg_iters * np * 4, hipMemcpyDeviceToHost, st));
        if (O.neff) PNR_HIP(hipMemcpyAsync(neff + t0 * dbg_iters, O.neff, (size_t)nt * dbg_iters * 4, hipMemcpyDeviceToHost, st));
        PNR_HIP(hipStreamSynchronize(st));
    }
    return PNR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Streaming trace + replay (the production form of the trace loop, Advantra_plugin.cpp:2658-2710): the scheduler is
// stream_sched.h (host logic, shared with the sharded multi-GPU form); this is its engine -- a window of trace slots stepped
// by the four phase kernels above, every trace at its own iteration, the per-iteration records written straight into pinned
// host memory by ph_update.
//
// The window can be shared by G groups of traces (option "groups", 1..4) that step independently, each on its own stream with
// its own active list: while the host collects, replays and refills one group, the other groups' steps keep the GPU busy, and
// launches with few traces (a chain's ordered sums take 0.25 ms however few there are) overlap.  Results do not depend on the
// grouping: a trace is a function of its seed and of the replayed map.  Two groups (the default) take 7 % off the tracing of the
// bench step -- with 22 instead of 40 sampling work-groups per CU and launch, because a launch now shares the CUs; a third group
// gains nothing (the streams share two hardware queues).  With several groups the launches of different groups overlap, so a
// kernel's duration (HIP events, rocprofv3) includes the time it shares the CUs; option "groups" = 1 gives the isolated figure.
// ---------------------------------------------------------------------------------------------------------
namespace {

struct PhasedEngine final : pnr::StreamEngine {
    pnr_ctx *c;
    PhEnv E;
    pnr_phased *h = nullptr;
    int NT = 0;
    TraceOut O{};
    struct Grp {
        PhState P; hipStream_t st; int lp = 0;
        int *h_snap, *h_flags, *h_cnt, *h_new, *d_new; float *h_new_s6, *d_new_s6;
        // what this turn's control() / density_update() / admit() asked for: sent as one dispatch by flush()
        bool p_ctl = false; int p_np = 0, p_nr = 0, p_m = 0; size_t p_nt = 0;
        long long poll_no = 0;
        hipEvent_t ev_state = nullptr; // the copies of flags and count of the last launch have landed
        int running = 0; // traces of the group's last launch still running (0 once its poll has been collected empty)
    };
    Grp grp[pnr_phased::MAXG];
    std::string msg;
    int ngroups = 1;    // trace groups the scheduler steps (what a launch shares the GPU with)
    int gcap = 0;       // list positions of the sample stash per trace group (grown when a launch holds more traces: grow_stash)
    // (with several trace groups a launch shares the CUs with the other groups' launches: fewer, fatter sampling work-groups -- 22
    // instead of 40 per 10 CUs -- and the form of ph_sums that fits beside them; decided per launch, see launch())

    explicit PhasedEngine(pnr_ctx *ctx) : c(ctx) {}
    const char *error() const override { return msg.c_str(); }
    int hip_fail(hipError_t e, const char *what)
    {
        msg = std::string(what) + " failed: " + hipGetErrorString(e);
        return PNR_E_HIP;
    }
#define PE_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_fail(e_, #call); } while (0)

    int init(int64_t window)
    {
        // the stash holds the traces of a LAUNCH (a trace's position in its group's list), not the window: 192 positions per group to
        // start with (the automatic target keeps 60 - 120 traces running per group), grown on demand
        int rc = phased_env(c, window, 0, false, false, false, E, (int64_t)ngroups * 192);
        if (rc) { msg = pnr_last_error(); return rc; }
        h = E.h;
        gcap = (int)std::max<int64_t>(1, h->cap_stash / ngroups);
        NT = (int)(E.NT - (E.NT & 1)); // slots come in pairs (the two directions of a seed)
        if (NT < 2) { msg = "not enough device memory for two trace slots"; return PNR_E_HIP; }
        const int ni = E.ni;
        if (h->stream_cap < NT || h->stream_ni != ni) {
            PE_HIP(hipDeviceSynchronize());
            if (h->h_xc) hipHostFree(h->h_xc);
            if (h->h_flags) hipHostFree(h->h_flags);
            if (h->h_new) hipHostFree(h->h_new);
            if (h->h_new_s6) hipHostFree(h->h_new_s6);
            hipFree(h->d_new); hipFree(h->d_new_s6);
            if (h->h_ctl) (void)hipHostFree(h->h_ctl);
            (void)hipFree(h->d_ctl);
            h->h_ctl = nullptr; h->d_ctl = nullptr;
            h->h_xc = nullptr; h->h_flags = nullptr; h->h_new = nullptr; h->h_new_s6 = nullptr; h->d_new = nullptr; h->d_new_s6 = nullptr;
            h->stream_cap = 0;
            PE_HIP(hipHostMalloc(&h->h_xc, (size_t)NT * ni * sizeof(pnr_xest)));
            constexpr int MG = pnr_phased::MAXG; // one set per trace group
            PE_HIP(hipHostMalloc(&h->h_flags, ((size_t)NT * FL_N + 2 * MG) * 4 * MG)); // per group: [2 MG counters | NT x FL_N flags]
            PE_HIP(hipHostMalloc(&h->h_new, (size_t)NT * 4 * MG));
            PE_HIP(hipHostMalloc(&h->h_new_s6, (size_t)NT * 24 * MG));
            PE_HIP(hipMalloc(&h->d_new, (size_t)NT * 4 * MG));
            PE_HIP(hipMalloc(&h->d_new_s6, (size_t)NT * 24 * MG));
            PE_HIP(hipHostMalloc(&h->h_ctl, (size_t)NT * 4 * 2 * MG));
            PE_HIP(hipMalloc(&h->d_ctl, (size_t)NT * 4 * 2 * MG));
            h->stream_cap = NT;
            h->stream_ni = ni;
        }
        O = h->O;
        O.xc = (float *)h->h_xc; // the per-iteration records go straight to pinned host memory (32 B per trace and iteration)
        O.dbg_iters = 0; O.xfilt = nullptr; O.idxres = nullptr; O.neff = nullptr;
        PE_HIP(hipMemsetAsync(E.P.cnt, 0, 2 * 4 * pnr_phased::MAXG, c->stream));
        PE_HIP(hipEventRecord(h->ev_start, c->stream)); // everything queued so far (volume, density map) precedes the other streams
        for (int g = 1; g < pnr_phased::MAXG; g++) PE_HIP(hipStreamWaitEvent(h->stg[g], h->ev_start, 0));
        PE_HIP(hipStreamWaitEvent(h->st_den, h->ev_start, 0));
        for (int g = 0; g < pnr_phased::MAXG; g++) {
            Grp &q = grp[g];
            q.P = E.P;
            q.P.stash_base = (g < ngroups ? g : 0) * gcap;
            q.P.list = E.P.list + (size_t)g * 2 * E.P.cap;
            q.P.cnt = E.P.cnt + 2 * g;
            q.st = g == 0 ? c->stream : h->stg[g];
            q.ev_state = h->ev_state[g];
            q.h_snap = h->h_flags + (size_t)g * ((size_t)h->stream_cap * FL_N + 2 * pnr_phased::MAXG);
            q.h_flags = q.h_snap + 2 * pnr_phased::MAXG; q.h_cnt = q.h_snap + 2 * g;
            q.h_new = h->h_new + (size_t)g * h->stream_cap; q.d_new = h->d_new + (size_t)g * h->stream_cap;
            q.h_new_s6 = h->h_new_s6 + (size_t)g * h->stream_cap * 6; q.d_new_s6 = h->d_new_s6 + (size_t)g * h->stream_cap * 6;
        }
        return PNR_OK;
    }
    int slots() const override { return NT; }
    int max_groups() const override { return pnr_phased::MAXG; }
    int admit(int g, const int *slots, const float *s6, int m) override
    {
        Grp &q = grp[g];
        PE_HIP(hipEventSynchronize(h->ev_ctl[g]));      // pinned staging of this group: its previous admission has been consumed (one event per ph_poll: flush())
        std::memcpy(q.h_new, slots, (size_t)m * 4);     // (normally long ago: a wait() lies between two admissions unless the group
        std::memcpy(q.h_new_s6, s6, (size_t)m * 24);    // had nothing to step)
        q.p_m = m; // (flush(): the kernel reads the pinned staging itself -- a few hundred bytes over the bus, no copies in front of it)
        return PNR_OK;
    }
    // a launch with more traces than a group's stash region holds: every group gets a larger region (rare: the stash starts with room
    // for the launches the automatic target produces)
    int grow_stash(int need)
    {
        const int ncap = (int)std::min<int64_t>(E.NT, std::max(need, gcap + gcap / 2));
        if (ncap < need) { msg = "a launch holds more traces than the window has slots"; return PNR_E_STATE; }
        const int rc = ensure_stash(c, h, (int64_t)ngroups * ncap, E.trace_floats);
        if (rc) { msg = pnr_last_error(); return rc; }
        gcap = ncap;
        E.P.stash = h->P.stash;
        if (E.P.cubes) E.P.cubes = h->P.cubes;
        for (int k = 0; k < pnr_phased::MAXG; k++) { grp[k].P.stash = h->P.stash; grp[k].P.cubes = E.P.cubes; grp[k].P.stash_base = (k < ngroups ? k : 0) * gcap; }
        return PNR_OK;
    }
    int launch(int g, int active, int poll, int lag) override
    {
        if (g >= ngroups) { msg = "trace group beyond the stash regions"; return PNR_E_STATE; }
        if (active > gcap) { const int rc = grow_stash(active); if (rc) return rc; }
        { const int rc = flush(g, /*followed_by_steps*/ true); if (rc) return rc; }
        Grp &q = grp[g];
        hipStream_t st = q.st;
        const PhState &P = q.P;
        const int np = E.np, ni = E.ni, S = E.S, np_pad = E.np_pad, ng = E.ng;
        // does this launch share the GPU with another group's steps?  (With the scheduler's `concentrate` the other groups run out.)
        q.running = active;
        int sharing = 1;
        for (int k = 0; k < ngroups; k++) sharing += (k != g && grp[k].running > 0) ? 1 : 0;
        const int x10 = c->opt.split_x10 > 0 ? c->opt.split_x10 : (sharing > 1 ? 22 : 40);
        const int pw = std::max(1, c->opt.profile_every);
        const bool prof = (q.poll_no++ % pw) == 0; // the kernel timers (pnr_set_profiling) look at every pw-th poll of the group
        for (int k = 0; k < poll; k++) { // `poll` SMC steps over the group's active list (every trace at its own iteration)
            const int lp = q.lp;
            const int nsplit = pick_nsplit(active, E.ncu, E.max_split, x10);
            if (prof) c->tic(st, k > 0); // (the first step of a poll follows the admission copies: its own opening event)
            hipLaunchKernelGGL(ph_predict, dim3(active), dim3(256), ph_predict_lds(np, P.dedup), st, E.T, E.X, P, (const float *)h->d_s6, E.V, np, ni, -1, lp, PH_CS, ph_tbl(np));
            if (prof) c->toc("smc_predict", 1, st, pw);
            if (P.cubes) {
                if (prof) c->tic(st, true);
                launch_cube(st, E.V, P, active, lp);
                if (prof) c->toc("smc_cube", 1, st, pw);
            }
            if (prof) c->tic(st, true);
            launch_sample(st, E.V, E.T, E.X, P, np, ni, -1, lp, active, nsplit, E.cube_bytes);
            if (prof) c->toc("smc", 1, st, pw);
            if (prof) c->tic(st, true);
            if (sums_deep(c, active, sharing))
                hipLaunchKernelGGL(ph_sums<true>, dim3((unsigned)(active * S * ng)), dim3(64), 0, st, E.T, E.X, P, np, np_pad, ni, -1, lp, ng);
            else
                hipLaunchKernelGGL(ph_sums<false>, dim3((unsigned)(active * S * ng)), dim3(64), 0, st, E.T, E.X, P, np, np_pad, ni, -1, lp, ng);
            if (prof) c->toc("smc_sums", 1, st, pw);
            if (prof) c->tic(st, true);
            hipLaunchKernelGGL(ph_update, dim3(active), dim3(256), E.upd_lds, st, E.V, E.T, P, np, np_pad, ni, -1, lp, c->prm.Kc, c->prm.znccth,
                               c->prm.neff_ratio, c->d_den, c->prm.nodepervol, O);
            if (prof) c->toc("smc_update", 1, st, pw);
            q.lp ^= 1;
            if (k == poll - 1 - lag) { // what wait() hands to the host: the state behind this step (the last `lag` steps run on meanwhile)
                // one copy: the counters of all step lists and the flags of all slots (every copy is a dispatch of its own in the stream,
                // ~10 us each on the chain of this group's steps)
                // (a kernel of our own that stores to the pinned snapshot, not hipMemcpyAsync: the runtime's copy is a blit dispatch between
                // two barrier packets -- 12 us before it and 6.5 us behind it on the chain of this group's steps, per poll; this is 2 us)
                {
                    const int n4 = (int)(((size_t)NT * FL_N + 2 * pnr_phased::MAXG) / 4);
                    hipLaunchKernelGGL(ph_snapshot, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const int4 *)E.P.cnt, (int4 *)q.h_snap, n4);
                }
                q.h_cnt = q.h_snap + 2 * g + q.lp; // the list of the step that follows
                PE_HIP(hipEventRecord(q.ev_state, st));
            }
        }
        return PNR_OK;
    }
    void idle(int g) override { grp[g].running = 0; }
    int wait(int g, int *active) override
    {
        PE_HIP(hipEventSynchronize(grp[g].ev_state)); // (the steps queued behind the snapshot may still be running; polling the event instead of sleeping on it changes nothing: A/B)
        PE_HIP(hipGetLastError());
        *active = grp[g].h_cnt[0];
        grp[g].running = *active;
        return PNR_OK;
    }
    bool finished(int g, int slot, int *T) const override
    {
        const int *fl = grp[g].h_flags + (size_t)slot * FL_N;
        *T = fl[FL_T];
        return fl[FL_DONE] != 0;
    }
    const pnr_xest *rows(int slot) const override { return h->h_xc + (size_t)slot * E.ni; }
    int progress(int g, int slot) const override
    {
        // ph_update of iteration `it` writes the estimate of iteration it - 1 (its corr is known one step late) and leaves FL_IT = it + 1
        return grp[g].h_flags[(size_t)slot * FL_N + FL_IT] - 1;
    }
    int control(int g, const int *pause, int np_, const int *resume, int nr) override
    {
        Grp &q = grp[g];
        int *hc = h->h_ctl + (size_t)g * 2 * h->stream_cap, *dc = h->d_ctl + (size_t)g * 2 * h->stream_cap;
        PE_HIP(hipEventSynchronize(h->ev_ctl[g]));                  // pinned staging of this group: its previous lists have been consumed
        if (np_ > 0) std::memcpy(hc, pause, (size_t)np_ * 4);
        if (nr > 0) std::memcpy(hc + h->stream_cap, resume, (size_t)nr * 4);
        (void)dc;
        q.p_ctl = true; q.p_np = np_; q.p_nr = nr;
        return PNR_OK;
    }
    // one dispatch for what control() / density_update() / admit() of this turn asked for, in that order
    int flush(int g, bool followed_by_steps = false)
    {
        Grp &q = grp[g];
        if (!q.p_ctl && q.p_m == 0 && q.p_nt == 0) return PNR_OK;
        int *hc = h->h_ctl + (size_t)g * 2 * h->stream_cap;
        hipLaunchKernelGGL(ph_poll, dim3((unsigned)(1 + (q.p_nt + 255) / 256)), dim3(256), 0, q.st, q.P, h->d_s6, q.p_ctl ? 1 : 0, (const int *)hc, q.p_np,
                           (const int *)(hc + h->stream_cap), q.p_nr, (const int *)q.h_new, (const float *)q.h_new_s6, q.p_m, q.lp, E.ni, c->d_den,
                           (const i64 *)h->h_den_idx[g], (const unsigned char *)h->h_den_val[g], (int)q.p_nt);
        PE_HIP(hipGetLastError());
        // The three pinned staging areas the dispatch reads (control lists, admissions, density cells) are rewritten in the group's next
        // turn at the earliest.  Where steps follow in this turn, that turn begins with wait(g) on a state snapshot queued BEHIND this
        // dispatch: it has run by then, and no event is needed (every hipEventRecord is a packet of its own, ~5 us on the chain of the
        // group's steps; there were up to three per poll).  Only a dispatch with nothing behind it (end_turn / settle of a group with
        // nothing to step) is followed by ONE event, which the three waits share.
        if (!followed_by_steps) PE_HIP(hipEventRecord(h->ev_ctl[g], q.st));
        q.p_ctl = false; q.p_np = q.p_nr = q.p_m = 0; q.p_nt = 0;
        return PNR_OK;
    }
    int end_turn(int g) override { return flush(g); }
    int settle(int g) override
    {
        { const int rc = flush(g); if (rc) return rc; }
        PE_HIP(hipStreamSynchronize(grp[g].st));
        return PNR_OK;
    }
    int density_update(const pnr::Replayer &r, int g) override
    {
        // Queued on the group's own stream, in front of its next steps, from staging of its own -- and nothing waits for it.  (Until
        // round 3 the update ran on a stream of its own and the host waited for it: that stream shares a hardware queue with a trace
        // group's stream, so the wait lasted until the OTHER group's whole poll had drained -- rocprofv3's timeline showed the two
        // groups taking turns instead of overlapping.)
        const size_t nt = r.touched.size();
        if (nt == 0) return PNR_OK;
        Grp &q = grp[g];
        if (h->den_cap[g] < nt) {
            PE_HIP(hipStreamSynchronize(q.st)); // (its last scatter may still read the old staging)
            if (h->h_den_idx[g]) (void)hipHostFree(h->h_den_idx[g]);
            if (h->h_den_val[g]) (void)hipHostFree(h->h_den_val[g]);
            (void)hipFree(h->d_den_idx[g]);
            (void)hipFree(h->d_den_val[g]);
            h->h_den_idx[g] = nullptr; h->h_den_val[g] = nullptr; h->d_den_idx[g] = nullptr; h->d_den_val[g] = nullptr; h->den_cap[g] = 0;
            const size_t cap = std::max<size_t>(2 * nt, 1 << 16);
            PE_HIP(hipHostMalloc(&h->h_den_idx[g], cap * 8));
            PE_HIP(hipHostMalloc(&h->h_den_val[g], cap));
            PE_HIP(hipMalloc(&h->d_den_idx[g], cap * 8));
            PE_HIP(hipMalloc(&h->d_den_val[g], cap));
            h->den_cap[g] = cap;
        }
        PE_HIP(hipEventSynchronize(h->ev_ctl[g])); // the previous update's cells have been read from the pinned staging (one event per ph_poll: flush())
        for (size_t i = 0; i < nt; i++) {
            h->h_den_idx[g][i] = r.touched[i];
            h->h_den_val[g][i] = (unsigned char)r.den_at(r.touched[i]); // final value: duplicates agree
        }
        q.p_nt = nt;
        return PNR_OK;
    }
    void drain() override
    {
        if (!h) return;
        for (int g = 0; g < pnr_phased::MAXG; g++) (void)hipStreamSynchronize(g == 0 ? c->stream : h->stg[g]);
        (void)hipStreamSynchronize(h->st_den);
    }
#undef PE_HIP
};

} // namespace

int pnr_trace_replay_stream(pnr_ctx *c, const pnr_seed *seeds, int64_t n, pnr::Replayer &r, const pnr::ShardSpec &sh, int64_t *iters_out)
{
    if (iters_out) *iters_out = 0;
    if (n == 0 && sh.world <= 1) return PNR_OK;
    pnr::SchedOptions o;
    o.window = c->opt.window; o.look0 = c->opt.look0; o.look_pct = c->opt.look_pct; o.poll = c->opt.poll; o.groups = c->opt.groups;
    o.timing = c->opt.trace_timing;
    o.tentative = c->opt.tentative;
    o.target = c->opt.target; o.overfill = c->opt.overfill; o.concentrate = c->opt.concentrate; o.lag = c->opt.lag;
    if (o.groups <= 0) o.groups = sh.world > 1 ? 1 : 2; // measured on 2 / 4 / 8 emulated ranks (scripts/emulate_ranks.py): 825 -> 774, 527 -> 499, 396 -> 353 ms
    const int64_t own = sh.world > 1 ? (n - sh.rank + sh.world - 1) / sh.world : n; // seeds of this rank
    if (o.window <= 0) o.window = (sh.world <= 1 && o.tentative) ? 1536 : 768; // automatic: without the pauses a wider window only buys speculation
    int64_t window = std::min<int64_t>(std::max(2, o.window), std::max<int64_t>(2, 2 * own));
    window += window & 1;
    PhasedEngine eng(c);
    eng.ngroups = std::min(std::max(1, o.groups), (int)pnr_phased::MAXG);
    int rc = eng.init(window);
    if (rc) { // (the other ranks are about to enter their first exchange: tell them)
        pnr::set_error("%s", eng.error());
        if (n > 0) pnr::abort_exchange(sh, c->prm.ni);
        return rc;
    }
    pnr::SchedStats st;
    std::string err;
    rc = pnr::run_stream(eng, seeds, n, c->prm.ni, o, sh, r, &st, err);
    if (rc) { pnr::set_error("%s", err.c_str()); return rc; }
    PNR_HIP(hipGetLastError());
    if (iters_out) *iters_out = st.iters;
    return PNR_OK;
}
