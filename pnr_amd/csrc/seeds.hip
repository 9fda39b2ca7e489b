// seeds.hip -- SeedExtractor::extractSeeds (seed.cpp:556-791) split the MI355X way:
//
//   GPU  K7a layer_minmax    per z-layer min/max of J8 (one block per layer, wave reductions)
//        K7b layer_maxima    8-neighbour local-maximum test for every pixel (seed.cpp:589-614),
//                            sort key (value<<32 | pixel) appended to the layer's candidate list
//        K7c gather_dirs     Vx,Vy,Vz at the accepted seed voxels only
//   host analyzeAndMarkMaxima (seed.cpp:643-782): the tolerance flood-fill is sequential and
//        order-dependent inside a layer (each accepted/rejected maximum marks pixels PROCESSED
//        for the ones after it); layers are independent, so they are spread over host threads.
//        This is the same split the trace bookkeeping uses (GPU for the data-parallel part,
//        host replay for the inherently sequential integer part) -- it is not a fallback.
#include "ctx.h"
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <chrono>
#include <cstdlib>
#include <thread>

namespace {
typedef long long i64;

// per-layer extremes of J8: LM_PARTS work-groups per layer, 16 bytes per lane and load, combined with atomics (vmin starts at
// 0x7f7f7f7f, vmax at 0: memsets)
constexpr int LM_PARTS = 8;
__global__ __launch_bounds__(256) void layer_minmax(const unsigned char *__restrict__ J8, i64 wh, int z0,
                                                     int *__restrict__ vmin, int *__restrict__ vmax)
{
    const int zl = blockIdx.x / LM_PARTS, part = blockIdx.x % LM_PARTS;
    const unsigned char *L = J8 + (i64)(z0 + zl) * wh;
    const i64 per = (wh + LM_PARTS - 1) / LM_PARTS;
    const i64 i0 = part * per, i1 = (i0 + per < wh) ? i0 + per : wh;
    int mn = 255, mx = 0;
    // bytes up to the first 16-byte boundary, then uint4 loads, then the rest
    i64 a = i0;
    const i64 mis = (16 - (i64)((uintptr_t)(L + i0) & 15)) & 15;
    const i64 head = (a + mis < i1) ? a + mis : i1;
    for (i64 i = a + threadIdx.x; i < head; i += 256) { const int v = L[i]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
    a = head;
    const i64 nvec = (i1 - a) / 16;
    const uint4 *V = (const uint4 *)(L + a);
    for (i64 k = threadIdx.x; k < nvec; k += 256) {
        const uint4 q = V[k];
        const unsigned int wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int b8 = 0; b8 < 4; b8++) {
                const int v = (int)((wds[c] >> (8 * b8)) & 0xffu);
                mn = v < mn ? v : mn;
                mx = v > mx ? v : mx;
            }
    }
    for (i64 i = a + nvec * 16 + threadIdx.x; i < i1; i += 256) { const int v = L[i]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int x = __shfl_xor(mn, o), y = __shfl_xor(mx, o);
        mn = x < mn ? x : mn;
        mx = y > mx ? y : mx;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&vmin[zl], mn);
        atomicMax(&vmax[zl], mx);
    }
}

// COUNT pass: counts per layer.  WRITE pass: keys into [off[zl], off[zl+1]).
template <bool WRITE>
__global__ __launch_bounds__(256) void layer_maxima(const unsigned char *__restrict__ J8, int w, int h, int z0, int tiles_x,
                                                     const int *__restrict__ vmin, const float *__restrict__ vfactor,
                                                     unsigned int *__restrict__ count, const i64 *__restrict__ off,
                                                     i64 *__restrict__ keys)
{
    i64 b = blockIdx.x;
    const int x = (int)(b % tiles_x) * 256 + threadIdx.x;
    b /= tiles_x;
    const int y = (int)(b % h);
    const int zl = (int)(b / h);
    if (x <= 0 || x >= w - 1 || y <= 0 || y >= h - 1) return; // border pixels are never maxima (seed.cpp:595)
    const unsigned char *L = J8 + (i64)(z0 + zl) * w * h;
    const int p = y * w + x;
    const int v = L[p];
    if (v == vmin[zl]) return; // seed.cpp:594
    const unsigned char *r0 = L + p - w, *r2 = L + p + w;
    const int m = max(max(max((int)r0[-1], (int)r0[0]), max((int)r0[1], (int)L[p - 1])),
                      max(max((int)L[p + 1], (int)r2[-1]), max((int)r2[0], (int)r2[1])));
    if (m > v) return;
    if (!WRITE) {
        atomicAdd(&count[zl], 1u);
    } else {
        // seed.cpp:616,626: iValue = (int)((fValue - globalMin) * vFactor), f32 arithmetic
        const float fValue = (float)v, gmin = (float)vmin[zl];
        const int iValue = (int)((fValue - gmin) * vfactor[zl]);
        const unsigned int slot = atomicAdd(&count[zl], 1u);
        keys[off[zl] + slot] = (i64)(((unsigned long long)(i64)iValue << 32) | (unsigned int)p);
    }
}

__global__ void gather_dirs(const unsigned char *__restrict__ Vx, const unsigned char *__restrict__ Vy,
                            const unsigned char *__restrict__ Vz, const i64 *__restrict__ idx, int n,
                            unsigned char *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const i64 v = idx[i];
    out[3 * i] = Vx[v];
    out[3 * i + 1] = Vy[v];
    out[3 * i + 2] = Vz[v];
}

// ---------------- host: tolerance flood-fill of one layer (ImageJ MaximumFinder) ----------------
enum : unsigned char { F_MAXIMUM = 1, F_LISTED = 2, F_PROCESSED = 4, F_MAX_AREA = 8, F_EQUAL = 16, F_MAX_POINT = 32, F_BORDER = 64 };

struct LayerFinder {
    int w, h;
    std::vector<unsigned char> flags;
    std::vector<int> list;
    LayerFinder(int w_, int h_) : w(w_), h(h_), flags((size_t)w_ * h_ + 8), list((size_t)w_ * h_) {}

    static inline bool inside(int x, int y, int d, int w, int h)
    {
        switch (d) { // neighbour d exists? (seed.cpp:1027-1049)
        case 0: return y > 0;
        case 1: return x < w - 1 && y > 0;
        case 2: return x < w - 1;
        case 3: return x < w - 1 && y < h - 1;
        case 4: return y < h - 1;
        case 5: return x > 0 && y < h - 1;
        case 6: return x > 0;
        default: return x > 0 && y > 0;
        }
    }

    // keys ascending; emits accepted maxima (pixel offsets) in processing order (highest first)
    void run(const unsigned char *L8, const i64 *keys, i64 nkeys, float tol, int layer_min, std::vector<int> &accepted)
    {
        const int step[8] = {-w, -w + 1, 1, w + 1, w, w - 1, -1, -w - 1};
        std::memset(flags.data(), 0, flags.size());
        for (int x = 0; x < w; x++) { flags[x] = F_BORDER; flags[(size_t)(h - 1) * w + x] = F_BORDER; } // image edge (seed.cpp: isWithin / edge tests)
        for (int y = 0; y < h; y++) { flags[(size_t)y * w] = F_BORDER; flags[(size_t)y * w + w - 1] = F_BORDER; }
        const uint32_t L3 = 0x010101u * F_LISTED;
        for (i64 q = nkeys - 1; q >= 0; --q) {
            int start = (int)(unsigned int)(keys[q] & 0xffffffffLL);
            if (flags[start] & F_PROCESSED) continue;
            int sx = start % w, sy = start / w;
            float v0 = (float)L8[start];
            // A maximum whose flood threshold v0 - tolerance does not exceed the layer's minimum can never be accepted: every
            // neighbour the fill looks at is either listed, or ends the candidate (PROCESSED, higher than v0, an edge pixel), or
            // qualifies -- so the fill only stops at one of those three, at the latest at the image edge.  The candidates come in
            // descending value, so none of the remaining ones can be accepted either, and the marks their fills would leave are
            // never looked at: the layer is finished (this is where the reference floods the whole background of every layer).
            if ((float)layer_min >= v0 - tol) break;
            bool retry;
            do {
                retry = false;
                list[0] = start;
                flags[start] |= (F_EQUAL | F_LISTED);
                int len = 1;
                bool edge = (sx == 0 || sx == w - 1 || sy == 0 || sy == h - 1);
                bool possible = true;
                double ex = sx, ey = sy;
                int neq = 1;
                for (int cur = 0; cur < len; ++cur) {
                    const int off = list[cur];
                    const unsigned char *f = flags.data() + off;
                    const bool inner = !(f[0] & F_BORDER);
                    int x = 0, y = 0;
                    if (inner) {
                        // all eight neighbours already in the list (the usual case inside a large flood): every direction
                        // of the loop below would `continue`
                        uint32_t a, b;
                        std::memcpy(&a, f - w - 1, 4);
                        std::memcpy(&b, f + w - 1, 4);
                        if ((a & L3) == L3 && (b & L3) == L3 && (f[-1] & f[1] & F_LISTED)) continue;
                    } else {
                        x = off % w; y = off / w;
                    }
                    for (int d = 0; d < 8; d++) {
                        const int o2 = off + step[d];
                        if (!(inner || inside(x, y, d, w, h))) continue;
                        const unsigned char f2 = flags[o2];
                        if (f2 & F_LISTED) continue;
                        if (f2 & F_PROCESSED) { possible = false; break; }
                        const float v2 = (float)L8[o2];
                        if (v2 > v0) { possible = false; break; } // maxSortingError == 0 (seed.cpp:634)
                        if (v2 >= v0 - tol) {
                            list[len++] = o2;
                            flags[o2] = f2 | F_LISTED;
                            if (f2 & F_BORDER) {
                                edge = true;
                                possible = false; // excludeEdgesNow
                                break;
                            }
                            if (v2 == v0) {
                                flags[o2] |= F_EQUAL;
                                ex += o2 % w; ey += o2 / w; neq++;
                            }
                        }
                    }
                }
                // (the sortingError branch of the original needs v2 > v0 past the test above: unreachable)
                const unsigned char keep = (unsigned char)~(possible ? F_LISTED : (F_LISTED | F_EQUAL));
                ex /= neq;
                ey /= neq;
                double best = 1e20;
                int besti = 0;
                for (int k = 0; k < len; k++) {
                    const int off = list[k];
                    flags[off] &= keep;
                    flags[off] |= F_PROCESSED;
                    if (possible) {
                        flags[off] |= F_MAX_AREA;
                        if (flags[off] & F_EQUAL) {
                            const int x = off % w, y = off / w;
                            const double d2 = (ex - x) * (double)(ex - x) + (ey - y) * (double)(ey - y);
                            if (d2 < best) { best = d2; besti = k; }
                        }
                    }
                }
                if (possible) {
                    const int off = list[besti];
                    flags[off] |= F_MAX_POINT;
                    if (!edge) accepted.push_back(off);
                }
            } while (retry);
        }
    }
};

} // namespace

int pnr_seeds_run(pnr_ctx *c, int64_t z0, int64_t z1)
{
    PNR_REQUIRE(c->have_j8, PNR_E_STATE, "pnr_extract_seeds: run pnr_frangi (or pnr_set_j8_v) first");
    PNR_REQUIRE(z0 >= 0 && z1 <= c->l && z0 <= z1, PNR_E_ARG, "layer range [%lld,%lld) outside [0,%lld)", (long long)z0, (long long)z1, (long long)c->l);
    c->seeds.clear();
    const bool timing = c->opt.seed_timing != 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = now();
    const int w = (int)c->w, h = (int)c->h;
    const int nl = (int)(z1 - z0);
    if (nl == 0) return PNR_OK;
    const i64 wh = (i64)w * h;

    int *d_min = nullptr, *d_max = nullptr;
    unsigned int *d_cnt = nullptr;
    float *d_vf = nullptr;
    i64 *d_off = nullptr, *d_keys = nullptr;
    int rc = c->scratch_get("seed_min", (size_t)nl, &d_min); // context-owned scratch, sized by the first pass
    if (!rc) rc = c->scratch_get("seed_max", (size_t)nl, &d_max);
    if (!rc) rc = c->scratch_get("seed_cnt", (size_t)nl, &d_cnt);
    if (!rc) rc = c->scratch_get("seed_vf", (size_t)nl, &d_vf);
    if (!rc) rc = c->scratch_get("seed_off", (size_t)nl + 1, &d_off);
    if (rc) return rc;
    std::vector<int> vmin(nl), vmax(nl);
    std::vector<unsigned int> cnt(nl);
    std::vector<float> vf(nl);
    std::vector<i64> off(nl + 1, 0);

    // start the J8 download for the host flood-fill while the kernels run
    if (c->h_j8_cap < (size_t)(wh * nl)) { // pinned staging buffer, kept across calls (allocating 1 GiB of pinned memory costs ~50 ms)
        if (c->h_j8) hipHostFree(c->h_j8);
        c->h_j8 = nullptr;
        c->h_j8_cap = 0;
        PNR_HIP(hipHostMalloc(&c->h_j8, (size_t)(wh * nl), hipHostMallocDefault));
        c->h_j8_cap = (size_t)(wh * nl);
    }
    unsigned char *h_j8 = c->h_j8;
    // ... on a second stream, in chunks of layers with an event each: the kernels below run beside it, and the fill of a layer
    // only waits for its own chunk (1 GiB takes 20 ms over PCIe, the fill of the first layers starts after ~6 ms)
    constexpr int NCH = pnr_ctx::J8_CHUNKS;
    if (!c->copy_stream) {
        PNR_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        PNR_HIP(hipEventCreateWithFlags(&c->j8_start, hipEventDisableTiming));
        for (int k = 0; k < NCH; k++) PNR_HIP(hipEventCreateWithFlags(&c->j8_ev[k], hipEventDisableTiming));
    }
    const int per_chunk = (nl + NCH - 1) / NCH;
    PNR_HIP(hipEventRecord(c->j8_start, c->stream)); // J8 is complete at this point of the context's stream
    PNR_HIP(hipStreamWaitEvent(c->copy_stream, c->j8_start, 0));
    for (int k = 0; k < NCH; k++) {
        const int l0 = k * per_chunk, l1 = std::min(nl, l0 + per_chunk);
        if (l0 < l1)
            PNR_HIP(hipMemcpyAsync(h_j8 + (size_t)l0 * wh, c->d_J8 + (z0 + l0) * wh, (size_t)(l1 - l0) * wh, hipMemcpyDeviceToHost, c->copy_stream));
        PNR_HIP(hipEventRecord(c->j8_ev[k], c->copy_stream));
    }

    c->tic();
    PNR_HIP(hipMemsetAsync(d_min, 0x7f, nl * 4, c->stream));
    PNR_HIP(hipMemsetAsync(d_max, 0, nl * 4, c->stream));
    hipLaunchKernelGGL(layer_minmax, dim3(nl * LM_PARTS), dim3(256), 0, c->stream, c->d_J8, wh, (int)z0, d_min, d_max);
    PNR_HIP(hipMemsetAsync(d_cnt, 0, nl * 4, c->stream));
    const int tiles_x = (w + 255) / 256;
    const unsigned nblk = (unsigned)((i64)tiles_x * h * nl);
    hipLaunchKernelGGL(layer_maxima<false>, dim3(nblk), dim3(256), 0, c->stream, c->d_J8, w, h, (int)z0, tiles_x, d_min,
                       (const float *)nullptr, d_cnt, (const i64 *)nullptr, (i64 *)nullptr);
    c->toc("seed_maxima", 2);
    PNR_HIP(hipMemcpyAsync(vmin.data(), d_min, nl * 4, hipMemcpyDeviceToHost, c->stream));
    PNR_HIP(hipMemcpyAsync(vmax.data(), d_max, nl * 4, hipMemcpyDeviceToHost, c->stream));
    PNR_HIP(hipMemcpyAsync(cnt.data(), d_cnt, nl * 4, hipMemcpyDeviceToHost, c->stream));
    PNR_HIP(hipStreamSynchronize(c->stream));
    for (int k = 0; k < nl; k++) {
        off[k + 1] = off[k] + cnt[k];
        vf[k] = (float)(2e9 / ((float)vmax[k] - (float)vmin[k])); // seed.cpp:616 (inf on flat layers: no maxima there)
    }
    const i64 total = off[nl];
    std::vector<i64> keys((size_t)total);
    if (total > 0) {
        rc = c->scratch_get("seed_keys", (size_t)total, &d_keys);
        if (rc) return rc;
        PNR_HIP(hipMemcpyAsync(d_off, off.data(), (nl + 1) * 8, hipMemcpyHostToDevice, c->stream));
        PNR_HIP(hipMemcpyAsync(d_vf, vf.data(), nl * 4, hipMemcpyHostToDevice, c->stream));
        PNR_HIP(hipMemsetAsync(d_cnt, 0, nl * 4, c->stream));
        c->tic();
        hipLaunchKernelGGL(layer_maxima<true>, dim3(nblk), dim3(256), 0, c->stream, c->d_J8, w, h, (int)z0, tiles_x, d_min,
                           d_vf, d_cnt, d_off, d_keys);
        c->toc("seed_maxima", 1);
        PNR_HIP(hipMemcpyAsync(keys.data(), d_keys, (size_t)total * 8, hipMemcpyDeviceToHost, c->stream));
    }
    PNR_HIP(hipGetLastError());
    PNR_HIP(hipStreamSynchronize(c->stream));

    const double t_gpu = now();
    // host: per-layer flood-fill on a thread pool; results kept per layer to preserve z-major order
    std::vector<std::vector<int>> acc(nl);
    {
        unsigned nt = (unsigned)pnr::host_threads(c->opt); // this process's share of the host's CPUs (options host_threads / local_ranks)
        if ((int)nt > nl) nt = nl;
        std::atomic<int> next(0);
        const float tol = (float)(double)c->prm.tolerance;
        auto work = [&]() {
            LayerFinder lf(w, h);
            for (;;) {
                const int k = next.fetch_add(1);
                if (k >= nl) break;
                if (cnt[k] == 0) continue;
                (void)hipEventSynchronize(c->j8_ev[k / per_chunk]); // this layer's bytes have arrived
                i64 *kb = keys.data() + off[k];
                std::sort(kb, kb + cnt[k]); // unique keys: order fully defined (seed.cpp:632)
                lf.run(h_j8 + (size_t)k * wh, kb, cnt[k], tol, vmin[k], acc[k]);
            }
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
    }
    PNR_HIP(hipStreamSynchronize(c->copy_stream)); // (layers without candidates never waited for their chunk)
    const double t_fill = now();

    // directions at the accepted voxels (seed.cpp:767-771)
    std::vector<i64> vox;
    for (int k = 0; k < nl; k++)
        for (int o : acc[k]) vox.push_back((z0 + k) * wh + o);
    const int ns = (int)vox.size();
    std::vector<unsigned char> dirs((size_t)ns * 3);
    if (ns > 0) {
        i64 *d_idx = nullptr;
        unsigned char *d_dirs = nullptr;
        rc = c->scratch_get("seed_idx", (size_t)ns, &d_idx);
        if (!rc) rc = c->scratch_get("seed_dirs", (size_t)ns * 3, &d_dirs);
        if (rc) return rc;
        PNR_HIP(hipMemcpyAsync(d_idx, vox.data(), (size_t)ns * 8, hipMemcpyHostToDevice, c->stream));
        const int how = pnr_seed_dirs(c, d_idx, ns, d_dirs); // solved at the seeds from the winning scale's smoothed volume ...
        if (how < 0) return how;
        if (how == 1) // ... or gathered from the direction volumes when those exist
            hipLaunchKernelGGL(gather_dirs, dim3((ns + 255) / 256), dim3(256), 0, c->stream, c->d_Vx, c->d_Vy, c->d_Vz, d_idx, ns,
                               d_dirs);
        PNR_HIP(hipMemcpyAsync(dirs.data(), d_dirs, (size_t)ns * 3, hipMemcpyDeviceToHost, c->stream));
        PNR_HIP(hipStreamSynchronize(c->stream));
    }
    c->seeds.resize(ns);
    for (int i = 0; i < ns; i++) {
        const i64 v = vox[i];
        const int z = (int)(v / wh);
        const int y = (int)((v - (i64)z * wh) / w), x = (int)(v % w);
        const float Ux = (((float)dirs[3 * i] / 255) * 2) - 1;
        const float Uy = (((float)dirs[3 * i + 1] / 255) * 2) - 1;
        const float Uz = (((float)dirs[3 * i + 2] / 255) * 2) - 1;
        const float Un = (float)std::sqrt((double)Ux * Ux + (double)Uy * Uy + (double)Uz * Uz); // pow(f,2): f64
        c->seeds[i] = pnr_seed{(float)x, (float)y, (float)z, Ux / Un, Uy / Un, Uz / Un, 0.f, 0.f};
    }
    if (timing)
        fprintf(stderr, "[pnr seeds] kernels + keys %.1f ms (J8 download overlapped), host fill %.1f ms (%u threads, %lld candidates), dirs+free %.1f ms\n",
                1e3 * (t_gpu - t_start), 1e3 * (t_fill - t_gpu), (unsigned)pnr::host_threads(c->opt), (long long)total, 1e3 * (now() - t_fill));
    return PNR_OK;
}
