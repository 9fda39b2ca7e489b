// soma.hip -- soma path for somaradius > 0 (Advantra_plugin.cpp:2426-2448, soma_extraction1 :1899-1915).
//
//   Frangi::imerode (xy)        frangi.cpp:880-968   erode_x / erode_y   u8 min filter, window 2*ceil(rad)+1, clamp-to-edge
//   Frangi::imgaussian (u8, xy) frangi.cpp:786-878   gauss_x_u8 (frangi.hip) / gauss_y_trunc_hist: the reference's y pass accumulates INTO the
//                                                    unsigned char output, i.e. converts the running sum back to u8 after
//                                                    every tap -- reproduced tap by tap, ascending
//   maxentropy_th               toolbox.cpp:657-737  256-bin histogram fused into the y pass, the entropy search on the host (libm logf/log)
//   binarise + conn3d           :1901-1908, toolbox.cpp:245-509  foreground voxels are compacted on the GPU in raster order;
//                               the region growing itself is sequential by definition (centroid / radius are running f32 means
//                               in LIFO visit order) and runs on the host over the foreground voxels only
//
// The four kernels are streaming u8 / f32 passes.  Results: one SOMA node per region, a
// sparse voxel -> node-index map used by the seed filter (:2561-2564), by the replay (tracker.cpp:858-869) and -- written
// as "saturated" into the GPU density map -- by the trace kernels' early stop.
#include "ctx.h"
#include "replay.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <unordered_map>

namespace {
typedef long long i64;
constexpr int SOMA_BLOCK = 256;

__global__ __launch_bounds__(SOMA_BLOCK) void erode_x(const unsigned char *__restrict__ I, unsigned char *__restrict__ K, int w, i64 n, int L)
{
    const i64 i = (i64)blockIdx.x * SOMA_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % w);
    const i64 row = i - x;
    unsigned m = I[i];
    for (int d = -L; d <= L; d++) {
        const int x1 = x + d < 0 ? 0 : (x + d > w - 1 ? w - 1 : x + d);
        m = min(m, (unsigned)I[row + x1]);
    }
    K[i] = (unsigned char)m;
}

__global__ __launch_bounds__(SOMA_BLOCK) void erode_y(const unsigned char *__restrict__ K, unsigned char *__restrict__ E, int w, int h, i64 n, int L)
{
    const i64 i = (i64)blockIdx.x * SOMA_BLOCK + threadIdx.x;
    if (i >= n) return;
    const i64 wh = (i64)w * h;
    const i64 z = i / wh, r = i - z * wh;
    const int y = (int)(r / w), x = (int)(r - (i64)y * w);
    unsigned m = K[i];
    for (int d = -L; d <= L; d++) {
        const int y1 = y + d < 0 ? 0 : (y + d > h - 1 ? h - 1 : y + d);
        m = min(m, (unsigned)K[z * wh + (i64)y1 * w + x]);
    }
    E[i] = (unsigned char)m;
}

// y pass of the u8 Gaussian: `I[i0] = 0; I[i0] += K[i1] * G[...]` -- the left operand is an unsigned char, so the running sum is
// truncated at every tap (frangi.cpp:839-872).  Tile: 64 (x, coalesced) x TYS (y) outputs, the TYS + 2L input rows staged in LDS;
// the 256-bin histogram of the result (maxentropy_th's input, toolbox.cpp:663-668) is taken on the way out, eight interleaved
// copies per work-group so that the dominant background value does not serialise on one LDS address.
constexpr int TYS = 32;
__global__ __launch_bounds__(256) void gauss_y_trunc_hist(const float *__restrict__ K, unsigned char *__restrict__ I, const float *__restrict__ G, int w,
                                                          int h, int tiles_x, int tiles_y, int L, unsigned long long *__restrict__ hist)
{
    extern __shared__ float smem[]; // [(TYS + 2L)][64] | taps[2L+1] | hist[8][256] (u32)
    float *s_in = smem;
    float *s_tap = smem + (TYS + 2 * L) * 64;
    unsigned int *s_hist = (unsigned int *)(s_tap + 2 * L + 1);
    i64 b = blockIdx.x;
    const int tx_tile = (int)(b % tiles_x);
    b /= tiles_x;
    const int ty_tile = (int)(b % tiles_y);
    const i64 z = b / tiles_y;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int x = tx_tile * 64 + lane, y0 = ty_tile * TYS;
    const float *base = K + z * (i64)w * h;
    const int rows = TYS + 2 * L;
    for (int t = threadIdx.x; t < 8 * 256; t += 256) s_hist[t] = 0;
    if (x < w)
        for (int r = grp; r < rows; r += 4) {
            int y = y0 - L + r;
            y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
            s_in[r * 64 + lane] = base[(i64)y * w + x];
        }
    for (int t = threadIdx.x; t < 2 * L + 1; t += 256) s_tap[t] = G[t];
    __syncthreads();
    if (x < w) {
        unsigned char *dst = I + z * (i64)w * h;
#pragma unroll 1
        for (int j = 0; j < TYS / 4; ++j) {
            const int ry = grp + 4 * j, y = y0 + ry;
            if (y >= h) break;
            unsigned a = 0;
            for (int k = 0; k <= 2 * L; ++k) a = (unsigned)(int)((float)a + s_in[(ry + k) * 64 + lane] * s_tap[k]) & 0xffu; // in [0, 255]: plain truncation
            dst[(i64)y * w + x] = (unsigned char)a;
            atomicAdd(&s_hist[(lane & 7) * 256 + a], 1u);
        }
    }
    __syncthreads();
    unsigned int tot = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) tot += s_hist[q * 256 + threadIdx.x];
    if (tot) atomicAdd(&hist[threadIdx.x], (unsigned long long)tot);
}

// foreground (E8 > th) voxels per (z,y) row, then their indices in raster order
__global__ __launch_bounds__(64) void row_count(const unsigned char *__restrict__ E, int w, int th, int *__restrict__ cnt)
{
    const i64 row = blockIdx.x;
    int c = 0;
    for (int x0 = 0; x0 < w; x0 += 64) {
        const int x = x0 + threadIdx.x;
        const bool f = x < w && (int)E[row * w + x] > th;
        c += __popcll(__builtin_amdgcn_ballot_w64(f));
    }
    if (threadIdx.x == 0) cnt[row] = c;
}

__global__ __launch_bounds__(64) void row_compact(const unsigned char *__restrict__ E, int w, int th, const i64 *__restrict__ off, i64 *__restrict__ vox)
{
    const i64 row = blockIdx.x;
    i64 o = off[row];
    for (int x0 = 0; x0 < w; x0 += 64) {
        const int x = x0 + threadIdx.x;
        const bool f = x < w && (int)E[row * w + x] > th;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(f);
        if (f) vox[o + __popcll(m & ((1ull << threadIdx.x) - 1ull))] = row * w + x;
        o += __popcll(m);
    }
}

// toolbox.cpp:657-737 on the histogram
unsigned char maxentropy_from_hist(const unsigned long long *h)
{
    float sum = 0;
    for (int i = 0; i < 256; i++) sum += (float)(int)h[i]; // `int hist[]` added into a float
    float nh[256], pT[256], hB[256], hW[256];
    for (int i = 0; i < 256; i++) nh[i] = (float)(int)h[i] / sum;
    pT[0] = nh[0];
    for (int i = 1; i < 256; i++) pT[i] = pT[i - 1] + nh[i];
    const float eps = FLT_MIN;
    for (int t = 0; t < 256; t++) {
        if (pT[t] > eps) {
            float hh = 0;
            for (int i = 0; i <= t; i++)
                if (nh[i] > eps) hh -= nh[i] / pT[t] * std::log(nh[i] / pT[t]); // float overload
            hB[t] = hh;
        } else {
            hB[t] = 0;
        }
        const double pTW = 1 - pT[t];
        if (pTW > eps) {
            float hh = 0;
            for (int i = t + 1; i < 256; i++)
                if (nh[i] > eps) hh = (float)((double)hh - nh[i] / pTW * std::log(nh[i] / pTW));
            hW[t] = hh;
        } else {
            hW[t] = 0;
        }
    }
    float jMax = hB[0] + hW[0];
    unsigned char tMax = 0;
    for (int t = 1; t < 256; t++) {
        const double j = hB[t] + hW[t];
        if (j > jMax) { jMax = (float)j; tMax = (unsigned char)t; }
    }
    return tMax;
}

// conn3d (toolbox.cpp:245-509) with diagonal = true, valuesOverDouble = 0, minRegSize = 1 on the binarised stack: every
// foreground voxel has the same value, so a region is a 26-connected component.  Regions are numbered in raster order of their
// first voxel; the growth is LIFO with neighbours pushed in z, y, x order; centroid and radius are running f32 means in visit order.
void grow_regions(const std::vector<i64> &vox, int w, int h, int l, std::vector<int32_t> &lab, std::vector<pnr_node> &nodes)
{
    const i64 wh = (i64)w * h;
    const size_t K = vox.size();
    std::unordered_map<i64, int32_t> at; // voxel -> position in vox
    at.reserve(K * 2 + 16);
    for (size_t k = 0; k < K; k++) at.emplace(vox[k], (int32_t)k);
    lab.assign(K, 0);
    std::vector<uint8_t> queued(K, 0);
    std::vector<int32_t> stack, reg;
    int32_t nreg = 0;
    for (size_t s = 0; s < K; s++) {
        if (queued[s]) continue; // already part of an earlier region
        stack.assign(1, (int32_t)s);
        queued[s] = 1;
        reg.clear();
        float xmean = 0, ymean = 0, zmean = 0;
        while (!stack.empty()) {
            const int32_t k = stack.back();
            stack.pop_back();
            const i64 idx = vox[(size_t)k];
            const int pz = (int)(idx / wh), py = (int)((idx % wh) / w), px = (int)((idx % wh) % w);
            reg.push_back(k);
            const i64 np = (i64)reg.size();
            const float t1 = (float)(np - 1) / (float)np, t2 = (float)(1.0 / (double)np);
            xmean = t1 * xmean + t2 * (float)px;
            ymean = t1 * ymean + t2 * (float)py;
            zmean = t1 * zmean + t2 * (float)pz;
            const int x0 = std::max(0, px - 1), y0 = std::max(0, py - 1), z0 = std::max(0, pz - 1);
            const int x1 = std::min(px + 1, w - 1), y1 = std::min(py + 1, h - 1), z1 = std::min(pz + 1, l - 1);
            for (int z = z0; z <= z1; z++)
                for (int y = y0; y <= y1; y++)
                    for (int x = x0; x <= x1; x++) {
                        if (x == px && y == py && z == pz) continue;
                        const auto it = at.find((i64)w * ((i64)z * h + y) + x);
                        if (it == at.end() || queued[(size_t)it->second]) continue;
                        queued[(size_t)it->second] = 1;
                        stack.push_back(it->second);
                    }
        }
        nreg++;
        float rmean = 0;
        for (size_t p = 1; p <= reg.size(); p++) {
            const i64 idx = vox[(size_t)reg[p - 1]];
            const int pz = (int)(idx / wh), py = (int)((idx % wh) / w), px = (int)((idx % wh) % w);
            const float t1 = (float)(p - 1) / (float)p, t2 = (float)(1.0 / (double)p);
            const double dx = (double)((float)px - xmean), dy = (double)((float)py - ymean), dz = (double)((float)pz - zmean);
            rmean = (float)((double)(t1 * rmean) + (double)t2 * std::sqrt(dx * dx + dy * dy + dz * dz));
        }
        for (int32_t k : reg) lab[(size_t)k] = nreg; // = index in the node list (node 0 is the dummy)
        pnr_node nd; // Node(x, y, z, r, SOMA): v = 0, corr = -FLT_MAX (node.cpp:68-79)
        std::memset(&nd, 0, sizeof(nd));
        nd.x = xmean; nd.y = ymean; nd.z = zmean; nd.sig = rmean;
        nd.corr = -FLT_MAX;
        nd.type = 1;
        nodes.push_back(nd);
    }
}

} // namespace

int pnr_soma_run(pnr_ctx *c, uint8_t *E8_out, int32_t *threshold)
{
    c->soma_nodes.clear(); c->soma_vox.clear(); c->soma_lab.clear(); c->soma_map.clear();
    c->have_soma = false;
    if (threshold) *threshold = 0;
    if (c->prm.somaradius <= 0) { c->have_soma = true; return PNR_OK; } // "no soma detection": empty map (:2482-2486)
    PNR_REQUIRE(c->d_img, PNR_E_STATE, "no volume set (pnr_set_volume)");
    int rc = pnr_ensure_frangi_buffers(c); // scratch: the Frangi buffers are free until pnr_frangi runs
    if (!rc) rc = pnr_ensure_tmpA(c);
    if (!rc) rc = pnr_ensure_v(c);
    if (rc) return rc;
    const int w = (int)c->w, h = (int)c->h, l = (int)c->l;
    const i64 n = c->N, rows = (i64)h * l;
    PNR_REQUIRE(rows < (1LL << 31), PNR_E_ARG, "too many rows");
    const float rad = (float)c->prm.somaradius;
    const int Le = (int)std::ceil(rad);
    std::vector<float> G;
    const int Lg = pnr::gaussian_taps(rad, G); // same kernel formula as the 3-D filter (frangi.cpp:791-803)
    hipStream_t st = c->stream;
    float *d_G = nullptr;
    unsigned long long *d_hist = nullptr;
    int *d_cnt = nullptr;
    i64 *d_off = nullptr, *d_vox = nullptr;
    { // a +0 in front of the taps and one behind: the register-tiled x pass reads the neighbours of a tap pairwise (gauss_sums_packed)
        std::vector<float> Gp(G.size() + 2, 0.f);
        std::copy(G.begin(), G.end(), Gp.begin() + 1);
        G.swap(Gp);
    }
    PNR_HIP(hipMalloc(&d_G, G.size() * 4));
    PNR_HIP(hipMalloc(&d_hist, 256 * 8));
    PNR_HIP(hipMalloc(&d_cnt, (size_t)rows * 4));
    PNR_HIP(hipMalloc(&d_off, (size_t)rows * 8));
    auto cleanup = [&]() { hipFree(d_G); hipFree(d_hist); hipFree(d_cnt); hipFree(d_off); hipFree(d_vox); };
    PNR_HIP(hipMemcpyAsync(d_G, G.data(), G.size() * 4, hipMemcpyHostToDevice, st));
    PNR_HIP(hipMemsetAsync(d_hist, 0, 256 * 8, st));
    unsigned char *d_K = c->d_Vx, *d_E = c->d_Vy;
    c->have_v = false; // the direction volumes serve as scratch here
    const unsigned nb = (unsigned)((n + SOMA_BLOCK - 1) / SOMA_BLOCK);
    c->tic();
    hipLaunchKernelGGL(erode_x, dim3(nb), dim3(SOMA_BLOCK), 0, st, c->d_img, d_K, w, n, Le);
    hipLaunchKernelGGL(erode_y, dim3(nb), dim3(SOMA_BLOCK), 0, st, (const unsigned char *)d_K, d_E, w, h, n, Le);
    rc = pnr_gauss_x_u8_launch(c, d_E, c->d_tmpA, d_G + 1, Lg); // K[i0] += I[i1] * G[...], taps ascending, clamp-to-edge (frangi.cpp:806-836)
    if (rc) { cleanup(); return rc; }
    {
        const int tiles_x = (w + 63) / 64, tiles_y = (h + TYS - 1) / TYS;
        const size_t sm = ((size_t)(TYS + 2 * Lg) * 64 + 2 * Lg + 1) * 4 + 8 * 256 * 4;
        hipLaunchKernelGGL(gauss_y_trunc_hist, dim3((unsigned)((i64)tiles_x * tiles_y * l)), dim3(256), sm, st, (const float *)c->d_tmpA, d_E,
                           (const float *)(d_G + 1), w, h, tiles_x, tiles_y, Lg, d_hist);
    }
    c->toc("soma", 4);
    unsigned long long hist[256];
    PNR_HIP(hipMemcpyAsync(hist, d_hist, sizeof(hist), hipMemcpyDeviceToHost, st));
    if (E8_out) PNR_HIP(hipMemcpyAsync(E8_out, d_E, (size_t)n, hipMemcpyDeviceToHost, st));
    PNR_HIP(hipStreamSynchronize(st));
    const int th = maxentropy_from_hist(hist);
    if (threshold) *threshold = th;
    hipLaunchKernelGGL(row_count, dim3((unsigned)rows), dim3(64), 0, st, (const unsigned char *)d_E, w, th, d_cnt);
    std::vector<int> cnt((size_t)rows);
    PNR_HIP(hipMemcpyAsync(cnt.data(), d_cnt, (size_t)rows * 4, hipMemcpyDeviceToHost, st));
    PNR_HIP(hipStreamSynchronize(st));
    std::vector<i64> off((size_t)rows);
    i64 K = 0;
    for (i64 r = 0; r < rows; r++) { off[(size_t)r] = K; K += cnt[(size_t)r]; }
    c->soma_vox.resize((size_t)K);
    if (K > 0) {
        if (hipMalloc(&d_vox, (size_t)K * 8) != hipSuccess) { cleanup(); PNR_REQUIRE(false, PNR_E_HIP, "out of device memory for %lld soma voxels", (long long)K); }
        PNR_HIP(hipMemcpyAsync(d_off, off.data(), (size_t)rows * 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(row_compact, dim3((unsigned)rows), dim3(64), 0, st, (const unsigned char *)d_E, w, th, (const i64 *)d_off, d_vox);
        PNR_HIP(hipMemcpyAsync(c->soma_vox.data(), d_vox, (size_t)K * 8, hipMemcpyDeviceToHost, st));
        PNR_HIP(hipStreamSynchronize(st));
    }
    PNR_HIP(hipGetLastError());
    cleanup();
    std::vector<i64> vox(c->soma_vox.begin(), c->soma_vox.end());
    grow_regions(vox, w, h, l, c->soma_lab, c->soma_nodes);
    c->soma_map.reserve((size_t)K * 2 + 16);
    for (size_t k = 0; k < (size_t)K; k++) c->soma_map.emplace(c->soma_vox[k], c->soma_lab[k]);
    c->have_soma = true;
    return PNR_OK;
}
