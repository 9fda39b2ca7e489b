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

// COUNT pass: candidate counts per layer; and, for the host's flood fill, which pixels lie above their layer's minimum -- one bit
// per pixel and their number per row.  WRITE pass: keys into [off[zl], off[zl+1]); the values of those pixels into
// [voff[zl] + rowoff[row] ...).  The host rebuilds a layer from the bits and the values (everything else IS the layer's minimum):
// 1 bit per pixel + 1 byte per pixel above the minimum cross PCIe instead of the whole J8 volume (174 MB instead of 1 GiB on the
// bench stack, 95.7 % of which is the minimum).
// A wavefront covers 256 consecutive pixels of a row, four per lane (one dword load); its four bitmap words are the ballots of
// the four byte positions: bit l of word b <-> pixel 256 * wt + 4 * l + b.  The values follow in (row, wave tile, b, bit) order --
// any fixed order serves, the host scatters them back by the same rule.  The WRITE pass is driven by the bitmap: a wavefront
// whose four words are zero (nine in ten on the bench stack) touches no pixel at all.
struct SparseJ8 {
    unsigned long long *bitmap; // [row][wpr] words, row = zl * h + y, wpr = 4 * ceil(w / 256)
    unsigned int *rowcnt;       // [row] pixels above the layer minimum (COUNT pass, atomics of the row's waves)
    const unsigned int *rowoff; // [row] exclusive prefix of rowcnt inside the layer (row_scan)
    const i64 *voff;            // [zl] first value of the layer
    unsigned char *vals;
    int wpr;
};

template <bool WRITE>
__global__ __launch_bounds__(256) void layer_maxima(const unsigned char *__restrict__ J8, int w, int h, int z0, int tiles_x,
                                                     const int *__restrict__ vmin, const float *__restrict__ vfactor,
                                                     unsigned int *__restrict__ count, const i64 *__restrict__ off,
                                                     i64 *__restrict__ keys, SparseJ8 SP)
{
    i64 b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wt = (int)(b % tiles_x) * 4 + (threadIdx.x >> 6); // wave tile of 256 pixels in the row
    b /= tiles_x;
    const int y = (int)(b % h);
    const int zl = (int)(b / h);
    if (wt * 256 >= w) return; // (wave-uniform)
    const unsigned char *L = J8 + (i64)(z0 + zl) * w * h;
    const int x0 = wt * 256 + 4 * lane;
    const i64 row = (i64)zl * h + y;
    const int lmin = vmin[zl];
    unsigned long long *bw = SP.bitmap + row * SP.wpr + wt * 4;
    unsigned long long words[4];
    if (WRITE) {
#pragma unroll
        for (int q = 0; q < 4; q++) words[q] = bw[q]; // (uniform address)
        if ((words[0] | words[1] | words[2] | words[3]) == 0ull) return; // nothing above the minimum here: no value, no candidate
    }
    // the lane's four pixels (beyond the row: the minimum)
    int v[4];
    {
        const i64 p0 = (i64)y * w + x0;
        typedef unsigned __attribute__((aligned(1))) u32u;
        if (x0 + 3 < w) {
            const unsigned q = *(const u32u *)(L + p0);
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = (int)((q >> (8 * k)) & 0xffu);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = x0 + k < w ? (int)L[p0 + k] : lmin;
        }
    }
    if (!WRITE) {
        unsigned int n = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            words[q] = __builtin_amdgcn_ballot_w64(v[q] > lmin);
            n += (unsigned int)__builtin_popcountll(words[q]);
        }
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 4; q++) bw[q] = words[q];
            if (n) atomicAdd(&SP.rowcnt[row], n);
        }
        if (n == 0) return;
    } else {
        // pixels of the row in front of this wave tile: the popcounts of the row's earlier words
        unsigned int before = 0;
        for (int k = lane; k < wt * 4; k += 64) before += (unsigned int)__builtin_popcountll(SP.bitmap[row * SP.wpr + k]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
        unsigned char *dst = SP.vals + SP.voff[zl] + SP.rowoff[row] + before;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (words[q] >> lane & 1ull) dst[__builtin_popcountll(words[q] & ((1ull << lane) - 1ull))] = (unsigned char)v[q];
            dst += __builtin_popcountll(words[q]);
        }
    }
    // 8-neighbour local maxima among the pixels above the minimum (seed.cpp:589-614)
    if (y <= 0 || y >= h - 1) return; // border pixels are never maxima (:595)
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int x = x0 + q;
        if (v[q] == lmin || x <= 0 || x >= w - 1) continue; // :594, :595 (pixels beyond the row carry the minimum)
        const int p = y * w + x;
        const unsigned char *r0 = L + p - w, *r2 = L + p + w;
        const int m = max(max(max((int)r0[-1], (int)r0[0]), max((int)r0[1], (int)L[p - 1])),
                          max(max((int)L[p + 1], (int)r2[-1]), max((int)r2[0], (int)r2[1])));
        if (m > v[q]) continue;
        if (!WRITE) {
            atomicAdd(&count[zl], 1u);
        } else {
            // seed.cpp:616,626: iValue = (int)((fValue - globalMin) * vFactor), f32 arithmetic
            const float fValue = (float)v[q], gmin = (float)lmin;
            const int iValue = (int)((fValue - gmin) * vfactor[zl]);
            const unsigned int slot = atomicAdd(&count[zl], 1u);
            keys[off[zl] + slot] = (i64)(((unsigned long long)(i64)iValue << 32) | (unsigned int)p);
        }
    }
}

// exclusive prefix of the row counts inside every layer (one work-group per layer) and the layer totals
__global__ __launch_bounds__(256) void row_scan(const unsigned int *__restrict__ rowcnt, int h, unsigned int *__restrict__ rowoff,
                                                 unsigned int *__restrict__ layer_total)
{
    __shared__ unsigned int wsum[4];
    __shared__ unsigned int carry;
    const int zl = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int y0 = 0; y0 < h; y0 += 256) {
        const int y = y0 + tid;
        const unsigned int c = y < h ? rowcnt[(i64)zl * h + y] : 0u;
        unsigned int incl = c; // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        unsigned int base = carry;
        for (int k = 0; k < wv; k++) base += wsum[k];
        if (y < h) rowoff[(i64)zl * h + y] = base + incl - c;
        __syncthreads();
        if (tid == 255) carry = base + incl;
        __syncthreads();
    }
    if (tid == 0) layer_total[zl] = carry;
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
    std::vector<unsigned char> img; // the layer, rebuilt from the sparse hand-over: the layer's minimum everywhere but at the set bits
    int img_fill = -1;
    LayerFinder(int w_, int h_) : w(w_), h(h_), flags((size_t)w_ * h_ + 8), list((size_t)w_ * h_), img((size_t)w_ * h_) {}

    // bits: [h][wpr] words; vals: the values of the set bits in raster order.  restore() puts the minimum back at exactly those pixels.
    const unsigned char *load(const unsigned long long *bits, int wpr, const unsigned char *vals, int layer_min)
    {
        if (img_fill != layer_min) { std::memset(img.data(), layer_min, img.size()); img_fill = layer_min; }
        for (int y = 0; y < h; y++) {
            unsigned char *row = img.data() + (size_t)y * w;
            const unsigned long long *bw = bits + (size_t)y * wpr;
            for (int k = 0; k < wpr; k++) { // word k = byte position k & 3 of wave tile k >> 2: bit l <-> pixel 256 (k >> 2) + 4 l + (k & 3)
                unsigned long long m = bw[k];
                unsigned char *base = row + (k >> 2) * 256 + (k & 3);
                while (m) {
                    const int bit = __builtin_ctzll(m);
                    m &= m - 1;
                    base[4 * bit] = *vals++;
                }
            }
        }
        return img.data();
    }
    void restore(const unsigned long long *bits, int wpr)
    {
        for (int y = 0; y < h; y++) {
            unsigned char *row = img.data() + (size_t)y * w;
            const unsigned long long *bw = bits + (size_t)y * wpr;
            for (int k = 0; k < wpr; k++) {
                unsigned long long m = bw[k];
                unsigned char *base = row + (k >> 2) * 256 + (k & 3);
                while (m) {
                    const int bit = __builtin_ctzll(m);
                    m &= m - 1;
                    base[4 * bit] = (unsigned char)img_fill;
                }
            }
        }
    }

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
    unsigned int *d_cnt = nullptr, *d_rowcnt = nullptr, *d_rowoff = nullptr, *d_ltot = nullptr;
    unsigned long long *d_bits = nullptr;
    unsigned char *d_vals = nullptr;
    float *d_vf = nullptr;
    i64 *d_off = nullptr, *d_voff = nullptr, *d_keys = nullptr;
    const int wpr = 4 * ((w + 255) / 256); // bitmap words per row: four per wave tile of 256 pixels
    const size_t nrows = (size_t)nl * h, nwords = nrows * (size_t)wpr;
    int rc = c->scratch_get("seed_min", (size_t)nl, &d_min); // context-owned scratch, sized by the first pass
    if (!rc) rc = c->scratch_get("seed_max", (size_t)nl, &d_max);
    if (!rc) rc = c->scratch_get("seed_cnt", (size_t)nl, &d_cnt);
    if (!rc) rc = c->scratch_get("seed_vf", (size_t)nl, &d_vf);
    if (!rc) rc = c->scratch_get("seed_off", (size_t)nl + 1, &d_off);
    if (!rc) rc = c->scratch_get("seed_voff", (size_t)nl + 1, &d_voff);
    if (!rc) rc = c->scratch_get("seed_ltot", (size_t)nl, &d_ltot);
    if (!rc) rc = c->scratch_get("seed_rowcnt", nrows, &d_rowcnt);
    if (!rc) rc = c->scratch_get("seed_rowoff", nrows, &d_rowoff);
    if (!rc) rc = c->scratch_get("seed_bits", nwords, &d_bits);
    if (rc) return rc;
    std::vector<int> vmin(nl), vmax(nl);
    std::vector<unsigned int> cnt(nl), ltot(nl);
    std::vector<float> vf(nl);
    std::vector<i64> off(nl + 1, 0), voff(nl + 1, 0);

    // The host's flood fill gets J8 in sparse form: one bit per pixel (above its layer's minimum or not) and the values of the
    // set pixels in raster order -- on a second stream, in chunks of layers with an event each, so that the fill of a layer only
    // waits for its own chunk.  Pinned staging buffers are kept across calls (allocating pinned memory costs ~50 ms per GiB).
    constexpr int NCH = pnr_ctx::J8_CHUNKS;
    if (!c->copy_stream) {
        PNR_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        PNR_HIP(hipEventCreateWithFlags(&c->j8_start, hipEventDisableTiming));
        for (int k = 0; k < NCH; k++) PNR_HIP(hipEventCreateWithFlags(&c->j8_ev[k], hipEventDisableTiming));
    }
    auto pinned = [&](unsigned char *&buf, size_t &cap, size_t need) -> int {
        if (cap >= need) return PNR_OK;
        if (buf) (void)hipHostFree(buf);
        buf = nullptr; cap = 0;
        need += need / 4 + 4096; // (the value count changes a little from stack to stack)
        PNR_HIP(hipHostMalloc(&buf, need, hipHostMallocDefault));
        cap = need;
        return PNR_OK;
    };
    rc = pinned(c->h_j8, c->h_j8_cap, nwords * 8);
    if (rc) return rc;
    const unsigned long long *h_bits = (const unsigned long long *)c->h_j8;
    const int per_chunk = (nl + NCH - 1) / NCH;

    c->tic();
    PNR_HIP(hipMemsetAsync(d_min, 0x7f, nl * 4, c->stream));
    PNR_HIP(hipMemsetAsync(d_max, 0, nl * 4, c->stream));
    hipLaunchKernelGGL(layer_minmax, dim3(nl * LM_PARTS), dim3(256), 0, c->stream, c->d_J8, wh, (int)z0, d_min, d_max);
    PNR_HIP(hipMemsetAsync(d_cnt, 0, nl * 4, c->stream));
    PNR_HIP(hipMemsetAsync(d_rowcnt, 0, nrows * 4, c->stream));
    const int tiles_x = (w + 1023) / 1024; // a work-group of four waves covers 1024 pixels of a row
    const unsigned nblk = (unsigned)((i64)tiles_x * h * nl);
    SparseJ8 SP{d_bits, d_rowcnt, d_rowoff, d_voff, nullptr, wpr};
    hipLaunchKernelGGL(layer_maxima<false>, dim3(nblk), dim3(256), 0, c->stream, c->d_J8, w, h, (int)z0, tiles_x, d_min,
                       (const float *)nullptr, d_cnt, (const i64 *)nullptr, (i64 *)nullptr, SP);
    hipLaunchKernelGGL(row_scan, dim3(nl), dim3(256), 0, c->stream, (const unsigned int *)d_rowcnt, h, d_rowoff, d_ltot);
    c->toc("seed_maxima", 3);
    PNR_HIP(hipMemcpyAsync(vmin.data(), d_min, nl * 4, hipMemcpyDeviceToHost, c->stream));
    PNR_HIP(hipMemcpyAsync(vmax.data(), d_max, nl * 4, hipMemcpyDeviceToHost, c->stream));
    PNR_HIP(hipMemcpyAsync(cnt.data(), d_cnt, nl * 4, hipMemcpyDeviceToHost, c->stream));
    PNR_HIP(hipMemcpyAsync(ltot.data(), d_ltot, nl * 4, hipMemcpyDeviceToHost, c->stream));
    // the bitmap is complete: its download (1 bit per pixel) starts now, beside the second pass
    PNR_HIP(hipEventRecord(c->j8_start, c->stream));
    PNR_HIP(hipStreamWaitEvent(c->copy_stream, c->j8_start, 0));
    PNR_HIP(hipMemcpyAsync(c->h_j8, d_bits, nwords * 8, hipMemcpyDeviceToHost, c->copy_stream));
    PNR_HIP(hipStreamSynchronize(c->stream));
    for (int k = 0; k < nl; k++) {
        off[k + 1] = off[k] + cnt[k];
        voff[k + 1] = voff[k] + ltot[k];
        vf[k] = (float)(2e9 / ((float)vmax[k] - (float)vmin[k])); // seed.cpp:616 (inf on flat layers: no maxima there)
    }
    const i64 total = off[nl], nvals = voff[nl];
    std::vector<i64> keys((size_t)total);
    rc = pinned(c->h_j8v, c->h_j8v_cap, (size_t)std::max<i64>(nvals, 1));
    if (rc) return rc;
    const unsigned char *h_vals = c->h_j8v;
    if (total > 0) { // (no candidate anywhere: nothing to fill, nothing to hand over)
        rc = c->scratch_get("seed_keys", (size_t)total, &d_keys);
        if (!rc) rc = c->scratch_get("seed_vals", (size_t)std::max<i64>(nvals, 1), &d_vals);
        if (rc) return rc;
        SP.vals = d_vals;
        PNR_HIP(hipMemcpyAsync(d_off, off.data(), (nl + 1) * 8, hipMemcpyHostToDevice, c->stream));
        PNR_HIP(hipMemcpyAsync(d_voff, voff.data(), (nl + 1) * 8, hipMemcpyHostToDevice, c->stream));
        PNR_HIP(hipMemcpyAsync(d_vf, vf.data(), nl * 4, hipMemcpyHostToDevice, c->stream));
        PNR_HIP(hipMemsetAsync(d_cnt, 0, nl * 4, c->stream));
        c->tic();
        hipLaunchKernelGGL(layer_maxima<true>, dim3(nblk), dim3(256), 0, c->stream, c->d_J8, w, h, (int)z0, tiles_x, d_min,
                           d_vf, d_cnt, d_off, d_keys, SP);
        c->toc("seed_maxima", 1);
        PNR_HIP(hipMemcpyAsync(keys.data(), d_keys, (size_t)total * 8, hipMemcpyDeviceToHost, c->stream));
        // the values follow the bitmap on the copy stream, in chunks of layers
        PNR_HIP(hipEventRecord(c->j8_start, c->stream));
        PNR_HIP(hipStreamWaitEvent(c->copy_stream, c->j8_start, 0));
        for (int k = 0; k < NCH; k++) {
            const int l0 = std::min(nl, k * per_chunk), l1 = std::min(nl, l0 + per_chunk);
            if (voff[l1] > voff[l0])
                PNR_HIP(hipMemcpyAsync(c->h_j8v + voff[l0], d_vals + voff[l0], (size_t)(voff[l1] - voff[l0]), hipMemcpyDeviceToHost, c->copy_stream));
            PNR_HIP(hipEventRecord(c->j8_ev[k], c->copy_stream));
        }
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
                i64 *kb = keys.data() + off[k];
                std::sort(kb, kb + cnt[k]); // unique keys: order fully defined (seed.cpp:632)
                (void)hipEventSynchronize(c->j8_ev[k / per_chunk]); // this layer's bits and values have arrived
                const unsigned long long *lb = h_bits + (size_t)k * h * wpr;
                const unsigned char *L8 = lf.load(lb, wpr, h_vals + voff[k], vmin[k]);
                lf.run(L8, kb, cnt[k], tol, vmin[k], acc[k]);
                lf.restore(lb, wpr);
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
