/* TEST INFRASTRUCTURE ONLY -- never linked into or called by the product (pnr_amd/): see pnr_oracle.h.
 *
 * CPU restatement of the soma path of the reference, plain C99:
 *   Frangi::imerode          frangi.cpp:880-968     separable xy erosion, window 2*ceil(rad)+1, clamp-to-edge
 *   Frangi::imgaussian (u8)  frangi.cpp:786-878     xy Gaussian: x pass accumulates f32, the y pass accumulates INTO the
 *                                                   unsigned char output (`I[i0] += K*G` truncates after every tap)
 *   maxentropy_th            toolbox.cpp:657-737    Kapur maximum-entropy threshold on the 256-bin histogram
 *   conn3d                   toolbox.cpp:245-509    26-connected regions of equal value by LIFO region growing, regions
 *                                                   numbered in raster order of their first voxel; centroid and mean radius
 *                                                   are running f32 means in VISIT order
 *   soma_extraction1         Advantra_plugin.cpp:1899-1915  binarise at > threshold, conn3d, one SOMA node per region
 *
 * Pinning: imerode and the u8 imgaussian are checked against the reference's own frangi.cpp compiled into
 * oracle/_ref/libpnr_ref.so (tests/test_oracle_golden.py).  toolbox.cpp includes the Vaa3D header v3d_message.h and
 * soma_extraction1 lives in Advantra_plugin.cpp (Qt + Vaa3D): neither can be built here, and the reference holds no
 * golden vectors for them -- maxentropy_th / conn3d / soma_extraction1 are PARITY UNPINNED restatements. */
#include "pnr_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;
static int clampi_(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* frangi.cpp:880-968.  The three loop regions of the reference (left-clamped / interior / right-clamped) are clamp-to-edge. */
void orc_imerode_xy(const uint8_t *I, int w, int h, int l, float rad, uint8_t *E)
{
    const int L = (int)ceil(rad);
    const i64 wh = (i64)w * h, n = wh * l;
    uint8_t *K = (uint8_t *)malloc((size_t)n);
    for (int z = 0; z < l; z++)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const i64 i0 = z * wh + (i64)y * w + x;
                uint8_t m = I[i0];
                for (int x1 = x - L; x1 <= x + L; x1++) {
                    const uint8_t v = I[z * wh + (i64)y * w + clampi_(x1, 0, w - 1)];
                    if (v < m) m = v;
                }
                K[i0] = m;
            }
    for (int z = 0; z < l; z++)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const i64 i0 = z * wh + (i64)y * w + x;
                uint8_t m = K[i0];
                for (int y1 = y - L; y1 <= y + L; y1++) {
                    const uint8_t v = K[z * wh + (i64)clampi_(y1, 0, h - 1) * w + x];
                    if (v < m) m = v;
                }
                E[i0] = m;
            }
    free(K);
}

/* frangi.cpp:786-878, in place */
void orc_imgaussian_u8_xy(uint8_t *I, int w, int h, int l, float sig)
{
    const int L = (int)ceil(3 * sig);
    const i64 wh = (i64)w * h, n = wh * l;
    float *G = (float *)malloc(sizeof(float) * (size_t)(2 * L + 1));
    float gn = 0;
    for (int i = -L; i <= L; i++) { /* exp(-(i*i)/(2*sig*sig)): int -> float, float division, the float overload of std::exp (:797) */
        G[i + L] = expf(-(float)(i * i) / (2 * sig * sig)); /* std::exp(float) */
        gn += G[i + L];
    }
    for (int i = 0; i < 2 * L + 1; i++) G[i] /= gn;
    float *K = (float *)malloc(sizeof(float) * (size_t)n);
    for (int z = 0; z < l; z++)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                float a = 0;
                for (int x1 = x - L; x1 <= x + L; x1++) a += I[z * wh + (i64)y * w + clampi_(x1, 0, w - 1)] * G[x1 - x + L];
                K[z * wh + (i64)y * w + x] = a;
            }
    for (int z = 0; z < l; z++)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                uint8_t a = 0; /* I[i0] = 0; I[i0] += K[i1] * G: the sum is converted back to unsigned char at every tap */
                for (int y1 = y - L; y1 <= y + L; y1++) a = (uint8_t)((float)a + K[z * wh + (i64)clampi_(y1, 0, h - 1) * w + x] * G[y1 - y + L]);
                I[z * wh + (i64)y * w + x] = a;
            }
    free(K);
    free(G);
}

/* toolbox.cpp:657-737 on a histogram (the reference builds it from the image first, :663-668) */
unsigned char orc_maxentropy_hist(const int64_t *hist)
{
    float sum = 0;
    for (int i = 0; i < 256; i++) sum += (float)(int)hist[i]; /* `int hist[]` added into a float (:672-674) */
    float nh[256], pT[256], hB[256], hW[256];
    for (int i = 0; i < 256; i++) nh[i] = (float)(int)hist[i] / sum;
    pT[0] = nh[0];
    for (int i = 1; i < 256; i++) pT[i] = pT[i - 1] + nh[i];
    const float eps = FLT_MIN;
    for (int t = 0; t < 256; t++) {
        if (pT[t] > eps) {
            float hh = 0;
            for (int i = 0; i <= t; i++)
                if (nh[i] > eps) hh -= nh[i] / pT[t] * logf(nh[i] / pT[t]); /* std::log(float) */
            hB[t] = hh;
        } else {
            hB[t] = 0;
        }
        const double pTW = 1 - pT[t]; /* f32 subtraction widened to double (:711) */
        if (pTW > eps) {
            float hh = 0;
            for (int i = t + 1; i < 256; i++)
                if (nh[i] > eps) hh = (float)((double)hh - nh[i] / pTW * log(nh[i] / pTW));
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

unsigned char orc_maxentropy_th(const uint8_t *img, int64_t size)
{
    int64_t hist[256];
    memset(hist, 0, sizeof(hist));
    for (i64 i = 0; i < size; i++) hist[img[i]]++;
    return orc_maxentropy_hist(hist);
}

/* toolbox.cpp:245-509 with maxNrRegions = INT_MAX.  Same visit order (LIFO, neighbours pushed in z,y,x order), same
 * running means; the whole-volume relabelling sweeps of the reference are replaced by the region's own voxel list. */
int64_t orc_conn3d(const uint8_t *inimg, int w, int h, int l, int32_t *lab, int diagonal, int values_over, int min_reg_size,
                   float *xc, float *yc, float *zc, float *rc, int64_t cap)
{
    const i64 wh = (i64)w * h, n = wh * l;
    uint8_t *state = (uint8_t *)calloc((size_t)n, 1); /* 0, 1 IN_QUEUE, 2 ADDED_TO_CURRENT_REGION, 3 IN_PREVIOUS_REGION */
    uint8_t *data = (uint8_t *)malloc((size_t)n);
    memcpy(data, inimg, (size_t)n);
    i64 qcap = 4096, *queue = (i64 *)malloc(sizeof(i64) * (size_t)qcap);
    i64 rcap = 4096, *reg = (i64 *)malloc(sizeof(i64) * (size_t)rcap);
    i64 nreg = 0, scan = 0;
    for (;;) {
        /* next starting point in raster order (:286-310); `scan` plays ignoreBefore{X,Y,Z} */
        i64 start = -1;
        for (; scan < n; scan++)
            if (state[scan] != 3 && (int)data[scan] > values_over) { start = scan; break; }
        if (start < 0) break;
        scan = start + 1;
        const int vint = data[start];
        i64 nq = 0, np = 0;
        state[start] = 1;
        queue[nq++] = start;
        float xmean = 0, ymean = 0, zmean = 0;
        while (nq > 0) {
            const i64 idx = queue[--nq];
            const int pz = (int)(idx / wh), py = (int)((idx % wh) / w), px = (int)((idx % wh) % w);
            state[idx] = 2;
            data[idx] = 0;
            if (np == rcap) { rcap *= 2; reg = (i64 *)realloc(reg, sizeof(i64) * (size_t)rcap); }
            reg[np++] = idx;
            const float t1 = (float)(np - 1) / (float)np, t2 = (float)(1.0 / (double)np);
            xmean = t1 * xmean + t2 * (float)px;
            ymean = t1 * ymean + t2 * (float)py;
            zmean = t1 * zmean + t2 * (float)pz;
            const int x0 = px - 1 < 0 ? 0 : px - 1, y0 = py - 1 < 0 ? 0 : py - 1, z0 = pz - 1 < 0 ? 0 : pz - 1;
            const int x1 = px + 1 > w - 1 ? w - 1 : px + 1, y1 = py + 1 > h - 1 ? h - 1 : py + 1, z1 = pz + 1 > l - 1 ? l - 1 : pz + 1;
            for (int z = z0; z <= z1; z++)
                for (int y = y0; y <= y1; y++)
                    for (int x = x0; x <= x1; x++) {
                        const int off = (x != px) + (y != py) + (z != pz);
                        if (off == 0 || (!diagonal && off > 1)) continue;
                        const i64 ni = (i64)w * ((i64)z * h + y) + x;
                        if ((int)data[ni] != vint || state[ni] != 0) continue;
                        state[ni] = 1;
                        if (nq == qcap) { qcap *= 2; queue = (i64 *)realloc(queue, sizeof(i64) * (size_t)qcap); }
                        queue[nq++] = ni;
                    }
        }
        if (np < min_reg_size) { /* too small: never looked at again, no label (:462-470) */
            for (i64 p = 0; p < np; p++) state[reg[p]] = 3;
            continue;
        }
        nreg++;
        float rmean = 0; /* mean distance to the centroid, running mean in visit order (:489-494) */
        for (i64 p = 1; p <= np; p++) {
            const i64 idx = reg[p - 1];
            const int pz = (int)(idx / wh), py = (int)((idx % wh) / w), px = (int)((idx % wh) % w);
            const float t1 = (float)(p - 1) / (float)p, t2 = (float)(1.0 / (double)p);
            const double dx = (double)((float)px - xmean), dy = (double)((float)py - ymean), dz = (double)((float)pz - zmean);
            rmean = (float)((double)(t1 * rmean) + (double)t2 * sqrt(dx * dx + dy * dy + dz * dz));
        }
        if (nreg <= cap) { xc[nreg - 1] = xmean; yc[nreg - 1] = ymean; zc[nreg - 1] = zmean; rc[nreg - 1] = rmean; }
        for (i64 p = 0; p < np; p++) { lab[reg[p]] = (int32_t)nreg; state[reg[p]] = 3; }
    }
    free(state); free(data); free(queue); free(reg);
    return nreg;
}

/* Advantra_plugin.cpp:2426-2448 + soma_extraction1 (:1899-1915): E8 = blur(erode(img)); threshold; regions -> smap labels
 * (1-based = index in the node list, which starts with the dummy node) and one (x, y, z, r) per region */
int64_t orc_soma_extract(const uint8_t *img, int w, int h, int l, int somaradius, uint8_t *E8, int *th_out, int32_t *smap,
                         float *nodes4, int64_t cap)
{
    const i64 n = (i64)w * h * l;
    orc_imerode_xy(img, w, h, l, (float)somaradius, E8);
    orc_imgaussian_u8_xy(E8, w, h, l, (float)somaradius);
    const unsigned char th = orc_maxentropy_th(E8, n);
    if (th_out) *th_out = th;
    uint8_t *bin = (uint8_t *)malloc((size_t)n);
    for (i64 i = 0; i < n; i++) { bin[i] = E8[i] > th ? 255 : 0; smap[i] = 0; }
    float *xc = (float *)malloc(sizeof(float) * 4 * (size_t)(cap > 0 ? cap : 1));
    float *yc = xc + cap, *zc = yc + cap, *rc = zc + cap;
    const i64 nreg = orc_conn3d(bin, w, h, l, smap, 1, 0, 1, xc, yc, zc, rc, cap);
    for (i64 k = 0; k < nreg && k < cap; k++) { nodes4[4 * k] = xc[k]; nodes4[4 * k + 1] = yc[k]; nodes4[4 * k + 2] = zc[k]; nodes4[4 * k + 3] = rc[k]; }
    free(bin); free(xc);
    return nreg;
}
