/*
 * pnr_oracle.c -- CPU restatement (plain C99) of the PNR/Advantra hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see pnr_oracle.h).  Single-threaded, scalar, the
 * same loop structure, precision choices (f32 / f64) and operation order as
 * the reference, so that integers come out identical and floats come out
 * bit-identical when built without FMA contraction:
 *     gcc -O2 -std=c99 -ffp-contract=off  (no -march=native, no -ffast-math)
 *
 * Intentional divergences from the reference (all documented in DESIGN.md):
 *   - 64-bit voxel indices (reference: int, overflows at 2^31 voxels).
 *   - systematic-resampling walk clamped to s <= np-1 (reference: unbounded,
 *     tracker.cpp:1087,1192).
 *   - RNG: the reference reseeds libc rand() with time(NULL) at the start of
 *     every SMC iteration (tracker.cpp:1003,1098); here the seed is a
 *     parameter, so every iteration replays the same glibc rand() stream.
 *   - seed sort ties (equal corr) broken by original index (std::sort in the
 *     reference is unstable, Advantra_plugin.cpp:2581).
 *
 * All file:line citations are relative to /root/reference/pnr-vaa3d/.
 */
#include "pnr_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;

/* Advantra_plugin.cpp:120-123 (the definition frangi.cpp:32 / tracker.cpp link to) */
static double round_half_away(double r) { return (r > 0.0) ? floor(r + 0.5) : ceil(r - 0.5); }

static int clampi(int x, int lo, int hi) { int c = (x < lo) ? lo : x; return (c > hi) ? hi : c; }

/* ====================================================================== */
/*  F1  Frangi::imgaussian (3-D)                     frangi.cpp:647-784   */
/* ====================================================================== */
static int gauss_taps(float sig, float *G /* >= 2*ceil(3 sig)+1 */)
{
    /* frangi.cpp:654-667: L = ceil(3 sig); G[i] = exp(-(i*i)/(2 sig sig)) in f32
     * (std::exp(float)), normalised by its f32 running sum. */
    int L = (int)ceilf(3 * sig);
    float norm = 0;
    for (int i = -L; i <= L; ++i) {
        G[i + L] = expf(-(i * i) / (2 * sig * sig));
        norm += G[i + L];
    }
    for (int i = 0; i < 2 * L + 1; ++i) G[i] /= norm;
    return L;
}

void orc_imgaussian3d(const uint8_t *I, int w, int h, int l, float sig, float zdist, float *F)
{
    float sigz = sig / zdist;                       /* frangi.cpp:651 */
    float *Gxy = (float *)malloc(sizeof(float) * (2 * (size_t)ceilf(3 * sig) + 3));
    float *Gz = (float *)malloc(sizeof(float) * (2 * (size_t)ceilf(3 * sigz) + 3));
    int Lxy = gauss_taps(sig, Gxy);
    int Lz = gauss_taps(sigz, Gz);                  /* frangi.cpp:669-680 */
    i64 wh = (i64)w * h, n = wh * l;
    float *K = (float *)malloc(sizeof(float) * (size_t)n);

    /* x pass, u8 -> f32 (frangi.cpp:683-714): clamp-to-edge, taps ascending,
     * separate multiply and add */
    for (int z = 0; z < l; ++z)
        for (int y = 0; y < h; ++y) {
            const uint8_t *row = I + z * wh + (i64)y * w;
            float *out = F + z * wh + (i64)y * w;
            for (int x = 0; x < w; ++x) {
                float acc = 0;
                for (int k = -Lxy; k <= Lxy; ++k)
                    acc += row[clampi(x + k, 0, w - 1)] * Gxy[k + Lxy];
                out[x] = acc;
            }
        }
    /* y pass F -> K (frangi.cpp:717-748) */
    for (int z = 0; z < l; ++z)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                float acc = 0;
                for (int k = -Lxy; k <= Lxy; ++k)
                    acc += F[z * wh + (i64)clampi(y + k, 0, h - 1) * w + x] * Gxy[k + Lxy];
                K[z * wh + (i64)y * w + x] = acc;
            }
    /* z pass K -> F with sigma/zdist (frangi.cpp:751-782) */
    for (int z = 0; z < l; ++z)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                float acc = 0;
                for (int k = -Lz; k <= Lz; ++k)
                    acc += K[(i64)clampi(z + k, 0, l - 1) * wh + (i64)y * w + x] * Gz[k + Lz];
                F[z * wh + (i64)y * w + x] = acc;
            }
    free(K);
    free(Gxy);
    free(Gz);
}

/* ====================================================================== */
/*  F2  Frangi::hessian3d                            frangi.cpp:291-390   */
/* ====================================================================== */
/* first difference along an axis of stride s at coordinate c in [0,n):
 * one-sided at the borders, 0.5*(f+ - f-) inside (frangi.cpp:307-311 etc.) */
static float diff1(const float *A, i64 i, i64 s, int c, int n)
{
    if (n == 1) return 0.0f; /* reference reads out of bounds here; never used (l>1,h>1,w>1) */
    if (c == 0) return A[i + s] - A[i];
    if (c < n - 1) return (float)(.5 * (A[i + s] - A[i - s]));
    return A[i] - A[i - s];
}

void orc_hessian3d(const float *F, int w, int h, int l, float sig,
                   float *Dzz, float *Dyy, float *Dyz, float *Dxx, float *Dxy, float *Dxz)
{
    i64 wh = (i64)w * h, n = wh * l;
    float *DD = (float *)malloc(sizeof(float) * (size_t)n);
    float s2 = sig * sig; /* "*= (sig*sig)" frangi.cpp:319 */
#define FORALL for (i64 i = 0; i < n; ++i) { int x = (int)(i % w); int z = (int)(i / wh); int y = (int)(i / w - (i64)z * h); (void)x; (void)y; (void)z;
    FORALL DD[i] = diff1(F, i, wh, z, l); }                               /* :305-311 */
    FORALL Dzz[i] = diff1(DD, i, wh, z, l); Dzz[i] *= s2; }               /* :313-320 */
    FORALL DD[i] = diff1(F, i, w, y, h); }                                /* :325-330 */
    FORALL Dyy[i] = diff1(DD, i, w, y, h); Dyy[i] *= s2;                  /* :332-339 */
           Dyz[i] = diff1(DD, i, wh, z, l); Dyz[i] *= s2; }               /* :340-346 */
    FORALL DD[i] = diff1(F, i, 1, x, w); }                                /* :352-357 */
    FORALL Dxx[i] = diff1(DD, i, 1, x, w); Dxx[i] *= s2;                  /* :361-368 */
           Dxy[i] = diff1(DD, i, w, y, h); Dxy[i] *= s2;                  /* :369-374 */
           Dxz[i] = diff1(DD, i, wh, z, l); Dxz[i] *= s2; }               /* :375-380 */
#undef FORALL
    free(DD);
}

/* ====================================================================== */
/*  F3  eigen_decomposition + tred2 + tql2          frangi.cpp:1269-1495  */
/*  (the public JAMA / EISPACK symmetric eigen-solver, n = 3, fp64)       */
/* ====================================================================== */
#define N3 3
static void householder_tridiag(double V[N3][N3], double d[N3], double e[N3])
{
    /* frangi.cpp:1309-1387 */
    for (int j = 0; j < N3; j++) d[j] = V[N3 - 1][j];
    for (int i = N3 - 1; i > 0; i--) {
        double scale = 0.0, h = 0.0;
        for (int k = 0; k < i; k++) scale = scale + fabs(d[k]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
            for (int j = 0; j < i; j++) {
                d[j] = V[i - 1][j];
                V[i][j] = 0.0;
                V[j][i] = 0.0;
            }
        } else {
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
            for (int j = 0; j < i; j++) e[j] = 0.0;
            for (int j = 0; j < i; j++) {
                f = d[j];
                V[j][i] = f;
                g = e[j] + V[j][j] * f;
                for (int k = j + 1; k <= i - 1; k++) {
                    g += V[k][j] * d[k];
                    e[k] += V[k][j] * f;
                }
                e[j] = g;
            }
            f = 0.0;
            for (int j = 0; j < i; j++) {
                e[j] /= h;
                f += e[j] * d[j];
            }
            double hh = f / (h + h);
            for (int j = 0; j < i; j++) e[j] -= hh * d[j];
            for (int j = 0; j < i; j++) {
                f = d[j];
                g = e[j];
                for (int k = j; k <= i - 1; k++) V[k][j] -= (f * e[k] + g * d[k]);
                d[j] = V[i - 1][j];
                V[i][j] = 0.0;
            }
        }
        d[i] = h;
    }
    for (int i = 0; i < N3 - 1; i++) {
        V[N3 - 1][i] = V[i][i];
        V[i][i] = 1.0;
        double h = d[i + 1];
        if (h != 0.0) {
            for (int k = 0; k <= i; k++) d[k] = V[k][i + 1] / h;
            for (int j = 0; j <= i; j++) {
                double g = 0.0;
                for (int k = 0; k <= i; k++) g += V[k][i + 1] * V[k][j];
                for (int k = 0; k <= i; k++) V[k][j] -= g * d[k];
            }
        }
        for (int k = 0; k <= i; k++) V[k][i + 1] = 0.0;
    }
    for (int j = 0; j < N3; j++) {
        d[j] = V[N3 - 1][j];
        V[N3 - 1][j] = 0.0;
    }
    V[N3 - 1][N3 - 1] = 1.0;
    e[0] = 0.0;
}

static double hyp2(double a, double b) { return sqrt(a * a + b * b); } /* frangi.cpp:1495 */

static void ql_implicit(double V[N3][N3], double d[N3], double e[N3])
{
    /* frangi.cpp:1390-1493 */
    for (int i = 1; i < N3; i++) e[i - 1] = e[i];
    e[N3 - 1] = 0.0;
    double f = 0.0, tst1 = 0.0;
    const double eps = 2.220446049250313e-16; /* pow(2.0,-52.0) */
    for (int l = 0; l < N3; l++) {
        double t = fabs(d[l]) + fabs(e[l]);
        tst1 = (tst1 > t) ? tst1 : t; /* MAX(a,b) ((a)>(b)?(a):(b)) */
        int m = l;
        while (m < N3) {
            if (fabs(e[m]) <= eps * tst1) break;
            m++;
        }
        if (m > l) {
            do {
                double g = d[l];
                double p = (d[l + 1] - g) / (2.0 * e[l]);
                double r = hyp2(p, 1.0);
                if (p < 0) r = -r;
                d[l] = e[l] / (p + r);
                d[l + 1] = e[l] * (p + r);
                double dl1 = d[l + 1];
                double h = g - d[l];
                for (int i = l + 2; i < N3; i++) d[i] -= h;
                f = f + h;
                p = d[m];
                double c = 1.0, c2 = c, c3 = c;
                double el1 = e[l + 1];
                double s = 0.0, s2 = 0.0;
                for (int i = m - 1; i >= l; i--) {
                    c3 = c2;
                    c2 = c;
                    s2 = s;
                    g = c * e[i];
                    h = c * p;
                    r = hyp2(p, e[i]);
                    e[i + 1] = s * r;
                    s = e[i] / r;
                    c = p / r;
                    p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    for (int k = 0; k < N3; k++) {
                        h = V[k][i + 1];
                        V[k][i + 1] = s * V[k][i] + c * h;
                        V[k][i] = c * V[k][i] - s * h;
                    }
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p;
                d[l] = c * p;
            } while (fabs(e[l]) > eps * tst1);
        }
        d[l] = d[l] + f;
        e[l] = 0.0;
    }
    /* ascending selection sort of eigenvalues with their columns */
    for (int i = 0; i < N3 - 1; i++) {
        int k = i;
        double p = d[i];
        for (int j = i + 1; j < N3; j++)
            if (d[j] < p) {
                k = j;
                p = d[j];
            }
        if (k != i) {
            d[k] = d[i];
            d[i] = p;
            for (int j = 0; j < N3; j++) {
                p = V[j][i];
                V[j][i] = V[j][k];
                V[j][k] = p;
            }
        }
    }
}

static void swap_cols(double V[N3][N3], double d[N3], double da[N3], int a, int b)
{
    double t = d[a]; d[a] = d[b]; d[b] = t;
    t = da[a]; da[a] = da[b]; da[b] = t;
    for (int r = 0; r < N3; r++) { t = V[r][a]; V[r][a] = V[r][b]; V[r][b] = t; }
}

static void eigen3(double A[N3][N3], double V[N3][N3], double d[N3])
{
    /* frangi.cpp:1269-1306 */
    double e[N3], da[N3];
    for (int i = 0; i < N3; i++)
        for (int j = 0; j < N3; j++) V[i][j] = A[i][j];
    householder_tridiag(V, d, e);
    ql_implicit(V, d, e);
    /* re-sort by |lambda| ascending (frangi.cpp:1286-1304): two conditional swaps */
    da[0] = fabs(d[0]); da[1] = fabs(d[1]); da[2] = fabs(d[2]);
    if ((da[0] >= da[1]) && (da[0] > da[2])) swap_cols(V, d, da, 0, 2);
    else if ((da[1] >= da[0]) && (da[1] > da[2])) swap_cols(V, d, da, 1, 2);
    if (da[0] > da[1]) swap_cols(V, d, da, 0, 1);
}

void orc_eigen3(const double A[9], double V[9], double d[3])
{
    double a[N3][N3], v[N3][N3];
    for (int i = 0; i < 9; i++) a[i / 3][i % 3] = A[i];
    eigen3(a, v, d);
    for (int i = 0; i < 9; i++) V[i] = v[i / 3][i % 3];
}

/* ====================================================================== */
/*  F4  vesselness + max over scales                 frangi.cpp:152-289   */
/* ====================================================================== */
static uint8_t quant_dir(double v)
{
    /* frangi.cpp:240-242 */
    int val = (int)round_half_away(((v + 1) / 2) * 255);
    val = (val < 0) ? 0 : (val > 255) ? 255 : val;
    return (uint8_t)val;
}

void orc_frangi3d(const uint8_t *I, int w, int h, int l, const float *sigs, int nsig, float zdist,
                  float alpha, float beta, float C,
                  float *J, float *Jmin, float *Jmax, uint8_t *Vx, uint8_t *Vy, uint8_t *Vz)
{
    i64 n = (i64)w * h * l;
    float *D[6], *F = (float *)malloc(sizeof(float) * (size_t)n);
    for (int k = 0; k < 6; k++) D[k] = (float *)malloc(sizeof(float) * (size_t)n);
    float *Dzz = D[0], *Dyy = D[1], *Dyz = D[2], *Dxx = D[3], *Dxy = D[4], *Dxz = D[5];
    *Jmin = FLT_MAX;
    *Jmax = -FLT_MAX;
    for (int si = 0; si < nsig; ++si) {
        orc_imgaussian3d(I, w, h, l, sigs[si], zdist, F);
        orc_hessian3d(F, w, h, l, sigs[si], Dzz, Dyy, Dyz, Dxx, Dxy, Dxz);
        for (i64 i = 0; i < n; ++i) {
            double Ma[3][3], Ve[3][3], ev[3];
            Ma[0][0] = Dxx[i]; Ma[0][1] = Dxy[i]; Ma[0][2] = Dxz[i];
            Ma[1][0] = Dxy[i]; Ma[1][1] = Dyy[i]; Ma[1][2] = Dyz[i];
            Ma[2][0] = Dxz[i]; Ma[2][1] = Dyz[i]; Ma[2][2] = Dzz[i];
            eigen3(Ma, Ve, ev);
            double L2 = ev[1], L3 = ev[2];
            double a1 = fabs(ev[0]), a2 = fabs(L2), a3 = fabs(L3);
            double Ra = a2 / a3;
            double Rb = a1 / sqrt(a2 * a3);
            double S = sqrt(a1 * a1 + a2 * a2 + a3 * a3);
            /* 2*alpha*alpha etc. are f32 products (frangi.cpp:214-216) */
            double expRa = (1 - exp(-((Ra * Ra) / (2 * alpha * alpha))));
            double expRb = exp(-((Rb * Rb) / (2 * beta * beta)));
            double expS = (1 - exp(-(S * S) / (2 * C * C)));
            double vox = expRa * expRb * expS;
            vox = (L2 > 0) ? 0 : vox; /* blackwhite == false, frangi.cpp:54,226-227 */
            vox = (L3 > 0) ? 0 : vox;
            vox = isnan(vox) ? 0 : vox;
            if (si == 0 || vox > J[i]) {
                J[i] = (float)vox;
                if (J[i] < *Jmin) *Jmin = J[i];
                if (J[i] > *Jmax) *Jmax = J[i];
                Vx[i] = quant_dir(Ve[0][0]);
                Vy[i] = quant_dir(Ve[1][0]);
                Vz[i] = quant_dir(Ve[2][0]);
            }
        }
    }
    free(F);
    for (int k = 0; k < 6; k++) free(D[k]);
}

/* ====================================================================== */
/*  F5  J -> J8                             Advantra_plugin.cpp:2499-2512 */
/* ====================================================================== */
void orc_j8(const float *J, i64 n, float Jmin, float Jmax, uint8_t *J8)
{
    if (fabsf(Jmax - Jmin) <= FLT_MIN) {
        memset(J8, 0, (size_t)n);
        return;
    }
    for (i64 i = 0; i < n; ++i) {
        int val = (int)round_half_away(((J[i] - Jmin) / (Jmax - Jmin)) * 255);
        val = (val < 0) ? 0 : (val > 255) ? 255 : val;
        J8[i] = (uint8_t)val;
    }
}

/* ====================================================================== */
/*  S1  SeedExtractor::extractSeeds (ImageJ MaximumFinder per z layer)    */
/*                                         seed.cpp:556-791, :1027-1059   */
/* ====================================================================== */
enum { T_MAXIMUM = 1, T_LISTED = 2, T_PROCESSED = 4, T_MAX_AREA = 8, T_EQUAL = 16, T_MAX_POINT = 32 };
static const int DX8[8] = {0, 1, 1, 1, 0, -1, -1, -1};
static const int DY8[8] = {-1, -1, 0, 1, 1, 1, 0, -1};

static int nb_inside(int x, int y, int d, int w, int h)
{
    int xm = w - 1, ym = h - 1; /* seed.cpp:1027-1049 */
    switch (d) {
    case 0: return y > 0;
    case 1: return x < xm && y > 0;
    case 2: return x < xm;
    case 3: return x < xm && y < ym;
    case 4: return y < ym;
    case 5: return x > 0 && y < ym;
    case 6: return x > 0;
    case 7: return x > 0 && y > 0;
    }
    return 0;
}

static int cmp_i64(const void *a, const void *b)
{
    i64 x = *(const i64 *)a, y = *(const i64 *)b;
    return (x > y) - (x < y);
}

i64 orc_extract_seeds(double tolerance, const uint8_t *J8, int w, int h, int l,
                      const uint8_t *Vx, const uint8_t *Vy, const uint8_t *Vz,
                      float *seeds_out, i64 cap)
{
    i64 wh = (i64)w * h, nseeds = 0;
    int dirOffset[8] = {-w, -w + 1, +1, +w + 1, +w, +w - 1, -1, -w - 1};
    uint8_t *types = (uint8_t *)malloc((size_t)wh);
    int *pList = (int *)malloc(sizeof(int) * (size_t)wh);
    i64 *maxPoints = (i64 *)malloc(sizeof(i64) * (size_t)wh);

    for (int z = 0; z < l; ++z) {
        const uint8_t *L8 = J8 + z * wh;
        memset(types, 0, (size_t)wh);
        float gmin = FLT_MAX, gmax = -FLT_MAX; /* seed.cpp:578-586 */
        for (i64 i = 0; i < wh; ++i) {
            float v = (float)(int)L8[i];
            if (gmin > v) gmin = v;
            if (gmax < v) gmax = v;
        }
        /* local maxima: 8-neighbour, border pixels and the layer minimum skipped (:589-614) */
        int nMax = 0;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                float v = L8[(i64)y * w + x];
                if (v == gmin) continue;
                if (x == 0 || x == w - 1 || y == 0 || y == h - 1) continue;
                int isMax = 1;
                for (int d = 0; d < 8; d++) {
                    float vn = L8[(i64)(y + DY8[d]) * w + (x + DX8[d])];
                    if (vn > v) { isMax = 0; break; }
                }
                if (isMax) { types[(i64)y * w + x] = T_MAXIMUM; nMax++; }
            }
        /* sort keys: value in the upper 32 bits, pixel offset in the lower (:616-632) */
        float vFactor = (float)(2e9 / (gmax - gmin));
        int iMax = 0;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int p = y * w + x;
                if (types[p] == T_MAXIMUM) {
                    float fValue = L8[p];
                    int iValue = (int)((fValue - gmin) * vFactor);
                    maxPoints[iMax++] = (i64)(((uint64_t)(i64)iValue << 32) | (uint32_t)p);
                }
            }
        qsort(maxPoints, (size_t)nMax, sizeof(i64), cmp_i64); /* keys unique => order defined */

        const float maxSortingError = 0;
        memset(pList, 0, sizeof(int) * (size_t)wh);
        for (iMax = nMax - 1; iMax >= 0; iMax--) { /* :643-782 */
            int offset0 = (int)maxPoints[iMax];
            if ((types[offset0] & T_PROCESSED) != 0) continue;
            int x0 = offset0 % w, y0 = offset0 / w;
            float v0 = L8[(i64)y0 * w + x0];
            int sortingError;
            do {
                pList[0] = offset0;
                types[offset0] |= (T_EQUAL | T_LISTED);
                int listLen = 1, listI = 0;
                int isEdgeMaximum = (x0 == 0 || x0 == w - 1 || y0 == 0 || y0 == h - 1);
                sortingError = 0;
                int maxPossible = 1;
                double xEqual = x0, yEqual = y0;
                int nEqual = 1;
                do {
                    int offset = pList[listI];
                    int x = offset % w, y = offset / w;
                    int isInner = (y != 0 && y != h - 1) && (x != 0 && x != w - 1);
                    for (int d = 0; d < 8; d++) {
                        int offset2 = offset + dirOffset[d];
                        if ((isInner || nb_inside(x, y, d, w, h)) && (types[offset2] & T_LISTED) == 0) {
                            if ((types[offset2] & T_PROCESSED) != 0) { maxPossible = 0; break; }
                            int x2 = x + DX8[d], y2 = y + DY8[d];
                            float v2 = L8[(i64)y2 * w + x2];
                            if (v2 > v0 + maxSortingError) { maxPossible = 0; break; }
                            else if (v2 >= v0 - (float)tolerance) {
                                if (v2 > v0) { /* unreachable with maxSortingError == 0; kept for fidelity */
                                    sortingError = 1;
                                    offset0 = offset2; v0 = v2; x0 = x2; y0 = y2;
                                }
                                pList[listLen] = offset2;
                                listLen++;
                                types[offset2] |= T_LISTED;
                                if (x2 == 0 || x2 == w - 1 || y2 == 0 || y2 == h - 1) {
                                    isEdgeMaximum = 1;
                                    maxPossible = 0; /* excludeEdgesNow */
                                    break;
                                }
                                if (v2 == v0) {
                                    types[offset2] |= T_EQUAL;
                                    xEqual += x2; yEqual += y2; nEqual++;
                                }
                            }
                        }
                    }
                    listI++;
                } while (listI < listLen);

                if (sortingError) {
                    for (listI = 0; listI < listLen; listI++) types[pList[listI]] = 0;
                } else {
                    int resetMask = ~(maxPossible ? T_LISTED : (T_LISTED | T_EQUAL));
                    xEqual /= nEqual;
                    yEqual /= nEqual;
                    double minDist2 = 1e20;
                    int nearestI = 0;
                    for (listI = 0; listI < listLen; listI++) {
                        int offset = pList[listI];
                        int x = offset % w, y = offset / w;
                        types[offset] &= resetMask;
                        types[offset] |= T_PROCESSED;
                        if (maxPossible) {
                            types[offset] |= T_MAX_AREA;
                            if ((types[offset] & T_EQUAL) != 0) {
                                double dist2 = (xEqual - x) * (double)(xEqual - x) + (yEqual - y) * (double)(yEqual - y);
                                if (dist2 < minDist2) { minDist2 = dist2; nearestI = listI; }
                            }
                        }
                    }
                    if (maxPossible) {
                        int offset = pList[nearestI];
                        types[offset] |= T_MAX_POINT;
                        if (!isEdgeMaximum) {
                            int x = offset % w, y = offset / w;
                            i64 si = z * wh + (i64)y * w + x;
                            /* direction decode, seed.cpp:767-771 */
                            float Ux = (((float)Vx[si] / 255) * 2) - 1;
                            float Uy = (((float)Vy[si] / 255) * 2) - 1;
                            float Uz = (((float)Vz[si] / 255) * 2) - 1;
                            float Un = (float)sqrt((double)Ux * (double)Ux + (double)Uy * (double)Uy + (double)Uz * (double)Uz);
                            if (nseeds < cap) {
                                float *s = seeds_out + nseeds * 8;
                                s[0] = (float)x; s[1] = (float)y; s[2] = (float)z;
                                s[3] = Ux / Un; s[4] = Uy / Un; s[5] = Uz / Un;
                                s[6] = 0; s[7] = 0;
                            }
                            nseeds++;
                        }
                    }
                }
            } while (sortingError);
        }
    }
    free(types);
    free(pList);
    free(maxPoints);
    return nseeds;
}

/* ====================================================================== */
/*  glibc rand() (TYPE_3 additive feedback, r[i] = r[i-3] + r[i-31])       */
/*  contract of SURVEY section 5 "RNG"; checked against libc in tests      */
/* ====================================================================== */
void orc_glibc_rand(uint32_t seed, int n, uint32_t *out)
{
    int32_t r[34];
    uint32_t *st = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(344 + n));
    if (seed == 0) seed = 1;
    r[0] = (int32_t)seed;
    for (int i = 1; i < 31; i++) {
        int64_t hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
        int64_t word = 16807 * lo - 2836 * hi;
        if (word < 0) word += 2147483647;
        r[i] = (int32_t)word;
    }
    for (int i = 0; i < 31; i++) st[i] = (uint32_t)r[i];
    for (int i = 31; i < 34; i++) st[i] = st[i - 31];
    for (int i = 34; i < 344 + n; i++) st[i] = st[i - 31] + st[i - 3];
    for (int k = 0; k < n; k++) out[k] = st[344 + k] >> 1;
    free(st);
}

/* ====================================================================== */
/*  T0  Tracker::Tracker tables                      tracker.cpp:79-527   */
/* ====================================================================== */
#define ORC_NDIR3D 50
#define ORC_NDIR2D 30 /* tracker.cpp:27 */
#define ORC_RAND_MAX 2147483647

struct orc_tracker {
    int nsig, step, npcles, niter, sz, ndir, nodespervol, is2d;
    float sig[16];
    float kappa, znccth, Kc, neff_ratio, zDist;
    /* model2_* (tracker.cpp:178-231) */
    int M[16];
    float *mvuw[16], *mwgt[16];
    float mavg[16];
    /* prediction tables */
    float *p, *u, *d, *d0, *w0, *w0_cws, *v, *w, *w_cws;
    uint32_t *rng;
    /* filter state */
    float *xfilt;   /* niter x np x 9 */
    int *idxres;    /* niter x np */
    float *neff;    /* niter */
    float *xc;      /* niter x 8 */
    float *prior, *lhood, *res_csw;
};

static double bessel_i0(double x)
{
    /* tracker.cpp:2254-2270 (Numerical-Recipes polynomial) */
    double ax = fabs(x), ans, y;
    if (ax < 3.75) {
        y = x / 3.75, y = y * y;
        ans = 1.0 + y * (3.5156229 + y * (3.0899424 + y * (1.2067492 + y * (0.2659732 + y * (0.360768e-1 + y * 0.45813e-2)))));
    } else {
        y = 3.75 / ax;
        ans = (exp(ax) / sqrt(ax)) * (0.39894228 + y * (0.1328592e-1 + y * (0.225319e-2 + y * (-0.157565e-2 + y * (0.916281e-2 + y * (-0.2057706e-1 + y * (0.2635537e-1 + y * (-0.1647633e-1 + y * 0.392377e-2))))))));
    }
    return ans;
}

orc_tracker *orc_tracker_new(const float *sigs, int nsig, int step, int npcles, int niter,
                             float kappa, float znccth, float Kc, float neff_ratio,
                             float zdist, int nodespervol, uint32_t rng_seed)
{
    return orc_tracker_new2(sigs, nsig, step, npcles, niter, kappa, znccth, Kc, neff_ratio, zdist, nodespervol, rng_seed, 0);
}

/* is2d = (P == 1) selects the 2-D branches of the constructor (Advantra_plugin.cpp:2526) */
orc_tracker *orc_tracker_new2(const float *sigs, int nsig, int step, int npcles, int niter,
                              float kappa, float znccth, float Kc, float neff_ratio,
                              float zdist, int nodespervol, uint32_t rng_seed, int is2d)
{
    if (nsig < 1 || nsig > 16) return NULL;
    orc_tracker *t = (orc_tracker *)calloc(1, sizeof(*t));
    t->nsig = nsig; t->step = step; t->npcles = npcles; t->niter = niter;
    t->kappa = kappa; t->znccth = znccth; t->Kc = Kc; t->neff_ratio = neff_ratio;
    t->zDist = zdist; t->nodespervol = nodespervol; t->is2d = is2d ? 1 : 0; t->ndir = is2d ? ORC_NDIR2D : ORC_NDIR3D;
    for (int i = 0; i < nsig; i++) t->sig[i] = sigs[i];

    /* ---- model2 templates, 3-D branch (tracker.cpp:210-231) ---- */
    const int model2_N = 12;
    for (int i = 0; i < nsig; i++) {
        int V2 = (int)roundf(1 * sigs[i]);
        int U2 = (int)roundf(3 * sigs[i]);
        int W2 = (int)roundf(3 * sigs[i]);
        float Vs = (float)((3.0 * sigs[i]) / model2_N);
        Vs = (Vs < 1.0) ? 1.0f : Vs;
        float vlim = V2 + FLT_MIN, ulim = U2 + FLT_MIN, wlim = W2 + FLT_MIN;
        int cnt = 0;
        if (is2d) { /* 2-D branch (tracker.cpp:191-208): offsets (vv, uu, 0), weight exp(-uu^2 / 2 sig^2) */
            for (float vv = (float)-V2; vv <= vlim; vv += Vs)
                for (float uu = (float)-U2; uu <= ulim; uu += Vs) cnt++;
            t->M[i] = cnt;
            t->mvuw[i] = (float *)malloc(sizeof(float) * 3 * (size_t)cnt);
            t->mwgt[i] = (float *)malloc(sizeof(float) * (size_t)cnt);
            float avg2 = 0.0f;
            int k2 = 0;
            for (float vv = (float)-V2; vv <= vlim; vv += Vs)
                for (float uu = (float)-U2; uu <= ulim; uu += Vs) {
                    float value = (float)exp(-(uu * uu) / (2 * pow((double)sigs[i], 2)));
                    t->mwgt[i][k2] = value;
                    t->mvuw[i][3 * k2 + 0] = vv;
                    t->mvuw[i][3 * k2 + 1] = uu;
                    t->mvuw[i][3 * k2 + 2] = 0;
                    avg2 += value;
                    k2++;
                }
            avg2 /= cnt;
            t->mavg[i] = avg2;
            continue;
        }
        for (float vv = (float)-V2; vv <= vlim; vv += Vs)
            for (float uu = (float)-U2; uu <= ulim; uu += Vs)
                for (float ww = (float)-W2; ww <= wlim; ww += Vs) cnt++;
        t->M[i] = cnt;
        t->mvuw[i] = (float *)malloc(sizeof(float) * 3 * (size_t)cnt);
        t->mwgt[i] = (float *)malloc(sizeof(float) * (size_t)cnt);
        float avg = 0.0f;
        int k = 0;
        for (float vv = (float)-V2; vv <= vlim; vv += Vs)
            for (float uu = (float)-U2; uu <= ulim; uu += Vs)
                for (float ww = (float)-W2; ww <= wlim; ww += Vs) {
                    /* exp(double): f32 numerator over 2*pow(sig,2) in f64 */
                    float value = (float)exp(-((uu * uu) + (ww * ww)) / (2 * pow((double)sigs[i], 2)));
                    t->mwgt[i][k] = value;
                    t->mvuw[i][3 * k + 0] = vv;
                    t->mvuw[i][3 * k + 1] = uu;
                    t->mvuw[i][3 * k + 2] = ww;
                    avg += value;
                    k++;
                }
        avg /= cnt;
        t->mavg[i] = avg;
    }

    /* ---- prediction offsets (tracker.cpp:375-438) ---- */
    int R = 2 * step, sz = 0;
    const int Rz = is2d ? 0 : R; /* 2-D: dz = 0 only (tracker.cpp:383-387) */
    for (int dx = -R; dx <= R; ++dx)
        for (int dy = -R; dy <= R; ++dy)
            for (int dz = -Rz; dz <= Rz; ++dz)
                if (dx * dx + dy * dy + dz * dz <= R * R && dx * dx + dy * dy + dz * dz > 0) sz++;
    t->sz = sz;
    t->p = (float *)malloc(sizeof(float) * 3 * (size_t)sz);
    t->u = (float *)malloc(sizeof(float) * 3 * (size_t)sz);
    t->d = (float *)malloc(sizeof(float) * (size_t)sz);
    t->d0 = (float *)malloc(sizeof(float) * (size_t)sz);
    t->w0 = (float *)malloc(sizeof(float) * (size_t)sz);
    t->w0_cws = (float *)malloc(sizeof(float) * (size_t)sz);
    float w0sum = 0;
    int i = 0;
    for (int dx = -R; dx <= R; ++dx)
        for (int dy = -R; dy <= R; ++dy)
            for (int dz = -Rz; dz <= Rz; ++dz) {
                if (!(dx * dx + dy * dy + dz * dz <= R * R && dx * dx + dy * dy + dz * dz > 0)) continue;
                float *p = t->p + 3 * i, *u = t->u + 3 * i;
                p[0] = (float)dx;
                p[1] = (float)dy;
                p[2] = dz / zdist;
                t->d[i] = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
                t->d0[i] = sqrtf((float)dx * dx + dy * dy + dz * dz);
                u[0] = p[0] / t->d[i];
                u[1] = p[1] / t->d[i];
                u[2] = p[2] / t->d[i];
                t->w0[i] = (float)exp(-pow((double)t->d[i], 2) / (2 * pow(step / 3.0, 2)));
                w0sum += t->w0[i];
                i++;
            }
    for (i = 0; i < sz; ++i) {
        t->w0[i] /= w0sum;
        t->w0_cws[i] = t->w0[i] + ((i == 0) ? 0 : t->w0_cws[i - 1]);
    }

    /* ---- 50 sphere directions (tracker.cpp:770-805) ---- */
    int nd = t->ndir;
    t->v = (float *)malloc(sizeof(float) * 3 * (size_t)nd);
    double phi_k = 0, phi_k_1 = 0;
    for (int k = 0; k < nd; k++) {
        if (is2d) { /* 30 directions on the circle (tracker.cpp:776-783): float angle, std::cos / std::sin float overloads */
            float ang1 = (float)(0.0 + k * ((2 * 3.14) / (float)nd));
            t->v[3 * k + 0] = cosf(ang1);
            t->v[3 * k + 1] = sinf(ang1);
            t->v[3 * k + 2] = 0;
            continue;
        }
        double h_k = 1 - 2 * ((double)k / (nd - 1));
        double theta_k = acos(h_k);
        if (k == 0 || k == (nd - 1)) {
            phi_k = 0;
            phi_k_1 = 0;
        } else {
            phi_k = phi_k_1 + 3.6 / (sqrtf((float)nd) * sqrt(1 - h_k * h_k));
            phi_k_1 = phi_k;
        }
        t->v[3 * k + 0] = (float)(sin(theta_k) * cos(phi_k));
        t->v[3 * k + 1] = (float)(sin(theta_k) * sin(phi_k));
        t->v[3 * k + 2] = (float)cos(theta_k);
    }

    /* ---- per-direction von Mises x radial prior (tracker.cpp:440-476) ---- */
    t->w = (float *)malloc(sizeof(float) * (size_t)nd * sz);
    t->w_cws = (float *)malloc(sizeof(float) * (size_t)nd * sz);
    for (int a = 0; a < nd; a++) {
        float wsum = 0;
        float *wa = t->w + (size_t)a * sz, *ca = t->w_cws + (size_t)a * sz;
        for (int j = 0; j < sz; j++) {
            double rad = exp(-pow((double)(t->d0[j] - step), 2) / (2 * pow(step / 3.0, 2)));
            double dotp = t->v[3 * a + 0] * t->u[3 * j + 0] + t->v[3 * a + 1] * t->u[3 * j + 1] + t->v[3 * a + 2] * t->u[3 * j + 2];
            dotp = (dotp > 1) ? 1 : (dotp < -1) ? -1 : dotp;
            double circ = exp(kappa * dotp) / (2.0 * 3.14 * bessel_i0(kappa));
            wa[j] = (float)(circ * rad);
            wsum += wa[j];
        }
        for (int j = 0; j < sz; j++) {
            wa[j] = wa[j] / wsum;
            ca[j] = wa[j] + ((j == 0) ? 0 : ca[j - 1]);
        }
    }

    t->rng = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(npcles + 1));
    orc_glibc_rand(rng_seed, npcles + 1, t->rng);

    t->xfilt = (float *)calloc((size_t)niter * npcles * 9, sizeof(float));
    t->idxres = (int *)malloc(sizeof(int) * (size_t)niter * npcles);
    for (i64 k = 0; k < (i64)niter * npcles; k++) t->idxres[k] = -2147483647;
    t->neff = (float *)calloc((size_t)niter, sizeof(float));
    t->xc = (float *)calloc((size_t)niter * 8, sizeof(float));
    t->prior = (float *)calloc((size_t)npcles, sizeof(float));
    t->lhood = (float *)calloc((size_t)npcles, sizeof(float));
    t->res_csw = (float *)calloc((size_t)npcles, sizeof(float));
    return t;
}

void orc_tracker_free(orc_tracker *t)
{
    if (!t) return;
    for (int i = 0; i < t->nsig; i++) { free(t->mvuw[i]); free(t->mwgt[i]); }
    free(t->p); free(t->u); free(t->d); free(t->d0); free(t->w0); free(t->w0_cws);
    free(t->v); free(t->w); free(t->w_cws); free(t->rng);
    free(t->xfilt); free(t->idxres); free(t->neff); free(t->xc);
    free(t->prior); free(t->lhood); free(t->res_csw);
    free(t);
}

int orc_tracker_sz(const orc_tracker *t) { return t->sz; }
int orc_tracker_ndir(const orc_tracker *t) { return t->ndir; }
const float *orc_tracker_p(const orc_tracker *t) { return t->p; }
const float *orc_tracker_u(const orc_tracker *t) { return t->u; }
const float *orc_tracker_w0(const orc_tracker *t) { return t->w0; }
const float *orc_tracker_w0_cws(const orc_tracker *t) { return t->w0_cws; }
const float *orc_tracker_v(const orc_tracker *t) { return t->v; }
const float *orc_tracker_w(const orc_tracker *t) { return t->w; }
const float *orc_tracker_w_cws(const orc_tracker *t) { return t->w_cws; }
int orc_tracker_model_count(const orc_tracker *t, int s) { return t->M[s]; }
const float *orc_tracker_model_vuw(const orc_tracker *t, int s) { return t->mvuw[s]; }
const float *orc_tracker_model_wgt(const orc_tracker *t, int s) { return t->mwgt[s]; }
float orc_tracker_model_avg(const orc_tracker *t, int s) { return t->mavg[s]; }
const uint32_t *orc_tracker_rng(const orc_tracker *t) { return t->rng; }

/* ====================================================================== */
/*  T2  Tracker::interp (3-D branch)               tracker.cpp:2138-2215  */
/* ====================================================================== */
static float clampf(float x, float lo, float hi) { float c = (x < lo) ? lo : x; return (c > hi) ? hi : c; }

float orc_interp(float _x, float _y, float _z, const uint8_t *img, int width, int height, int length)
{
    /* upper clamp bound "dim-1.001" is a double literal expression converted to float */
    float xc = clampf(_x, 0, (float)(width - 1.001));
    int x1 = (int)xc, x2 = x1 + 1;
    float xf = xc - x1;
    float yc = clampf(_y, 0, (float)(height - 1.001));
    int y1 = (int)yc, y2 = y1 + 1;
    float yf = yc - y1;
    if (length == 1) /* 2-D branch: _z is not used (tracker.cpp:2152-2175) */
        return (1 - yf) * ((1 - xf) * img[(i64)y1 * width + x1] + xf * img[(i64)y1 * width + x2]) +
               (yf) * ((1 - xf) * img[(i64)y2 * width + x1] + xf * img[(i64)y2 * width + x2]);
    float zc = clampf(_z, 0, (float)(length - 1.001));
    int z1 = (int)zc, z2 = z1 + 1;
    float zf = zc - z1;
    i64 wh = (i64)width * height;
    const uint8_t *a = img + z1 * wh, *b = img + z2 * wh;
    i64 r1 = (i64)y1 * width, r2 = (i64)y2 * width;
    return (1 - zf) * ((1 - yf) * ((1 - xf) * a[r1 + x1] + xf * a[r1 + x2]) +
                       (yf) * ((1 - xf) * a[r2 + x1] + xf * a[r2 + x2])) +
           (zf) * ((1 - yf) * ((1 - xf) * b[r1 + x1] + xf * b[r1 + x2]) +
                   (yf) * ((1 - xf) * b[r2 + x1] + xf * b[r2 + x2]));
}

/* ====================================================================== */
/*  T1  Tracker::znccBBB                           tracker.cpp:1891-1964  */
/* ====================================================================== */
float orc_zncc(orc_tracker *t, float _x, float _y, float _z, float _vx, float _vy, float _vz,
               const uint8_t *img, int w, int h, int l, float *sig_out)
{
    float ux, uy, uz, wx, wy, wz;
    float nrm = (float)sqrt((double)_vx * (double)_vx + (double)_vy * (double)_vy); /* pow(f,2) is f64 */
    if (nrm > 0.0001) {
        int sg = (_vy < 0) ? -1 : 1;
        ux = sg * (_vy / nrm);
        uy = -sg * (_vx / nrm);
        uz = 0;
    } else {
        ux = 1; uy = 0; uz = 0;
    }
    if (t->is2d) { /* tracker.cpp:1908-1912 */
        wx = 0; wy = 0; wz = 0;
    } else {
        wx = uy * _vz - uz * _vy;
        wy = -ux * _vz + uz * _vx;
        wz = ux * _vy - uy * _vx;
    }

    float out_corr = -FLT_MAX;
    static float *buf = NULL;
    static int bufcap = 0;
    for (int s = 0; s < t->nsig; ++s) {
        int M = t->M[s];
        if (M > bufcap) { free(buf); buf = (float *)malloc(sizeof(float) * (size_t)M); bufcap = M; }
        const float *vuw = t->mvuw[s], *wgt = t->mwgt[s];
        float ag = 0;
        for (int k = 0; k < M; ++k) {
            float ov = vuw[3 * k], ou = vuw[3 * k + 1], ow = vuw[3 * k + 2];
            float x = _x + ov * (-_vx) + ou * ux + ow * wx;
            float y = _y + ov * (-_vy) + ou * uy + ow * wy;
            float z = _z + ov * (-_vz) + ou * uz + ow * wz;
            buf[k] = orc_interp(x, y, z, img, w, h, l);
            ag += buf[k];
        }
        ag /= M;
        float corra = 0, corrb = 0, corrc = 0;
        float avg = t->mavg[s];
        for (int k = 0; k < M; ++k) {
            corra += (buf[k] - ag) * (wgt[k] - avg);
            float di = buf[k] - ag, dw = wgt[k] - avg;
            corrb = (float)(corrb + (double)di * (double)di); /* corrb += pow(f32,2) : f64 add, f32 store */
            corrc = (float)(corrc + (double)dw * (double)dw);
        }
        float corr_val = (corrb * corrc > FLT_MIN) ? corra / sqrtf(corrb * corrc) : 0;
        if (corr_val > out_corr) {
            out_corr = corr_val;
            *sig_out = t->sig[s];
        }
    }
    return out_corr;
}

/* ====================================================================== */
/*  T3-T7  iter0New / iterINew                     tracker.cpp:1001-1198  */
/* ====================================================================== */
enum { XF_X, XF_Y, XF_Z, XF_VX, XF_VY, XF_VZ, XF_W, XF_CORR, XF_SIG };   /* struct X, tracker.h:13-17 */
enum { XC_X, XC_Y, XC_Z, XC_VX, XC_VY, XC_VZ, XC_SIG, XC_CORR };        /* struct X_est, tracker.h:19-23 */

static int getdirection(const orc_tracker *t, float vx, float vy, float vz)
{
    int idx = -1; /* tracker.cpp:751-768: first maximum wins */
    float maxdotp = -FLT_MAX;
    for (int i = 0; i < t->ndir; i++) {
        float c = vx * t->v[3 * i] + vy * t->v[3 * i + 1] + vz * t->v[3 * i + 2];
        if (c > maxdotp) { maxdotp = c; idx = i; }
    }
    return idx;
}

/* shared tail of iter0New/iterINew: normalise, N_eff, CDF, centroid, stop tests, resampling.
 * `prevw` NULL => weights start from 1/np. rdraw = RNG draw used for resampling. */
static int smc_finish(orc_tracker *t, int it, const float *prevw, float wnorm_prior,
                      const uint8_t *img, int w, int h, int l, uint32_t rdraw, int *stop)
{
    int np = t->npcles;
    float *xf = t->xfilt + (size_t)it * np * 9;
    float *xc = t->xc + (size_t)it * 8;
    float wnorm_posterior = 0;
    for (int k = 0; k < np; ++k) {
        double base = prevw ? (double)prevw[k] : (1.0 / np);
        xf[k * 9 + XF_W] = (float)(base * (t->prior[k] / wnorm_prior) * t->lhood[k]);
        wnorm_posterior += xf[k * 9 + XF_W];
    }
    float neff = 0;
    for (int k = 0; k < np; ++k) {
        xf[k * 9 + XF_W] /= wnorm_posterior;
        neff = (float)(neff + (double)xf[k * 9 + XF_W] * (double)xf[k * 9 + XF_W]);
        t->res_csw[k] = xf[k * 9 + XF_W] + ((k > 0) ? t->res_csw[k - 1] : 0);
    }
    neff = (float)(1.0 / neff);
    t->neff[it] = neff;

    for (int c = 0; c < 8; c++) xc[c] = 0;
    for (int k = 0; k < np; ++k) {
        const float *q = xf + k * 9;
        xc[XC_X] += q[XF_W] * q[XF_X];
        xc[XC_Y] += q[XF_W] * q[XF_Y];
        xc[XC_Z] += q[XF_W] * q[XF_Z];
        xc[XC_VX] += q[XF_W] * q[XF_VX];
        xc[XC_VY] += q[XF_W] * q[XF_VY];
        xc[XC_VZ] += q[XF_W] * q[XF_VZ];
        xc[XC_SIG] += q[XF_W] * q[XF_SIG];
    }
    float vnorm = (float)sqrt((double)xc[XC_VX] * xc[XC_VX] + (double)xc[XC_VY] * xc[XC_VY] + (double)xc[XC_VZ] * xc[XC_VZ]);
    xc[XC_VX] /= vnorm;
    xc[XC_VY] /= vnorm;
    xc[XC_VZ] /= vnorm;
    xc[XC_CORR] = orc_zncc(t, xc[XC_X], xc[XC_Y], xc[XC_Z], xc[XC_VX], xc[XC_VY], xc[XC_VZ], img, w, h, l, &xc[XC_SIG]);

    int x1 = (int)roundf(xc[XC_X]), y1 = (int)roundf(xc[XC_Y]), z1 = (int)roundf(xc[XC_Z]);
    if (x1 < 0 || x1 >= w || y1 < 0 || y1 >= h || z1 < 0 || z1 >= l) { *stop = 1; return 0; }
    if (xc[XC_CORR] < t->znccth) { *stop = 2; return 0; }

    if (neff / np < t->neff_ratio) {
        float u1 = (float)((1.0 / np) * ((float)rdraw / ORC_RAND_MAX));
        int s = 0;
        int *idx = t->idxres + (size_t)it * np;
        for (int k = 0; k < np; ++k) {
            float ui = (float)(u1 + k * (1.0 / np));
            while (ui > t->res_csw[s] && s < np - 1) s++; /* clamp: documented divergence */
            idx[k] = s;
        }
    }
    return 1;
}

static int iter0(orc_tracker *t, const float seed[6], const uint8_t *img, int w, int h, int l, int *stop)
{
    int np = t->npcles, sz = t->sz;
    float *xf = t->xfilt;
    float wnorm_prior = 0;
    float u1 = (t->w0_cws[sz - 1] / np) * ((float)t->rng[0] / ORC_RAND_MAX);
    int s = 0;
    for (int i = 0; i < np; ++i) {
        float ui = u1 + i * (t->w0_cws[sz - 1] / np);
        while (ui > t->w0_cws[s] && s < (sz - 1)) s++;
        float *q = xf + i * 9;
        q[XF_X] = seed[0] + t->p[3 * s + 0];
        q[XF_Y] = seed[1] + t->p[3 * s + 1];
        q[XF_Z] = seed[2] + t->p[3 * s + 2];
        q[XF_VX] = isnan(seed[3]) ? t->u[3 * s + 0] : seed[3];
        q[XF_VY] = isnan(seed[4]) ? t->u[3 * s + 1] : seed[4];
        q[XF_VZ] = isnan(seed[5]) ? t->u[3 * s + 2] : seed[5];
        t->prior[i] = t->w0[s];
        wnorm_prior += t->prior[i];
        q[XF_CORR] = orc_zncc(t, q[XF_X], q[XF_Y], q[XF_Z], q[XF_VX], q[XF_VY], q[XF_VZ], img, w, h, l, &q[XF_SIG]);
        t->lhood[i] = expf(t->Kc * q[XF_CORR]);
    }
    return smc_finish(t, 0, NULL, wnorm_prior, img, w, h, l, t->rng[1], stop);
}

static int iterI(orc_tracker *t, int it, const uint8_t *img, int w, int h, int l, int *stop)
{
    int np = t->npcles, sz = t->sz;
    const float *xp = t->xfilt + (size_t)(it - 1) * np * 9;
    float *xf = t->xfilt + (size_t)it * np * 9;
    const int *idxp = t->idxres + (size_t)(it - 1) * np;
    int resampled = (t->neff[it - 1] / np < t->neff_ratio);
    float wnorm_prior = 0;
    for (int k = 0; k < np; ++k) {
        int k1 = resampled ? idxp[k] : k;
        const float *par = xp + k1 * 9;
        int vi = getdirection(t, par[XF_VX], par[XF_VY], par[XF_VZ]);
        const float *cws = t->w_cws + (size_t)vi * sz;
        float u1 = (cws[sz - 1]) * ((float)t->rng[k] / ORC_RAND_MAX);
        int s = 0;
        while (u1 > cws[s] && s < sz - 1) s++;
        float *q = xf + k * 9;
        q[XF_X] = par[XF_X] + t->p[3 * s + 0];
        q[XF_Y] = par[XF_Y] + t->p[3 * s + 1];
        q[XF_Z] = par[XF_Z] + t->p[3 * s + 2];
        q[XF_VX] = t->u[3 * s + 0];
        q[XF_VY] = t->u[3 * s + 1];
        q[XF_VZ] = t->u[3 * s + 2];
        t->prior[k] = t->w[(size_t)vi * sz + s];
        wnorm_prior += t->prior[k];
        q[XF_CORR] = orc_zncc(t, q[XF_X], q[XF_Y], q[XF_Z], q[XF_VX], q[XF_VY], q[XF_VZ], img, w, h, l, &q[XF_SIG]);
        t->lhood[k] = expf(t->Kc * q[XF_CORR]);
    }
    /* previous weights: of slot k, used only when step it-1 did not resample (tracker.cpp:1143) */
    float *prevw = NULL;
    if (!resampled) {
        prevw = (float *)malloc(sizeof(float) * (size_t)np);
        for (int k = 0; k < np; k++) prevw[k] = xp[k * 9 + XF_W];
    }
    int ok = smc_finish(t, it, prevw, wnorm_prior, img, w, h, l, t->rng[np], stop);
    free(prevw);
    return ok;
}

int orc_trace(orc_tracker *t, const float seed[6], const uint8_t *img, int w, int h, int l,
              float *xc_out, int *stop, int max_dbg, float *xfilt_out, int *idxres_out,
              float *neff_out)
{
    int T = t->niter, st = 0, np = t->npcles;
    for (int i = 0; i < t->niter; ++i) {
        int ok = (i == 0) ? iter0(t, seed, img, w, h, l, &st) : iterI(t, i, img, w, h, l, &st);
        memcpy(xc_out + (size_t)i * 8, t->xc + (size_t)i * 8, sizeof(float) * 8);
        if (i < max_dbg) {
            if (xfilt_out) memcpy(xfilt_out + (size_t)i * np * 9, t->xfilt + (size_t)i * np * 9, sizeof(float) * 9 * (size_t)np);
            if (idxres_out) memcpy(idxres_out + (size_t)i * np, t->idxres + (size_t)i * np, sizeof(int) * (size_t)np);
            if (neff_out) neff_out[i] = t->neff[i];
        }
        if (!ok) { T = i; break; }
    }
    if (stop) *stop = st;
    return T;
}

/* ====================================================================== */
/*  T8/T9  trackPos bookkeeping + trace loop                              */
/*         tracker.cpp:825-933, Advantra_plugin.cpp:2602-2710             */
/* ====================================================================== */
static int neighbours(i64 i, int N, int M, int P, int vol, i64 *out)
{
    /* Advantra_plugin.cpp:2605-2648, including the clampi(y+-1,0,N-1) typos at :2632-2637 */
    int x = (int)(i % N), z = (int)(i / ((i64)N * M)), y = (int)(i / N - (i64)z * M);
    i64 NM = (i64)N * M;
    int n = 0;
    if (vol == 1) return 0;
#define IDX(zz, yy, xx) ((i64)(zz) * NM + (i64)(yy) * N + (xx))
    int xm = clampi(x - 1, 0, N - 1), xp = clampi(x + 1, 0, N - 1);
    int ym = clampi(y - 1, 0, M - 1), yp = clampi(y + 1, 0, M - 1);
    int zm = clampi(z - 1, 0, P - 1), zp = clampi(z + 1, 0, P - 1);
    int ymN = clampi(y - 1, 0, N - 1), ypN = clampi(y + 1, 0, N - 1);
    if (vol >= 5) { out[n++] = IDX(z, y, xm); out[n++] = IDX(z, y, xp); out[n++] = IDX(z, ym, x); out[n++] = IDX(z, yp, x); }
    if (vol >= 9) { out[n++] = IDX(z, ym, xm); out[n++] = IDX(z, ym, xp); out[n++] = IDX(z, yp, xm); out[n++] = IDX(z, yp, xp); }
    if (vol >= 11) { out[n++] = IDX(zm, y, x); out[n++] = IDX(zp, y, x); }
    if (vol >= 19) {
        out[n++] = IDX(zm, y, xm); out[n++] = IDX(zm, y, xp); out[n++] = IDX(zm, ymN, x); out[n++] = IDX(zm, ypN, x);
        out[n++] = IDX(zp, y, xm); out[n++] = IDX(zp, y, xp); out[n++] = IDX(zp, ymN, x); out[n++] = IDX(zp, ypN, x);
    }
    if (vol >= 27) {
        out[n++] = IDX(zm, ym, xm); out[n++] = IDX(zm, ym, xp); out[n++] = IDX(zm, yp, xm); out[n++] = IDX(zm, yp, xp);
        out[n++] = IDX(zp, ym, xm); out[n++] = IDX(zp, ym, xp); out[n++] = IDX(zp, yp, xm); out[n++] = IDX(zp, yp, xp);
    }
#undef IDX
    return n;
}

static i64 replay_impl(const float *seeds, i64 nseeds, const int *T, const float *xc, int niter,
                       int w, int h, int l, int nodespervol, int vol, int max_trace_count, const int32_t *smap,
                       const float *soma4, i64 n_soma,
                       orc_node *nodes, i64 cap_nodes, int32_t *links, i64 cap_links,
                       i64 *nlinks, i64 *ntraces_used);

i64 orc_replay(const float *seeds, i64 nseeds, const int *T, const float *xc, int niter,
               int w, int h, int l, int nodespervol, int vol, int max_trace_count,
               orc_node *nodes, i64 cap_nodes, int32_t *links, i64 cap_links,
               i64 *nlinks, i64 *ntraces_used)
{
    return replay_impl(seeds, nseeds, T, xc, niter, w, h, l, nodespervol, vol, max_trace_count, NULL, NULL, 0, nodes, cap_nodes, links,
                       cap_links, nlinks, ntraces_used);
}

i64 orc_replay_soma(const float *seeds, i64 nseeds, const int *T, const float *xc, int niter,
                    int w, int h, int l, int nodespervol, int vol, int max_trace_count, const int32_t *smap,
                    const float *soma4, i64 n_soma,
                    orc_node *nodes, i64 cap_nodes, int32_t *links, i64 cap_links,
                    i64 *nlinks, i64 *ntraces_used)
{
    return replay_impl(seeds, nseeds, T, xc, niter, w, h, l, nodespervol, vol, max_trace_count, smap, soma4, n_soma, nodes, cap_nodes,
                       links, cap_links, nlinks, ntraces_used);
}

static i64 replay_impl(const float *seeds, i64 nseeds, const int *T, const float *xc, int niter,
                       int w, int h, int l, int nodespervol, int vol, int max_trace_count, const int32_t *smap,
                       const float *soma4, i64 n_soma,
                       orc_node *nodes, i64 cap_nodes, int32_t *links, i64 cap_links,
                       i64 *nlinks, i64 *ntraces_used)
{
    i64 size = (i64)w * h * l, nn = 0, nl = 0;
    uint8_t *den = (uint8_t *)calloc((size_t)size, 1);
    int32_t *nidx = (int32_t *)calloc((size_t)size, sizeof(int32_t));
    /* dummy node 0 (Advantra_plugin.cpp:2416-2419; Node() in node.cpp:43-54) */
    if (cap_nodes > 0) {
        memset(&nodes[0], 0, sizeof(orc_node));
        nodes[0].corr = -FLT_MAX;
        nodes[0].type = 7;
    }
    nn = 1;
    for (i64 k = 0; k < n_soma; k++) { /* Node(x, y, z, r, SOMA): v = 0, corr = -FLT_MAX (node.cpp:68-79) */
        if (nn < cap_nodes) {
            memset(&nodes[nn], 0, sizeof(orc_node));
            nodes[nn].x = soma4[4 * k]; nodes[nn].y = soma4[4 * k + 1]; nodes[nn].z = soma4[4 * k + 2]; nodes[nn].sig = soma4[4 * k + 3];
            nodes[nn].corr = -FLT_MAX;
            nodes[nn].type = 1;
        }
        nn++;
    }
    int trace_count = 0;
#define LINK(a, b) do { if (nl < cap_links) { links[2 * nl] = (int32_t)(a); links[2 * nl + 1] = (int32_t)(b); } nl++; } while (0)
    for (i64 s = 0; s < nseeds; ++s) {
        const float *sd = seeds + s * 8;
        i64 si = (i64)(int)roundf(sd[2]) * w * h + (i64)(int)roundf(sd[1]) * w + (int)roundf(sd[0]);
        if (!((int)den[si] < nodespervol)) continue;
        trace_count++;
        for (int dir = 0; dir < 2; dir++) {
            i64 j = 2 * s + dir;
            const float *X = xc + j * niter * 8;
            int ti_limit = niter;
            for (int i = 0; i < niter; ++i) {
                if (i < T[j]) {
                    const float *e = X + i * 8;
                    i64 crd = (i64)(int)roundf(e[XC_Z]) * w * h + (i64)(int)roundf(e[XC_Y]) * w + (int)roundf(e[XC_X]);
                    if (smap && smap[crd] > 0) { /* SOMA reached: link to its node and stop (tracker.cpp:858-869) */
                        if (i > 0) LINK(smap[crd], nn - 1);
                        ti_limit = i;
                        break;
                    }
                    if ((int)den[crd] >= nodespervol) { /* DENSITY */
                        if (i > 0) LINK(nidx[crd], nn - 1);
                        ti_limit = i;
                        break;
                    }
                    if (nn < cap_nodes) {
                        orc_node *nd = &nodes[nn];
                        nd->x = e[XC_X]; nd->y = e[XC_Y]; nd->z = e[XC_Z];
                        nd->vx = e[XC_VX]; nd->vy = e[XC_VY]; nd->vz = e[XC_VZ];
                        nd->corr = e[XC_CORR]; nd->sig = e[XC_SIG];
                        nd->type = (i == 0) ? 7 : 2; /* UNDEFINED : AXON */
                    }
                    nn++;
                    den[crd] = (uint8_t)((int)den[crd] + 1);
                    nidx[crd] = (int32_t)(nn - 1);
                    if (vol > 1) {
                        i64 nb[26];
                        int c = neighbours(crd, w, h, l, vol, nb);
                        for (int q = 0; q < c; q++) {
                            den[nb[q]] = (uint8_t)((int)den[nb[q]] + 1);
                            nidx[nb[q]] = (int32_t)(nn - 1);
                        }
                    }
                    if (i > 0) LINK(nn - 1, nn - 2);
                } else { /* filter returned false */
                    ti_limit = i;
                    break;
                }
            }
            if (ti_limit > 1 && nn - 1 < cap_nodes) nodes[nn - 1].type = 6; /* END: applied to nodelist.back() */
        }
        if (trace_count > max_trace_count) break;
    }
#undef LINK
    free(den);
    free(nidx);
    if (nlinks) *nlinks = nl;
    if (ntraces_used) *ntraces_used = trace_count;
    return nn;
}
