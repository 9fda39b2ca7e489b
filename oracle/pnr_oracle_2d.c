/* TEST INFRASTRUCTURE ONLY -- never linked into or called by the product (pnr_amd/): see pnr_oracle.h.
 *
 * CPU restatement of the 2-D Frangi branch the plugin takes for single-slice stacks, P == 1 (SURVEY 8f-4;
 * Advantra_plugin.cpp:2496-2497):
 *   Frangi::imgaussian (2-D)  frangi.cpp:576-645   separable xy Gaussian, clamp-to-edge, f32 accumulate, taps ascending
 *   Frangi::hessian2d         frangi.cpp:508-574   first differences applied twice (one-sided at the borders), x sigma^2
 *   Frangi::frangi2d          frangi.cpp:392-506   closed-form 2x2 eigen-analysis, Rb / S2 vesselness, max over scales
 * Pinned: frangi.cpp is built into oracle/_ref (ref_frangi2d); tests/test_oracle_golden.py compares bit for bit.
 * The 2-D branches of the tracker (tables, bilinear interp, w = 0 in znccBBB) live in pnr_oracle.c (orc_tracker_new2). */
#include "pnr_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>

typedef int64_t i64;
static int clampi2(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

static void imgaussian2d(const uint8_t *I, int w, int h, float sig, float *F)
{
    const int L = (int)ceil(3 * sig);
    float *G = (float *)malloc(sizeof(float) * (size_t)(2 * L + 1));
    float gn = 0;
    for (int i = -L; i <= L; i++) { G[i + L] = expf(-(float)(i * i) / (2 * sig * sig)); gn += G[i + L]; } /* std::exp(float) */
    for (int i = 0; i < 2 * L + 1; i++) G[i] /= gn;
    float *K = (float *)malloc(sizeof(float) * (size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float a = 0;
            for (int x1 = x - L; x1 <= x + L; x1++) a += I[(i64)y * w + clampi2(x1, 0, w - 1)] * G[x1 - x + L];
            K[(i64)y * w + x] = a;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float a = 0;
            for (int y1 = y - L; y1 <= y + L; y1++) a += K[(i64)clampi2(y1, 0, h - 1) * w + x] * G[y1 - y + L];
            F[(i64)y * w + x] = a;
        }
    free(K);
    free(G);
}

/* derivative along y (stride w) or x (stride 1), one-sided at the borders; .5*(a-b) is an exact scaling of the f32 difference */
static void diff_axis(const float *F, float *D, int w, int h, int along_y)
{
    for (i64 i = 0; i < (i64)w * h; i++) {
        const int x = (int)(i % w), y = (int)(i / w);
        const int c = along_y ? y : x, n = along_y ? h : w;
        const i64 s = along_y ? w : 1;
        if (c == 0) D[i] = F[i + s] - F[i];
        else if (c < n - 1) D[i] = (float)(.5 * (F[i + s] - F[i - s]));
        else D[i] = F[i] - F[i - s];
    }
}

void orc_hessian2d(const uint8_t *I, int w, int h, float sig, float *Dyy, float *Dxy, float *Dxx)
{
    const i64 n = (i64)w * h;
    float *F = (float *)malloc(sizeof(float) * (size_t)n), *DD = (float *)malloc(sizeof(float) * (size_t)n);
    imgaussian2d(I, w, h, sig, F);
    diff_axis(F, DD, w, h, 1);
    diff_axis(DD, Dyy, w, h, 1);
    for (i64 i = 0; i < n; i++) Dyy[i] *= (sig * sig);
    diff_axis(F, DD, w, h, 0);
    diff_axis(DD, Dxx, w, h, 0);
    diff_axis(DD, Dxy, w, h, 1);
    for (i64 i = 0; i < n; i++) { Dxx[i] *= (sig * sig); Dxy[i] *= (sig * sig); }
    free(F);
    free(DD);
}

static uint8_t quant2(float v, float n) /* round((((v/n)+1)/2)*255.0), clamped (frangi.cpp:463-468) */
{
    const double r = round((double)(((v / n) + 1) / 2) * 255.0);
    if (!(r == r)) return 0; /* NaN (zero vector): (int)NaN is INT_MIN on the reference's platform, clamped to 0 */
    const int val = (int)r;
    return (uint8_t)(val < 0 ? 0 : (val > 255 ? 255 : val));
}

void orc_frangi2d(const uint8_t *I, int w, int h, const float *sigs, int nsig, float BetaOne, float BetaTwo,
                  float *J, float *Jmin, float *Jmax, uint8_t *Vx, uint8_t *Vy, uint8_t *Vz)
{
    const i64 n = (i64)w * h;
    float *Dxx = (float *)malloc(sizeof(float) * (size_t)n), *Dxy = (float *)malloc(sizeof(float) * (size_t)n),
          *Dyy = (float *)malloc(sizeof(float) * (size_t)n);
    const float beta = (float)(2 * pow((double)BetaOne, 2)), c = (float)(2 * pow((double)BetaTwo, 2));
    *Jmin = FLT_MAX;
    *Jmax = -FLT_MAX;
    for (int si = 0; si < nsig; si++) {
        orc_hessian2d(I, w, h, sigs[si], Dyy, Dxy, Dxx);
        for (i64 i = 0; i < n; i++) {
            const float dd = Dxx[i] - Dyy[i];
            const float tmp = (float)sqrt((double)dd * (double)dd + 4 * ((double)Dxy[i] * (double)Dxy[i]));
            float v2x = 2 * Dxy[i];
            float v2y = Dyy[i] - Dxx[i] + tmp;
            const float mag = (float)sqrt((double)v2x * (double)v2x + (double)v2y * (double)v2y);
            if (mag > 0) { v2x /= mag; v2y /= mag; }
            const float v1x = -v2y, v1y = v2x;
            const float mu1 = (float)(0.5 * (Dxx[i] + Dyy[i] + tmp));
            const float mu2 = (float)(0.5 * (Dxx[i] + Dyy[i] - tmp));
            const int check = fabsf(mu1) < fabsf(mu2);
            float L1 = check ? mu2 : mu1;
            const float L2 = check ? mu1 : mu2;
            const float Vecx = check ? v2x : v1x, Vecy = check ? v2y : v1y;
            L1 = (L1 == 0) ? FLT_MIN : L1;
            const float q = L2 / L1;
            const float Rb = (float)((double)q * (double)q);
            const float S2 = (float)((double)L1 * (double)L1 + (double)L2 * (double)L2);
            float If = expf(-Rb / beta) * (1 - expf(-S2 / c));
            If = (L1 > 0) ? 0 : If; /* blackwhite == false */
            if (si == 0 || If > J[i]) {
                J[i] = If;
                if (J[i] < *Jmin) *Jmin = J[i];
                if (J[i] > *Jmax) *Jmax = J[i];
                const float Vecn = sqrtf(Vecx * Vecx + Vecy * Vecy);
                Vx[i] = quant2(Vecx, Vecn);
                Vy[i] = quant2(Vecy, Vecn);
                Vz[i] = 0;
            }
        }
    }
    free(Dxx); free(Dxy); free(Dyy);
}
