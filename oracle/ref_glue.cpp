// ref_glue.cpp -- extern "C" entry points onto the REFERENCE's own translation
// units (frangi.cpp, seed.cpp, node.cpp), compiled where they lie under
// /root/reference/pnr-vaa3d by oracle/Makefile into oracle/_ref/libpnr_ref.so.
//
// TEST INFRASTRUCTURE ONLY: used to validate oracle/pnr_oracle.c and to
// generate tests/golden/*.npz (tests/golden/make_golden.py).  This file is our
// own code; no reference source is copied -- the reference headers are found
// through -I at build time, in this container only.  Neither the reference's
// sources nor this compiled library leave the container: the licence forbids
// redistribution (pnr-vaa3d/LICENSE:4-5), oracle/_ref/ is listed in .gitignore
// AND .gpurunignore, and the GPU-side tests run on the committed fixtures
// (tests/golden/*.npz) that this library produced here (tests/orc.py:load_ref
// returns None where it is absent).
#include "frangi.h"
#include "seed.h"
#include <cstdint>
#include <iostream>
#include <sstream>
#include <vector>

namespace {
struct MuteStdout { // the reference prints progress on std::cout
    std::streambuf *old;
    std::ostringstream sink;
    MuteStdout() : old(std::cout.rdbuf(sink.rdbuf())) {}
    ~MuteStdout() { std::cout.rdbuf(old); }
};
} // namespace

extern "C" {

void ref_imgaussian3d(unsigned char *I, int w, int h, int l, float sig, float zdist, float *F)
{
    Frangi::imgaussian(I, w, h, l, sig, zdist, F); // frangi.cpp:647
}

void ref_frangi2d(unsigned char *I, int w, int h, const float *sigs, int nsig, float betaone, float betatwo,
                  float *J, float *Jmin, float *Jmax, unsigned char *Vx, unsigned char *Vy, unsigned char *Vz)
{
    MuteStdout m;
    std::vector<float> s(sigs, sigs + nsig);
    Frangi f(s, 1.f, .5f, .5f, 500.f, betaone, betatwo); // Advantra_plugin.cpp:2488 (frangi_betaone, frangi_betatwo)
    f.frangi2d(I, w, h, 1, J, *Jmin, *Jmax, Vx, Vy, Vz); // frangi.cpp:392, the P == 1 branch of :2496-2497
}

void ref_hessian2d(unsigned char *I, int w, int h, float sig, float *Dyy, float *Dxy, float *Dxx)
{
    MuteStdout m;
    std::vector<float> s(1, sig);
    Frangi f(s, 1.f, .5f, .5f, 500.f, .5f, 15.f);
    f.hessian2d(I, w, h, sig, Dyy, Dxy, Dxx); // frangi.cpp:508
}

void ref_imerode_xy(unsigned char *I, int w, int h, int l, float rad, unsigned char *E)
{
    Frangi::imerode(I, w, h, l, rad, E); // frangi.cpp:880 (the xy erosion the soma path calls, Advantra_plugin.cpp:2431)
}

void ref_imgaussian_u8_xy(unsigned char *I, int w, int h, int l, float sig)
{
    Frangi::imgaussian(I, w, h, l, sig); // frangi.cpp:786, in place (Advantra_plugin.cpp:2437)
}

void ref_hessian3d(unsigned char *I, int w, int h, int l, float sig, float zdist,
                   float *Dzz, float *Dyy, float *Dyz, float *Dxx, float *Dxy, float *Dxz)
{
    MuteStdout m;
    std::vector<float> s(1, sig);
    Frangi f(s, zdist, .5f, .5f, 500.f, .5f, 15.f);
    f.hessian3d(I, w, h, l, sig, zdist, Dzz, Dyy, Dyz, Dxx, Dxy, Dxz); // frangi.cpp:291
}

void ref_eigen3(const double *A, double *V, double *d)
{
    MuteStdout m;
    std::vector<float> s(1, 1.f);
    Frangi f(s, 1.f, .5f, .5f, 500.f, .5f, 15.f);
    double a[3][3], v[3][3];
    for (int i = 0; i < 9; i++) a[i / 3][i % 3] = A[i];
    f.eigen_decomposition(a, v, d); // frangi.cpp:1269
    for (int i = 0; i < 9; i++) V[i] = v[i / 3][i % 3];
}

void ref_frangi3d(unsigned char *I, int w, int h, int l, const float *sigs, int nsig, float zdist,
                  float alpha, float beta, float C,
                  float *J, float *Jmin, float *Jmax, unsigned char *Vx, unsigned char *Vy, unsigned char *Vz)
{
    MuteStdout m;
    std::vector<float> s(sigs, sigs + nsig);
    Frangi f(s, zdist, alpha, beta, C, .5f, 15.f); // Advantra_plugin.cpp:2488
    f.frangi3d(I, w, h, l, J, *Jmin, *Jmax, Vx, Vy, Vz); // frangi.cpp:152
}

int64_t ref_extract_seeds(double tolerance, unsigned char *J8, int w, int h, int l,
                          unsigned char *Vx, unsigned char *Vy, unsigned char *Vz,
                          float *seeds_out, int64_t cap)
{
    MuteStdout m;
    std::vector<seed> sd;
    SeedExtractor::extractSeeds(tolerance, J8, w, h, l, Vx, Vy, Vz, sd); // seed.cpp:556
    for (int64_t i = 0; i < (int64_t)sd.size() && i < cap; i++) {
        float *o = seeds_out + i * 8;
        o[0] = sd[i].x; o[1] = sd[i].y; o[2] = sd[i].z;
        o[3] = sd[i].vx; o[4] = sd[i].vy; o[5] = sd[i].vz;
        o[6] = sd[i].score; o[7] = sd[i].corr;
    }
    return (int64_t)sd.size();
}

} // extern "C"
