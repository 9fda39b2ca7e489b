// tables.cpp -- host-side, one-off lookup tables of the SMC tracker and the Frangi Gaussian
// taps.  Product code (NOT the oracle): restates Tracker::Tracker (tracker.cpp:79-527),
// Tracker::generate_directions (:770-805), Tracker::bessi0 (:2254-2270) and the kernel set-up
// of Frangi::imgaussian (frangi.cpp:651-680), keeping the reference's mixed f32/f64 arithmetic
// so the tables are bit-identical to the reference's.  Built with -ffp-contract=off.
#include "ctx.h"
#include <cfloat>
#include <cmath>

namespace pnr {

// glibc rand() after srand(seed): TYPE_3 additive feedback generator.  The reference reseeds
// with srand(time(NULL)) at the top of every SMC iteration (tracker.cpp:1003,1098), so with a
// pinned seed every iteration sees draws #0..#np of this stream.
void glibc_rand_stream(uint32_t seed, int n, uint32_t *out)
{
    std::vector<uint32_t> r(344 + (size_t)n);
    if (seed == 0) seed = 1;
    int32_t word = (int32_t)seed;
    r[0] = (uint32_t)word;
    for (int i = 1; i < 31; i++) {
        // Park-Miller via Schrage, as initstate/srandom_r do
        int64_t hi = word / 127773, lo = word % 127773;
        int64_t nx = 16807 * lo - 2836 * hi;
        if (nx < 0) nx += 2147483647;
        word = (int32_t)nx;
        r[i] = (uint32_t)word;
    }
    for (int i = 31; i < 34; i++) r[i] = r[i - 31];
    for (size_t i = 34; i < r.size(); i++) r[i] = r[i - 31] + r[i - 3];
    for (int k = 0; k < n; k++) out[k] = r[344 + k] >> 1;
}

int gaussian_taps(float sig, std::vector<float> &g)
{
    const int L = (int)std::ceil(3 * sig); // f32 ceil
    g.assign(2 * L + 1, 0.f);
    float norm = 0;
    for (int i = -L; i <= L; ++i) {
        g[i + L] = std::exp(-(i * i) / (2 * sig * sig)); // std::exp(float)
        norm += g[i + L];
    }
    for (float &x : g) x /= norm;
    return L;
}

static double i0_poly(double x)
{
    const double ax = std::fabs(x);
    if (ax < 3.75) {
        double y = x / 3.75;
        y = y * y;
        return 1.0 + y * (3.5156229 + y * (3.0899424 + y * (1.2067492 + y * (0.2659732 + y * (0.360768e-1 + y * 0.45813e-2)))));
    }
    const double y = 3.75 / ax;
    return (std::exp(ax) / std::sqrt(ax)) *
           (0.39894228 + y * (0.1328592e-1 + y * (0.225319e-2 + y * (-0.157565e-2 + y * (0.916281e-2 + y * (-0.2057706e-1 + y * (0.2635537e-1 + y * (-0.1647633e-1 + y * 0.392377e-2))))))));
}

static void build_templates(const pnr_params &P, bool is2d, Tables &t)
{
    // model2_* of tracker.cpp:178-231; model2_N = 12 samples per 3*sigma.  2-D branch (:191-208): offsets (vv, uu, 0),
    // weight exp(-uu^2 / 2 sig^2) -- the same product grid with a single w value 0
    t.M.clear(); t.moff.clear(); t.tmpl.clear(); t.mwgt.clear(); t.mavg.clear(); t.corrc.clear();
    t.grid.clear(); t.axes.clear(); t.axes_off.clear(); t.wd.clear(); t.ext_v = t.ext_uw = 0;
    t.ext_vs.clear(); t.ext_uws.clear();
    int off = 0;
    for (int s = 0; s < P.nsig; s++) {
        const float sg = P.sig[s];
        const int V2 = (int)std::round(1 * sg), U2 = (int)std::round(3 * sg), W2 = U2;
        float Vs = (float)((3.0 * sg) / 12);
        if (Vs < 1.0) Vs = 1.0f;
        std::vector<float> vuw, wgt;
        float avg = 0.f;
        // the three nested float-stepped loops visit a product grid: record its axes once, so the
        // kernel can hoist the outer partial sums of the sample position
        std::vector<float> av, au, aw;
        for (float vv = (float)-V2; vv <= V2 + FLT_MIN; vv += Vs) av.push_back(vv);
        for (float uu = (float)-U2; uu <= U2 + FLT_MIN; uu += Vs) au.push_back(uu);
        if (is2d) aw.push_back(0.f);
        else
            for (float ww = (float)-W2; ww <= W2 + FLT_MIN; ww += Vs) aw.push_back(ww);
        t.grid.push_back((int)av.size()); t.grid.push_back((int)au.size()); t.grid.push_back((int)aw.size()); t.grid.push_back(off);
        t.axes_off.push_back((int)t.axes.size());
        t.axes.insert(t.axes.end(), av.begin(), av.end());
        t.axes.insert(t.axes.end(), au.begin(), au.end());
        t.axes.insert(t.axes.end(), aw.begin(), aw.end());
        for (float x : av) t.ext_v = std::fmax(t.ext_v, std::fabs(x));
        for (float x : au) t.ext_uw = std::fmax(t.ext_uw, std::fabs(x));
        for (float x : aw) t.ext_uw = std::fmax(t.ext_uw, std::fabs(x));
        {
            float ev = 0, eu = 0;
            for (float x : av) ev = std::fmax(ev, std::fabs(x));
            for (float x : au) eu = std::fmax(eu, std::fabs(x));
            for (float x : aw) eu = std::fmax(eu, std::fabs(x));
            t.ext_vs.push_back(ev); t.ext_uws.push_back(eu);
        }
        for (float vv = (float)-V2; vv <= V2 + FLT_MIN; vv += Vs)
            for (float uu = (float)-U2; uu <= U2 + FLT_MIN; uu += Vs) {
                if (is2d) {
                    const float value = (float)std::exp((double)(-(uu * uu)) / (2 * std::pow((double)sg, 2)));
                    wgt.push_back(value);
                    vuw.push_back(vv); vuw.push_back(uu); vuw.push_back(0.f);
                    avg += value;
                    continue;
                }
                for (float ww = (float)-W2; ww <= W2 + FLT_MIN; ww += Vs) {
                    const float value = (float)std::exp((double)(-((uu * uu) + (ww * ww))) / (2 * std::pow((double)sg, 2)));
                    wgt.push_back(value);
                    vuw.push_back(vv); vuw.push_back(uu); vuw.push_back(ww);
                    avg += value;
                }
            }
        const int M = (int)wgt.size();
        avg /= (float)M;
        // (wgt - avg) and corrc = sum pow(wgt-avg,2) do not depend on the image: the reference
        // recomputes them in every znccBBB call (tracker.cpp:1947-1953); hoisted here with the
        // same accumulation (f64 add, f32 store, ascending sample order).
        float cc = 0.f;
        for (int k = 0; k < M; k++) {
            const float wd = wgt[k] - avg;
            t.tmpl.push_back(vuw[3 * k]); t.tmpl.push_back(vuw[3 * k + 1]); t.tmpl.push_back(vuw[3 * k + 2]);
            t.tmpl.push_back(wd);
            t.mwgt.push_back(wgt[k]);
            t.wd.push_back(wd);
            cc = (float)((double)cc + (double)wd * (double)wd);
        }
        t.M.push_back(M);
        t.moff.push_back(off);
        off += M;
        t.mavg.push_back(avg);
        t.corrc.push_back(cc);
    }
}

static void build_prediction(const pnr_params &P, bool is2d, Tables &t)
{
    // tracker.cpp:375-438: integer offsets inside the radius-2*step ball (2-D: disc, dz = 0), z scaled by 1/zdist
    const int R = 2 * P.step, Rz = is2d ? 0 : R;
    std::vector<int> px, py, pz;
    for (int dx = -R; dx <= R; ++dx)
        for (int dy = -R; dy <= R; ++dy)
            for (int dz = -Rz; dz <= Rz; ++dz) {
                const int r2 = dx * dx + dy * dy + dz * dz;
                if (r2 <= R * R && r2 > 0) { px.push_back(dx); py.push_back(dy); pz.push_back(dz); }
            }
    const int sz = t.sz = (int)px.size();
    t.p.resize(3 * sz); t.u.resize(3 * sz); t.d0.resize(sz); t.w0.resize(sz); t.w0_cws.resize(sz);
    std::vector<float> d(sz);
    float w0sum = 0;
    for (int i = 0; i < sz; i++) {
        float *p = &t.p[3 * i], *u = &t.u[3 * i];
        p[0] = (float)px[i];
        p[1] = (float)py[i];
        p[2] = pz[i] / P.zdist;
        d[i] = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
        t.d0[i] = std::sqrt((float)px[i] * px[i] + py[i] * py[i] + pz[i] * pz[i]);
        u[0] = p[0] / d[i]; u[1] = p[1] / d[i]; u[2] = p[2] / d[i];
        t.w0[i] = (float)std::exp(-std::pow((double)d[i], 2) / (2 * std::pow(P.step / 3.0, 2)));
        w0sum += t.w0[i];
    }
    for (int i = 0; i < sz; i++) {
        t.w0[i] /= w0sum;
        t.w0_cws[i] = t.w0[i] + (i == 0 ? 0.f : t.w0_cws[i - 1]);
    }
}

static void build_directions(bool is2d, Tables &t)
{
    // tracker.cpp:785-799 (3-D): spiral points on the sphere, poles at phi = 0; :776-783 (2-D): 30 angles on the circle
    const int n = t.ndir;
    t.v.resize(3 * n);
    if (is2d) {
        for (int k = 0; k < n; k++) {
            const float ang1 = (float)(0.0 + k * ((2 * 3.14) / (float)n)); // "3.14" as in the reference
            t.v[3 * k + 0] = std::cos(ang1); // float overloads
            t.v[3 * k + 1] = std::sin(ang1);
            t.v[3 * k + 2] = 0.f;
        }
        return;
    }
    double phi = 0, phi_prev = 0;
    for (int k = 0; k < n; k++) {
        const double hk = 1 - 2 * ((double)k / (n - 1));
        const double th = std::acos(hk);
        if (k == 0 || k == n - 1) {
            phi = 0; phi_prev = 0;
        } else {
            phi = phi_prev + 3.6 / (std::sqrt((float)n) * std::sqrt(1 - hk * hk));
            phi_prev = phi;
        }
        t.v[3 * k + 0] = (float)(std::sin(th) * std::cos(phi));
        t.v[3 * k + 1] = (float)(std::sin(th) * std::sin(phi));
        t.v[3 * k + 2] = (float)std::cos(th);
    }
}

static void build_oriented_priors(const pnr_params &P, Tables &t)
{
    // tracker.cpp:453-476: von Mises (kappa) x radial Gaussian around |offset| = step, per direction
    const int sz = t.sz, n = t.ndir;
    t.w.resize((size_t)n * sz);
    t.w_cws.resize((size_t)n * sz);
    for (int a = 0; a < n; a++) {
        float *wa = &t.w[(size_t)a * sz], *ca = &t.w_cws[(size_t)a * sz];
        float wsum = 0;
        for (int j = 0; j < sz; j++) {
            const double rad = std::exp(-std::pow((double)(t.d0[j] - P.step), 2) / (2 * std::pow(P.step / 3.0, 2)));
            double dotp = t.v[3 * a] * t.u[3 * j] + t.v[3 * a + 1] * t.u[3 * j + 1] + t.v[3 * a + 2] * t.u[3 * j + 2];
            dotp = dotp > 1 ? 1 : (dotp < -1 ? -1 : dotp);
            const double circ = std::exp(P.kappa * dotp) / (2.0 * 3.14 * i0_poly(P.kappa)); // "3.14" as in the reference
            wa[j] = (float)(circ * rad);
            wsum += wa[j];
        }
        for (int j = 0; j < sz; j++) {
            wa[j] = wa[j] / wsum;
            ca[j] = wa[j] + (j == 0 ? 0.f : ca[j - 1]);
        }
    }
}

void build_tables(const pnr_params &P, bool is2d, Tables &t)
{
    t.nsig = P.nsig;
    t.is2d = is2d;
    t.ndir = is2d ? 30 : 50; // Tracker::ndirs2d / ndirs3d (tracker.cpp:27-28)
    static_assert(50 <= 64 && 30 <= 64, "ph_predict keeps the directions in an LDS array of 64");
    build_templates(P, is2d, t);
    build_prediction(P, is2d, t);
    build_directions(is2d, t);
    build_oriented_priors(P, t);
    t.rng.resize((size_t)P.np + 1);
    glibc_rand_stream(P.rng_seed, P.np + 1, t.rng.data());
    t.gxy.resize(P.nsig);
    t.gz.resize(P.nsig);
    for (int s = 0; s < P.nsig; s++) {
        gaussian_taps(P.sig[s], t.gxy[s]);
        gaussian_taps(P.sig[s] / P.zdist, t.gz[s]); // sigz = sig/zdist (frangi.cpp:651)
    }
}

} // namespace pnr
