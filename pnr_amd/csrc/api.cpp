// api.cpp -- the extern "C" boundary of libpnr_hip.so (include/pnr_hip.h): context life-cycle,
// parameter validation (Advantra::dofunc, Advantra_plugin.cpp:317-326), uploads/read-backs, the
// seed filter/sort (:2561-2586) and the host replay of the trace bookkeeping
// (tracker.cpp:825-933 + Advantra_plugin.cpp:2602-2710).
#include "ctx.h"
#include "replay.h"
#include "stream_sched.h"
#include "../host/reconstruct.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <unordered_map>
#include <fstream>
#include <thread>
#include <sched.h>

namespace pnr {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// worker threads for the host-side stages (seed flood fill, reconstruct()): the CPUs this process may run on (affinity mask,
// cgroup quota), shared between the ranks of this host (one process per GPU)
int host_threads(const Options &o)
{
    if (o.host_threads > 0) return o.host_threads;
    int n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = (int)std::thread::hardware_concurrency();
    std::ifstream f("/sys/fs/cgroup/cpu.max"); // "<quota> <period>" or "max <period>"
    std::string q;
    long long per = 0;
    if (f >> q >> per && q != "max" && per > 0) {
        const long long lim = (atoll(q.c_str()) + per - 1) / per;
        if (lim > 0 && lim < n) n = (int)lim;
    }
    n /= std::max(1, o.local_ranks);
    return std::min(std::max(n, 1), 32); // the fills of a stack's layers stop scaling long before that
}
} // namespace pnr

using pnr::set_error;

template <typename T>
static int upload(T **dst, const std::vector<T> &src, hipStream_t s)
{
    PNR_HIP(hipMalloc(dst, std::max<size_t>(src.size(), 1) * sizeof(T)));
    if (!src.empty()) PNR_HIP(hipMemcpyAsync(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, s));
    return PNR_OK;
}

template <typename T>
static int download(pnr_ctx *c, T *dst, const T *src, size_t n)
{
    if (!dst) return PNR_OK;
    PNR_HIP(hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToHost, c->stream));
    return PNR_OK;
}

// tracker tables for the 3-D (default) or the 2-D (single-slice stack, P == 1) branch of Tracker::Tracker: built on the host,
// resident on the device.  Rebuilt when pnr_set_volume changes the dimensionality.
static int load_tables(pnr_ctx *c, bool is2d)
{
    hipFree(c->d_p); hipFree(c->d_u); hipFree(c->d_w0); hipFree(c->d_w0cws); hipFree(c->d_v); hipFree(c->d_w);
    hipFree(c->d_wcws); hipFree(c->d_tmpl); hipFree(c->d_corrc); hipFree(c->d_sig); hipFree(c->d_M); hipFree(c->d_moff);
    hipFree(c->d_rng); hipFree(c->d_grid); hipFree(c->d_axes); hipFree(c->d_axes_off); hipFree(c->d_wd);
    c->d_p = c->d_u = c->d_w0 = c->d_w0cws = c->d_v = c->d_w = c->d_wcws = c->d_tmpl = c->d_corrc = c->d_sig = nullptr;
    c->d_M = c->d_moff = nullptr; c->d_rng = nullptr; c->d_grid = nullptr; c->d_axes = nullptr; c->d_axes_off = nullptr; c->d_wd = nullptr;
    pnr::build_tables(c->prm, is2d, c->tab);
    const pnr::Tables &t = c->tab;
    std::vector<float> sig(c->prm.sig, c->prm.sig + c->prm.nsig);
    int rc = upload(&c->d_p, t.p, c->stream);
    if (!rc) rc = upload(&c->d_u, t.u, c->stream);
    if (!rc) rc = upload(&c->d_w0, t.w0, c->stream);
    if (!rc) rc = upload(&c->d_w0cws, t.w0_cws, c->stream);
    if (!rc) rc = upload(&c->d_v, t.v, c->stream);
    if (!rc) rc = upload(&c->d_w, t.w, c->stream);
    if (!rc) rc = upload(&c->d_wcws, t.w_cws, c->stream);
    if (!rc) rc = upload(&c->d_tmpl, t.tmpl, c->stream);
    if (!rc) rc = upload(&c->d_corrc, t.corrc, c->stream);
    if (!rc) rc = upload(&c->d_sig, sig, c->stream);
    if (!rc) rc = upload(&c->d_M, t.M, c->stream);
    if (!rc) rc = upload(&c->d_moff, t.moff, c->stream);
    if (!rc) rc = upload(&c->d_rng, t.rng, c->stream);
    if (!rc) rc = upload(&c->d_grid, t.grid, c->stream);
    if (!rc) rc = upload(&c->d_axes, t.axes, c->stream);
    if (!rc) rc = upload(&c->d_axes_off, t.axes_off, c->stream);
    if (!rc) rc = upload(&c->d_wd, t.wd, c->stream);
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = PNR_E_HIP;
    return rc;
}

extern "C" {

const char *pnr_last_error(void) { return pnr::g_err; }

void pnr_default_params(pnr_params *p)
{
    std::memset(p, 0, sizeof(*p));
    p->sig[0] = 2; p->sig[1] = 4; p->sig[2] = 6; // README.md:17
    p->nsig = 3;
    p->somaradius = 0;
    p->tolerance = 5;
    p->znccth = 0.3f;
    p->kappa = 3;
    p->step = 2;
    p->ni = 200;
    p->np = 20;
    p->zdist = 2;
    p->nodepervol = 4;
    p->vol = 1;
    p->Kc = 20.0f;
    p->neff_ratio = 0.8f;
    p->alpha = .5f;
    p->beta = .5f;
    p->C = 500;
    p->rng_seed = 42;
    p->max_trace_count = 5000;
}

static int validate(const pnr_params &p)
{
    PNR_REQUIRE(p.nsig >= 1 && p.nsig <= PNR_MAX_SIGMAS, PNR_E_ARG, "neuritesigmas: need 1..%d sigmas", PNR_MAX_SIGMAS);
    for (int i = 0; i < p.nsig; i++) {
        PNR_REQUIRE(p.sig[i] > 0 && p.sig[i] <= 20, PNR_E_ARG, "neuritesigmas out of range");
        PNR_REQUIRE(i == 0 || p.sig[i] >= p.sig[i - 1], PNR_E_ARG, "neuritesigmas must be sorted ascending");
    }
    // messages follow Advantra_plugin.cpp:317-326
    PNR_REQUIRE(p.somaradius >= 0, PNR_E_ARG, "somaradius out of range");
    PNR_REQUIRE(p.somaradius <= 21, PNR_E_ARG, "somaradius %d too large (Gaussian radius 3*somaradius > 64)", p.somaradius);
    PNR_REQUIRE(p.tolerance >= 0, PNR_E_ARG, "tolerance out of range");
    PNR_REQUIRE(p.znccth >= 0 && p.znccth <= 1, PNR_E_ARG, "znccth out of range");
    PNR_REQUIRE(p.kappa >= 0 && p.kappa <= 5, PNR_E_ARG, "kappa out of range");
    PNR_REQUIRE(p.step >= 1 && p.step <= 8, PNR_E_ARG, "step out of range");
    PNR_REQUIRE(p.ni > 0, PNR_E_ARG, "ni out of range");
    PNR_REQUIRE(p.np > 0 && p.np <= 4096, PNR_E_ARG, "np out of range");
    PNR_REQUIRE(p.zdist >= 1, PNR_E_ARG, "zdist out of range");
    PNR_REQUIRE(p.nodepervol > 2 && p.nodepervol <= 20, PNR_E_ARG, "nodepervol out of range");
    PNR_REQUIRE(p.vol == 1 || p.vol == 5 || p.vol == 9 || p.vol == 11 || p.vol == 19 || p.vol == 27, PNR_E_ARG,
                "vol can be 1,5,9,11,19,27");
    return PNR_OK;
}

int pnr_create(const pnr_params *p, int device, pnr_ctx **out)
{
    PNR_REQUIRE(p && out, PNR_E_ARG, "null argument");
    *out = nullptr;
    int rc = validate(*p);
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device: libpnr_hip has no CPU path (MI355X / gfx950 required)");
        return PNR_E_NODEVICE;
    }
    PNR_REQUIRE(device >= 0 && device < ndev, PNR_E_NODEVICE, "device %d not present (%d devices)", device, ndev);
    PNR_HIP(hipSetDevice(device));
    pnr_ctx *c = new pnr_ctx();
    c->prm = *p;
    if (c->prm.max_trace_count <= 0) c->prm.max_trace_count = 5000;
    c->device = device;
    if (hipStreamCreate(&c->own_stream) != hipSuccess) {
        delete c;
        set_error("hipStreamCreate failed");
        return PNR_E_HIP;
    }
    c->stream = c->own_stream;
    (void)hipEventCreate(&c->ev0);
    (void)hipEventCreate(&c->ev1);
    rc = load_tables(c, /*is2d*/ false);
    if (!rc && hipMalloc(&c->d_minmax, 8) != hipSuccess) rc = PNR_E_HIP;
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = PNR_E_HIP;
    if (rc) {
        pnr_destroy(c);
        return rc;
    }
    *out = c;
    return PNR_OK;
}

void pnr_destroy(pnr_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    pnr_job_destroy(c->job);
    pnr_phased_destroy(c->phased);
    if (c->h_j8) (void)hipHostFree(c->h_j8);
    if (c->h_j8v) (void)hipHostFree(c->h_j8v);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (int k = 0; k < pnr_ctx::J8_CHUNKS; k++)
        if (c->j8_ev[k]) (void)hipEventDestroy(c->j8_ev[k]);
    if (c->j8_start) (void)hipEventDestroy(c->j8_start);
    hipFree(c->d_img_owned); hipFree(c->d_stash); hipFree(c->d_slot_busy); hipFree(c->d_den); hipFree(c->d_den_idx); hipFree(c->d_den_val);
    hipFree(c->d_tmpA); hipFree(c->d_tmpB); hipFree(c->d_J);
    hipFree(c->d_Vx); hipFree(c->d_Vy); hipFree(c->d_Vz); hipFree(c->d_J8); hipFree(c->d_minmax);
    for (int s = 0; s < PNR_MAX_SIGMAS; s++) hipFree(c->d_F[s]);
    hipFree(c->d_scale); hipFree(c->d_taps); hipFree(c->d_qh); hipFree(c->d_qidx); hipFree(c->d_qcount);
    hipFree(c->d_p); hipFree(c->d_u); hipFree(c->d_w0); hipFree(c->d_w0cws); hipFree(c->d_v); hipFree(c->d_w);
    hipFree(c->d_wcws); hipFree(c->d_tmpl); hipFree(c->d_corrc); hipFree(c->d_sig); hipFree(c->d_M); hipFree(c->d_moff);
    hipFree(c->d_rng); hipFree(c->d_grid); hipFree(c->d_axes); hipFree(c->d_axes_off); hipFree(c->d_wd);
    for (auto &kv : c->scratch) hipFree(kv.second.p);
    c->resolve_timers();
    for (hipEvent_t e : c->free_events) (void)hipEventDestroy(e);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int pnr_set_stream(pnr_ctx *c, void *s)
{
    PNR_REQUIRE(c, PNR_E_ARG, "null ctx");
    PNR_HIP(hipStreamSynchronize(c->stream));
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return PNR_OK;
}

int pnr_synchronize(pnr_ctx *c)
{
    PNR_REQUIRE(c, PNR_E_ARG, "null ctx");
    PNR_HIP(hipStreamSynchronize(c->stream));
    return PNR_OK;
}

static int set_dims(pnr_ctx *c, int64_t w, int64_t h, int64_t l)
{
    PNR_REQUIRE(c, PNR_E_ARG, "null ctx");
    // P == 1 takes the reference's frangi2d / 2-D tracker branch (Advantra_plugin.cpp:2496-2497, :2526)
    PNR_REQUIRE(w >= 2 && h >= 2 && l >= 1, PNR_E_ARG, "volume must be at least 2x2x1");
    PNR_REQUIRE(w <= 1 << 20 && h <= 1 << 20 && l <= 1 << 20 && w * h < (1LL << 31), PNR_E_ARG, "volume extent too large");
    PNR_HIP(hipSetDevice(c->device));
    c->w = w; c->h = h; c->l = l;
    c->N = w * h * l;
    c->have_j8 = false;
    c->have_v = c->have_scale = false;
    c->frangi_pruned = false;
    c->seeds.clear();
    c->have_soma = false;
    if ((l == 1) != c->tab.is2d) { // the tracker tables depend on the dimensionality (Tracker(..., P == 1, ...))
        PNR_HIP(hipDeviceSynchronize());
        const int rc = load_tables(c, l == 1);
        if (rc) return rc;
    }
    return PNR_OK;
}

int pnr_set_volume(pnr_ctx *c, const uint8_t *img, int64_t w, int64_t h, int64_t l)
{
    PNR_REQUIRE(img, PNR_E_ARG, "null image");
    int rc = set_dims(c, w, h, l);
    if (rc) return rc;
    c->d_img = nullptr; // never leave the context pointing at a freed image if the allocation below fails
    if (c->img_owned_cap < (size_t)c->N) {
        hipFree(c->d_img_owned);
        c->d_img_owned = nullptr;
        c->img_owned_cap = 0;
        PNR_HIP(hipMalloc(&c->d_img_owned, (size_t)c->N));
        c->img_owned_cap = (size_t)c->N;
    }
    PNR_HIP(hipMemcpyAsync(c->d_img_owned, img, (size_t)c->N, hipMemcpyHostToDevice, c->stream));
    PNR_HIP(hipStreamSynchronize(c->stream));
    c->d_img = c->d_img_owned;
    return PNR_OK;
}

int pnr_set_volume_device(pnr_ctx *c, const void *dev_img, int64_t w, int64_t h, int64_t l)
{
    PNR_REQUIRE(dev_img, PNR_E_ARG, "null image");
    int rc = set_dims(c, w, h, l);
    if (rc) return rc;
    hipFree(c->d_img_owned);
    c->d_img_owned = nullptr;
    c->d_img = (const uint8_t *)dev_img;
    return PNR_OK;
}

int pnr_frangi(pnr_ctx *c, float *Jmin, float *Jmax)
{
    PNR_REQUIRE(c && c->d_img, PNR_E_STATE, "pnr_frangi: no volume set");
    PNR_HIP(hipSetDevice(c->device));
    return pnr_frangi_run(c, Jmin, Jmax);
}

int pnr_frangi_slab(pnr_ctx *c, int64_t z_keep0, int64_t z_keep1, float *Jmin, float *Jmax)
{
    PNR_REQUIRE(c && c->d_img, PNR_E_STATE, "pnr_frangi_slab: no volume set");
    PNR_REQUIRE(c->l > 1, PNR_E_ARG, "pnr_frangi_slab: a single-slice stack has no z-slabs");
    PNR_REQUIRE(z_keep0 >= 0 && z_keep0 < z_keep1 && z_keep1 <= c->l, PNR_E_ARG, "kept planes [%lld,%lld) outside [0,%lld)", (long long)z_keep0,
                (long long)z_keep1, (long long)c->l);
    PNR_HIP(hipSetDevice(c->device));
    return pnr_frangi_run_range(c, z_keep0, z_keep1, /*finish*/ false, Jmin, Jmax);
}

int pnr_quantise_j8(pnr_ctx *c, float Jmin, float Jmax)
{
    PNR_REQUIRE(c && c->d_img, PNR_E_STATE, "pnr_quantise_j8: no volume set");
    PNR_HIP(hipSetDevice(c->device));
    if (c->frangi_pruned && !(Jmin == 0.f && Jmax >= c->Jmax_run)) {
        // the last run skipped the solver below the first J8 level of ITS extremes (Jmin = 0, Jmax at least its own maximum): the
        // global extremes of a sharded stack satisfy that; anything else needs the exact response
        c->frangi_exact_once = true;
        c->frangi_recomputes++;
        if (c->opt.trace_timing || c->opt.seed_timing) fprintf(stderr, "[pnr frangi] pnr_quantise_j8(%g, %g): extremes the pruned run did not assume -- exact re-run of every scale\n", Jmin, Jmax);
        const int rc = pnr_frangi_run_range(c, c->fr_zs0, c->fr_zs1, false, nullptr, nullptr);
        if (rc) return rc;
    }
    return pnr_j8_run(c, Jmin, Jmax);
}

int pnr_get_frangi(pnr_ctx *c, float *J, uint8_t *J8, uint8_t *Vx, uint8_t *Vy, uint8_t *Vz)
{
    PNR_REQUIRE(c && c->have_j8 && c->d_J, PNR_E_STATE, "pnr_get_frangi: run pnr_frangi first");
    const size_t n = (size_t)c->N;
    int rc = PNR_OK;
    if ((J || Vx || Vy || Vz) && c->frangi_pruned) {
        // the last run skipped the solver where the response could not reach J8 = 1 (option frangi_prune): J8, the extremes and the
        // seeds are exact, the f32 J and the winning scale of J8 = 0 voxels are not -- recompute without the shortcut, same extremes
        const float jmin = c->Jmin, jmax = c->Jmax;
        c->frangi_exact_once = true;
        c->frangi_recomputes++;
        if (c->opt.trace_timing || c->opt.seed_timing) fprintf(stderr, "[pnr frangi] pnr_get_frangi: J / V asked for after a pruned run -- exact re-run of every scale\n");
        rc = pnr_frangi_run_range(c, c->fr_zs0, c->fr_zs1, false, nullptr, nullptr);
        if (!rc) rc = pnr_j8_run(c, jmin, jmax);
        if (rc) return rc;
    }
    if (Vx || Vy || Vz) rc = pnr_frangi_materialise_v(c); // the pipeline itself only needs the directions at the seeds
    if (!rc) rc = download(c, J, c->d_J, n);
    if (!rc) rc = download(c, J8, c->d_J8, n);
    if (!rc) rc = download(c, Vx, c->d_Vx, n);
    if (!rc) rc = download(c, Vy, c->d_Vy, n);
    if (!rc) rc = download(c, Vz, c->d_Vz, n);
    if (rc) return rc;
    PNR_HIP(hipStreamSynchronize(c->stream));
    return PNR_OK;
}

int pnr_gaussian(pnr_ctx *c, float sig, float *F)
{
    PNR_REQUIRE(c && c->d_img && F, PNR_E_STATE, "pnr_gaussian: no volume set");
    PNR_REQUIRE(sig > 0, PNR_E_ARG, "sigma must be positive");
    int rc = pnr_ensure_frangi_buffers(c);
    if (!rc) rc = pnr_ensure_tmpA(c);
    if (rc) return rc;
    rc = pnr_gaussian_run(c, sig, c->d_tmpA);
    if (rc) return rc;
    PNR_HIP(hipMemcpy(F, c->d_tmpA, (size_t)c->N * 4, hipMemcpyDeviceToHost));
    return PNR_OK;
}

int pnr_hessian(pnr_ctx *c, float sig, float *Dzz, float *Dyy, float *Dyz, float *Dxx, float *Dxy, float *Dxz)
{
    PNR_REQUIRE(c && c->d_img, PNR_E_STATE, "pnr_hessian: no volume set");
    PNR_REQUIRE(c->l > 1, PNR_E_ARG, "pnr_hessian taps the 3-D Hessian: not defined for a single-slice stack");
    PNR_REQUIRE(sig > 0, PNR_E_ARG, "sigma must be positive");
    int rc = pnr_ensure_frangi_buffers(c);
    if (rc) return rc;
    float *d[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int k = 0; k < 6; k++)
        if (hipMalloc(&d[k], (size_t)c->N * 4) != hipSuccess) {
            for (int j = 0; j < k; j++) hipFree(d[j]);
            set_error("hipMalloc failed for Hessian tap");
            return PNR_E_NOMEM;
        }
    rc = pnr_hessian_run(c, sig, d);
    float *hst[6] = {Dzz, Dyy, Dyz, Dxx, Dxy, Dxz};
    for (int k = 0; k < 6 && !rc; k++)
        if (hst[k] && hipMemcpy(hst[k], d[k], (size_t)c->N * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = PNR_E_HIP;
    for (int k = 0; k < 6; k++) hipFree(d[k]);
    c->have_j8 = false; // tmp buffers were reused
    return rc;
}

int pnr_set_j8_v(pnr_ctx *c, const uint8_t *J8, const uint8_t *Vx, const uint8_t *Vy, const uint8_t *Vz)
{
    PNR_REQUIRE(c && c->N > 0, PNR_E_STATE, "pnr_set_j8_v: set a volume first (for the dimensions)");
    PNR_REQUIRE(J8 && Vx && Vy && Vz, PNR_E_ARG, "null argument");
    int rc = pnr_ensure_frangi_buffers(c);
    if (!rc) rc = pnr_ensure_v(c);
    if (rc) return rc;
    const size_t n = (size_t)c->N;
    PNR_HIP(hipMemcpyAsync(c->d_J8, J8, n, hipMemcpyHostToDevice, c->stream));
    PNR_HIP(hipMemcpyAsync(c->d_Vx, Vx, n, hipMemcpyHostToDevice, c->stream));
    PNR_HIP(hipMemcpyAsync(c->d_Vy, Vy, n, hipMemcpyHostToDevice, c->stream));
    PNR_HIP(hipMemcpyAsync(c->d_Vz, Vz, n, hipMemcpyHostToDevice, c->stream));
    PNR_HIP(hipStreamSynchronize(c->stream));
    c->have_j8 = true;
    c->have_v = true;
    c->have_scale = false;
    c->frangi_pruned = false; // what is in HBM now is the caller's, not a run of pnr_frangi
    return PNR_OK;
}

int pnr_extract_seeds_range(pnr_ctx *c, int64_t z0, int64_t z1, const pnr_seed **seeds, int64_t *n)
{
    PNR_REQUIRE(c && seeds && n, PNR_E_ARG, "null argument");
    PNR_HIP(hipSetDevice(c->device));
    int rc = pnr_seeds_run(c, z0, z1);
    if (rc) return rc;
    *seeds = c->seeds.data();
    *n = (int64_t)c->seeds.size();
    return PNR_OK;
}

int pnr_extract_seeds(pnr_ctx *c, const pnr_seed **seeds, int64_t *n)
{
    PNR_REQUIRE(c, PNR_E_ARG, "null ctx");
    return pnr_extract_seeds_range(c, 0, c->l, seeds, n);
}

int pnr_zncc_batch(pnr_ctx *c, const float *pos_dir, int64_t n, float *corr, float *sig)
{
    PNR_REQUIRE(c && (n == 0 || (pos_dir && corr)), PNR_E_ARG, "null argument");
    PNR_REQUIRE(n >= 0, PNR_E_ARG, "negative count");
    PNR_HIP(hipSetDevice(c->device));
    return pnr_zncc_run(c, pos_dir, n, corr, sig);
}

// the three steps of the seed filter (Advantra_plugin.cpp:2561-2586); score_only / presorted let several GPUs score their own
// seeds and sort the merged list
static int seed_filter(pnr_ctx *c, pnr_seed *seeds, int64_t n, int64_t *n_out, bool score, bool sort)
{
    PNR_REQUIRE(c && n_out && (n == 0 || seeds), PNR_E_ARG, "null argument");
    *n_out = 0;
    if (n == 0) return PNR_OK;
    if (score && c->prm.somaradius > 0) { // seeds inside a soma are dropped before they are scored (:2561-2564)
        PNR_REQUIRE(c->have_soma, PNR_E_STATE, "somaradius > 0: call pnr_soma before the seeds are filtered");
        int64_t m = 0;
        for (int64_t i = 0; i < n; i++) {
            const int64_t j = (int64_t)(int)std::round(seeds[i].z) * c->w * c->h + (int64_t)(int)std::round(seeds[i].y) * c->w + (int)std::round(seeds[i].x);
            if (c->soma_map.find(j) == c->soma_map.end()) seeds[m++] = seeds[i];
        }
        n = m;
        if (n == 0) return PNR_OK;
    }
    if (score) {
        std::vector<float> pd((size_t)n * 6), corr((size_t)n);
        for (int64_t i = 0; i < n; i++) {
            float *q = &pd[(size_t)i * 6];
            q[0] = seeds[i].x; q[1] = seeds[i].y; q[2] = seeds[i].z;
            q[3] = seeds[i].vx; q[4] = seeds[i].vy; q[5] = seeds[i].vz;
        }
        int rc = pnr_zncc_batch(c, pd.data(), n, corr.data(), nullptr);
        if (rc) return rc;
        for (int64_t i = 0; i < n; i++) seeds[i].corr = corr[i];
    }
    std::vector<pnr_seed> kept;
    for (int64_t i = 0; i < n; i++)
        if (!(seeds[i].corr < c->prm.znccth)) kept.push_back(seeds[i]); // erase if corr < znccth (:2571)
    std::vector<int64_t> order(kept.size());
    std::iota(order.begin(), order.end(), 0);
    // std::sort with CompareSeedCorr is unstable in the reference; ties are broken by original index here
    if (sort) std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return kept[a].corr > kept[b].corr; });
    for (size_t i = 0; i < order.size(); i++) seeds[i] = kept[order[i]];
    *n_out = (int64_t)kept.size();
    return PNR_OK;
}

int pnr_score_filter_sort_seeds(pnr_ctx *c, pnr_seed *seeds, int64_t n, int64_t *n_out) { return seed_filter(c, seeds, n, n_out, true, true); }
int pnr_score_filter_seeds(pnr_ctx *c, pnr_seed *seeds, int64_t n, int64_t *n_out) { return seed_filter(c, seeds, n, n_out, true, false); }
int pnr_sort_seeds(pnr_ctx *c, pnr_seed *seeds, int64_t n, int64_t *n_out) { return seed_filter(c, seeds, n, n_out, false, true); }

int pnr_trace_batch(pnr_ctx *c, const pnr_seed *seeds, int64_t n, int32_t *T, int32_t *stop, pnr_xest *xc, int dbg_iters,
                    float *xfilt, int32_t *idxres, float *neff)
{
    PNR_REQUIRE(c && (n == 0 || (seeds && T && stop && xc)), PNR_E_ARG, "null argument");
    PNR_REQUIRE(n >= 0, PNR_E_ARG, "negative count");
    PNR_HIP(hipSetDevice(c->device));
    return pnr_trace_run(c, seeds, n, T, stop, xc, dbg_iters, xfilt, idxres, neff, /*use_density*/ 0);
}

// ---- soma path ---------------------------------------------------------------------------
int pnr_soma(pnr_ctx *c, uint8_t *E8_out, int32_t *threshold, int64_t *n_soma)
{
    PNR_REQUIRE(c, PNR_E_ARG, "null ctx");
    PNR_HIP(hipSetDevice(c->device));
    int rc = pnr_soma_run(c, E8_out, threshold);
    if (rc) return rc;
    if (n_soma) *n_soma = (int64_t)c->soma_nodes.size();
    return PNR_OK;
}

int pnr_get_soma(pnr_ctx *c, pnr_node *nodes, int64_t cap_nodes, int64_t *n_nodes, int64_t *vox, int32_t *lab, int64_t cap_vox, int64_t *n_vox)
{
    PNR_REQUIRE(c, PNR_E_ARG, "null ctx");
    PNR_REQUIRE(c->have_soma, PNR_E_STATE, "pnr_soma has not run");
    if (n_nodes) *n_nodes = (int64_t)c->soma_nodes.size();
    if (n_vox) *n_vox = (int64_t)c->soma_vox.size();
    if (nodes) std::memcpy(nodes, c->soma_nodes.data(), sizeof(pnr_node) * (size_t)std::min<int64_t>(cap_nodes, (int64_t)c->soma_nodes.size()));
    const size_t m = (size_t)std::min<int64_t>(cap_vox, (int64_t)c->soma_vox.size());
    if (vox) std::memcpy(vox, c->soma_vox.data(), 8 * m);
    if (lab) std::memcpy(lab, c->soma_lab.data(), 4 * m);
    return PNR_OK;
}

// ---- host replay ------------------------------------------------------------------------
int pnr_replay_traces_ctx(pnr_ctx *c, const pnr_seed *seeds, int64_t n, const int32_t *T, const pnr_xest *xc, pnr_node *nodes,
                          int64_t cap_nodes, int64_t *n_nodes, int32_t *links, int64_t cap_links, int64_t *n_links, int64_t *n_traces_used)
{
    PNR_REQUIRE(c && n_nodes && n_links && (n == 0 || (seeds && T && xc)), PNR_E_ARG, "null argument");
    PNR_REQUIRE(c->w > 0, PNR_E_STATE, "no volume set");
    PNR_REQUIRE(c->prm.somaradius == 0 || c->have_soma, PNR_E_STATE, "somaradius > 0: call pnr_soma first");
    pnr::Replayer r(c->prm, c->w, c->h, c->l);
    r.set_soma(&c->soma_map, c->soma_nodes);
    r.add(seeds, n, T, xc);
    *n_nodes = (int64_t)r.nodes.size();
    *n_links = (int64_t)r.links.size() / 2;
    if (nodes) std::memcpy(nodes, r.nodes.data(), sizeof(pnr_node) * (size_t)std::min<int64_t>(cap_nodes, *n_nodes));
    if (links) std::memcpy(links, r.links.data(), 8 * (size_t)std::min<int64_t>(cap_links, *n_links));
    if (n_traces_used) *n_traces_used = r.trace_count;
    return PNR_OK;
}

int pnr_replay_traces(const pnr_params *p, int64_t w, int64_t h, int64_t l, const pnr_seed *seeds, int64_t n,
                      const int32_t *T, const pnr_xest *xc, pnr_node *nodes, int64_t cap_nodes, int64_t *n_nodes,
                      int32_t *links, int64_t cap_links, int64_t *n_links, int64_t *n_traces_used)
{
    PNR_REQUIRE(p && n_nodes && n_links && (n == 0 || (seeds && T && xc)), PNR_E_ARG, "null argument");
    PNR_REQUIRE(w > 0 && h > 0 && l > 0, PNR_E_ARG, "bad dimensions");
    pnr::Replayer r(*p, w, h, l);
    r.add(seeds, n, T, xc);
    *n_nodes = (int64_t)r.nodes.size();
    *n_links = (int64_t)r.links.size() / 2;
    if (nodes) std::memcpy(nodes, r.nodes.data(), sizeof(pnr_node) * (size_t)std::min<int64_t>(cap_nodes, *n_nodes));
    if (links) std::memcpy(links, r.links.data(), 8 * (size_t)std::min<int64_t>(cap_links, *n_links));
    if (n_traces_used) *n_traces_used = r.trace_count;
    return PNR_OK;
}

// the graph of the last pnr_trace_replay / pnr_trace_replay_sharded stays in the context (pnr_get_graph): a caller whose buffers
// were too small fetches it again instead of tracing again
static int store_graph(pnr_ctx *c, pnr::Replayer &r, pnr_node *nodes, int64_t cap_nodes, int64_t *n_nodes, int32_t *links, int64_t cap_links,
                       int64_t *n_links, int64_t *n_traces_used)
{
    c->graph_nodes.swap(r.nodes);
    c->graph_links.swap(r.links);
    c->graph_traces = r.trace_count;
    c->have_graph = true;
    *n_nodes = (int64_t)c->graph_nodes.size();
    *n_links = (int64_t)c->graph_links.size() / 2;
    if (nodes) std::memcpy(nodes, c->graph_nodes.data(), sizeof(pnr_node) * (size_t)std::min<int64_t>(cap_nodes, *n_nodes));
    if (links) std::memcpy(links, c->graph_links.data(), 8 * (size_t)std::min<int64_t>(cap_links, *n_links));
    if (n_traces_used) *n_traces_used = c->graph_traces;
    return PNR_OK;
}

int pnr_get_graph(pnr_ctx *c, pnr_node *nodes, int64_t cap_nodes, int64_t *n_nodes, int32_t *links, int64_t cap_links, int64_t *n_links)
{
    PNR_REQUIRE(c && n_nodes && n_links, PNR_E_ARG, "null argument");
    PNR_REQUIRE(c->have_graph, PNR_E_STATE, "no node graph: pnr_trace_replay has not run");
    *n_nodes = (int64_t)c->graph_nodes.size();
    *n_links = (int64_t)c->graph_links.size() / 2;
    if (nodes) std::memcpy(nodes, c->graph_nodes.data(), sizeof(pnr_node) * (size_t)std::min<int64_t>(cap_nodes, *n_nodes));
    if (links) std::memcpy(links, c->graph_links.data(), 8 * (size_t)std::min<int64_t>(cap_links, *n_links));
    return PNR_OK;
}

int pnr_get_trace_log(pnr_ctx *c, int32_t *rec, int64_t cap, int64_t *n)
{
    PNR_REQUIRE(c && n, PNR_E_ARG, "null argument");
    PNR_REQUIRE(c->have_graph, PNR_E_STATE, "no node graph: pnr_trace_replay has not run");
    *n = (int64_t)c->graph_log.size() / 5;
    if (rec) std::memcpy(rec, c->graph_log.data(), 20 * (size_t)std::min<int64_t>(cap, *n));
    return PNR_OK;
}

static int trace_replay_impl(pnr_ctx *c, const pnr_seed *seeds, int64_t n, int64_t first_batch, const pnr::ShardSpec &sh, pnr_node *nodes,
                             int64_t cap_nodes, int64_t *n_nodes, int32_t *links, int64_t cap_links, int64_t *n_links, int64_t *n_traces_used,
                             int64_t *n_iterations)
{
    PNR_REQUIRE(c && n_nodes && n_links && (n == 0 || seeds), PNR_E_ARG, "null argument");
    const int ni = c->prm.ni;
    // Sharded: a rank that fails here, before its scheduler runs, still owes the other ranks the exchange they are about to enter
    // (include/pnr_hip.h: "a rank that fails says so in one last exchange") -- one block with the abort word set.  n == 0 is the
    // same on every rank and exchanges nothing.
    auto prepare = [&]() -> int {
        PNR_REQUIRE(c->d_img, PNR_E_STATE, "no volume set");
        PNR_HIP(hipSetDevice(c->device));
        PNR_REQUIRE(c->prm.somaradius == 0 || c->have_soma, PNR_E_STATE, "somaradius > 0: call pnr_soma first");
        PNR_REQUIRE((c->smc_driver == 0 && !c->opt.replay_batches) || sh.world <= 1, PNR_E_STATE,
                    "sharded tracing needs the phased driver with the streaming scheduler");
        return pnr_density_reset(c);
    };
    c->have_graph = false;
    c->graph_log.clear();
    int rc = prepare();
    if (rc) {
        if (n > 0) pnr::abort_exchange(sh, ni);
        return rc;
    }
    pnr::Replayer r(c->prm, c->w, c->h, c->l);
    r.set_soma(&c->soma_map, c->soma_nodes);
    std::vector<pnr::Replayer::TraceEnd> ends;
    if (c->opt.trace_log) r.log = &ends;
    int64_t iters = 0;
    // phased driver: a window of trace slots refilled as traces stop (smc_phased.hip); `first_batch` has no meaning there
    const bool streaming = c->smc_driver == 0 && !c->opt.replay_batches;
    if (streaming) {
        rc = pnr_trace_replay_stream(c, seeds, n, r, sh, &iters);
        if (rc) return rc;
        n = 0; // nothing left for the batch loop below
    }
    // persistent driver: strictly sequential seed-rank batches (128, 256, ... 1024 seeds): the freshest map a batch scheme can
    // have.  (Several batches in flight on separate streams were slower: they must be collected in rank order, so the in-flight
    // count collapses behind every slow batch, and a staler map costs iterations.)
    int64_t batch = first_batch > 0 ? first_batch : 128;
    const int64_t growth_pct = std::max(100, c->opt.batch_growth);
    const int64_t batch_max = std::max(1, c->opt.batch_max);
    std::vector<pnr_seed> bs;
    std::vector<int32_t> T, stop;
    std::vector<pnr_xest> xc;
    for (int64_t next = 0; next < n && !r.stopped;) {
        const int64_t i1 = std::min(n, next + std::min(batch, batch_max));
        bs.clear();
        for (int64_t i = next; i < i1; i++) // a seed on a saturated voxel is skipped by the replay whatever its traces are: not launched
            if (!r.seed_saturated(seeds[i])) bs.push_back(seeds[i]);
        next = i1;
        if (batch < batch_max) batch = std::max<int64_t>(batch + 1, batch * growth_pct / 100);
        const int64_t m = (int64_t)bs.size();
        T.assign((size_t)(2 * m), 0);
        stop.assign((size_t)(2 * m), 0);
        xc.resize((size_t)(2 * m) * ni);
        rc = pnr_trace_run(c, bs.data(), m, T.data(), stop.data(), xc.data(), 0, nullptr, nullptr, nullptr, /*use_density*/ 1);
        if (rc) return rc;
        int64_t bi = 0, bmax = 0;
        for (int64_t j = 0; j < 2 * m; j++) {
            const int64_t e = std::min<int64_t>(T[(size_t)j] + 1, ni);
            bi += e;
            bmax = std::max(bmax, e);
        }
        iters += bi;
        if (c->opt.trace_timing)
            fprintf(stderr, "[pnr trace] batch of %lld seeds launched: %lld iterations, longest trace %lld, nodes so far %zu\n", (long long)m,
                    (long long)bi, (long long)bmax, r.nodes.size());
        r.touched.clear();
        r.log_base = (int64_t)ends.size() / 2; // (batch mode: rank among the launched seeds)
        r.add(bs.data(), m, T.data(), xc.data());
        rc = pnr_density_update(c, r);
        if (rc) return rc;
    }
    for (const auto &e : ends) c->graph_log.insert(c->graph_log.end(), {e.seed, e.dir, e.ti_limit, e.reason, e.value});
    if (n_iterations) *n_iterations = iters;
    return store_graph(c, r, nodes, cap_nodes, n_nodes, links, cap_links, n_links, n_traces_used);
}

// Trace + replay with early DENSITY stops (the production form of the trace loop, Advantra_plugin.cpp:2658-2710).
// Tracing every seed to its map-free end wastes most GPU iterations: in the reference a trace stops as
// soon as it runs into voxels that earlier traces already filled (DENSITY stop, tracker.cpp:855,870-882),
// and a seed on a filled voxel is never traced (:2669-2670).  The density map produced by the replay of
// lower-ranked seeds is kept on the GPU; the kernel ends a trace at the first iteration whose centroid
// voxel is already saturated in that (stale) map.  A stale map only under-counts, so a trace is never cut
// earlier than the sequential reference would cut it, and the replay -- which applies the true map --
// produces exactly the same nodes and links as the one-shot form.
int pnr_trace_replay(pnr_ctx *c, const pnr_seed *seeds, int64_t n, int64_t first_batch, pnr_node *nodes, int64_t cap_nodes,
                     int64_t *n_nodes, int32_t *links, int64_t cap_links, int64_t *n_links, int64_t *n_traces_used,
                     int64_t *n_iterations)
{
    return trace_replay_impl(c, seeds, n, first_batch, pnr::ShardSpec{}, nodes, cap_nodes, n_nodes, links, cap_links, n_links, n_traces_used, n_iterations);
}

int pnr_trace_replay_sharded(pnr_ctx *c, const pnr_seed *seeds, int64_t n, int rank, int world, pnr_allgather_fn exchange, void *user,
                             pnr_node *nodes, int64_t cap_nodes, int64_t *n_nodes, int32_t *links, int64_t cap_links, int64_t *n_links,
                             int64_t *n_traces_used, int64_t *n_iterations)
{
    PNR_REQUIRE(c, PNR_E_ARG, "null ctx");
    PNR_REQUIRE(world >= 1 && rank >= 0 && rank < world, PNR_E_ARG, "rank %d outside a world of %d", rank, world);
    PNR_REQUIRE(world == 1 || exchange, PNR_E_ARG, "sharded tracing needs an exchange callback");
    pnr::ShardSpec sh;
    sh.rank = rank; sh.world = world; sh.exchange = exchange; sh.user = user; sh.block_bytes = c->opt.exchange_block;
    return trace_replay_impl(c, seeds, n, 0, sh, nodes, cap_nodes, n_nodes, links, cap_links, n_links, n_traces_used, n_iterations);
}

// ---- the scheduler of pnr_trace_replay[_sharded] over a HOST engine that plays map-free traces (no GPU involved) -----------
namespace {
struct PlaybackEngine final : pnr::StreamEngine {
    // (s_*: the state wait() hands to the scheduler -- the one behind the first poll - lag steps of the last launch, as PhasedEngine's)
    struct Slot { bool used = false, done = false, s_done = false; int it = 0, T = 0, Tfree = 0, s_it = 0, s_T = 0; std::vector<pnr_xest> xc; };
    const pnr_params &prm;
    int64_t W, H;
    int ni, nslots;
    pnr_trace_fn fn;
    void *user;
    std::vector<Slot> sl;
    std::vector<std::vector<int>> active; // per group
    // stream order, modelled: a control() is queued on its group's stream and has only run for sure once that group has been waited
    // for (or settled); until then no OTHER group may be handed a slot it names (the HIP engine would race: advisor finding, round 3)
    std::vector<int> ctl_pending; // per slot: the group whose control() names it and may not have run yet, or -1
    int s_active[4] = {0, 0, 0, 0};
    std::unordered_map<int64_t, uint8_t> den;
    std::string msg;
    int64_t steps = 0;
    PlaybackEngine(const pnr_params &p, int64_t w, int64_t h, int window, pnr_trace_fn f, void *u)
        : prm(p), W(w), H(h), ni(p.ni), nslots(window), fn(f), user(u), sl((size_t)window), active(4), ctl_pending((size_t)window, -1) {}
    const char *error() const override { return msg.c_str(); }
    int slots() const override { return nslots; }
    int max_groups() const override { return 4; }
    int admit(int g, const int *slots, const float *s6, int m) override
    {
        for (int j = 0; j < m; j++) {
            if (ctl_pending[(size_t)slots[j]] >= 0 && ctl_pending[(size_t)slots[j]] != g) {
                msg = "slot handed to another trace group before the control() that frees it has run";
                return PNR_E_STATE;
            }
            Slot &s = sl[(size_t)slots[j]];
            s = Slot();
            s.used = true;
            s.xc.assign((size_t)ni, pnr_xest{});
            int32_t T = 0;
            const int rc = fn(user, s6 + (size_t)j * 6, &T, s.xc.data());
            if (rc) { msg = "trace callback failed"; return PNR_E_STATE; }
            s.Tfree = T;
            active[(size_t)g].push_back(slots[j]);
        }
        return PNR_OK;
    }
    void snapshot(int g)
    {
        s_active[g] = (int)active[(size_t)g].size();
        for (Slot &s : sl) { s.s_done = s.done; s.s_it = s.it; s.s_T = s.T; } // (all slots: PhasedEngine copies the whole flag array, too)
    }
    int launch(int g, int, int poll, int lag) override
    {
        // what ph_update does with a trace, on the recorded map-free result: iteration `it` fails (T = it), or succeeds and its
        // centroid voxel is saturated in the replayed map (DENSITY stop, T = it + 1), or the trace goes on
        for (int k = 0; k < poll; k++) {
            std::vector<int> keep;
            for (int slot : active[(size_t)g]) {
                Slot &s = sl[(size_t)slot];
                if (s.it >= s.Tfree || s.it >= ni) { s.T = s.Tfree; s.done = true; continue; }
                const pnr_xest &e = s.xc[(size_t)s.it];
                const int64_t v = (int64_t)(int)std::round(e.z) * W * H + (int64_t)(int)std::round(e.y) * W + (int)std::round(e.x);
                auto f = den.find(v);
                s.it++;
                if (f != den.end() && (int)f->second >= prm.nodepervol) { s.T = s.it; s.done = true; continue; }
                keep.push_back(slot);
            }
            active[(size_t)g].swap(keep);
            steps++;
            if (k == poll - 1 - lag) snapshot(g);
        }
        return PNR_OK;
    }
    int wait(int g, int *act) override
    {
        *act = s_active[g];
        for (int &p : ctl_pending) if (p == g) p = -1;
        return PNR_OK;
    }
    int settle(int g) override
    {
        for (int &p : ctl_pending) if (p == g) p = -1;
        return PNR_OK;
    }
    bool finished(int, int slot, int *T) const override { *T = sl[(size_t)slot].s_T; return sl[(size_t)slot].s_done; }
    const pnr_xest *rows(int slot) const override { return sl[(size_t)slot].xc.data(); }
    int progress(int, int slot) const override { return sl[(size_t)slot].s_it; }
    int control(int g, const int *pause, int np, const int *resume, int nr) override
    {
        std::vector<int> &a = active[(size_t)g];
        for (int j = 0; j < np; j++) {
            ctl_pending[(size_t)pause[j]] = g;
            auto f = std::find(a.begin(), a.end(), pause[j]);
            if (f == a.end()) { // (stopped by itself in the steps that ran behind the state the scheduler decided on)
                if (!sl[(size_t)pause[j]].done) { msg = "pause of a trace that is not stepped"; return PNR_E_STATE; }
                continue;
            }
            a.erase(f);
        }
        for (int j = 0; j < nr; j++) {
            if (std::find(a.begin(), a.end(), resume[j]) != a.end()) { msg = "resume of a trace that is stepped"; return PNR_E_STATE; }
            a.push_back(resume[j]);
        }
        return PNR_OK;
    }
    int density_update(const pnr::Replayer &r, int) override
    {
        for (int64_t v : r.touched) den[v] = (uint8_t)r.den_at(v);
        return PNR_OK;
    }
    void drain() override {}
};
} // namespace

int pnr_sched_playback(const pnr_params *p, int64_t w, int64_t h, int64_t l, const pnr_seed *seeds, int64_t n, int rank, int world,
                       pnr_allgather_fn exchange, void *xuser, int64_t block_bytes, pnr_trace_fn trace, void *tuser, int window, int groups,
                       int poll, int look0, int look_pct, pnr_node *nodes, int64_t cap_nodes, int64_t *n_nodes, int32_t *links, int64_t cap_links,
                       int64_t *n_links, int64_t *n_traces_used, int64_t *n_iterations)
{
    if (window < 2) window = 2; // (this entry point has always clamped; only pnr_sched_playback2 treats a window below 2 as a set-up failure)
    return pnr_sched_playback2(p, w, h, l, seeds, n, rank, world, exchange, xuser, block_bytes, trace, tuser, window, groups, poll, look0, look_pct,
                               /*tentative*/ 1, /*target*/ -1, /*lag*/ -1, nodes, cap_nodes, n_nodes, links, cap_links, n_links, n_traces_used, n_iterations);
}

int pnr_sched_playback2(const pnr_params *p, int64_t w, int64_t h, int64_t l, const pnr_seed *seeds, int64_t n, int rank, int world,
                        pnr_allgather_fn exchange, void *xuser, int64_t block_bytes, pnr_trace_fn trace, void *tuser, int window, int groups,
                        int poll, int look0, int look_pct, int tentative, int target, int lag, pnr_node *nodes, int64_t cap_nodes, int64_t *n_nodes,
                        int32_t *links, int64_t cap_links, int64_t *n_links, int64_t *n_traces_used, int64_t *n_iterations)
{
    PNR_REQUIRE(p && trace && n_nodes && n_links && (n == 0 || seeds), PNR_E_ARG, "null argument");
    PNR_REQUIRE(w > 0 && h > 0 && l > 0, PNR_E_ARG, "bad dimensions");
    PNR_REQUIRE(world >= 1 && rank >= 0 && rank < world && (world == 1 || exchange), PNR_E_ARG, "bad rank / world / exchange");
    pnr::ShardSpec sh;
    sh.rank = rank; sh.world = world; sh.exchange = exchange; sh.user = xuser; sh.block_bytes = block_bytes;
    if (window < 2) { // the playback engine's "set-up failure" (what an out-of-memory GPU is to the HIP engine): the other ranks must hear of it
        set_error("playback engine: a window of %d trace slots", window);
        if (n > 0) pnr::abort_exchange(sh, p->ni);
        return PNR_E_ARG;
    }
    pnr::Replayer r(*p, w, h, l);
    window &= ~1;
    PlaybackEngine eng(*p, w, h, window, trace, tuser);
    pnr::SchedOptions o;
    o.window = window; o.groups = std::max(1, groups); o.poll = std::max(1, poll);
    o.look0 = std::max(0, look0); o.look_pct = look_pct;
    o.tentative = tentative != 0;
    o.target = std::max(-1, target);
    o.lag = std::max(-1, lag);
    pnr::SchedStats st;
    std::string err;
    const int rc = pnr::run_stream(eng, seeds, n, p->ni, o, sh, r, &st, err);
    if (rc) { set_error("%s", err.c_str()); return rc; }
    *n_nodes = (int64_t)r.nodes.size();
    *n_links = (int64_t)r.links.size() / 2;
    if (nodes) std::memcpy(nodes, r.nodes.data(), sizeof(pnr_node) * (size_t)std::min<int64_t>(cap_nodes, *n_nodes));
    if (links) std::memcpy(links, r.links.data(), 8 * (size_t)std::min<int64_t>(cap_links, *n_links));
    if (n_traces_used) *n_traces_used = r.trace_count;
    if (n_iterations) *n_iterations = st.iters;
    return PNR_OK;
}

// ---- options ----------------------------------------------------------------------------
namespace {
struct OptEntry { const char *key; int pnr::Options::*i32; int64_t pnr::Options::*i64; int64_t lo, hi; };
const OptEntry OPTS[] = {
    {"window", &pnr::Options::window, nullptr, 0, 1 << 20},      {"look0", &pnr::Options::look0, nullptr, 0, 1 << 24},
    {"look_pct", &pnr::Options::look_pct, nullptr, -1, 100000},   {"poll", &pnr::Options::poll, nullptr, 0, 1024},
    {"groups", &pnr::Options::groups, nullptr, 0, 4},            {"split_x10", &pnr::Options::split_x10, nullptr, 0, 10000},
    {"max_split", &pnr::Options::max_split, nullptr, 1, 4096},   {"stash_mb", nullptr, &pnr::Options::stash_mb, 1, 1 << 20},
    {"host_threads", &pnr::Options::host_threads, nullptr, 0, 1024}, {"local_ranks", &pnr::Options::local_ranks, nullptr, 1, 1024},
    {"trace_timing", &pnr::Options::trace_timing, nullptr, 0, 1}, {"seed_timing", &pnr::Options::seed_timing, nullptr, 0, 1},
    {"trace_log", &pnr::Options::trace_log, nullptr, 0, 1},
    {"replay_batches", &pnr::Options::replay_batches, nullptr, 0, 1}, {"batch_growth", &pnr::Options::batch_growth, nullptr, 100, 100000},
    {"batch_max", &pnr::Options::batch_max, nullptr, 1, 1 << 24}, {"no_stash", &pnr::Options::no_stash, nullptr, 0, 1},
    {"exchange_block", nullptr, &pnr::Options::exchange_block, 0, 1 << 28}, {"frangi_prune", &pnr::Options::frangi_prune, nullptr, 0, 1},
    {"cube_copy", &pnr::Options::cube_copy, nullptr, 0, 1},      {"gauss_march", &pnr::Options::gauss_march, nullptr, 0, 1},
    {"tentative", &pnr::Options::tentative, nullptr, 0, 1},      {"target", &pnr::Options::target, nullptr, -1, 1 << 20},
    {"sums_deep", &pnr::Options::sums_deep, nullptr, -1, 1},      {"sums_deep_max", &pnr::Options::sums_deep_max, nullptr, 0, 1 << 20},
    {"lag", &pnr::Options::lag, nullptr, -1, 1023},               {"profile_every", &pnr::Options::profile_every, nullptr, 1, 1024},
    {"overfill", &pnr::Options::overfill, nullptr, 0, 1},        {"concentrate", &pnr::Options::concentrate, nullptr, 0, 100},
};
} // namespace

int pnr_set_option(pnr_ctx *c, const char *key, int64_t value)
{
    PNR_REQUIRE(c && key, PNR_E_ARG, "null argument");
    if (std::strcmp(key, "recon_timing") == 0) { // process-wide: pnr_reconstruct takes no context
        PNR_REQUIRE(value == 0 || value == 1, PNR_E_ARG, "option recon_timing = %lld outside [0, 1]", (long long)value);
        advantra::set_recon_timing(value != 0);
        return PNR_OK;
    }
    for (const OptEntry &e : OPTS)
        if (std::strcmp(e.key, key) == 0) {
            PNR_REQUIRE(value >= e.lo && value <= e.hi, PNR_E_ARG, "option %s = %lld outside [%lld, %lld]", key, (long long)value, (long long)e.lo, (long long)e.hi);
            if (e.i32) c->opt.*(e.i32) = (int)value; else c->opt.*(e.i64) = value;
            return PNR_OK;
        }
    set_error("unknown option '%s'", key);
    return PNR_E_ARG;
}

int pnr_get_option(pnr_ctx *c, const char *key, int64_t *value)
{
    PNR_REQUIRE(c && key && value, PNR_E_ARG, "null argument");
    if (std::strcmp(key, "recon_timing") == 0) { *value = advantra::recon_timing() ? 1 : 0; return PNR_OK; }
    if (std::strcmp(key, "host_threads_effective") == 0) { *value = pnr::host_threads(c->opt); return PNR_OK; }
    if (std::strcmp(key, "frangi_recomputes") == 0) { *value = c->frangi_recomputes; return PNR_OK; } // exact Frangi re-runs so far (one pnr_frangi of GPU time each)
    for (const OptEntry &e : OPTS)
        if (std::strcmp(e.key, key) == 0) {
            *value = e.i32 ? (int64_t)(c->opt.*(e.i32)) : c->opt.*(e.i64);
            return PNR_OK;
        }
    set_error("unknown option '%s'", key);
    return PNR_E_ARG;
}

int pnr_reconstruct(const pnr_node *nodes, int64_t n_nodes, const int32_t *links, int64_t n_links, float trace_rsmpl,
                    float sig2radius, int refine_iter, float epsilon2, float group_radius, int tree_size_min,
                    pnr_node *out_nodes, int32_t *out_parent, int64_t cap, int64_t *n_out)
{
    PNR_REQUIRE(nodes && n_nodes >= 1 && n_out && (n_links == 0 || links), PNR_E_ARG, "null argument");
    for (int64_t k = 0; k < 2 * n_links; k++) PNR_REQUIRE(links[k] >= 0 && links[k] < n_nodes, PNR_E_ARG, "link index out of range");
    advantra::ReconParams rp;
    if (trace_rsmpl > 0) rp.trace_rsmpl = trace_rsmpl;
    if (sig2radius > 0) rp.sig2radius = sig2radius;
    if (refine_iter > 0) rp.refine_iter = refine_iter;
    if (epsilon2 > 0) rp.epsilon2 = epsilon2;
    if (group_radius > 0) rp.group_radius = group_radius;
    if (tree_size_min > 0) rp.tree_size_min = tree_size_min;
    rp.single_tree = tree_size_min < 0;
    rp.threads = pnr::host_threads(pnr::Options()); // rank 0 post-processes alone: every CPU this process may use
    std::vector<pnr_node> in(nodes, nodes + n_nodes), out;
    std::vector<int32_t> lk(links, links + 2 * n_links), par;
    advantra::reconstruct(in, lk, rp, out, par);
    *n_out = (int64_t)out.size();
    const size_t m = (size_t)std::min<int64_t>(cap, *n_out);
    if (out_nodes) std::memcpy(out_nodes, out.data(), sizeof(pnr_node) * m);
    if (out_parent) std::memcpy(out_parent, par.data(), 4 * m);
    return PNR_OK;
}

int pnr_reconstruct_stage(const pnr_node *nodes, int64_t n_nodes, const int32_t *links, int64_t n_links, float trace_rsmpl, float sig2radius,
                          int refine_iter, float epsilon2, float group_radius, int stage, pnr_node *out_nodes, int64_t cap_nodes,
                          int64_t *n_out_nodes, int32_t *out_links, int64_t cap_links, int64_t *n_out_links)
{
    PNR_REQUIRE(nodes && n_nodes >= 1 && n_out_nodes && n_out_links && (n_links == 0 || links), PNR_E_ARG, "null argument");
    PNR_REQUIRE(stage >= advantra::RECON_N0RES && stage <= advantra::RECON_N2TREE, PNR_E_ARG, "stage %d outside [1, 4]", stage);
    for (int64_t k = 0; k < 2 * n_links; k++) PNR_REQUIRE(links[k] >= 0 && links[k] < n_nodes, PNR_E_ARG, "link index out of range");
    advantra::ReconParams rp;
    if (trace_rsmpl > 0) rp.trace_rsmpl = trace_rsmpl;
    if (sig2radius > 0) rp.sig2radius = sig2radius;
    if (refine_iter > 0) rp.refine_iter = refine_iter;
    if (epsilon2 > 0) rp.epsilon2 = epsilon2;
    if (group_radius > 0) rp.group_radius = group_radius;
    rp.threads = pnr::host_threads(pnr::Options());
    std::vector<pnr_node> in(nodes, nodes + n_nodes), out;
    std::vector<int32_t> lk(links, links + 2 * n_links), par, sl;
    advantra::reconstruct(in, lk, rp, out, par, stage, &sl);
    *n_out_nodes = (int64_t)out.size();
    *n_out_links = (int64_t)sl.size() / 2;
    if (out_nodes) std::memcpy(out_nodes, out.data(), sizeof(pnr_node) * (size_t)std::min<int64_t>(cap_nodes, *n_out_nodes));
    if (out_links) std::memcpy(out_links, sl.data(), 8 * (size_t)std::min<int64_t>(cap_links, *n_out_links));
    return PNR_OK;
}

int pnr_get_table(pnr_ctx *c, const char *name, void *out, int64_t cap, int64_t *n)
{
    PNR_REQUIRE(c && name && n, PNR_E_ARG, "null argument");
    const pnr::Tables &t = c->tab;
    const void *src = nullptr;
    size_t cnt = 0;
    std::string nm(name);
    auto fv = [&](const std::vector<float> &v) { src = v.data(); cnt = v.size(); };
    std::vector<float> tmp;
    if (nm == "p") fv(t.p);
    else if (nm == "u") fv(t.u);
    else if (nm == "w0") fv(t.w0);
    else if (nm == "w0_cws") fv(t.w0_cws);
    else if (nm == "v") fv(t.v);
    else if (nm == "w") fv(t.w);
    else if (nm == "w_cws") fv(t.w_cws);
    else if (nm == "model_avg") fv(t.mavg);
    else if (nm == "rng") { src = t.rng.data(); cnt = t.rng.size(); }
    else if (nm.rfind("model_vuw", 0) == 0 || nm.rfind("model_wgt", 0) == 0 || nm.rfind("gauss_xy", 0) == 0 ||
             nm.rfind("gauss_z", 0) == 0) {
        const size_t pre = (nm[0] == 'm') ? 9 : (nm[6] == 'x' ? 8 : 7);
        const int s = std::atoi(nm.c_str() + pre);
        PNR_REQUIRE(s >= 0 && s < t.nsig, PNR_E_ARG, "sigma index out of range in '%s'", name);
        if (nm[0] == 'g') fv(nm[6] == 'x' ? t.gxy[s] : t.gz[s]);
        else if (nm[6] == 'w') { src = t.mwgt.data() + t.moff[s]; cnt = (size_t)t.M[s]; }
        else {
            tmp.resize((size_t)t.M[s] * 3);
            for (int k = 0; k < t.M[s]; k++)
                for (int q = 0; q < 3; q++) tmp[(size_t)k * 3 + q] = t.tmpl[((size_t)t.moff[s] + k) * 4 + q];
            fv(tmp);
        }
    } else {
        set_error("unknown table '%s'", name);
        return PNR_E_ARG;
    }
    *n = (int64_t)cnt;
    if (out && cap > 0) std::memcpy(out, src, std::min<size_t>(cnt, (size_t)cap) * 4);
    return PNR_OK;
}

int pnr_set_smc_driver(pnr_ctx *c, int driver)
{
    PNR_REQUIRE(c, PNR_E_ARG, "null ctx");
    PNR_REQUIRE(driver == 0 || driver == 1, PNR_E_ARG, "smc driver must be 0 (phased) or 1 (persistent), got %d", driver);
    c->smc_driver = driver;
    return PNR_OK;
}

int pnr_set_profiling(pnr_ctx *c, int enable)
{
    PNR_REQUIRE(c, PNR_E_ARG, "null ctx");
    c->profiling = enable != 0;
    return PNR_OK;
}

int pnr_get_kernel_ms(pnr_ctx *c, const char *group, double *ms, int64_t *launches)
{
    PNR_REQUIRE(c && group && ms, PNR_E_ARG, "null argument");
    c->resolve_timers();
    auto it = c->timers.find(group);
    *ms = (it == c->timers.end()) ? 0.0 : it->second.ms;
    if (launches) *launches = (it == c->timers.end()) ? 0 : it->second.launches;
    return PNR_OK;
}

int pnr_reset_kernel_ms(pnr_ctx *c)
{
    PNR_REQUIRE(c, PNR_E_ARG, "null ctx");
    c->resolve_timers();
    c->timers.clear();
    return PNR_OK;
}

int pnr_expf_batch(pnr_ctx *c, const float *x, int64_t n, float *y)
{
    PNR_REQUIRE(c && (n == 0 || (x && y)), PNR_E_ARG, "null argument");
    return pnr_expf_run(c, x, n, y);
}

int pnr_eigen_batch(pnr_ctx *c, const double *A, int64_t n, double *V, double *d)
{
    PNR_REQUIRE(c && n >= 0 && (n == 0 || (A && d)), PNR_E_ARG, "null argument");
    return pnr_eigen_run(c, A, n, V, d);
}

} // extern "C"
