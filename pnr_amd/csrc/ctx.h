// ctx.h -- internal state of one pnr_ctx (one GPU) and small helpers shared by the
// translation units of libpnr_hip.so.  Not part of the C ABI.
#pragma once
#include "../../include/pnr_hip.h"
#include "../../include/pnr_hip_test.h" // the test taps (implemented beside the product entry points)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

namespace pnr {

void set_error(const char *fmt, ...);

#define PNR_HIP(call)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            pnr::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return PNR_E_HIP;                                                                \
        }                                                                                    \
    } while (0)

#define PNR_REQUIRE(cond, code, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            pnr::set_error(__VA_ARGS__);  \
            return (code);                \
        }                                 \
    } while (0)

// ---- host tables (tables.cpp): Tracker::Tracker (tracker.cpp:79-527) + Gaussian taps ----
struct Tables {
    int sz = 0, ndir = 50, nsig = 0;
    bool is2d = false; // built for a single-slice stack (P == 1)
    std::vector<float> p, u, d0, w0, w0_cws, v, w, w_cws; // p,u: sz x 3; w,w_cws: ndir x sz
    std::vector<int> M;                                   // samples per sigma
    std::vector<int> moff;                                // prefix offsets into tmpl
    std::vector<float> tmpl;                              // sum(M) x 4 : (v_off, u_off, w_off, wgt - avg)
    std::vector<float> mwgt;                              // sum(M)     : raw template weights
    std::vector<float> mavg, corrc;                       // per sigma
    std::vector<int> grid;                                // per sigma: nv, nu, nw, first sample
    std::vector<float> axes;                              // per sigma: vv[nv] | uu[nu] | ww[nw]
    std::vector<int> axes_off;                            // per sigma offset into axes
    std::vector<float> wd;                                // sum(M): wgt - avg
    float ext_v = 0, ext_uw = 0;                          // largest |vv| and |uu|,|ww|
    std::vector<float> ext_vs, ext_uws;                   // the same per sigma
    std::vector<uint32_t> rng;                            // np + 1 glibc rand() draws
    std::vector<std::vector<float>> gxy, gz;              // Gaussian taps per sigma
};
void build_tables(const pnr_params &p, bool is2d, Tables &t);
void glibc_rand_stream(uint32_t seed, int n, uint32_t *out);
int gaussian_taps(float sig, std::vector<float> &g); // returns radius L

// scheduler / host-side knobs of one context (pnr_set_option); none of them changes a result
struct Options {
    int window = 0;           // trace slots the streaming tracer keeps busy (0: automatic -- 1536 on one GPU with the tentative replay, 768 otherwise)
    int look0 = 0, look_pct = -1; // admission lookahead max(look0, frontier * look_pct / 100); 0 / -1 = automatic (stream_sched.h)
    int target = -1;          // running traces the admission keeps up (0: off, -1: automatic -- 120 on one GPU with the tentative replay)
    int sums_deep = -1;       // form of the ordered sums: -1 automatic (smc_phased.hip sums_deep()), 0 two buffers folded, 1 four buffers in turn
    int sums_deep_max = 96;   // ... automatic with several trace groups: the four-buffer form for launches of at most this many traces (round 4: 64 -> 96, -1 % on both bench workloads)
    int lag = -1;             // steps of a poll that run while the host works on the state in front of them (-1 automatic: stream_sched.h)
    int concentrate = 1;      // several trace groups: new seeds go to group 0 only while few traces survive a poll (experiment switch)
    int overfill = 1;         // the target is the mean over a poll, not the count at its start (experiment switch)
    int poll = 0;             // SMC steps between two polls; 0 = automatic (stream_sched.h run_stream: 2 on one or two GPUs, 4 from four ranks on)
    int groups = 0;           // trace groups on separate streams (0: automatic -- 2 on one GPU: one group's ordered sums overlap the other's sampling;
                              // 1 sharded: every poll is then an exchange, and small launches gain nothing from sharing the CUs)
    int split_x10 = 0;        // sampling work-groups per CU x 10 and launch; 0 = automatic (40 with one trace group, 22 with several)
    int max_split = 96;       // ... and at most this many per trace (round 4: 24 -> 96: the last few traces of a late poll are spread over the whole chip)
    int64_t stash_mb = 65536; // sample-stash budget
    int host_threads = 0;     // host worker threads of the seed flood fill / reconstruct(); 0 = hardware threads / local_ranks
    int local_ranks = 1;      // processes that share this host (one per GPU)
    int trace_timing = 0, seed_timing = 0; // statistics on stderr
    int profile_every = 1;    // with pnr_set_profiling: the streaming tracer times the kernels of every n-th poll of a trace group only and counts
                              // them n-fold (the event pairs around ~9000 launches per stack are 1.2 % of the bench step; every 4th: 0.3 %)
    int trace_log = 0;        // keep how every replayed trace ended (pnr_get_trace_log)
    int replay_batches = 0;   // 1: seed-rank batches instead of the streaming window (always so with the persistent driver)
    int batch_growth = 200, batch_max = 1024;
    int no_stash = 0;         // persistent driver: in-lane two-pass sums
    int frangi_prune = 1;       // skip the eigen-solver where the response cannot reach the first non-zero J8 level (frangi.hip)
    int tentative = 1;          // streaming scheduler: pause traces that a tentative replay of everything recorded so far cuts (stream_sched.h)
    int gauss_march = 1;        // the fused x-y Gaussian marches down strips of a slice (gauss_xy_u8_m; 0: one 64 x 64 tile per work-group)
    int cube_copy = 1;          // phased driver: the cube of a trace is fetched from the image once per step (ph_cube) and copied by its sampling work-groups (0: each stages it itself)
    int64_t exchange_block = 0; // bytes per rank and exchange of the sharded tracer; 0 = automatic (256 KB / world, at least 32 KB)
};
int host_threads(const Options &o); // worker threads to use on this host

struct KernelTimer {
    double ms = 0;
    int64_t launches = 0;
};

} // namespace pnr

struct pnr_ctx {
    pnr_params prm;
    pnr::Options opt;
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    pnr::Tables tab;

    // volume
    int64_t w = 0, h = 0, l = 0, N = 0;
    const uint8_t *d_img = nullptr; // device
    uint8_t *d_img_owned = nullptr;
    size_t img_owned_cap = 0;

    // Frangi state (HBM)
    float *d_tmpA = nullptr, *d_tmpB = nullptr, *d_J = nullptr;
    uint8_t *d_Vx = nullptr, *d_Vy = nullptr, *d_Vz = nullptr, *d_J8 = nullptr;
    unsigned int *d_minmax = nullptr; // [0]=min bits, [1]=max bits
    float *d_F[PNR_MAX_SIGMAS] = {};  // smoothed volume of every scale, kept for the direction bytes (frangi.hip)
    uint8_t *d_scale = nullptr;       // per voxel: the scale whose response is in J
    bool frangi_pruned = false, frangi_exact_once = false; // J / the winning scale of J8 = 0 voxels are not exact (option frangi_prune); next run without it
    float Jmax_run = 0.f;                                   // the maximum the last Frangi run found itself (before pnr_quantise_j8)
    int64_t frangi_recomputes = 0;                          // exact re-runs pnr_get_frangi / pnr_quantise_j8 had to make (pnr_get_option "frangi_recomputes")
    int64_t fr_zs0 = 0, fr_zs1 = 0;                         // planes the extremes of the last Frangi run were taken over
    bool have_scale = false, have_v = false; // d_scale + d_F valid / the direction volumes Vx, Vy, Vz are filled
    float *d_taps = nullptr;          // Gaussian taps of all scales (frangi.hip: TAPS_SLOT floats per pass, zero-padded)
    std::vector<float> taps_stage;    // their host staging (asynchronous uploads)
    float *d_qh = nullptr;            // survivor queue of the Hessian stage: [region][6][entries]
    unsigned int *d_qidx = nullptr, *d_qcount = nullptr;
    size_t q_regions = 0;
    int64_t frangi_cap = 0;           // voxels the buffers above were sized for
    bool have_j8 = false;
    float Jmin = 0, Jmax = 0;

    // device tables
    float *d_p = nullptr, *d_u = nullptr, *d_w0 = nullptr, *d_w0cws = nullptr, *d_v = nullptr, *d_w = nullptr,
          *d_wcws = nullptr, *d_tmpl = nullptr, *d_corrc = nullptr, *d_sig = nullptr;
    int *d_M = nullptr, *d_moff = nullptr, *d_grid = nullptr, *d_axes_off = nullptr;
    float *d_axes = nullptr, *d_wd = nullptr;
    uint32_t *d_rng = nullptr;

    // SMC pass-1 sample stash (HBM scratch, sized at the first trace batch)
    float *d_stash = nullptr;
    int *d_slot_busy = nullptr;
    size_t stash_bytes = 0;
    int stash_slots = 0;

    // node-density map of earlier trace batches (u8 per voxel), read by smc_trace for early DENSITY stops
    uint8_t *d_den = nullptr;
    int64_t den_cap = 0;
    long long *d_den_idx = nullptr; // staging for the per-batch scatter of touched voxels
    uint8_t *d_den_val = nullptr;
    size_t den_stage_cap = 0;

    struct pnr_trace_job *job = nullptr; // device buffers of pnr_trace_batch / the persistent driver's batches (ctx stream)
    struct pnr_phased *phased = nullptr; // state of the launch-per-phase SMC driver (smc_phased.hip)
    int smc_driver = 0;                  // 0: launch per phase (default), 1: one persistent work-group per trace

    // soma path (somaradius > 0): one SOMA node per region, sparse voxel -> node-index map (soma.hip)
    std::vector<pnr_node> soma_nodes;
    std::vector<int64_t> soma_vox;   // foreground voxels in raster order
    std::vector<int32_t> soma_lab;   // their region = index in the node list (1-based)
    std::unordered_map<int64_t, int32_t> soma_map;
    bool have_soma = false;

    // seeds
    unsigned char *h_j8 = nullptr, *h_j8v = nullptr; // pinned staging of the sparse J8 hand-over to the flood fill: bitmap | values (seeds.hip)
    size_t h_j8_cap = 0, h_j8v_cap = 0;
    static constexpr int J8_CHUNKS = 16;   // the download is cut into chunks of layers so that the flood fill starts on the first ones
    hipStream_t copy_stream = nullptr;
    hipEvent_t j8_ev[J8_CHUNKS] = {}, j8_start = nullptr;
    std::vector<pnr_seed> seeds;

    // node graph of the last pnr_trace_replay[_sharded] (pnr_get_graph)
    std::vector<pnr_node> graph_nodes;
    std::vector<int32_t> graph_links;
    int graph_traces = 0;
    bool have_graph = false;
    std::vector<int32_t> graph_log; // 5 ints per replayed trace (option "trace_log")

    // profiling: HIP event pairs recorded on the ctx stream around each kernel group, resolved
    // lazily (no host sync inside the timed region)
    bool profiling = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr; // unused legacy pair (kept for create/destroy symmetry)
    struct Pending {
        std::string group;
        hipEvent_t a, b;
        int launches;
        bool a_shared = false; // a is the b of the entry before (chained timers): not freed twice
        int weight = 1;        // sampled timers (option "profile_every"): this measurement stands for `weight` launches like it
    };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> free_events;
    hipEvent_t cur_a = nullptr, last_b = nullptr; // last_b: the closing event of the last toc(), reusable as the next opening event
    hipStream_t last_b_stream = nullptr;
    bool cur_a_shared = false;
    std::map<std::string, pnr::KernelTimer> timers;

    // grow-only named device scratch (per-layer tables, candidate keys, ...): nothing is allocated or freed inside a timed stage
    // once the first pass has sized it
    struct DevScratch { void *p = nullptr; size_t cap = 0; };
    std::map<std::string, DevScratch> scratch;
    template <typename T>
    int scratch_get(const char *name, size_t count, T **out)
    {
        DevScratch &b = scratch[name];
        const size_t need = std::max<size_t>(count, 1) * sizeof(T);
        if (b.cap < need) {
            if (b.p) (void)hipFree(b.p); // (synchronises the device)
            b.p = nullptr;
            b.cap = 0;
            const size_t cap = need + need / 2;
            if (hipMalloc(&b.p, cap) != hipSuccess) {
                pnr::set_error("hipMalloc of %zu B for scratch '%s' failed", cap, name);
                return PNR_E_NOMEM;
            }
            b.cap = cap;
        }
        *out = (T *)b.p;
        return PNR_OK;
    }

    hipEvent_t get_event()
    {
        if (!free_events.empty()) {
            hipEvent_t e = free_events.back();
            free_events.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    // chain = true: nothing was queued on the stream since the last toc() on it -- its closing event opens this timer as well (one
    // event per kernel instead of two in the SMC loops, where the events are ~2 % of a step)
    void tic(hipStream_t on = nullptr, bool chain = false)
    {
        if (!profiling) return;
        hipStream_t s = on ? on : stream;
        if (chain && last_b && last_b_stream == s) {
            cur_a = last_b;
            cur_a_shared = true;
            return;
        }
        cur_a = get_event();
        cur_a_shared = false;
        (void)hipEventRecord(cur_a, s);
    }
    void toc(const char *group, int launches = 1, hipStream_t on = nullptr, int weight = 1)
    {
        if (!profiling || !cur_a) return;
        hipStream_t s = on ? on : stream;
        hipEvent_t b = get_event();
        (void)hipEventRecord(b, s);
        pending.push_back(Pending{group, cur_a, b, launches, cur_a_shared, weight});
        cur_a = nullptr;
        last_b = b;
        last_b_stream = s;
    }
    void resolve_timers()
    {
        for (auto &p : pending) {
            (void)hipEventSynchronize(p.b);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, p.a, p.b);
            auto &t = timers[p.group];
            t.ms += (double)ms * p.weight;
            t.launches += (int64_t)p.launches * p.weight;
            if (!p.a_shared) free_events.push_back(p.a);
            free_events.push_back(p.b);
        }
        pending.clear();
        last_b = nullptr;
    }
};

// stage entry points implemented in the .hip files
int pnr_frangi_run(pnr_ctx *c, float *Jmin, float *Jmax);
int pnr_frangi_run_range(pnr_ctx *c, int64_t zs0, int64_t zs1, bool finish, float *Jmin, float *Jmax);
int pnr_j8_run(pnr_ctx *c, float jmin, float jmax);
int pnr_gaussian_run(pnr_ctx *c, float sig, float *d_out /*device N floats*/);
int pnr_hessian_run(pnr_ctx *c, float sig, float *const d_out[6]);
int pnr_seeds_run(pnr_ctx *c, int64_t z0, int64_t z1);
int pnr_soma_run(pnr_ctx *c, uint8_t *E8_out, int32_t *threshold);
int pnr_gauss_x_u8_launch(pnr_ctx *c, const uint8_t *src, float *dst, const float *d_taps, int L);
int pnr_ensure_tmpA(pnr_ctx *c); // the second Gaussian scratch volume / the direction volumes: allocated where they are used (frangi.hip)
int pnr_ensure_v(pnr_ctx *c);
int pnr_zncc_run(pnr_ctx *c, const float *h_pos_dir, int64_t n, float *h_corr, float *h_sig);
int pnr_trace_run(pnr_ctx *c, const pnr_seed *seeds, int64_t n, int32_t *T, int32_t *stop, pnr_xest *xc,
                  int dbg_iters, float *xfilt, int32_t *idxres, float *neff, int use_density);
namespace pnr { struct Replayer; struct ShardSpec; }
struct pnr_trace_job;
pnr_trace_job *pnr_job_create(pnr_ctx *c, bool own_stream);
void pnr_job_destroy(pnr_trace_job *j);
int pnr_job_launch(pnr_ctx *c, pnr_trace_job *j, const pnr_seed *seeds, int64_t n, int dbg_iters, bool want_xfilt, bool want_idxres,
                   bool want_neff, int use_density);
int pnr_job_finish(pnr_ctx *c, pnr_trace_job *j, int32_t *T, int32_t *stop, pnr_xest *xc, float *xfilt, int32_t *idxres, float *neff);
struct pnr_phased;
int pnr_trace_run_phased(pnr_ctx *c, const pnr_seed *seeds, int64_t n, int32_t *T, int32_t *stop, pnr_xest *xc, int dbg_iters,
                         float *xfilt, int32_t *idxres, float *neff, int use_density);
void pnr_phased_destroy(pnr_phased *h);
int pnr_trace_replay_stream(pnr_ctx *c, const pnr_seed *seeds, int64_t n, pnr::Replayer &r, const pnr::ShardSpec &sh, int64_t *iters);
int pnr_density_reset(pnr_ctx *c);                       // zero the device density map (allocating it on first use)
int pnr_density_update(pnr_ctx *c, const pnr::Replayer &r, hipStream_t on = nullptr); // push the voxels touched since Replayer::touched was cleared
int pnr_density_scatter_async(pnr_ctx *c, const long long *d_idx, const unsigned char *d_val, size_t n, hipStream_t st); // (staging owned by the caller, no wait)
int pnr_expf_run(pnr_ctx *c, const float *x, int64_t n, float *y);
int pnr_eigen_run(pnr_ctx *c, const double *A, int64_t n, double *V, double *d); // test tap (pnr_hip_test.h)
int pnr_ensure_frangi_buffers(pnr_ctx *c);
int pnr_frangi_materialise_v(pnr_ctx *c);
int pnr_seed_dirs(pnr_ctx *c, const long long *d_idx, int n, unsigned char *d_dirs); // 0: done, 1: gather from the volumes, < 0: error
