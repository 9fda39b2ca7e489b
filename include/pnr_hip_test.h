/*
 * pnr_hip_test.h -- TEST TAPS of libpnr_hip.so.  Not part of the drop-in boundary (include/pnr_hip.h): nothing a host of the
 * reference's pipeline calls.  These entry points expose single stages of the device code so that tests/ can compare them with the
 * oracle and with the reference-generated fixtures (tests/golden/), and let the multi-process tests drive the scheduler with a
 * host engine.  Same conventions as pnr_hip.h (int status, pnr_last_error()); file:line = /root/reference/pnr-vaa3d/.
 */
#ifndef PNR_HIP_TEST_H
#define PNR_HIP_TEST_H
#include "pnr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test taps: Frangi::imgaussian (frangi.cpp:647) and Frangi::hessian3d (:291) for one sigma.
 * Host outputs, N floats each; Hessian order Dzz,Dyy,Dyz,Dxx,Dxy,Dxz (any may be NULL). */
int pnr_gaussian(pnr_ctx *ctx, float sig, float *F);
int pnr_hessian(pnr_ctx *ctx, float sig, float *Dzz, float *Dyy, float *Dyz, float *Dxx, float *Dxy, float *Dxz);
/* Feed externally produced J8/V (host, N each) instead of pnr_frangi's: lets extractSeeds be
 * tested in isolation exactly like SeedExtractor::extractSeeds(tolerance,J8,...,Vx,Vy,Vz). */
int pnr_set_j8_v(pnr_ctx *ctx, const uint8_t *J8, const uint8_t *Vx, const uint8_t *Vy, const uint8_t *Vz);

/* Tracker tables for parity tests: name in {"p","u","w0","w0_cws","v","w","w_cws","rng",
 * "model_vuw<s>","model_wgt<s>","model_avg","gauss_xy<s>","gauss_z<s>"}.  Copies up to cap
 * 4-byte words into out; *n receives the element count. */
int pnr_get_table(pnr_ctx *ctx, const char *name, void *out, int64_t cap, int64_t *n);

/* expf used for the particle likelihood exp(Kc*corr) (tracker.cpp:1029,1136), exposed so the
 * tests can compare the device implementation with the host libm over many inputs. */
int pnr_expf_batch(pnr_ctx *ctx, const float *x, int64_t n, float *y);

/* Frangi::eigen_decomposition (frangi.cpp:1269-1306: tred2 :1309, tql2 :1390, the |lambda| re-sort :1286-1304) of n symmetric
 * 3 x 3 matrices through the DEVICE solver (frangi.hip eigen3), host arrays of row-major doubles: A [n][3][3] in, d [n][3] the
 * eigenvalues by ascending |lambda|, V [n][3][3] the eigenvectors in columns (column 0 = the axis direction, sign as the solver
 * leaves it).  V == NULL runs the eigenvalues-only form that the vesselness kernel (eigen_queue) compiles; V != NULL the full solver
 * of the direction kernel (vdir_points). */
int pnr_eigen_batch(pnr_ctx *ctx, const double *A, int64_t n, double *V, double *d);

/* The scheduler behind pnr_trace_replay[_sharded] (stream_sched.h) over a HOST engine that plays back map-free traces which
 * `trace(user, pos_dir[6], &T, xc[ni])` supplies (0 = ok; rows 0..min(T, ni)-1 of xc valid) -- pure host code, no GPU: the
 * multi-process tests drive the window / admission / exchange / replay logic with it, and a recorded workload can be
 * re-scheduled offline.  Same outputs as pnr_trace_replay_sharded.  look0 / look_pct: the admission lookahead (options of the same
 * name; 0 / -1 = automatic). */
typedef int (*pnr_trace_fn)(void *user, const float *pos_dir, int32_t *T, pnr_xest *xc);
int pnr_sched_playback(const pnr_params *p, int64_t w, int64_t h, int64_t l, const pnr_seed *seeds, int64_t n, int rank,
                       int world, pnr_allgather_fn exchange, void *exchange_user, int64_t block_bytes, pnr_trace_fn trace,
                       void *trace_user, int window, int groups, int poll, int look0, int look_pct, pnr_node *nodes, int64_t cap_nodes,
                       int64_t *n_nodes, int32_t *links, int64_t cap_links, int64_t *n_links, int64_t *n_traces_used,
                       int64_t *n_iterations_here);

/* The same with the scheduler's tentative replay switched on or off (pnr_sched_playback: on, as in pnr_trace_replay[_sharded]; option
 * "tentative"), the number of running traces the admission keeps up (option "target"; -1 = automatic, 0 = off) and the steps of a poll
 * that run on while the host works (option "lag"; -1 = automatic): results are identical either way, only the number of iterations run differs. */
int pnr_sched_playback2(const pnr_params *p, int64_t w, int64_t h, int64_t l, const pnr_seed *seeds, int64_t n, int rank,
                        int world, pnr_allgather_fn exchange, void *exchange_user, int64_t block_bytes, pnr_trace_fn trace,
                        void *trace_user, int window, int groups, int poll, int look0, int look_pct, int tentative, int target, int lag,
                        pnr_node *nodes, int64_t cap_nodes, int64_t *n_nodes, int32_t *links, int64_t cap_links, int64_t *n_links,
                        int64_t *n_traces_used, int64_t *n_iterations_here);

#ifdef __cplusplus
}
#endif
#endif /* PNR_HIP_TEST_H */
