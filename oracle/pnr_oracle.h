/*
 * pnr_oracle.h -- CPU restatement (plain C99) of the PNR/Advantra hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker / timed CPU baseline.  The product
 * (pnr_amd/) never links, imports or calls this file.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - Frangi3D / J8 / extractSeeds: pinned against the reference's own
 *     frangi.cpp / seed.cpp compiled into oracle/_ref (tests/golden/ fixtures and
 *     live comparison when oracle/_ref/libpnr_ref.so is present).
 *   - Tracker tables / znccBBB / iter0New / iterINew / trackPos replay:
 *     PARITY UNPINNED -- tracker.cpp cannot be built here (tracker.h:11 needs
 *     the un-vendored Vaa3D header v3d_interface.h), and the reference ships
 *     no tests or golden vectors.  Restated from the source text, each
 *     function citing the lines it follows.
 *
 * All file:line citations are relative to /root/reference/pnr-vaa3d/.
 */
#ifndef PNR_ORACLE_H
#define PNR_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------- Frangi (frangi.cpp) ---------- */
void orc_imgaussian3d(const uint8_t *I, int w, int h, int l, float sig, float zdist, float *F);
void orc_hessian3d(const float *F, int w, int h, int l, float sig,
                   float *Dzz, float *Dyy, float *Dyz, float *Dxx, float *Dxy, float *Dxz);
void orc_eigen3(const double A[9], double V[9], double d[3]);
void orc_frangi3d(const uint8_t *I, int w, int h, int l, const float *sigs, int nsig, float zdist,
                  float alpha, float beta, float C,
                  float *J, float *Jmin, float *Jmax, uint8_t *Vx, uint8_t *Vy, uint8_t *Vz);
void orc_j8(const float *J, int64_t n, float Jmin, float Jmax, uint8_t *J8);

/* ---------- seeds (seed.cpp) ---------- */
/* seeds_out: cap x 8 floats (x,y,z,vx,vy,vz,score,corr); returns the number of
 * seeds found (may exceed cap; only the first cap are written). */
int64_t orc_extract_seeds(double tolerance, const uint8_t *J8, int w, int h, int l,
                          const uint8_t *Vx, const uint8_t *Vy, const uint8_t *Vz,
                          float *seeds_out, int64_t cap);

/* ---------- tracker (tracker.cpp) ---------- */
typedef struct orc_tracker orc_tracker;

orc_tracker *orc_tracker_new(const float *sigs, int nsig, int step, int npcles, int niter,
                             float kappa, float znccth, float Kc, float neff_ratio,
                             float zdist, int nodespervol, uint32_t rng_seed);
/* the same with is2d = (P == 1): 2-D templates (v, u, 0), in-plane prediction offsets, 30 directions, bilinear interp */
orc_tracker *orc_tracker_new2(const float *sigs, int nsig, int step, int npcles, int niter,
                             float kappa, float znccth, float Kc, float neff_ratio,
                             float zdist, int nodespervol, uint32_t rng_seed, int is2d);
void orc_tracker_free(orc_tracker *t);

/* table access for parity tests (pointers stay owned by the tracker) */
int orc_tracker_sz(const orc_tracker *t);
int orc_tracker_ndir(const orc_tracker *t);
const float *orc_tracker_p(const orc_tracker *t);       /* sz x 3 */
const float *orc_tracker_u(const orc_tracker *t);       /* sz x 3 */
const float *orc_tracker_w0(const orc_tracker *t);      /* sz */
const float *orc_tracker_w0_cws(const orc_tracker *t);  /* sz */
const float *orc_tracker_v(const orc_tracker *t);       /* ndir x 3 */
const float *orc_tracker_w(const orc_tracker *t);       /* ndir x sz */
const float *orc_tracker_w_cws(const orc_tracker *t);   /* ndir x sz */
int orc_tracker_model_count(const orc_tracker *t, int sig_idx);
const float *orc_tracker_model_vuw(const orc_tracker *t, int sig_idx); /* M x 3 (v,u,w) */
const float *orc_tracker_model_wgt(const orc_tracker *t, int sig_idx); /* M */
float orc_tracker_model_avg(const orc_tracker *t, int sig_idx);
const uint32_t *orc_tracker_rng(const orc_tracker *t);  /* npcles + 1 draws of glibc rand() */

/* glibc rand() stream after srand(seed): first n draws */
void orc_glibc_rand(uint32_t seed, int n, uint32_t *out);

float orc_interp(float x, float y, float z, const uint8_t *img, int w, int h, int l);
float orc_zncc(orc_tracker *t, float x, float y, float z, float vx, float vy, float vz,
               const uint8_t *img, int w, int h, int l, float *sig_out);

/* Map-independent part of trackPos: run the particle filter from a seed until
 * iter0New/iterINew returns false or niter is reached.
 *   xc_out     : niter x 8 floats (x,y,z,vx,vy,vz,sig,corr), rows 0..T written
 *                (row T holds the failing estimate when T < niter)
 *   returns T  = number of successful iterations (ti_limit of a map-free run)
 *   stop       : 0 = niter reached, 1 = out of volume, 2 = corr < znccth
 * Optional debug dumps (may be NULL), sized for max_dbg iterations:
 *   xfilt_out  : max_dbg x npcles x 9 (x,y,z,vx,vy,vz,w,corr,sig)
 *   idxres_out : max_dbg x npcles (only meaningful where neff/np < neff_ratio)
 *   neff_out   : max_dbg
 */
int orc_trace(orc_tracker *t, const float seed[6], const uint8_t *img, int w, int h, int l,
              float *xc_out, int *stop, int max_dbg, float *xfilt_out, int *idxres_out,
              float *neff_out);

/* ---------- host bookkeeping (tracker.cpp:825-933, Advantra_plugin.cpp:2602-2710) ---------- */
typedef struct {
    float x, y, z, vx, vy, vz, corr, sig;
    int type;
} orc_node;

/* Replays trackPos/trackNeg bookkeeping over pre-computed map-free traces.
 *   seeds   : nseeds x 8 (sorted); traces for seed i: pos = 2*i, neg = 2*i+1
 *   T, xc   : per trace, T[j] successful iterations, xc + j*niter*8
 *   nodes   : out, capacity cap_nodes (index 0 = dummy node)
 *   links   : out, pairs (a,b) appended in push order, capacity cap_links pairs
 * returns number of nodes (incl. dummy); *nlinks = number of directed nbr pushes / 2 pairs
 */
int64_t orc_replay(const float *seeds, int64_t nseeds, const int *T, const float *xc, int niter,
                   int w, int h, int l, int nodespervol, int vol, int max_trace_count,
                   orc_node *nodes, int64_t cap_nodes, int32_t *links, int64_t cap_links,
                   int64_t *nlinks, int64_t *ntraces_used);

/* the same with a soma map (tracker.cpp:858-869): smap[voxel] > 0 = index of the SOMA node that owns the voxel; the node
 * list starts with the dummy node followed by the n_soma soma nodes (x, y, z, r each; Advantra_plugin.cpp:1911-1914) */
int64_t orc_replay_soma(const float *seeds, int64_t nseeds, const int *T, const float *xc, int niter,
                        int w, int h, int l, int nodespervol, int vol, int max_trace_count, const int32_t *smap,
                        const float *soma4, int64_t n_soma,
                        orc_node *nodes, int64_t cap_nodes, int32_t *links, int64_t cap_links,
                        int64_t *nlinks, int64_t *ntraces_used);

/* ---------- 2-D Frangi for single-slice stacks, pnr_oracle_2d.c ---------- */
void orc_hessian2d(const uint8_t *I, int w, int h, float sig, float *Dyy, float *Dxy, float *Dxx);
void orc_frangi2d(const uint8_t *I, int w, int h, const float *sigs, int nsig, float BetaOne, float BetaTwo,
                  float *J, float *Jmin, float *Jmax, uint8_t *Vx, uint8_t *Vy, uint8_t *Vz);

/* ---------- soma path, pnr_oracle_soma.c ---------- */
void orc_imerode_xy(const uint8_t *I, int w, int h, int l, float rad, uint8_t *E);
void orc_imgaussian_u8_xy(uint8_t *I, int w, int h, int l, float sig);
unsigned char orc_maxentropy_hist(const int64_t *hist256);
unsigned char orc_maxentropy_th(const uint8_t *img, int64_t size);
int64_t orc_conn3d(const uint8_t *inimg, int w, int h, int l, int32_t *lab, int diagonal, int values_over, int min_reg_size,
                   float *xc, float *yc, float *zc, float *rc, int64_t cap);
int64_t orc_soma_extract(const uint8_t *img, int w, int h, int l, int somaradius, uint8_t *E8, int *th_out, int32_t *smap,
                         float *nodes4, int64_t cap);

/* ---------- graph post-processing (Advantra_plugin.cpp:2096-2181), pnr_oracle_recon.c ---------- */
int64_t orc_reconstruct(const orc_node *nodes, int64_t n_nodes, const int32_t *links, int64_t n_links, float trace_rsmpl,
                        float sig2radius, int refine_iter, float epsilon2, float group_radius, int tree_size_min,
                        orc_node *out, int32_t *parent, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif
