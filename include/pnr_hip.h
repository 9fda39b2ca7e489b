/*
 * pnr_hip.h -- C ABI of the MI355X (gfx950) implementation of the PNR / Advantra hot path:
 * multi-scale Frangi vesselness -> J8 -> per-layer seed extraction -> ZNCC seed scoring ->
 * batched SMC particle tracing -> host replay of the trace bookkeeping.
 *
 * This is the drop-in boundary: plain C types, no exceptions, no torch types.  Every entry
 * point names the reference interface it replaces (file:line relative to
 * /root/reference/pnr-vaa3d/).  The reference has no FFI of its own (it is one C++ plugin);
 * the call sites a maintainer re-points are Advantra_plugin.cpp:2488-2497 (Frangi),
 * :2499-2512 (J8), :2549 (extractSeeds), :2561-2586 (seed filter/sort), :2658-2710 (trace loop).
 * INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every function returns 0 on success, a negative PNR_E_* code on failure;
 *     pnr_last_error() returns a thread-local message for the last failure.
 *   - volumes are uint8, x fastest: i = z*w*h + y*w + x (frangi.cpp:307).
 *   - one pnr_ctx = one GPU = one host thread at a time (the reference is single-threaded).
 *   - the library FAILS (PNR_E_NODEVICE) when no HIP device is present: there is no CPU path.
 */
#ifndef PNR_HIP_H
#define PNR_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PNR_MAX_SIGMAS 8

enum {
    PNR_OK = 0,
    PNR_E_ARG = -1,      /* invalid argument / parameter out of range */
    PNR_E_NODEVICE = -2, /* no HIP device / device init failed */
    PNR_E_HIP = -3,      /* HIP runtime error (message has the call) */
    PNR_E_STATE = -4,    /* call order violated (e.g. seeds before frangi) */
    PNR_E_NOMEM = -5
};

/* input_PARA (Advantra_plugin.cpp:88-103) + the hard-wired constants of :63-83 */
typedef struct pnr_params {
    float sig[PNR_MAX_SIGMAS]; /* neuritesigmas, ascending (parse_csv_string :1885-1897) */
    int nsig;
    int somaradius;  /* > 0: soma path, call pnr_soma after pnr_set_volume (Advantra_plugin.cpp:2426-2448) */
    float tolerance; /* MaximumFinder tolerance on J8 */
    float znccth;
    float kappa;
    int step;
    int ni; /* SMC iterations per trace direction */
    int np; /* particles */
    float zdist;
    int nodepervol;
    int vol; /* 1,5,9,11,19,27 */
    float Kc;         /* 20   (:63) */
    float neff_ratio; /* 0.8  (:64) */
    float alpha;      /* 0.5  frangi_alfa (:66) */
    float beta;       /* 0.5  frangi_beta (:67) */
    float C;          /* 500  frangi_C    (:68) */
    uint32_t rng_seed;    /* replaces srand(time(NULL)) of tracker.cpp:1003,1098 */
    int max_trace_count;  /* 5000 MAX_TRACE_COUNT (:72) */
} pnr_params;

/* struct seed (seed.h:33-39) */
typedef struct pnr_seed {
    float x, y, z, vx, vy, vz, score, corr;
} pnr_seed;

/* struct X_est (tracker.h:19-23) */
typedef struct pnr_xest {
    float x, y, z, vx, vy, vz, sig, corr;
} pnr_xest;

/* class Node without its nbr vector (node.h:5-44); links are returned separately */
typedef struct pnr_node {
    float x, y, z, vx, vy, vz, corr, sig;
    int32_t type; /* node.cpp:14-21: AXON=2, END=6, UNDEFINED=7 */
} pnr_node;

typedef struct pnr_ctx pnr_ctx;

const char *pnr_last_error(void);
void pnr_default_params(pnr_params *p); /* README.md:17 example + plugin constants */

/* Validates like Advantra::dofunc (Advantra_plugin.cpp:317-326) and builds the Tracker tables
 * (Tracker::Tracker, tracker.cpp:79-527) on the host, uploads them.  device = HIP ordinal. */
int pnr_create(const pnr_params *p, int device, pnr_ctx **out);
void pnr_destroy(pnr_ctx *ctx);

/* Use an externally owned HIP stream (e.g. torch's current stream); NULL = ctx's own stream. */
int pnr_set_stream(pnr_ctx *ctx, void *hip_stream);
int pnr_synchronize(pnr_ctx *ctx);

/* data1d + in_sz of reconstruction_func (Advantra_plugin.cpp:2241-2255).  Host pointer is
 * borrowed for the call and copied to HBM; the _device variant borrows a device pointer that
 * must stay valid until the next set_volume/destroy (no copy).  l == 1 (a single slice) selects the reference's 2-D mode:
 * Frangi::frangi2d (frangi.cpp:392) and the is2d branches of the tracker (Advantra_plugin.cpp:2496-2497, :2526); the tracker
 * tables are rebuilt whenever the dimensionality changes.  2-D stacks are traced by the phased SMC driver only. */
int pnr_set_volume(pnr_ctx *ctx, const uint8_t *img, int64_t w, int64_t h, int64_t l);
int pnr_set_volume_device(pnr_ctx *ctx, const void *dev_img, int64_t w, int64_t h, int64_t l);

/* Frangi::frangi3d (frangi.cpp:152-289; called at Advantra_plugin.cpp:2496) followed by the
 * J -> J8 rule (:2499-2512).  Results stay in HBM; Jmin/Jmax are returned. */
int pnr_frangi(pnr_ctx *ctx, float *Jmin, float *Jmax);
/* z-slab sharding of Frangi / seed extraction over several GPUs: the context's volume is a slab of the stack WITH a
 * halo of ceil(3*sigma_max/zdist) + 2 planes on every cut side (the z pass of the Gaussian plus the radius-2 Hessian stencil), so
 * that the planes [z_keep0, z_keep1) of the slab are exactly what the whole stack would give.  pnr_frangi_slab runs every scale and
 * returns Jmin / Jmax over the kept planes only; after the ranks have reduced them to the global extremes, pnr_quantise_j8 applies
 * the J -> J8 rule (Advantra_plugin.cpp:2499-2512) and pnr_extract_seeds_range(z_keep0, z_keep1) gives the slab's seeds. */
int pnr_frangi_slab(pnr_ctx *ctx, int64_t z_keep0, int64_t z_keep1, float *Jmin, float *Jmax);
int pnr_quantise_j8(pnr_ctx *ctx, float Jmin, float Jmax);

/* Optional read-back of the Frangi outputs (any pointer may be NULL); N = w*h*l each.  J8 is always what pnr_frangi left in HBM.
 * With option frangi_prune (default 1) pnr_frangi skips the eigen-solver where the response provably cannot reach the first
 * non-zero J8 level: J8, Jmin / Jmax and everything at voxels with J8 > 0 (all seeds) are exact, but the f32 J and the winning
 * scale of J8 = 0 voxels are not -- so asking for J or for the direction volumes first recomputes the response without that
 * shortcut (one more pnr_frangi worth of GPU time; tests and diagnostics), and so does pnr_quantise_j8 when it is given extremes
 * the shortcut did not assume (Jmin other than 0, or a Jmax below the run's own maximum; the global extremes of a sharded stack
 * never are). */
int pnr_get_frangi(pnr_ctx *ctx, float *J, uint8_t *J8, uint8_t *Vx, uint8_t *Vy, uint8_t *Vz);
/* SeedExtractor::extractSeeds (seed.cpp:556-791; Advantra_plugin.cpp:2549).  Library-owned
 * array, valid until the next call or pnr_destroy.  z-major, value-descending-per-layer order. */
int pnr_extract_seeds(pnr_ctx *ctx, const pnr_seed **seeds, int64_t *n);
/* Same, restricted to layers [z0, z1): the unit of multi-GPU Frangi/seed sharding. */
int pnr_extract_seeds_range(pnr_ctx *ctx, int64_t z0, int64_t z1, const pnr_seed **seeds, int64_t *n);

/* Tracker::znccBBB (tracker.cpp:1891-1964) for n (pos,dir) pairs: pos_dir = n x 6 floats. */
int pnr_zncc_batch(pnr_ctx *ctx, const float *pos_dir, int64_t n, float *corr, float *sig);

/* Seed filter + sort (Advantra_plugin.cpp:2561-2586): corr = znccBBB(seed); drop corr < znccth;
 * sort by corr descending (ties: original order).  In place; *n_out <= n. */
int pnr_score_filter_sort_seeds(pnr_ctx *ctx, pnr_seed *seeds, int64_t n, int64_t *n_out);
/* The same in two halves, for one stack on several GPUs: every rank scores and filters the seeds of its own z-slab (order kept),
 * the merged list -- in the z-major order of the unsharded extraction -- is then sorted on every rank: the result is the list
 * pnr_score_filter_sort_seeds gives on one GPU (the sort is stable, ties keep the z-major order). */
int pnr_score_filter_seeds(pnr_ctx *ctx, pnr_seed *seeds, int64_t n, int64_t *n_out);
int pnr_sort_seeds(pnr_ctx *ctx, pnr_seed *seeds, int64_t n, int64_t *n_out);

/* Map-independent part of Tracker::trackPos / trackNeg (tracker.cpp:819-933 -> iter0New :1001,
 * iterINew :1096) for n seeds x 2 directions, all on the GPU.  Trace j = 2*i + dir (dir 1 =
 * negated seed direction).  Outputs (host, caller-allocated):
 *   T    [2n]        successful iterations (= ti_limit of a run without density/soma maps)
 *   stop [2n]        0 = ni reached, 1 = left the volume, 2 = centroid corr < znccth
 *   xc   [2n*ni]     centroid estimates; rows 0..min(T,ni-1) are valid
 * Optional debug taps for the first dbg_iters iterations of every trace (NULL to skip):
 *   xfilt [2n*dbg_iters*np*9]  particles (x,y,z,vx,vy,vz,w,corr,sig) (struct X, tracker.h:13-17)
 *   idxres[2n*dbg_iters*np]    resampled indices, neff [2n*dbg_iters] */
int pnr_trace_batch(pnr_ctx *ctx, const pnr_seed *seeds, int64_t n, int32_t *T, int32_t *stop,
                    pnr_xest *xc, int dbg_iters, float *xfilt, int32_t *idxres, float *neff);

/* Host replay of the sequential bookkeeping (trackPos :848-931 + trace loop
 * Advantra_plugin.cpp:2658-2710) over map-free traces, in seed order.  Pure host integer work.
 *   nodes: capacity cap_nodes (node 0 = dummy, :2416-2419); links: pairs (a,b) meaning
 *   a.nbr.push_back(b); b.nbr.push_back(a) in push order.  Returns counts through pointers. */
int pnr_replay_traces(const pnr_params *p, int64_t w, int64_t h, int64_t l, const pnr_seed *seeds,
                      int64_t n, const int32_t *T, const pnr_xest *xc, pnr_node *nodes,
                      int64_t cap_nodes, int64_t *n_nodes, int32_t *links, int64_t cap_links,
                      int64_t *n_links, int64_t *n_traces_used);

/* Soma path (params.somaradius > 0; Advantra_plugin.cpp:2426-2448 and soma_extraction1 :1899-1915): xy erosion
 * (Frangi::imerode, frangi.cpp:880), xy Gaussian of the u8 stack (Frangi::imgaussian, frangi.cpp:786), maxentropy_th
 * (toolbox.cpp:657), binarise at > threshold, conn3d (toolbox.cpp:245) -> one SOMA node (x, y, z, sig = mean radius,
 * corr = -FLT_MAX, type 1) per 26-connected region and the map voxel -> node index the seed filter, the replay and the
 * trace kernels' early stop use.  Must run after pnr_set_volume and before pnr_frangi (it uses the Frangi scratch) whenever
 * somaradius > 0; with somaradius = 0 it records "no soma".  E8 (nullable, N bytes) receives the eroded + blurred stack. */
int pnr_soma(pnr_ctx *ctx, uint8_t *E8, int32_t *threshold, int64_t *n_soma);
/* soma nodes (in node-list order, index k+1) and the sparse label map: foreground voxels in raster order with their node index */
int pnr_get_soma(pnr_ctx *ctx, pnr_node *nodes, int64_t cap_nodes, int64_t *n_nodes, int64_t *vox, int32_t *label,
                 int64_t cap_vox, int64_t *n_vox);
/* pnr_replay_traces with the context's dimensions, parameters and soma (the node list then starts dummy, somas, ...) */
int pnr_replay_traces_ctx(pnr_ctx *ctx, const pnr_seed *seeds, int64_t n, const int32_t *T, const pnr_xest *xc,
                          pnr_node *nodes, int64_t cap_nodes, int64_t *n_nodes, int32_t *links, int64_t cap_links,
                          int64_t *n_links, int64_t *n_traces_used);

/* Production form of the trace loop (Advantra_plugin.cpp:2658-2710): trace + replay with early DENSITY stops.  The
 * node-density map produced by the replay of lower-ranked seeds is kept on the GPU, which ends a trace at the first
 * iteration whose centroid voxel is already saturated there (what the reference's DENSITY stop, tracker.cpp:855, would
 * do at the latest), and seeds on saturated voxels are not launched (:2669-2670).  The GPU map only holds replayed
 * nodes, so it only under-counts and the node graph is identical to pnr_trace_batch + pnr_replay_traces.
 * Phased driver (default): a window of traces is kept full, finished traces are replayed in seed order and their slots
 * handed to the next seeds (first_batch is ignored).  Persistent driver: seed-rank batches of first_batch seeds,
 * doubling up to 1024 (<= 0: default 128).  *n_iterations = SMC iterations run. */
int pnr_trace_replay(pnr_ctx *ctx, const pnr_seed *seeds, int64_t n, int64_t first_batch, pnr_node *nodes,
                     int64_t cap_nodes, int64_t *n_nodes, int32_t *links, int64_t cap_links, int64_t *n_links,
                     int64_t *n_traces_used, int64_t *n_iterations);

/* The node graph of the last pnr_trace_replay / pnr_trace_replay_sharded stays in the context: a caller whose buffers were too
 * small (n_nodes > cap_nodes) allocates and fetches it here instead of tracing again. */
int pnr_get_graph(pnr_ctx *ctx, pnr_node *nodes, int64_t cap_nodes, int64_t *n_nodes, int32_t *links, int64_t cap_links,
                  int64_t *n_links);

/* How every replayed trace of the last pnr_trace_replay[_sharded] ended -- what the reference prints per trace at
 * tracker.cpp:866,879,908,916 -- when the option "trace_log" is set: 5 ints per trace, in replay order:
 * {seed rank, direction (0 = trackPos, 1 = trackNeg), ti_limit, reason, value}; reason 0 = TRACK LIMIT (value: bits of the last
 * corr), 1 = success=0 (value: bits of the failing iteration's corr), 2 = DENSITY (value: nodepervol), 3 = SOMA (value: node). */
int pnr_get_trace_log(pnr_ctx *ctx, int32_t *rec, int64_t cap, int64_t *n);

/* ---- one stack, several GPUs: the sorted seeds sharded over `world` processes (one pnr_ctx per GPU; BASELINE configs[3]) ----
 * The reference has no distributed code; what is kept is the result of its sequential trace loop
 * (Advantra_plugin.cpp:2658-2710).  Rank r traces the seeds r, r + world, ... of the SAME sorted list in its own window of trace
 * slots; after every poll the ranks all-gather the records of the traces that finished (one fixed-size block per rank) and every
 * rank replays them in global seed order, so that every GPU's density map holds the replayed nodes of all ranks: the early
 * DENSITY stops (tracker.cpp:855) and the seed skip rule (:2669-2670) work as on one GPU and every rank returns the same node
 * graph -- the graph of pnr_trace_replay on one GPU.
 *
 * The transport is the host's: `exchange(user, send, recv, bytes)` must behave like an all-gather of `bytes` bytes per rank
 * (recv = world x bytes, rank order) over whatever joins the processes -- RCCL / torch.distributed (pnr_amd/multigpu.py), MPI, a
 * thread barrier in tests -- and return 0.  It is called once per poll by every rank, the same number of times on all of them.
 * A rank that fails says so in one last exchange, so the others return PNR_E_STATE instead of waiting for it. */
typedef int (*pnr_allgather_fn)(void *user, const void *send, void *recv, int64_t bytes_per_rank);
int pnr_trace_replay_sharded(pnr_ctx *ctx, const pnr_seed *seeds, int64_t n, int rank, int world, pnr_allgather_fn exchange,
                             void *user, pnr_node *nodes, int64_t cap_nodes, int64_t *n_nodes, int32_t *links,
                             int64_t cap_links, int64_t *n_links, int64_t *n_traces_used, int64_t *n_iterations_here);

/* A ready-made exchange for ranks that share ONE host (the 8 GPUs of a node): an all-gather through POSIX shared memory.  The
 * records being exchanged live in pinned host memory and are consumed by the host replay, so a host-side transport saves the
 * host -> device -> xGMI -> device -> host round trip of a device collective (a few microseconds per exchange).  `name` must be the
 * same on all ranks and unique to the job; rank 0 creates the segment, the call returns when all `world` ranks are attached.
 * Pass pnr_shm_allgather as `exchange` and the handle as `user`.  capacity_bytes >= the largest block ever exchanged. */
typedef struct pnr_shm_exchange pnr_shm_exchange;
int pnr_shm_exchange_open(const char *name, int rank, int world, int64_t capacity_bytes, pnr_shm_exchange **out);
int pnr_shm_allgather(void *user, const void *send, void *recv, int64_t bytes_per_rank);
void pnr_shm_exchange_close(pnr_shm_exchange *x);

/* The same collectives over RCCL (xGMI between the GPUs of a node, the network across nodes) for hosts that are not Python
 * (advantra_cli --ranks N --exchange rccl; pnr_amd/multigpu.py reaches RCCL through torch.distributed instead): an ncclAllGather of one
 * fixed-size block per rank -- pass pnr_rccl_allgather as `exchange` and the handle as `user` -- and the 2-float all-reduce of
 * (Jmin, Jmax) of the z-slab Frangi (one ncclAllReduce(ncclMax) over (-min, max)).  The payloads are host data: every call stages
 * through pinned host memory and a device buffer on the exchange's own stream.  One rank calls pnr_rccl_unique_id and hands the 128
 * bytes to the others by any means (advantra_cli: the shared-memory segment; across nodes: the launcher); every rank then opens the
 * exchange with ITS device -- a collective call, one rank per GPU (RCCL refuses two ranks on one device).  librccl is opened at run
 * time, when the first of these calls is made; without it they fail with PNR_E_STATE.  capacity_bytes >= the largest block. */
typedef struct pnr_rccl_exchange pnr_rccl_exchange;
int pnr_rccl_unique_id(void *id128);
int pnr_rccl_exchange_open(const void *id128, int rank, int world, int device, int64_t capacity_bytes, pnr_rccl_exchange **out);
int pnr_rccl_allgather(void *user, const void *send, void *recv, int64_t bytes_per_rank);
int pnr_rccl_allreduce_minmax(pnr_rccl_exchange *x, float *min_inout, float *max_inout);
void pnr_rccl_exchange_close(pnr_rccl_exchange *x);

/* reconstruct() chain of the plugin (Advantra_plugin.cpp:2096-2181), pure host: link resampling
 * (TRACE_RSMPL) -> mean-shift refinement (SIG2RADIUS, REFINE_ITER, EPSILON2) -> sphere grouping (GROUP_RADIUS) ->
 * BFS trees -> drop trees < TREE_SIZE_MIN -> tree resampling.  Input: the node graph of pnr_trace_replay /
 * pnr_replay_traces.  Output: the tree list save_nodelist writes (node 0 dummy; parent -1 = root).  Values <= 0
 * select the plugin constants (1.0, 1.5, 4, 1e-4, 2.0, 10) -- except tree_size_min < 0, which selects the plugin's ENFORCE_SINGLE_TREE
 * branch (:81, :2142-2152): only the largest tree is kept (extract_largest_tree :546-589; the plugin names that file _Advantra1.swc). */
int pnr_reconstruct(const pnr_node *nodes, int64_t n_nodes, const int32_t *links, int64_t n_links, float trace_rsmpl,
                    float sig2radius, int refine_iter, float epsilon2, float group_radius, int tree_size_min,
                    pnr_node *out_nodes, int32_t *out_parent, int64_t cap, int64_t *n_out);

/* The plugin's saveMidres taps inside reconstruct() (:2098-2141): the node list as it stands behind stage 1 = interpolate_nodelist
 * (_n0res_), 2 = non_blurring (_n1_), 3 = group1 (_n2_), 4 = compute_trees (_n2tree_).  out_links: pairs, every undirected link once
 * (stage 4: (child, parent)); counts are returned even when the buffers are too small. */
int pnr_reconstruct_stage(const pnr_node *nodes, int64_t n_nodes, const int32_t *links, int64_t n_links, float trace_rsmpl,
                          float sig2radius, int refine_iter, float epsilon2, float group_radius, int stage, pnr_node *out_nodes,
                          int64_t cap_nodes, int64_t *n_out_nodes, int32_t *out_links, int64_t cap_links, int64_t *n_out_links);

/* How pnr_trace_batch / pnr_trace_replay schedule the particle filter on the GPU (results are bit-identical):
 * 0 = one launch per SMC phase over all active traces of a batch (default), 1 = one persistent work-group per trace. */
int pnr_set_smc_driver(pnr_ctx *ctx, int driver);

/* Scheduling and host-side knobs of a context; none of them changes a result (the library reads no environment variable).
 *   window (0 = automatic: 1536 on one GPU, 768 sharded or without the tentative replay) trace slots kept busy | look0, look_pct (0 / -1 = automatic) admission lookahead max(look0, frontier*look_pct/100)
 *   target (-1 = automatic: 120 on one GPU, 96 per rank sharded, 0 = off without the tentative replay) seeds are admitted only while fewer traces than this are running |
 *   lag (-1 = automatic: half a poll when no other trace group covers the host's share of a poll, else one step) steps of a poll that run on while the host works on the state in front of them |
 *   overfill (1) the target is the mean over a poll | concentrate (1) with several trace groups new seeds go to one group while few traces survive a poll |
 *   sums_deep (-1 = automatic: launches of at most sums_deep_max (96) traces, or one trace group; 0 / 1) form of the ordered sums (four chunk buffers in turn) |
 *   profile_every (1) with pnr_set_profiling: the streaming tracer times every n-th poll of a trace group and counts it n-fold | poll (0 = automatic: 2 on one or two GPUs, 4 from four ranks on) SMC steps between polls | groups (0 = automatic: 2 on one GPU, 1 sharded; 1..4) trace groups on separate streams | split_x10 (0 = automatic), max_split (96) sampling
 *   work-groups per CU x 10 / per trace | stash_mb (65536) sample-stash budget | host_threads (0 = CPUs of this process /
 *   local_ranks) workers of the seed flood fill and of pnr_reconstruct_ctx | local_ranks (1) processes sharing this host |
 *   trace_timing, seed_timing (0/1) statistics on stderr | recon_timing (0/1) stage times of every pnr_reconstruct on stderr (process-wide: that call takes no context) | trace_log (0/1) keep every trace's end for pnr_get_trace_log | replay_batches (0/1), batch_growth, batch_max: rank batches instead
 *   of the streaming window | no_stash (0/1) persistent driver without the sample stash | exchange_block (0 = 256 KB / world) bytes per rank
 *   and exchange of pnr_trace_replay_sharded | frangi_prune (1) skip the eigen-solver below the first J8 level (pnr_get_frangi) |
 *   cube_copy (1) phased driver: a trace's cube is fetched from the image once per step and copied by its sampling work-groups (0: each stages it itself) |
 *   gauss_march (1) the fused x-y Gaussian marches down strips of a slice (0: one 64 x 64 tile per work-group; the same bits) |
 *   tentative (1) the streaming scheduler pauses traces that a tentative replay of everything recorded so far cuts, and ends them
 *   itself once that verdict is final (fewer wasted SMC iterations; same graph).
 *   pnr_get_option also knows "host_threads_effective" and "frangi_recomputes" (how often pnr_get_frangi / pnr_quantise_j8 had to
 *   re-run Frangi without the frangi_prune shortcut -- one pnr_frangi worth of GPU time each; also printed with trace_timing /
 *   seed_timing).  The kernel timers (pnr_get_kernel_ms) include those re-runs. */
int pnr_set_option(pnr_ctx *ctx, const char *key, int64_t value);
int pnr_get_option(pnr_ctx *ctx, const char *key, int64_t *value);

/* Per-kernel-group device time (HIP events on the ctx stream) accumulated since the last reset:
 * groups: "gauss","hessian_eigen","j8","seed_maxima","soma","zncc","smc" (sampling kernel; the whole trace kernel of the persistent driver),"smc_sums","smc_predict","smc_update","smc_cube" (the traces' cubes fetched once per step).  Enabled by set_profiling. */
int pnr_set_profiling(pnr_ctx *ctx, int enable);
int pnr_get_kernel_ms(pnr_ctx *ctx, const char *group, double *ms, int64_t *launches);
int pnr_reset_kernel_ms(pnr_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* PNR_HIP_H */
