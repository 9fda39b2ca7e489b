// advantra_host.h -- C++ host side above the C ABI, mirroring the reference's plugin entry for the
// accelerated path: Advantra::dofunc("advantra_func", ...) -> reconstruction_func
// (/root/reference/pnr-vaa3d/Advantra_plugin.cpp:274-337, 2183-2731).  Same contract: input[0] = list of
// file names, input[1] = the 11 positional parameters as strings; returns false on a usage error
// (missing image / wrong parameter count, after printing the help), true otherwise (range errors print
// the reference's v3d_msg text and stop, as `return 0` does there).
// The Vaa3D/Qt shell itself (V3DPluginInterface2_1, QObject) cannot be built without the Vaa3D SDK;
// this is the same logic behind a plain C++ signature, driven by advantra_cli.
#pragma once
#include "../../include/pnr_hip.h"
#include <string>
#include <vector>

namespace advantra {

struct Stack {
    std::vector<unsigned char> data; // x fastest: i = z*w*h + y*w + x (TIFF: the pages copied out of the file)
    long long w = 0, h = 0, l = 0;
    // a raw stack is mapped, not copied: 1 GiB through a zero-filled vector cost a quarter of a second that the upload pays anyway
    const unsigned char *view = nullptr;
    size_t map_len = 0;
    const unsigned char *bytes() const { return view ? view : data.data(); }
    Stack() = default;
    Stack(const Stack &) = delete;
    Stack &operator=(const Stack &) = delete;
    ~Stack();
};

struct Result {
    float Jmin = 0, Jmax = 0;
    long long n_seeds_init = 0, n_seeds = 0, n_traces = 0, n_iterations = 0;
    std::vector<pnr_node> nodes;   // nodes[0] = dummy
    std::vector<int32_t> links;    // pairs
    std::vector<pnr_node> tree;    // reconstruct() output: tree list, tree[0] = dummy
    std::vector<int32_t> parent;   // parent index per tree node, -1 = root
    std::string swc_path;
    double t_frangi = 0, t_seeds = 0, t_select = 0, t_trace = 0, t_recon = 0;
    double t_load = 0, t_setup = 0, t_write = 0, t_total = 0; // stack file -> memory; context + upload (+ soma); SWC file; advantra_func as a whole
};

// switches of the head-less driver (advantra_cli flags; the plugin's compile-time TRACING_VERBOSE / saveMidres taps)
struct Settings {
    bool verbose = false;     // -v: per-trace progress and stop reasons in the reference's words (tracker.cpp:866,879,908,916; Advantra_plugin.cpp:2677)
    bool timing = false;      // --timing: the library's stage statistics on stderr (options trace_timing, seed_timing, recon_timing)
    bool save_midres = false; // --save-midres: also write the plugin's saveMidres node lists (:2098-2141): <inimg>_n0_.swc (the node graph before
                              // reconstruct()), _n0res_, _n1_, _n2_, _n2tree_
    bool single_tree = false; // --single-tree: the plugin's ENFORCE_SINGLE_TREE branch (:81, :2142-2152): only the largest tree, written to <inimg>_Advantra1.swc
    uint32_t rng_seed = 42;   // --rng-seed: replaces srand(time(NULL)) of tracker.cpp:1003,1098
    // --ranks N: this process is rank `rank` of `world` processes of one host, one GPU each, that reconstruct ONE stack together
    // (z-slabs of Frangi / seeds, sorted seeds dealt round-robin, finished traces exchanged through shared memory: INTEGRATION.md);
    // rank 0 writes the SWC.  The reference has no counterpart.
    int rank = 0, world = 1;
    pnr_shm_exchange *exchange = nullptr; // the ranks of one host: bootstrap and default transport (shared memory)
    // --exchange rccl: the collectives of the sharded path -- ncclAllReduce of (Jmin, Jmax), ncclAllGather of the scored seeds and of the
    // finished trace records at every poll -- over RCCL / xGMI instead (one rank per GPU; the shared-memory segment only carries the
    // 128-byte ncclUniqueId).  force_shard: the sharded code path also for a world of one (the only RCCL world a one-GPU box can form).
    pnr_rccl_exchange *rccl = nullptr;
    bool force_shard = false;
};
Settings &settings();

void print_help();
// simple_loadimage_wrapper's role (Advantra_plugin.cpp:2241): 8-bit multi-page uncompressed TIFF, or
// ".raw" (u8, dims from `raw_dims` = "w,h,l").  Returns false with a message in `err`.
bool load_stack(const std::string &path, const std::string &raw_dims, Stack &out, std::string &err);
// save_nodelist (Advantra_plugin.cpp:480-523)
bool save_nodelist(const std::vector<pnr_node> &nodes, const std::vector<int32_t> &links, const std::string &swcname,
                   int type = -1, float sig2r = 1.f, const std::string &name = "", const std::string &comment = "");
// the same writer for a tree list (each node has 0 or 1 link: its parent)
bool save_treelist(const std::vector<pnr_node> &tree, const std::vector<int32_t> &parent, const std::string &swcname, int type = -1,
                   float sig2r = 1.f, const std::string &name = "", const std::string &comment = "");
// 0 = ok, -1 = usage error (dofunc returns false), -2 = range error (dofunc "return 0"), -3 = runtime failure
int parse_params(const std::vector<std::string> &paras, pnr_params &p, std::string &err);
// reconstruction_func (Advantra_plugin.cpp:2183-2731) from the point where the stack is in memory (:2255): the caller keeps
// ownership of `data1d` (u8, x fastest).  Writes <inimg_file>_Advantra.swc; false = a library call failed (message printed).
bool reconstruction_func(const unsigned char *data1d, long long w, long long h, long long l, const std::string &inimg_file,
                         const std::vector<std::string> &paras, pnr_params p, int device = 0, Result *result = nullptr);
bool advantra_func(const std::vector<char *> &infiles, const std::vector<char *> &paras, int device = 0,
                   const std::string &raw_dims = "", Result *result = nullptr);

} // namespace advantra
