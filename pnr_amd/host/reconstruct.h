// reconstruct.h -- host-side graph post-processing of the trace graph n0 into the final tree list, the
// reference's reconstruct() chain (/root/reference/pnr-vaa3d/Advantra_plugin.cpp:2096-2181):
//   interpolate_nodelist (:780-861) -> non_blurring mean-shift (:968-1052) -> group1 (:1566-1642) + check_nbr
//   (:1532-1564) -> compute_trees = bfs2 (:379-478) -> extract_trees (:591-629) -> interpolate_treelist (:714-778)
// SURVEY.md 8(f) rank 1.  The reference scans all node pairs (O(n^2), minutes at 10^5 nodes); here the
// neighbour searches go through a uniform grid but visit candidates in ascending node index, so every f32 sum
// is accumulated in the reference's order and the result is the same as the plain scan.
#pragma once
#include "../../include/pnr_hip.h"
#include <cstdint>
#include <vector>

namespace advantra {

struct ReconParams {          // Advantra_plugin.cpp:72-83
    float trace_rsmpl = 1.0f;  // TRACE_RSMPL
    float sig2radius = 1.5f;   // SIG2RADIUS
    int refine_iter = 4;       // REFINE_ITER
    float epsilon2 = 0.0001f;  // EPSILON2
    float group_radius = 2.0f; // GROUP_RADIUS
    int tree_size_min = 10;    // TREE_SIZE_MIN
    int threads = 0;           // host threads of the mean-shift (its nodes are independent); 0 = one per CPU this process may use
    bool single_tree = false;  // ENFORCE_SINGLE_TREE (:81, :2142-2152): keep the largest tree only (extract_largest_tree :546-589)
};

// the node lists reconstruct() passes through, in the order of the plugin's saveMidres taps (:2098-2141)
enum { RECON_FINAL = 0, RECON_N0RES = 1 /* after interpolate_nodelist */, RECON_N1 = 2 /* after non_blurring */, RECON_N2 = 3 /* after group1 */,
       RECON_N2TREE = 4 /* compute_trees: the BFS forest, (child, parent) links */ };

// nodes[0] is the dummy; links = pairs (a,b): a.nbr.push_back(b); b.nbr.push_back(a).
// out_nodes/out_parent: the tree list (index 0 dummy; parent -1 = root), as save_nodelist would write it.
// stage wall times of every reconstruct() on stderr: a process-wide switch (pnr_reconstruct takes no context), set through
// pnr_set_option(ctx, "recon_timing", 0 / 1) like every other diagnostic -- the library reads no environment variable
void set_recon_timing(bool on);
bool recon_timing();

// stop_after != RECON_FINAL (with stage_links): out_nodes / *stage_links receive the list behind that stage instead (links as pairs:
// every undirected link once; for RECON_N2TREE (child, parent)), out_parent is left empty.
void reconstruct(const std::vector<pnr_node> &nodes, const std::vector<int32_t> &links, const ReconParams &rp,
                 std::vector<pnr_node> &out_nodes, std::vector<int32_t> &out_parent, int stop_after = RECON_FINAL,
                 std::vector<int32_t> *stage_links = nullptr);

} // namespace advantra
