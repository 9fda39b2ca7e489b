// reconstruct.cpp -- see reconstruct.h.  Pure host code (the reference keeps this stage on the host as well).
#include "reconstruct.h"
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <sched.h>
#include <thread>
#include <unordered_map>

namespace advantra {
namespace {

struct N {
    float x, y, z, vx, vy, vz, corr, sig;
    int type;
    std::vector<int> nbr;
};
typedef std::vector<N> List;

inline float length(const N &a, const N &b)
{
    // sqrt(pow(dx,2)+pow(dy,2)+pow(dz,2)) with f32 differences: f64 arithmetic, f32 result
    const double dx = (double)(b.x - a.x), dy = (double)(b.y - a.y), dz = (double)(b.z - a.z);
    return (float)std::sqrt(dx * dx + dy * dy + dz * dz);
}

N between(const N &a, const N &b, int k, int Nseg, float vnorm, float vx, float vy, float vz)
{
    N m;
    m.x = a.x + k * (vnorm / Nseg) * vx;
    m.y = a.y + k * (vnorm / Nseg) * vy;
    m.z = a.z + k * (vnorm / Nseg) * vz;
    m.vx = vx; m.vy = vy; m.vz = vz;
    m.corr = a.corr + (b.corr - a.corr) * (k / (float)Nseg);
    m.sig = a.sig + (b.sig - a.sig) * (k / (float)Nseg);
    m.type = (k <= Nseg / 2) ? a.type : b.type;
    return m;
}

// :780-861 -- every bidirectional link longer than `step` gets evenly spaced nodes
void resample_links(List &g, float step)
{
    const size_t init = g.size();
    std::vector<std::vector<char>> done(init);
    for (size_t i = 0; i < init; i++) done[i].assign(g[i].nbr.size(), 0);
    for (size_t i = 1; i < init; i++)
        for (size_t j = 0; j < g[i].nbr.size(); j++) {
            if (done[i][j]) continue;
            const int i1 = g[i].nbr[j];
            const size_t j1 = std::find(g[i1].nbr.begin(), g[i1].nbr.end(), (int)i) - g[i1].nbr.begin();
            if (j1 >= g[i1].nbr.size()) continue; // no link back: left alone
            done[i][j] = 1;
            done[i1][j1] = 1;
            const float vnorm = length(g[i], g[i1]);
            const float vx = (g[i1].x - g[i].x) / vnorm, vy = (g[i1].y - g[i].y) / vnorm, vz = (g[i1].z - g[i].z) / vnorm;
            const int Nseg = (int)std::ceil(vnorm / step);
            for (int k = 1; k < Nseg; k++) {
                g.push_back(between(g[i], g[i1], k, Nseg, vnorm, vx, vy, vz));
                const int last = (int)g.size() - 1;
                if (k == 1) { g[last].nbr.push_back((int)i); g[i].nbr[j] = last; }
                else { g[last].nbr.push_back(last - 1); g[last - 1].nbr.push_back(last); }
                if (k == Nseg - 1) { g[last].nbr.push_back(i1); g[i1].nbr[j1] = last; }
            }
        }
}

// uniform grid over node positions (indices >= 1).  The nodes are kept cell by cell (ascending index inside a cell) with their
// coordinates packed beside them, so a query streams contiguous memory; cells are found through an open-addressing table.
struct Grid {
    float cell;
    std::vector<int> idx;          // node indices, sorted by (cell, index)
    std::vector<float> px, py, pz; // their coordinates, in the same order
    std::vector<long long> hkey;   // table: cell key (-1 = empty) -> [hbeg, hend) in idx
    std::vector<int> hbeg, hend;
    size_t mask = 0;
    static long long key(int a, int b, int c) { return ((long long)(a + (1 << 20)) << 42) | ((long long)(b + (1 << 20)) << 21) | (long long)(c + (1 << 20)); }
    static size_t mix(long long k) { return (size_t)(((unsigned long long)k * 0x9E3779B97F4A7C15ull) >> 20); }
    Grid(const List &g, float cell_) : cell(cell_)
    {
        const size_t n = g.size() > 0 ? g.size() - 1 : 0;
        std::vector<std::pair<long long, int>> ki(n);
        for (size_t i = 1; i < g.size(); i++) ki[i - 1] = {key(bin(g[i].x), bin(g[i].y), bin(g[i].z)), (int)i};
        std::sort(ki.begin(), ki.end());
        idx.resize(n); px.resize(n); py.resize(n); pz.resize(n);
        size_t ncell = 0;
        for (size_t k = 0; k < n; k++) {
            const N &v = g[ki[k].second];
            idx[k] = ki[k].second; px[k] = v.x; py[k] = v.y; pz[k] = v.z;
            if (k == 0 || ki[k].first != ki[k - 1].first) ncell++;
        }
        size_t cap = 16;
        while (cap < 2 * ncell) cap <<= 1;
        mask = cap - 1;
        hkey.assign(cap, -1); hbeg.assign(cap, 0); hend.assign(cap, 0);
        for (size_t k = 0; k < n;) {
            size_t e = k + 1;
            while (e < n && ki[e].first == ki[k].first) e++;
            size_t h = mix(ki[k].first) & mask;
            while (hkey[h] != -1) h = (h + 1) & mask;
            hkey[h] = ki[k].first; hbeg[h] = (int)k; hend[h] = (int)e;
            k = e;
        }
    }
    int bin(float v) const { return (int)std::floor(v / cell); }
    // calls visit(first, last) -- a range of positions in idx / px / py / pz -- for every non-empty cell that intersects the cube
    // of half-width R around (x,y,z); the caller filters by distance first and orders what is left (the sums of the reference
    // run over ascending node index)
    template <class F>
    void for_cells(float x, float y, float z, float R, F &&visit) const
    {
        const int a0 = bin(x - R), a1 = bin(x + R), b0 = bin(y - R), b1 = bin(y + R), c0 = bin(z - R), c1 = bin(z + R);
        for (int a = a0; a <= a1; a++)
            for (int b = b0; b <= b1; b++)
                for (int c = c0; c <= c1; c++) {
                    const long long k = key(a, b, c);
                    for (size_t h = mix(k) & mask; hkey[h] != -1; h = (h + 1) & mask)
                        if (hkey[h] == k) { visit(hbeg[h], hend[h]); break; }
                }
    }
};

// CPUs this process may run on (affinity mask), at most 32
unsigned usable_cpus()
{
    cpu_set_t set;
    int n = sched_getaffinity(0, sizeof(set), &set) == 0 ? CPU_COUNT(&set) : (int)std::thread::hardware_concurrency();
    return (unsigned)std::min(std::max(n, 1), 32);
}

// :968-1052 -- non-blurring mean-shift of (x,y,z,sig) with a kernel radius SIG2RAD*sig
void mean_shift(List &g, float SIG2RAD, int MAXITER, float EPS2, int threads)
{
    // in place: every trajectory reads the positions as they were on entry (kept packed, in index order), the results are
    // written back once all of them are known
    struct P4 { float x, y, z, s; };
    const size_t n = g.size();
    std::vector<P4> src(n), res(n);
    float smax = 0;
    for (size_t i = 0; i < n; i++) src[i] = P4{g[i].x, g[i].y, g[i].z, g[i].sig};
    for (size_t i = 1; i < n; i++) smax = std::max(smax, src[i].s);
    const Grid grid(g, std::max(4.0f, SIG2RAD * smax));
    // the nodes are independent: host threads take them in blocks from a shared counter (tubes are dense in places)
    std::atomic<size_t> next_block{1};
    constexpr size_t BLOCK = 256;
    auto shift_blocks = [&]() {
        std::vector<int> cand;
        for (;;) {
            const size_t i0 = next_block.fetch_add(BLOCK), i1 = std::min(n, i0 + BLOCK);
            if (i0 >= n) break;
            for (size_t i = i0; i < i1; i++) {
                float conv[4] = {src[i].x, src[i].y, src[i].z, src[i].s}, next[4];
                int iter = 0, cnt;
                float d2;
                do {
                    next[0] = next[1] = next[2] = next[3] = 0;
                    const float r2 = (float)std::pow((double)(SIG2RAD * conv[3]), 2);
                    // the members of the ball, then in ascending index: summed in the order of the reference's full scan
                    cand.clear();
                    grid.for_cells(conv[0], conv[1], conv[2], std::sqrt(r2) * 1.0001f + 1e-3f, [&](int p, int e) {
                        for (; p < e; p++) {
                            const float x2 = (float)std::pow((double)(grid.px[p] - conv[0]), 2);
                            if (!(x2 <= r2)) continue;
                            const float y2 = (float)std::pow((double)(grid.py[p] - conv[1]), 2);
                            if (!(x2 + y2 <= r2)) continue;
                            const float z2 = (float)std::pow((double)(grid.pz[p] - conv[2]), 2);
                            if (x2 + y2 + z2 <= r2) cand.push_back(grid.idx[p]);
                        }
                    });
                    std::sort(cand.begin(), cand.end());
                    for (int j : cand) {
                        next[0] += src[j].x; next[1] += src[j].y; next[2] += src[j].z; next[3] += src[j].s;
                    }
                    cnt = (int)cand.size();
                    next[0] /= cnt; next[1] /= cnt; next[2] /= cnt; next[3] /= cnt;
                    d2 = (float)(std::pow((double)(next[0] - conv[0]), 2) + std::pow((double)(next[1] - conv[1]), 2) + std::pow((double)(next[2] - conv[2]), 2));
                    for (int q = 0; q < 4; q++) conv[q] = next[q];
                    iter++;
                } while (iter < MAXITER && d2 > EPS2);
                res[i] = P4{conv[0], conv[1], conv[2], conv[3]};
            }
        }
    };
    unsigned nt = threads > 0 ? (unsigned)threads : usable_cpus();
    if (n < 4096) nt = 1;
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(shift_blocks);
    shift_blocks();
    for (auto &t : th) t.join();
    for (size_t i = 1; i < n; i++) { g[i].x = res[i].x; g[i].y = res[i].y; g[i].z = res[i].z; g[i].sig = res[i].s; }
}

// :1532-1564 -- unique neighbour lists, no self links, links made bidirectional
void tidy_links(List &g)
{
    for (size_t i = 1; i < g.size(); i++) {
        auto &nb = g[i].nbr;
        std::sort(nb.begin(), nb.end());
        nb.erase(std::unique(nb.begin(), nb.end()), nb.end());
        auto self = std::find(nb.begin(), nb.end(), (int)i);
        if (self != nb.end()) nb.erase(self);
    }
    for (size_t i = 1; i < g.size(); i++)
        for (size_t j = 0; j < g[i].nbr.size(); j++) {
            auto &other = g[g[i].nbr[j]].nbr;
            if (std::find(other.begin(), other.end(), (int)i) == other.end()) other.push_back((int)i);
        }
}

// :1566-1642 -- greedy sphere grouping in order of decreasing corr; a group is the running mean of its members
void group_spheres(List &src, List &dst, float rad)
{
    const size_t n = src.size();
    src[0].corr = FLT_MAX;
    // the reference's std::sort by corr is unstable; equal corr keeps index order here (the keys are sorted beside their
    // indices: a comparator that reads the nodes themselves spends its time on cache misses)
    std::vector<std::pair<float, int>> keyed(n);
    for (size_t i = 0; i < n; i++) keyed[i] = {src[i].corr, (int)i};
    // A NaN corr (tracker.cpp:1079,1184 stop on corr < znccth only, which a NaN passes) would make "a.corr > b.corr" no strict weak
    // order -- undefined behaviour in std::sort, here as in the reference (:1571).  The order is made total: numbers by decreasing
    // corr, NaNs behind all of them, equal keys by index.
    std::sort(keyed.begin(), keyed.end(), [](const std::pair<float, int> &a, const std::pair<float, int> &b) {
        const bool an = a.first != a.first, bn = b.first != b.first;
        if (an != bn) return bn;
        if (!an && a.first != b.first) return a.first > b.first;
        return a.second < b.second;
    });
    std::vector<int> order(n);
    for (size_t i = 0; i < n; i++) order[i] = keyed[i].second;
    std::vector<int> to(n, -1);
    to[0] = 0;
    dst.clear();
    dst.push_back(src[0]);
    for (size_t i = 1; i < n; i++) // soma nodes are groups of their own, ahead of everything else (:1581-1588)
        if (src[i].type == 1) {
            to[i] = (int)dst.size();
            dst.push_back(src[i]);
        }
    const Grid grid(src, std::max(2.0f, rad));
    std::vector<int> cand;
    const float r2 = rad * rad;
    for (size_t oi = 1; oi < n; oi++) {
        const int ci = order[oi];
        if (to[ci] != -1) continue;
        to[ci] = (int)dst.size();
        N grp = src[ci];
        float members = 1;
        cand.clear();
        grid.for_cells(src[ci].x, src[ci].y, src[ci].z, rad * 1.0001f + 1e-3f, [&](int p, int e) {
            for (; p < e; p++) {
                const int j = grid.idx[p];
                if (j == ci || to[j] != -1) continue;
                float d2 = (float)std::pow((double)(grid.px[p] - src[ci].x), 2);
                if (!(d2 <= r2)) continue;
                d2 = (float)(d2 + std::pow((double)(grid.py[p] - src[ci].y), 2));
                if (!(d2 <= r2)) continue;
                d2 = (float)(d2 + std::pow((double)(grid.pz[p] - src[ci].z), 2));
                if (d2 <= r2) cand.push_back(j);
            }
        });
        std::sort(cand.begin(), cand.end());
        for (int j : cand) {
            to[j] = (int)dst.size();
            grp.nbr.insert(grp.nbr.end(), src[j].nbr.begin(), src[j].nbr.end());
            members++;
            const float a = (members - 1) / members;
            const float b = (float)(1.0 / members);
            grp.x = a * grp.x + b * src[j].x;
            grp.y = a * grp.y + b * src[j].y;
            grp.z = a * grp.z + b * src[j].z;
            grp.sig = a * grp.sig + b * src[j].sig;
            grp.corr = a * grp.corr + b * src[j].corr;
        }
        grp.type = 2; // Node::AXON
        dst.push_back(grp);
    }
    for (size_t i = 1; i < dst.size(); i++)
        for (int &v : dst[i].nbr) v = to[v];
    tidy_links(dst);
}

// :379-478 -- breadth-first forest: every node keeps at most one link (to its BFS parent); single-node trees dropped
void bfs_forest(const List &g, List &tree)
{
    const size_t n = g.size();
    std::vector<char> seen(n, 0);
    std::vector<int> where(n, -1), parent(n, -1), queue;
    tree.clear();
    tree.push_back(g[0]);
    int trees = 0;
    size_t scan = 1; // first never-discovered index only moves forward
    for (;;) {
        while (scan < n && seen[scan]) scan++;
        if (scan >= n) break;
        trees++;
        queue.assign(1, (int)scan);
        seen[scan] = 1;
        size_t head = 0;
        int in_tree = 0;
        while (head < queue.size()) {
            const int cur = queue[head++];
            N t = g[cur];
            t.nbr.clear();
            if (t.type != 1) t.type = trees + 2;
            if (parent[cur] > 0) t.nbr.push_back(where[parent[cur]]);
            where[cur] = (int)tree.size();
            tree.push_back(t);
            in_tree++;
            for (int adj : g[cur].nbr)
                if (!seen[adj] && adj != 0) { seen[adj] = 1; parent[adj] = cur; queue.push_back(adj); }
            if (in_tree == 1 && head == queue.size()) { tree.pop_back(); where[cur] = -1; }
        }
    }
}

// :591-629 -- drop trees with fewer than min_size nodes (a tree = run of nodes starting at a parentless one)
void drop_small_trees(const List &X, List &Y, int min_size)
{
    const size_t n = X.size();
    std::vector<char> drop(n + 1, 0);
    size_t root_cur = 1, root_prev = 1;
    for (size_t i = 1; i <= n; i++)
        if (i == n || X[i].nbr.empty()) {
            root_prev = root_cur;
            root_cur = i;
            if ((long)(root_cur - root_prev) < (long)min_size)
                for (size_t j = root_prev; j < root_cur; j++) drop[j] = 1;
        }
    std::vector<int> to(n, -1);
    Y.clear();
    for (size_t i = 0; i < n; i++)
        if (!drop[i]) { to[i] = (int)Y.size(); Y.push_back(X[i]); }
    for (size_t i = 1; i < Y.size(); i++)
        for (int &v : Y[i].nbr) v = to[v];
}

// :546-589 (extract_largest_tree, the ENFORCE_SINGLE_TREE branch of reconstruct() :2142-2152) -- keep the largest tree of a tree list
// only.  The reference measures a tree by the distance between consecutive roots, starting with root_curr = root_prev = 1: mirrored.
void keep_largest_tree(const List &X, List &Y)
{
    const long n = (long)X.size();
    long root_cur = 1, root_prev = 1, best = -(long)INT32_MAX, beg = -(long)INT32_MAX, end = -(long)INT32_MAX;
    for (long i = 1; i <= n; i++)
        if (i == n || X[(size_t)i].nbr.empty()) {
            root_prev = root_cur;
            root_cur = i;
            if (root_cur - root_prev > best) { best = root_cur - root_prev; beg = root_prev; end = root_cur; }
        }
    std::vector<int> to((size_t)n, -1);
    Y.clear();
    for (long i = 0; i < n; i++)
        if (i == 0 || (i >= beg && i < end)) { to[(size_t)i] = (int)Y.size(); Y.push_back(X[(size_t)i]); }
    for (size_t i = 1; i < Y.size(); i++)
        for (int &v : Y[i].nbr) v = to[(size_t)v];
}

// :714-778 -- resample the (one-directional) tree links
void resample_tree(List &t, float step, int type)
{
    const size_t init = t.size();
    for (size_t i = 1; i < init; i++) {
        if (type >= 0 && t[i].type != 1) t[i].type = type;
        for (size_t j = 0; j < t[i].nbr.size(); j++) {
            const int i1 = t[i].nbr[j];
            const float vnorm = length(t[i], t[i1]);
            const float vx = (t[i1].x - t[i].x) / vnorm, vy = (t[i1].y - t[i].y) / vnorm, vz = (t[i1].z - t[i].z) / vnorm;
            const int Nseg = (int)std::ceil(vnorm / step);
            for (int k = 1; k < Nseg; k++) {
                t.push_back(between(t[i], t[i1], k, Nseg, vnorm, vx, vy, vz));
                const int last = (int)t.size() - 1;
                if (k == 1) t[i].nbr[j] = last;
                else t[last - 1].nbr.push_back(last);
                if (k == Nseg - 1) t[last].nbr.push_back(i1);
            }
        }
    }
}

} // namespace

static std::atomic<bool> g_recon_timing{false};
void set_recon_timing(bool on) { g_recon_timing.store(on, std::memory_order_relaxed); }
bool recon_timing() { return g_recon_timing.load(std::memory_order_relaxed); }

static void export_list(const List &g, bool tree, std::vector<pnr_node> &out_nodes, std::vector<int32_t> &out_links)
{
    out_nodes.resize(g.size());
    out_links.clear();
    for (size_t i = 0; i < g.size(); i++) {
        const N &s = g[i];
        out_nodes[i] = pnr_node{s.x, s.y, s.z, s.vx, s.vy, s.vz, s.corr, s.sig, s.type};
        for (int j : s.nbr)
            if (tree || (size_t)j > i || std::find(g[(size_t)j].nbr.begin(), g[(size_t)j].nbr.end(), (int)i) == g[(size_t)j].nbr.end()) {
                out_links.push_back((int32_t)i); // tree lists: (child, parent); node lists: every undirected link once
                out_links.push_back((int32_t)j);
            }
    }
}

void reconstruct(const std::vector<pnr_node> &nodes, const std::vector<int32_t> &links, const ReconParams &rp,
                 std::vector<pnr_node> &out_nodes, std::vector<int32_t> &out_parent, int stop_after, std::vector<int32_t> *stage_links)
{
    List n0(nodes.size());
    for (size_t i = 0; i < nodes.size(); i++) {
        const pnr_node &s = nodes[i];
        n0[i] = N{s.x, s.y, s.z, s.vx, s.vy, s.vz, s.corr, s.sig, s.type, {}};
    }
    for (size_t k = 0; k + 1 < links.size(); k += 2) {
        n0[links[k]].nbr.push_back(links[k + 1]);
        n0[links[k + 1]].nbr.push_back(links[k]);
    }
    List n2, forest, kept;
    const bool timing = recon_timing(); // option "recon_timing": the stages' wall times on stderr
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what, size_t n) {
        if (!timing) return;
        const auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[pnr reconstruct] %-14s %8.2f ms  (%zu nodes)\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count(), n);
        t_prev = t;
    };
    lap("graph", n0.size());
    // stop_after (the saveMidres taps of :2098-2141): the list as it stands behind that stage, with its links
    auto tap = [&](int stage, const List &g, bool tree) {
        if (stop_after != stage || !stage_links) return false;
        export_list(g, tree, out_nodes, *stage_links);
        out_parent.clear();
        return true;
    };
    resample_links(n0, rp.trace_rsmpl);
    lap("resample_links", n0.size());
    if (tap(RECON_N0RES, n0, false)) return;
    mean_shift(n0, rp.sig2radius, rp.refine_iter, rp.epsilon2, rp.threads);
    lap("mean_shift", n0.size());
    if (tap(RECON_N1, n0, false)) return;
    group_spheres(n0, n2, rp.group_radius);
    lap("group_spheres", n2.size());
    if (tap(RECON_N2, n2, false)) return;
    bfs_forest(n2, forest);
    lap("bfs_forest", forest.size());
    if (tap(RECON_N2TREE, forest, true)) return;
    if (rp.single_tree) keep_largest_tree(forest, kept); // ENFORCE_SINGLE_TREE (:2142-2152)
    else drop_small_trees(forest, kept, rp.tree_size_min);
    lap(rp.single_tree ? "largest_tree" : "drop_small", kept.size());
    resample_tree(kept, 1.0f, 2 /* Node::AXON */);
    lap("resample_tree", kept.size());
    out_nodes.resize(kept.size());
    out_parent.resize(kept.size());
    for (size_t i = 0; i < kept.size(); i++) {
        const N &s = kept[i];
        out_nodes[i] = pnr_node{s.x, s.y, s.z, s.vx, s.vy, s.vz, s.corr, s.sig, s.type};
        out_parent[i] = (i > 0 && !s.nbr.empty()) ? s.nbr[0] : -1;
    }
}

} // namespace advantra
