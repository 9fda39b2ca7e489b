// replay.h -- host replay of the sequential trace bookkeeping: the part of Tracker::trackPos that
// touches the shared maps and the node list (tracker.cpp:848-931) plus the trace loop of
// reconstruction_func (Advantra_plugin.cpp:2658-2710), applied IN SEED ORDER to traces whose particle
// filters were run on the GPU.  Incremental: traces may arrive in batches (pnr_trace_replay), the
// result is the same as replaying them all at once.
#pragma once
#include "../../include/pnr_hip.h"
#include <cfloat>
#include <cmath>
#include <cstring>
#include <unordered_map>
#include <vector>

namespace pnr {

inline int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

// neighbour voxels of the density pattern `vol` (Advantra_plugin.cpp:2609-2648), computed on the
// fly instead of the reference's 8 B/voxel pointer table.  The reference clamps y+-1 with N-1
// (the x extent) in the vol>=19 rows (:2632-2637); reproduced literally.
inline int density_neighbours(int64_t i, int N, int M, int P, int vol, int64_t *out)
{
    if (vol == 1) return 0;
    const int64_t NM = (int64_t)N * M;
    const int x = (int)(i % N), z = (int)(i / NM), y = (int)(i / N - (int64_t)z * M);
    auto at = [&](int zz, int yy, int xx) { return (int64_t)zz * NM + (int64_t)yy * N + xx; };
    const int xm = clampi(x - 1, 0, N - 1), xp = clampi(x + 1, 0, N - 1);
    const int ym = clampi(y - 1, 0, M - 1), yp = clampi(y + 1, 0, M - 1);
    const int zm = clampi(z - 1, 0, P - 1), zp = clampi(z + 1, 0, P - 1);
    const int ymN = clampi(y - 1, 0, N - 1), ypN = clampi(y + 1, 0, N - 1);
    int n = 0;
    out[n++] = at(z, y, xm); out[n++] = at(z, y, xp); out[n++] = at(z, ym, x); out[n++] = at(z, yp, x);
    if (vol >= 9) { out[n++] = at(z, ym, xm); out[n++] = at(z, ym, xp); out[n++] = at(z, yp, xm); out[n++] = at(z, yp, xp); }
    if (vol >= 11) { out[n++] = at(zm, y, x); out[n++] = at(zp, y, x); }
    if (vol >= 19) {
        out[n++] = at(zm, y, xm); out[n++] = at(zm, y, xp); out[n++] = at(zm, ymN, x); out[n++] = at(zm, ypN, x);
        out[n++] = at(zp, y, xm); out[n++] = at(zp, y, xp); out[n++] = at(zp, ymN, x); out[n++] = at(zp, ypN, x);
    }
    if (vol >= 27) {
        out[n++] = at(zm, ym, xm); out[n++] = at(zm, ym, xp); out[n++] = at(zm, yp, xm); out[n++] = at(zm, yp, xp);
        out[n++] = at(zp, ym, xm); out[n++] = at(zp, ym, xp); out[n++] = at(zp, yp, xm); out[n++] = at(zp, yp, xp);
    }
    return n;
}

struct Replayer {
    // npervol_map / nidx_map of the reference are dense N-voxel arrays (5 B/voxel, 5 GiB at 1024^3);
    // only voxels that received a node are ever non-zero, so a hash map holds the same state
    struct Cell { uint8_t den = 0; int32_t nidx = 0; };
    // ... open addressing, voxel indices are >= 0: the scheduler's tentative replay (stream_sched.h) looks up every recorded node of
    // every unreplayed trace at every poll, millions of finds per stack
    struct CellMap {
        std::vector<int64_t> key;
        std::vector<Cell> val;
        size_t used = 0;
        CellMap() { key.assign((size_t)1 << 17, -1); val.assign((size_t)1 << 17, Cell()); }
        static size_t hash(int64_t v) { return (size_t)(((uint64_t)v * 0x9E3779B97F4A7C15ull) >> 24); }
        void prefetch(int64_t v) const { __builtin_prefetch(&key[hash(v) & (key.size() - 1)]); }
        const Cell *find(int64_t v) const
        {
            const size_t mask = key.size() - 1;
            for (size_t h = hash(v) & mask;; h = (h + 1) & mask) {
                if (key[h] == v) return &val[h];
                if (key[h] < 0) return nullptr;
            }
        }
        Cell &operator[](int64_t v)
        {
            if (2 * (used + 1) > key.size()) grow();
            const size_t mask = key.size() - 1;
            for (size_t h = hash(v) & mask;; h = (h + 1) & mask) {
                if (key[h] == v) return val[h];
                if (key[h] < 0) { key[h] = v; used++; return val[h]; }
            }
        }
        void grow()
        {
            std::vector<int64_t> k0((size_t)2 * key.size(), -1);
            std::vector<Cell> v0((size_t)2 * key.size(), Cell());
            k0.swap(key); v0.swap(val);
            const size_t mask = key.size() - 1;
            for (size_t i = 0; i < k0.size(); i++)
                if (k0[i] >= 0) {
                    size_t h = hash(k0[i]) & mask;
                    while (key[h] >= 0) h = (h + 1) & mask;
                    key[h] = k0[i]; val[h] = v0[i];
                }
        }
    };
    pnr_params prm;
    int W, H, L;
    CellMap cells;
    std::vector<pnr_node> nodes;
    std::vector<int32_t> links;    // pairs (a,b): a.nbr.push_back(b); b.nbr.push_back(a)
    std::vector<int64_t> touched;  // voxels whose density changed since clear_touched()
    int trace_count = 0;
    bool stopped = false;          // MAX_TRACE_COUNT reached (:2702)
    const std::unordered_map<int64_t, int32_t> *soma = nullptr; // smap: voxel -> index of its SOMA node (> 0)
    // optional per-trace record of how the trace ended, what the reference prints at tracker.cpp:866,879,908,916:
    // {seed rank, direction, ti_limit, reason (0 TRACK LIMIT, 1 success=0, 2 DENSITY, 3 SOMA), value (corr bits / nodespervol / soma index)}
    struct TraceEnd { int32_t seed, dir, ti_limit, reason, value; };
    std::vector<TraceEnd> *log = nullptr;
    int64_t log_base = 0; // rank of seeds[0] of the next add()

    Replayer(const pnr_params &p, int64_t w, int64_t h, int64_t l) : prm(p), W((int)w), H((int)h), L((int)l)
    {
        pnr_node d; // n0[0]: dummy Node() (node.cpp:43-54; Advantra_plugin.cpp:2416-2419)
        std::memset(&d, 0, sizeof(d));
        d.corr = -FLT_MAX;
        d.type = 7;
        nodes.push_back(d);
    }
    // soma nodes follow the dummy node (soma_extraction1, Advantra_plugin.cpp:1911-1914); call before the first add()
    void set_soma(const std::unordered_map<int64_t, int32_t> *map, const std::vector<pnr_node> &soma_nodes)
    {
        soma = (map && !map->empty()) ? map : nullptr;
        nodes.insert(nodes.end(), soma_nodes.begin(), soma_nodes.end());
    }
    int soma_at(int64_t v) const
    {
        if (!soma) return 0;
        auto it = soma->find(v);
        return it == soma->end() ? 0 : (int)it->second;
    }
    int64_t voxel(float x, float y, float z) const
    {
        return (int64_t)(int)std::round(z) * W * H + (int64_t)(int)std::round(y) * W + (int)std::round(x);
    }
    int den_at(int64_t v) const
    {
        const Cell *c = cells.find(v);
        return c ? (int)c->den : 0;
    }
    bool seed_saturated(const pnr_seed &s) const { return !(den_at(voxel(s.x, s.y, s.z)) < prm.nodepervol); }
    void bump(int64_t v, int32_t node)
    {
        Cell &c = cells[v];
        c.den = (uint8_t)((int)c.den + 1);
        c.nidx = node;
        touched.push_back(v);
    }
    // traces of seed s: T[2*s+dir], xc + (2*s+dir)*ni; seeds in rank order
    void add(const pnr_seed *seeds, int64_t n, const int32_t *T, const pnr_xest *xc)
    {
        const int ni = prm.ni;
        const int maxtr = prm.max_trace_count > 0 ? prm.max_trace_count : 5000;
        for (int64_t s = 0; s < n && !stopped; s++) {
            if (seed_saturated(seeds[s])) continue; // :2669-2670
            trace_count++;
            for (int dir = 0; dir < 2; dir++) {
                const int64_t j = 2 * s + dir;
                const pnr_xest *X = xc + j * ni;
                int ti_limit = ni;
                int why = 0, val = 0; // TRACK LIMIT unless something ends the trace before
                if (ni > 0) std::memcpy(&val, &X[ni - 1].corr, 4);
                for (int i = 0; i < ni; i++) {
                    if (i >= T[j]) { ti_limit = i; why = 1; std::memcpy(&val, &X[i].corr, 4); break; } // iter*New returned false
                    const pnr_xest &e = X[i];
                    const int64_t crd = voxel(e.x, e.y, e.z);
                    if (const int sn = soma_at(crd)) { // soma reached: link with its node, stop the trace (tracker.cpp:858-869)
                        if (i > 0) { links.push_back(sn); links.push_back((int32_t)(nodes.size() - 1)); }
                        ti_limit = i; why = 3; val = sn;
                        break;
                    }
                    if (den_at(crd) >= prm.nodepervol) { // density limit: link to the node that owns the voxel
                        if (i > 0) { links.push_back(cells[crd].nidx); links.push_back((int32_t)(nodes.size() - 1)); }
                        ti_limit = i; why = 2; val = prm.nodepervol;
                        break;
                    }
                    nodes.push_back(pnr_node{e.x, e.y, e.z, e.vx, e.vy, e.vz, e.corr, e.sig, (i == 0) ? 7 : 2});
                    const int32_t me = (int32_t)(nodes.size() - 1);
                    bump(crd, me);
                    if (prm.vol > 1) {
                        int64_t nb[26];
                        const int cnt = density_neighbours(crd, W, H, L, prm.vol, nb);
                        for (int q = 0; q < cnt; q++) bump(nb[q], me);
                    }
                    if (i > 0) { links.push_back(me); links.push_back(me - 1); }
                }
                if (ti_limit > 1) nodes.back().type = 6; // END (tracker.cpp:930-931)
                if (log) log->push_back(TraceEnd{(int32_t)(log_base + s), dir, ti_limit, why, val});
            }
            if (trace_count > maxtr) stopped = true; // :2702
        }
    }
};

} // namespace pnr
