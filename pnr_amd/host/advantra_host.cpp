// advantra_host.cpp -- see advantra_host.h.  Host orchestration only: every compute stage is a call
// through the C ABI into libpnr_hip.so.
#include "advantra_host.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace advantra {

static const int nrInputParams = 11; // Advantra_plugin.cpp:60

Settings &settings()
{
    static Settings s;
    return s;
}

void print_help()
{
    // wording of the reference's print_help (Advantra_plugin.cpp:125-148), shortened to the call contract
    printf("**** usage of Advantra tracing ****\n");
    printf("vaa3d -x Advantra -f advantra_func -i <inimg_file> -p <neuritesigmas> <somaradius> <tolerance> <znccth> <kappa> "
           "<step> <ni> <np> <zdist> <nodepervol> <vol>\n");
    printf("inimg_file     The input image (8-bit multi-page TIFF)\n");
    printf("neuritesigmas  Comma delimited list of gaussian cross-section sigmas, e.g. 2,4,6\n");
    printf("somaradius     Soma radius (0: no soma)\n");
    printf("tolerance      Seed extraction (find maxima) tolerance\n");
    printf("znccth         Correlation threshold [0,1]\n");
    printf("kappa          Von Mises kappa [0,5]\n");
    printf("step           Prediction step\n");
    printf("ni             Number of iterations\n");
    printf("np             Number of particles\n");
    printf("zdist          Distance between layers in pixels\n");
    printf("nodepervol     Node density limit (2,20]\n");
    printf("vol            Volume pattern: 1,5,9,11,19,27\n");
    printf("outswc_file    <inimg_file>_Advantra.swc\n");
}

// ---- minimal baseline TIFF reader: 8-bit grayscale, uncompressed strips, any number of pages ----
namespace {
struct Reader {
    std::vector<unsigned char> buf;
    bool be = false;
    uint16_t u16(size_t o) const { return be ? (uint16_t)(buf[o] << 8 | buf[o + 1]) : (uint16_t)(buf[o] | buf[o + 1] << 8); }
    uint32_t u32(size_t o) const
    {
        return be ? ((uint32_t)buf[o] << 24 | (uint32_t)buf[o + 1] << 16 | (uint32_t)buf[o + 2] << 8 | buf[o + 3])
                  : ((uint32_t)buf[o + 3] << 24 | (uint32_t)buf[o + 2] << 16 | (uint32_t)buf[o + 1] << 8 | buf[o]);
    }
    std::vector<uint32_t> values(uint16_t type, uint32_t count, size_t field) const
    {
        const size_t sz = (type == 3) ? 2 : (type == 4 ? 4 : 1);
        size_t off = (sz * count <= 4) ? field : u32(field);
        if (off > buf.size() || (size_t)count > (buf.size() - off) / sz) return {}; // a count the file cannot hold: nothing is allocated for it
        std::vector<uint32_t> v(count);
        for (uint32_t i = 0; i < count; i++) {
            if (off + sz > buf.size()) return {};
            v[i] = (type == 3) ? u16(off) : (type == 4 ? u32(off) : buf[off]);
            off += sz;
        }
        return v;
    }
};
} // namespace

static bool load_tiff(const std::string &path, Stack &out, std::string &err)
{
    Reader r;
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "cannot open " + path; return false; }
    r.buf.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    if (r.buf.size() < 8) { err = "not a TIFF"; return false; }
    if (r.buf[0] == 'M' && r.buf[1] == 'M') r.be = true;
    else if (!(r.buf[0] == 'I' && r.buf[1] == 'I')) { err = "not a TIFF"; return false; }
    if (r.u16(2) != 42) { err = "not a baseline TIFF (BigTIFF is not supported)"; return false; }
    size_t ifd = r.u32(4);
    out.data.clear();
    out.w = out.h = out.l = 0;
    std::vector<size_t> seen; // IFD offsets visited: a chain that revisits one would never end
    while (ifd != 0) {
        if (ifd + 2 > r.buf.size()) { err = "truncated TIFF"; return false; }
        if (std::find(seen.begin(), seen.end(), ifd) != seen.end()) { err = "TIFF directory chain loops"; return false; }
        seen.push_back(ifd);
        if (seen.size() > (1u << 20)) { err = "TIFF with more than 2^20 pages"; return false; }
        const uint16_t n = r.u16(ifd);
        uint32_t w = 0, h = 0, bps = 1, comp = 1, spp = 1, rps = 0xffffffffu;
        std::vector<uint32_t> soff, scnt;
        for (uint16_t e = 0; e < n; e++) {
            const size_t o = ifd + 2 + 12 * (size_t)e;
            if (o + 12 > r.buf.size()) { err = "truncated TIFF"; return false; }
            const uint16_t tag = r.u16(o), type = r.u16(o + 2);
            const uint32_t cnt = r.u32(o + 4);
            const std::vector<uint32_t> v = r.values(type, cnt, o + 8);
            if (v.empty()) continue;
            switch (tag) {
            case 256: w = v[0]; break;
            case 257: h = v[0]; break;
            case 258: bps = v[0]; break;
            case 259: comp = v[0]; break;
            case 273: soff = v; break;
            case 277: spp = v[0]; break;
            case 278: rps = v[0]; break;
            case 279: scnt = v; break;
            }
        }
        if (bps != 8 || spp != 1) { err = "only 8-bit single-channel stacks are supported (the reference assumes uint8, Advantra_plugin.cpp:2255)"; return false; }
        if (comp != 1) { err = "compressed TIFF is not supported"; return false; }
        if (w == 0 || h == 0 || (uint64_t)w * h > r.buf.size()) { err = "TIFF page larger than the file"; return false; } // uncompressed: w*h bytes must be in the file
        if (out.l == 0) { out.w = w; out.h = h; }
        else if (w != out.w || h != out.h) { err = "pages of different size"; return false; }
        (void)rps;
        size_t got = 0;
        const size_t page = (size_t)w * h;
        const size_t base = out.data.size();
        out.data.resize(base + page);
        for (size_t s = 0; s < soff.size() && got < page; s++) {
            size_t c = (s < scnt.size()) ? scnt[s] : page - got;
            c = std::min(c, page - got);
            if ((size_t)soff[s] + c > r.buf.size()) { err = "truncated TIFF strip"; return false; }
            std::memcpy(out.data.data() + base + got, r.buf.data() + soff[s], c);
            got += c;
        }
        if (got != page) { err = "TIFF page shorter than width*height"; return false; }
        out.l++;
        const size_t nx = ifd + 2 + 12 * (size_t)n;
        if (nx + 4 > r.buf.size()) break;
        ifd = r.u32(nx);
    }
    return out.l > 0;
}

bool load_stack(const std::string &path, const std::string &raw_dims, Stack &out, std::string &err)
{
    const bool raw = path.size() > 4 && path.substr(path.size() - 4) == ".raw";
    if (!raw) return load_tiff(path, out, err);
    long long w = 0, h = 0, l = 0;
    if (sscanf(raw_dims.c_str(), "%lld,%lld,%lld", &w, &h, &l) != 3 || w <= 0 || h <= 0 || l <= 0) {
        err = "raw stacks need -d w,h,l";
        return false;
    }
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "cannot open " + path; return false; }
    out.data.resize((size_t)(w * h * l));
    f.read((char *)out.data.data(), (std::streamsize)out.data.size());
    if ((long long)f.gcount() != w * h * l) { err = "raw file shorter than w*h*l"; return false; }
    out.w = w; out.h = h; out.l = l;
    return true;
}

bool save_nodelist(const std::vector<pnr_node> &nodes, const std::vector<int32_t> &links, const std::string &swcname, int type,
                   float sig2r, const std::string &name, const std::string &comment)
{
    // one line per (node, neighbour) pair, ids repeat; isolated nodes get parent -1; node 0 is the dummy
    std::vector<std::vector<int>> nbr(nodes.size());
    for (size_t k = 0; k + 1 < links.size(); k += 2) { // a.nbr.push_back(b); b.nbr.push_back(a)
        nbr[links[k]].push_back(links[k + 1]);
        nbr[links[k + 1]].push_back(links[k]);
    }
    FILE *f = fopen(swcname.c_str(), "w");
    if (!f) return false;
    if (!name.empty()) fprintf(f, "#name %s\n", name.c_str());
    if (!comment.empty()) {
        std::stringstream ss(comment);
        std::string ln;
        bool first = true;
        while (std::getline(ss, ln)) {
            if (ln.empty()) continue;
            fprintf(f, "%s%s\n", (first || ln[0] != '#') ? "#comment " : "", ln.c_str());
            first = false;
        }
    }
    fprintf(f, "##n,type,x,y,z,radius,parent\n");
    for (size_t i = 1; i < nodes.size(); i++) {
        const pnr_node &nd = nodes[i];
        const int t = (type == -1) ? nd.type : type;
        if (nbr[i].empty()) fprintf(f, "%zu %d %.3f %.3f %.3f %.3f %d\n", i, t, nd.x, nd.y, nd.z, sig2r * nd.sig, -1);
        for (int par : nbr[i]) fprintf(f, "%zu %d %.3f %.3f %.3f %.3f %d\n", i, t, nd.x, nd.y, nd.z, sig2r * nd.sig, par);
    }
    fclose(f);
    return true;
}

bool save_treelist(const std::vector<pnr_node> &tree, const std::vector<int32_t> &parent, const std::string &swcname, int type,
                   float sig2r, const std::string &name, const std::string &comment)
{
    FILE *f = fopen(swcname.c_str(), "w");
    if (!f) return false;
    if (!name.empty()) fprintf(f, "#name %s\n", name.c_str());
    if (!comment.empty()) {
        std::stringstream ss(comment);
        std::string ln;
        bool first = true;
        while (std::getline(ss, ln)) {
            if (ln.empty()) continue;
            fprintf(f, "%s%s\n", (first || ln[0] != '#') ? "#comment " : "", ln.c_str());
            first = false;
        }
    }
    fprintf(f, "##n,type,x,y,z,radius,parent\n");
    for (size_t i = 1; i < tree.size(); i++) {
        const pnr_node &nd = tree[i];
        fprintf(f, "%zu %d %.3f %.3f %.3f %.3f %d\n", i, (type == -1) ? nd.type : type, nd.x, nd.y, nd.z, sig2r * nd.sig, parent[i]);
    }
    fclose(f);
    return true;
}

int parse_params(const std::vector<std::string> &paras, pnr_params &p, std::string &err)
{
    if ((int)paras.size() != nrInputParams) { // Advantra_plugin.cpp:295-299
        err = "Needs 11 input parameters.";
        return -1;
    }
    pnr_default_params(&p);
    { // parse_csv_string (:1885-1897): comma separated floats, sorted ascending
        std::vector<float> sig;
        std::stringstream ss(paras[0]);
        float v;
        while (ss >> v) {
            sig.push_back(v);
            if (ss.peek() == ',') ss.ignore();
        }
        std::sort(sig.begin(), sig.end());
        if (sig.empty() || sig.size() > PNR_MAX_SIGMAS) { err = "neuritesigmas out of range"; return -2; }
        p.nsig = (int)sig.size();
        for (int i = 0; i < p.nsig; i++) p.sig[i] = sig[i];
    }
    p.somaradius = atoi(paras[1].c_str());
    p.tolerance = (float)atof(paras[2].c_str());
    p.znccth = (float)atof(paras[3].c_str());
    p.kappa = (float)atof(paras[4].c_str());
    p.step = atoi(paras[5].c_str());
    p.ni = atoi(paras[6].c_str());
    p.np = atoi(paras[7].c_str());
    p.zdist = (float)atof(paras[8].c_str());
    p.nodepervol = atoi(paras[9].c_str());
    p.vol = atoi(paras[10].c_str());
    // range checks and messages of Advantra_plugin.cpp:317-326
    if (p.somaradius < 0) { err = "somaradius out of range"; return -2; }
    if (p.tolerance < 0) { err = "tolerance out of range"; return -2; }
    if (p.znccth < 0 || p.znccth > 1) { err = "znccth out of range"; return -2; }
    if (p.kappa < 0 || p.kappa > 5) { err = "kappa out of range"; return -2; }
    if (p.step < 1) { err = "step out of range"; return -2; }
    if (p.ni <= 0) { err = "ni out of range"; return -2; }
    if (p.np <= 0) { err = "np out of range"; return -2; }
    if (p.zdist < 1) { err = "zdist out of range"; return -2; }
    if (p.nodepervol <= 2 || p.nodepervol > 20) { err = "nodepervol out of range"; return -2; }
    if (!(p.vol == 1 || p.vol == 5 || p.vol == 9 || p.vol == 11 || p.vol == 19 || p.vol == 27)) { err = "vol can be 1,5,9,11,19,27"; return -2; }
    return 0;
}

static std::string swc_comment(const std::vector<std::string> &paras, const pnr_params &p)
{
    static const char *keys[] = {"neuritesigmas", "somaradius", "tolerance", "znccth", "kappa", "step", "ni", "np", "zdist", "nodepervol", "vol"};
    std::stringstream c;
    c << "email: miro@braincadet.com\n#params:\n#channel=1"; // Advantra_plugin.cpp:2276-2306
    for (int i = 0; i < nrInputParams; i++) c << "\n#" << keys[i] << "=" << paras[i];
    c << "\n#------------------------\n#Kc=" << p.Kc << "\n#neff_ratio=" << p.neff_ratio << "\n#frangi_alfa=" << p.alpha
      << "\n#frangi_beta=" << p.beta << "\n#frangi_C=" << p.C << "\n#MAX_TRACE_COUNT=" << p.max_trace_count
      << "\n#EPSILON2=0.0001\n#REFINE_ITER=4\n#SIG2RADIUS=1.5\n#TRACE_RSMPL=1\n#GROUP_RADIUS=2\n#ENFORCE_SINGLE_TREE=0\n#TREE_SIZE_MIN=10\n#TAIL_SIZE_MIN=2";
    return c.str();
}

bool advantra_func(const std::vector<char *> &infiles, const std::vector<char *> &paras_c, int device, const std::string &raw_dims,
                   Result *result)
{
    if (infiles.empty()) {
        fprintf(stderr, "Need input image. \n"); // :286-289
        return false;
    }
    std::vector<std::string> paras(paras_c.begin(), paras_c.end());
    pnr_params p;
    std::string err;
    const int pr = parse_params(paras, p, err);
    if (pr == -1) {
        fprintf(stderr, "\nNeeds %d input parameters.\n\n", nrInputParams);
        print_help();
        return false;
    }
    if (pr == -2) {
        fprintf(stderr, "%s\n", err.c_str()); // v3d_msg(...); return 0
        return true;
    }
    Stack st;
    if (!load_stack(infiles[0], raw_dims, st, err)) {
        fprintf(stderr, "%s\n", err.c_str());
        return true;
    }
    reconstruction_func(st.data.data(), st.w, st.h, st.l, infiles[0], paras, p, device, result);
    return true;
}

bool reconstruction_func(const unsigned char *data1d, long long w, long long h, long long l, const std::string &inimg_file,
                         const std::vector<std::string> &paras, pnr_params p, int device, Result *result)
{
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    p.rng_seed = settings().rng_seed;
    printf("-------------  ADVANTRA  -------------\n");
    pnr_ctx *ctx = nullptr;
    if (pnr_create(&p, device, &ctx) != PNR_OK || pnr_set_volume(ctx, data1d, w, h, l) != PNR_OK) {
        fprintf(stderr, "%s\n", pnr_last_error());
        pnr_destroy(ctx);
        return false;
    }
    Result R;
    bool ok = true;
    if (p.somaradius > 0) { // SOMA EXTR. (:2426-2486): erosion, xy blur, max-entropy threshold, regions -> soma nodes
        auto ts = clk::now();
        int32_t th = 0;
        int64_t nsoma = 0;
        ok = pnr_soma(ctx, nullptr, &th, &nsoma) == PNR_OK;
        printf("imerode(%d) imgaussian(%d) maxentropy_th() %d  %lld soma regions  %.3f sec.\n", p.somaradius, p.somaradius, (int)th, (long long)nsoma,
               std::chrono::duration<double>(clk::now() - ts).count());
    } else {
        printf("no soma detection\n");
    }
    auto t0 = clk::now();
    ok = ok && pnr_frangi(ctx, &R.Jmin, &R.Jmax) == PNR_OK; // :2496-2512
    auto t1 = clk::now();
    const pnr_seed *found = nullptr;
    int64_t nfound = 0;
    ok = ok && pnr_extract_seeds(ctx, &found, &nfound) == PNR_OK; // :2549
    auto t2 = clk::now();
    std::vector<pnr_seed> seeds;
    int64_t nseeds = 0;
    if (ok) {
        seeds.assign(found, found + nfound);
        printf("seed extraction... %gk seeds,  %g sec.\n", nfound / 1000.0, secs(t1, t2));
        ok = pnr_score_filter_sort_seeds(ctx, seeds.data(), nfound, &nseeds) == PNR_OK; // :2561-2586
        seeds.resize((size_t)nseeds);
    }
    auto t3 = clk::now();
    int64_t nn = 0, nl = 0, used = 0, iters = 0;
    if (ok) {
        printf("seed selection & sorting... %gk seeds, %g sec.\ntracing...\n", nseeds / 1000.0, secs(t2, t3));
        // :2658-2710: the particle filters on the GPU (a window of traces refilled as they stop), the bookkeeping replayed on the
        // host in seed order; the graph stays in the context and is fetched once its size is known
        if (settings().verbose) pnr_set_option(ctx, "trace_log", 1);
        ok = pnr_trace_replay(ctx, seeds.data(), nseeds, 0, nullptr, 0, &nn, nullptr, 0, &nl, &used, &iters) == PNR_OK;
        if (ok) {
            R.nodes.resize((size_t)nn);
            R.links.resize((size_t)(2 * nl));
            ok = pnr_get_graph(ctx, R.nodes.data(), nn, &nn, R.links.data(), nl, &nl) == PNR_OK;
        }
        if (ok && settings().verbose) { // what the reference prints while it traces (TRACING_VERBOSE :2677; tracker.cpp:866,879,908,916)
            int64_t nlog = 0;
            pnr_get_trace_log(ctx, nullptr, 0, &nlog);
            std::vector<int32_t> lg((size_t)nlog * 5);
            pnr_get_trace_log(ctx, lg.data(), nlog, &nlog);
            int trace_count = 0;
            for (int64_t k = 0; k < nlog; k++) {
                const int32_t *e = &lg[(size_t)k * 5];
                const pnr_seed &sd = seeds[(size_t)e[0]];
                if (e[1] == 0)
                    printf("\nTrace: %6d\t [%4.1f, %4.1f, %4.1f]\t sc=%6.2f\t corr=%3.2f\t progress %3.2f%%\t", ++trace_count, sd.x, sd.y, sd.z, sd.score, sd.corr,
                           (100.0 * e[0]) / (double)nseeds);
                float cv;
                std::memcpy(&cv, &e[4], 4);
                switch (e[3]) {
                case 3: printf("\n--%d[%d], SOMA, idx=%d", e[2], p.ni, e[4]); break;
                case 2: printf("\n--%d[%d], DENSITY, nodespervol=%d", e[2], p.ni, e[4]); break;
                case 1: printf("\n--%d[%d], success=0, corr=%1.2f", e[2], p.ni, cv); break;
                default: printf("\n--%d[%d], TRACK LIMIT, niter=%d", e[2], p.ni, p.ni); break;
                }
            }
            printf("\n");
        }
    }
    auto t4 = clk::now();
    if (!ok) {
        fprintf(stderr, "%s\n", pnr_last_error());
        pnr_destroy(ctx);
        return false;
    }
    R.n_seeds_init = nfound; R.n_seeds = nseeds; R.n_traces = used; R.n_iterations = iters;
    R.t_frangi = secs(t0, t1); R.t_seeds = secs(t1, t2); R.t_select = secs(t2, t3); R.t_trace = secs(t3, t4);
    printf("\n-----\n%g%% seeds used \n", nseeds ? 100.0 * used / nseeds : 0.0);
    { // reconstruct(n0, ...) :2729 -> :2096-2181 (host): refinement, grouping, trees, final resampling
        int64_t cap = std::max<int64_t>(16, 4 * nn), nt = 0;
        for (;;) {
            R.tree.resize((size_t)cap);
            R.parent.resize((size_t)cap);
            if (pnr_reconstruct(R.nodes.data(), nn, R.links.data(), nl, 0, 0, 0, 0, 0, 0, R.tree.data(), R.parent.data(), cap, &nt) != PNR_OK) {
                fprintf(stderr, "%s\n", pnr_last_error());
                pnr_destroy(ctx);
                return false;
            }
            if (nt <= cap) break;
            cap = nt;
        }
        R.tree.resize((size_t)nt);
        R.parent.resize((size_t)nt);
    }
    auto t5 = clk::now();
    R.t_recon = secs(t4, t5);
    R.swc_path = inimg_file + "_Advantra.swc"; // :2164
    save_treelist(R.tree, R.parent, R.swc_path, -1, 1.f, "Advantra", swc_comment(paras, p));
    if (settings().save_midres) save_nodelist(R.nodes, R.links, inimg_file + "_n0_.swc"); // saveMidres tap (:2099)
    printf("%s\n%lld trace nodes, %lld traces, %lld SMC iterations, %zu tree nodes | frangi %.3f s, seeds %.3f s, selection %.3f s, "
           "tracing %.3f s, reconstruct %.3f s\n",
           R.swc_path.c_str(), (long long)nn - 1, (long long)used, (long long)iters, R.tree.size() - 1, R.t_frangi, R.t_seeds, R.t_select,
           R.t_trace, R.t_recon);
    pnr_destroy(ctx);
    if (result) *result = R;
    return true;
}

} // namespace advantra
