// advantra_host.cpp -- see advantra_host.h.  Host orchestration only: every compute stage is a call
// through the C ABI into libpnr_hip.so.
#include "advantra_host.h"
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cmath>
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
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) { err = "cannot open " + path; return false; }
    struct stat sb;
    if (fstat(fd, &sb) != 0 || (long long)sb.st_size < w * h * l) { close(fd); err = "raw file shorter than w*h*l"; return false; }
    void *m = mmap(nullptr, (size_t)(w * h * l), PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { // (a file system that cannot map: read it)
        std::ifstream f(path, std::ios::binary);
        if (!f) { err = "cannot open " + path; return false; }
        out.data.resize((size_t)(w * h * l));
        f.read((char *)out.data.data(), (std::streamsize)out.data.size());
        if ((long long)f.gcount() != w * h * l) { err = "raw file shorter than w*h*l"; return false; }
    } else {
        out.view = (const unsigned char *)m;
        out.map_len = (size_t)(w * h * l);
        (void)madvise(m, out.map_len, MADV_SEQUENTIAL);
    }
    out.w = w; out.h = h; out.l = l;
    return true;
}

Stack::~Stack()
{
    if (view) munmap((void *)view, map_len);
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
    const auto tl0 = std::chrono::steady_clock::now();
    if (!load_stack(infiles[0], raw_dims, st, err)) {
        fprintf(stderr, "%s\n", err.c_str());
        return true;
    }
    Result local;
    Result *R = result ? result : &local;
    const double t_load = std::chrono::duration<double>(std::chrono::steady_clock::now() - tl0).count();
    // a failure of the device library (no GPU, out of memory, a failed exchange) has no counterpart in the reference's contract:
    // it is reported on stderr by reconstruction_func and makes the function -- and the CLI's exit status -- fail
    if (!reconstruction_func(st.bytes(), st.w, st.h, st.l, infiles[0], paras, p, device, R)) return false;
    if (settings().rank == 0) {
        // what a user of advantra_func waits for (Advantra_plugin.cpp:2241 load, :2183-2731 reconstruction_func, :2164 the SWC)
        R->t_load = t_load;
        R->t_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - tl0).count();
        printf("wall: load %.3f s, context + upload %.3f s, frangi %.3f s, seeds %.3f s, selection %.3f s, tracing %.3f s, reconstruct %.3f s, "
               "write %.3f s | total %.3f s for %lld voxels\n",
               R->t_load, R->t_setup, R->t_frangi, R->t_seeds, R->t_select, R->t_trace, R->t_recon, R->t_write, R->t_total, (long long)(st.w * st.h * st.l));
    }
    return true;
}

// ---- the few collectives of the sharded path, over the transport the ranks were joined with: the shared-memory all-gather of the
// ranks of this host (default) or RCCL (--exchange rccl) -----------------------------------------------------------------------------
namespace {
constexpr size_t XCHUNK = 1 << 18; // bytes per rank and call (both transports are opened with this capacity)
bool allgather_bytes(pnr_allgather_fn fn, void *user, int world, const void *send, size_t n, std::vector<unsigned char> &out)
{
    out.assign(n * (size_t)world, 0);
    std::vector<unsigned char> sb(XCHUNK), rb(XCHUNK * (size_t)world);
    for (size_t off = 0; off < n || off == 0; off += XCHUNK) {
        const size_t m = std::min(XCHUNK, n - off);
        if (m) std::memcpy(sb.data(), (const unsigned char *)send + off, m);
        if (fn(user, sb.data(), rb.data(), (int64_t)m) != PNR_OK) return false;
        for (int r = 0; r < world; r++)
            if (m) std::memcpy(out.data() + (size_t)r * n + off, rb.data() + (size_t)r * m, m);
        if (n == 0) break;
    }
    return true;
}
} // namespace

bool reconstruction_func(const unsigned char *data1d, long long w, long long h, long long l, const std::string &inimg_file,
                         const std::vector<std::string> &paras, pnr_params p, int device, Result *result)
{
    const int rank = settings().rank, world = settings().world;
    // the transport of the sharded path's collectives
    pnr_rccl_exchange *const RX = settings().rccl;
    const pnr_allgather_fn xfn = RX ? pnr_rccl_allgather : pnr_shm_allgather;
    void *const X = RX ? (void *)RX : (void *)settings().exchange;
    const bool sharded = world > 1 || settings().force_shard;
    if (sharded && (!X || l < 2)) {
        fprintf(stderr, "--ranks needs a stack of at least 2 planes and an open exchange\n");
        return false;
    }
    if (rank != 0) { // only rank 0 talks
        if (!freopen("/dev/null", "w", stdout)) return false;
    }
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    p.rng_seed = settings().rng_seed;
    printf("-------------  ADVANTRA  -------------\n");
    const auto t_begin = clk::now();
    pnr_ctx *ctx = nullptr;
    if (pnr_create(&p, device, &ctx) != PNR_OK) {
        fprintf(stderr, "%s\n", pnr_last_error());
        return false;
    }
    pnr_set_option(ctx, "local_ranks", world);
    const auto t_created = clk::now();
    Result R;
    bool ok = true;
    auto run_soma = [&]() { // SOMA EXTR. (:2426-2486): erosion, xy blur, max-entropy threshold, regions -> soma nodes
        if (p.somaradius <= 0) { printf("no soma detection\n"); return; }
        auto ts = clk::now();
        int32_t th = 0;
        int64_t nsoma = 0;
        ok = ok && pnr_soma(ctx, nullptr, &th, &nsoma) == PNR_OK;
        printf("imerode(%d) imgaussian(%d) maxentropy_th() %d  %lld soma regions  %.3f sec.\n", p.somaradius, p.somaradius, (int)th, (long long)nsoma,
               std::chrono::duration<double>(clk::now() - ts).count());
    };
    std::vector<pnr_seed> seeds;
    int64_t nfound = 0, nseeds = 0;
    auto t0 = clk::now(), t1 = t0, t2 = t0;
    if (!sharded) {
        ok = pnr_set_volume(ctx, data1d, w, h, l) == PNR_OK;
        if (settings().timing) fprintf(stderr, "[pnr host] context %.3f s, upload of %.2f GB %.3f s\n", secs(t_begin, t_created), (double)(w * h * l) / 1e9, secs(t_created, clk::now()));
        if (ok) run_soma();
        t0 = clk::now();
        ok = ok && pnr_frangi(ctx, &R.Jmin, &R.Jmax) == PNR_OK; // :2496-2512
        t1 = clk::now();
        const pnr_seed *found = nullptr;
        ok = ok && pnr_extract_seeds(ctx, &found, &nfound) == PNR_OK; // :2549
        t2 = clk::now();
        if (ok) {
            seeds.assign(found, found + nfound);
            printf("seed extraction... %gk seeds,  %g sec.\n", nfound / 1000.0, secs(t1, t2));
            ok = pnr_score_filter_sort_seeds(ctx, seeds.data(), nfound, &nseeds) == PNR_OK; // :2561-2586
            seeds.resize((size_t)nseeds);
        }
    } else {
        // this rank's z-slab with its halo (the z pass of the widest Gaussian + the radius-2 Hessian stencil): exact Frangi / seeds
        // of the planes it owns, no halo exchange -- every rank has the whole stack
        float smax = 0;
        for (int i = 0; i < p.nsig; i++) smax = std::max(smax, p.sig[i]);
        const long long halo = (long long)std::ceil(3 * (smax / p.zdist)) + 2;
        const long long z0 = l * rank / world, z1 = l * (rank + 1) / world, zlo = std::max(0LL, z0 - halo), zhi = std::min(l, z1 + halo);
        float mm[2] = {FLT_MAX, -FLT_MAX}; // (min, max) of J over the owned planes
        const pnr_seed *found = nullptr;
        int64_t nmine = 0;
        std::vector<pnr_seed> mine;
        if (z1 > z0) {
            ok = pnr_set_volume(ctx, data1d + zlo * w * h, w, h, zhi - zlo) == PNR_OK;
            ok = ok && pnr_frangi_slab(ctx, z0 - zlo, z1 - zlo, &mm[0], &mm[1]) == PNR_OK;
        }
        // the 2-float all-reduce: one ncclAllReduce over RCCL, an all-gather + local reduction through shared memory
        R.Jmin = FLT_MAX; R.Jmax = -FLT_MAX;
        if (RX) {
            R.Jmin = mm[0]; R.Jmax = mm[1];
            if (pnr_rccl_allreduce_minmax(RX, &R.Jmin, &R.Jmax) != PNR_OK) ok = false;
        } else {
            std::vector<unsigned char> all;
            if (!allgather_bytes(xfn, X, world, mm, sizeof(mm), all)) ok = false;
            for (int r = 0; r < world && ok; r++) {
                float q[2];
                std::memcpy(q, all.data() + (size_t)r * sizeof(q), sizeof(q));
                R.Jmin = std::min(R.Jmin, q[0]); R.Jmax = std::max(R.Jmax, q[1]);
            }
        }
        t1 = clk::now();
        if (ok && z1 > z0) {
            ok = pnr_quantise_j8(ctx, R.Jmin, R.Jmax) == PNR_OK && pnr_extract_seeds_range(ctx, z0 - zlo, z1 - zlo, &found, &nmine) == PNR_OK;
            if (ok) {
                mine.assign(found, found + nmine);
                for (auto &sd : mine) sd.z += (float)zlo;
            }
        }
        t2 = clk::now();
        ok = ok && pnr_set_volume(ctx, data1d, w, h, l) == PNR_OK; // scoring and tracing see the whole stack
        if (ok) run_soma();
        int64_t cnt[2] = {nmine, 0};
        if (ok && nmine) ok = pnr_score_filter_seeds(ctx, mine.data(), nmine, &cnt[1]) == PNR_OK; // this slab's seeds: znccBBB + threshold
        if (!ok) cnt[0] = -1; // a rank that failed says so: every collective up to here was entered by everybody, none after is
        std::vector<unsigned char> counts;
        if (!allgather_bytes(xfn, X, world, cnt, sizeof(cnt), counts)) ok = false;
        int64_t mx = 0;
        std::vector<int64_t> kept((size_t)world);
        for (int r = 0; r < world; r++) {
            int64_t q[2] = {-1, 0};
            if (counts.size() >= (size_t)(r + 1) * sizeof(q)) std::memcpy(q, counts.data() + (size_t)r * sizeof(q), sizeof(q));
            if (q[0] < 0) { if (ok) fprintf(stderr, "rank %d failed\n", r); ok = false; continue; }
            nfound += q[0]; kept[(size_t)r] = q[1]; mx = std::max(mx, q[1]);
        }
        mine.resize((size_t)mx); // padded payloads
        std::vector<unsigned char> pay;
        if (ok && !allgather_bytes(xfn, X, world, mine.data(), (size_t)mx * sizeof(pnr_seed), pay)) ok = false;
        for (int r = 0; r < world && ok; r++) { // rank order = the z-major order of the unsharded extraction
            const pnr_seed *q = (const pnr_seed *)(pay.data() + (size_t)r * (size_t)mx * sizeof(pnr_seed));
            seeds.insert(seeds.end(), q, q + kept[(size_t)r]);
        }
        if (ok) {
            printf("seed extraction... %gk seeds,  %g sec.\n", nfound / 1000.0, secs(t1, t2));
            nseeds = (int64_t)seeds.size();
            if (nseeds) ok = pnr_sort_seeds(ctx, seeds.data(), nseeds, &nseeds) == PNR_OK; // the list one GPU would have
            seeds.resize((size_t)nseeds);
        }
    }
    auto t3 = clk::now();
    int64_t nn = 0, nl = 0, used = 0, iters = 0;
    if (ok) {
        printf("seed selection & sorting... %gk seeds, %g sec.\ntracing...\n", nseeds / 1000.0, secs(t2, t3));
        // :2658-2710: the particle filters on the GPU (a window of traces refilled as they stop), the bookkeeping replayed on the
        // host in seed order; the graph stays in the context and is fetched once its size is known
        if (settings().verbose) pnr_set_option(ctx, "trace_log", 1);
        if (settings().timing) { pnr_set_option(ctx, "trace_timing", 1); pnr_set_option(ctx, "recon_timing", 1); }
        if (!sharded)
            ok = pnr_trace_replay(ctx, seeds.data(), nseeds, 0, nullptr, 0, &nn, nullptr, 0, &nl, &used, &iters) == PNR_OK;
        else // every rank traces seeds rank, rank + world, ...; finished traces are exchanged and replayed in seed order on every rank
            ok = pnr_trace_replay_sharded(ctx, seeds.data(), nseeds, rank, world, xfn, X, nullptr, 0, &nn, nullptr, 0, &nl, &used, &iters) == PNR_OK;
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
    if (rank != 0) { // every rank holds the same graph; rank 0 post-processes and writes it
        pnr_destroy(ctx);
        if (result) *result = R;
        return true;
    }
    R.t_frangi = secs(t0, t1); R.t_seeds = secs(t1, t2); R.t_select = secs(t2, t3); R.t_trace = secs(t3, t4);
    R.t_setup = secs(t_begin, t0); // context, upload of the stack, soma path
    printf("\n-----\n%g%% seeds used \n", nseeds ? 100.0 * used / nseeds : 0.0);
    { // reconstruct(n0, ...) :2729 -> :2096-2181 (host): refinement, grouping, trees, final resampling
        int64_t cap = std::max<int64_t>(16, 4 * nn), nt = 0;
        for (;;) {
            R.tree.resize((size_t)cap);
            R.parent.resize((size_t)cap);
            if (pnr_reconstruct(R.nodes.data(), nn, R.links.data(), nl, 0, 0, 0, 0, 0, settings().single_tree ? -1 : 0, R.tree.data(), R.parent.data(), cap, &nt) != PNR_OK) {
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
    R.swc_path = inimg_file + (settings().single_tree ? "_Advantra1.swc" : "_Advantra.swc"); // :2152 / :2164
    save_treelist(R.tree, R.parent, R.swc_path, -1, 1.f, "Advantra", swc_comment(paras, p));
    if (settings().save_midres) { // the saveMidres taps of reconstruct() (:2098-2141)
        save_nodelist(R.nodes, R.links, inimg_file + "_n0_.swc");
        static const char *const names[] = {"", "_n0res_.swc", "_n1_.swc", "_n2_.swc", "_n2tree_.swc"};
        for (int stage = 1; stage <= 4; stage++) {
            int64_t sn = 0, sl = 0;
            if (pnr_reconstruct_stage(R.nodes.data(), nn, R.links.data(), nl, 0, 0, 0, 0, 0, stage, nullptr, 0, &sn, nullptr, 0, &sl) != PNR_OK) break;
            std::vector<pnr_node> tn((size_t)sn);
            std::vector<int32_t> tl((size_t)(2 * sl));
            if (pnr_reconstruct_stage(R.nodes.data(), nn, R.links.data(), nl, 0, 0, 0, 0, 0, stage, tn.data(), sn, &sn, tl.data(), sl, &sl) != PNR_OK) break;
            if (stage < 4) save_nodelist(tn, tl, inimg_file + names[stage]);
            else { // a tree list: every node carries its parent (or none)
                std::vector<int32_t> par((size_t)sn, -1);
                for (int64_t k = 0; k < sl; k++) par[(size_t)tl[(size_t)(2 * k)]] = tl[(size_t)(2 * k + 1)];
                save_treelist(tn, par, inimg_file + names[stage], -1, 1.f, "", "");
            }
        }
    }
    R.t_write = secs(t5, clk::now());
    printf("%s\n%lld trace nodes, %lld traces, %lld SMC iterations, %zu tree nodes | frangi %.3f s, seeds %.3f s, selection %.3f s, "
           "tracing %.3f s, reconstruct %.3f s\n",
           R.swc_path.c_str(), (long long)nn - 1, (long long)used, (long long)iters, R.tree.size() - 1, R.t_frangi, R.t_seeds, R.t_select,
           R.t_trace, R.t_recon);
    pnr_destroy(ctx);
    if (result) *result = R;
    return true;
}

} // namespace advantra
