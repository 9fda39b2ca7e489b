// advantra_cli -- head-less driver with the reference's command-line contract:
//   vaa3d -x Advantra -f advantra_func -i <inimg_file> -p <11 parameters>      (README.md:15-18)
// becomes
//   advantra_cli [-v] [--save-midres] [--rng-seed N] [-g device] [-d w,h,l for .raw] -f advantra_func -i <inimg_file> -p <11 parameters>
// (-p takes the rest of the line, as in vaa3d: put the driver's own flags before it)
//   --ranks N [--share-gpu]: N processes of this host, one GPU each (device = -g + rank; --share-gpu: all on -g, for rehearsals),
//   reconstruct the ONE stack together; the processes are forked before anything touches a GPU and joined through shared memory.
// Exit code: 0 = dofunc returned true, 1 = dofunc returned false (usage error).
#include "advantra_host.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

int main(int argc, char **argv)
{
    std::vector<char *> infiles, paras;
    std::string func = "advantra_func", raw_dims;
    int device = 0, ranks = 1;
    bool share_gpu = false;
    std::string transport = "shm";
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--ranks") && i + 1 < argc) { ranks = atoi(argv[++i]); continue; }
        if (!strcmp(argv[i], "--share-gpu")) { share_gpu = true; continue; }
        if (!strcmp(argv[i], "--exchange") && i + 1 < argc) { transport = argv[++i]; continue; }
        if (!strcmp(argv[i], "-x") && i + 1 < argc) { i++; continue; } // plugin name: ignored
        if (!strcmp(argv[i], "-f") && i + 1 < argc) { func = argv[++i]; continue; }
        if (!strcmp(argv[i], "-g") && i + 1 < argc) { device = atoi(argv[++i]); continue; }
        if (!strcmp(argv[i], "-v")) { advantra::settings().verbose = true; continue; }
        if (!strcmp(argv[i], "--timing")) { advantra::settings().timing = true; continue; }
        if (!strcmp(argv[i], "--save-midres")) { advantra::settings().save_midres = true; continue; }
        if (!strcmp(argv[i], "--single-tree")) { advantra::settings().single_tree = true; continue; }
        if (!strcmp(argv[i], "--rng-seed") && i + 1 < argc) { advantra::settings().rng_seed = (uint32_t)strtoul(argv[++i], nullptr, 10); continue; }
        if (!strcmp(argv[i], "-d") && i + 1 < argc) { raw_dims = argv[++i]; continue; }
        if (!strcmp(argv[i], "-i")) { while (i + 1 < argc && argv[i + 1][0] != '-') infiles.push_back(argv[++i]); continue; }
        if (!strcmp(argv[i], "-p")) { while (i + 1 < argc) paras.push_back(argv[++i]); continue; }
    }
    if (func == "help") { // funclist(): advantra_func, help (Advantra_plugin.cpp:157-162)
        advantra::print_help();
        return 0;
    }
    if (func != "advantra_func") return 1; // dofunc: unknown function -> false
    if (transport != "shm" && transport != "rccl") { fprintf(stderr, "--exchange: shm or rccl\n"); return 1; }
    const bool rccl = transport == "rccl";
    if (rccl && share_gpu && ranks > 1) { fprintf(stderr, "--exchange rccl needs one GPU per rank (RCCL refuses two ranks on one device)\n"); return 1; }
    // (--ranks 1 --exchange rccl: the sharded code path with its RCCL collectives on a world of one -- what a one-GPU box can rehearse)
    if (ranks <= 1 && !rccl) return advantra::advantra_func(infiles, paras, device, raw_dims) ? 0 : 1;
    if (ranks < 1) ranks = 1;
    if (ranks > 64) { fprintf(stderr, "--ranks: at most 64\n"); return 1; }
    // one process per GPU, forked here -- nothing has touched a GPU yet -- and joined through a shared-memory segment
    // unique to this job, not only to this pid (a recycled pid must never meet the segment of a crashed earlier job)
    struct timespec now;
    clock_gettime(CLOCK_REALTIME, &now);
    const std::string name = "pnr_cli_" + std::to_string((long long)getpid()) + "_" + std::to_string((long long)now.tv_sec) + "_" + std::to_string((long long)now.tv_nsec);
    std::vector<pid_t> kids;
    for (int r = 0; r < ranks; r++) {
        const pid_t pid = fork();
        if (pid < 0) { perror("fork"); return 1; }
        if (pid == 0) {
            advantra::Settings &S = advantra::settings();
            S.rank = r; S.world = ranks;
            if (pnr_shm_exchange_open(name.c_str(), r, ranks, 1 << 18, &S.exchange) != PNR_OK) {
                fprintf(stderr, "rank %d: %s\n", r, pnr_last_error());
                _exit(1);
            }
            const int dev = share_gpu ? device : device + r;
            if (rccl) { // the ncclUniqueId of rank 0 reaches the others through the shared-memory segment; then every rank joins with its GPU
                unsigned char id[128] = {0};
                std::vector<unsigned char> all((size_t)128 * ranks);
                int okid = (r != 0 || pnr_rccl_unique_id(id) == PNR_OK) ? 1 : 0;
                if (!okid) fprintf(stderr, "rank 0: %s\n", pnr_last_error());
                if (pnr_shm_allgather(S.exchange, id, all.data(), 128) != PNR_OK) { fprintf(stderr, "rank %d: %s\n", r, pnr_last_error()); _exit(1); }
                bool zero = true; // (rank 0 failed to make an id: everybody leaves)
                for (int b = 0; b < 128; b++) zero = zero && all[(size_t)b] == 0;
                if (zero || pnr_rccl_exchange_open(all.data(), r, ranks, dev, 1 << 18, &S.rccl) != PNR_OK) {
                    if (!zero) fprintf(stderr, "rank %d: %s\n", r, pnr_last_error());
                    _exit(1);
                }
                S.force_shard = true;
            }
            const bool okr = advantra::advantra_func(infiles, paras, dev, raw_dims);
            pnr_rccl_exchange_close(S.rccl);
            pnr_shm_exchange_close(S.exchange);
            fflush(stdout); fflush(stderr);
            _exit(okr ? 0 : 1);
        }
        kids.push_back(pid);
    }
    int worst = 0;
    for (pid_t k : kids) {
        int st = 0;
        if (waitpid(k, &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) worst = 1;
    }
    return worst;
}
