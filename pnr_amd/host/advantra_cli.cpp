// advantra_cli -- head-less driver with the reference's command-line contract:
//   vaa3d -x Advantra -f advantra_func -i <inimg_file> -p <11 parameters>      (README.md:15-18)
// becomes
//   advantra_cli [-v] [--save-midres] [--rng-seed N] [-g device] [-d w,h,l for .raw] -f advantra_func -i <inimg_file> -p <11 parameters>
// (-p takes the rest of the line, as in vaa3d: put the driver's own flags before it)
// Exit code: 0 = dofunc returned true, 1 = dofunc returned false (usage error).
#include "advantra_host.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>

int main(int argc, char **argv)
{
    std::vector<char *> infiles, paras;
    std::string func = "advantra_func", raw_dims;
    int device = 0;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "-x") && i + 1 < argc) { i++; continue; } // plugin name: ignored
        if (!strcmp(argv[i], "-f") && i + 1 < argc) { func = argv[++i]; continue; }
        if (!strcmp(argv[i], "-g") && i + 1 < argc) { device = atoi(argv[++i]); continue; }
        if (!strcmp(argv[i], "-v")) { advantra::settings().verbose = true; continue; }
        if (!strcmp(argv[i], "--save-midres")) { advantra::settings().save_midres = true; continue; }
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
    return advantra::advantra_func(infiles, paras, device, raw_dims) ? 0 : 1;
}
