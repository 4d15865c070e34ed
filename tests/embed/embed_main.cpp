// embed_main.cpp -- TEST ONLY: Trim_Single / Trim_Paired used as a LIBRARY, the way the reference's own main uses
// them (reference src/sickle.cpp:61-80: a trimmer on the stack, parse_args, trim_main, return) -- several runs in ONE
// process, no front process, sickle_leave_fast left false, so that close_streams() releases the readers,
// close_device() the pinned staging and sk_destroy() the device side between the runs, and the destructor runs at
// the end of each scope.  After every run one line goes to stderr:
//   [embed] run K rc R device_free_bytes F rss_kb S
// (device_free_bytes from hipMemGetInfo in the GPU build; -1 in the host build against the oracle-backed shim).
//
// usage: embed_main pe <args...> -- se <args...> -- pe <args...>
#include <getopt.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "sickle.h"
#include "trim_paired.h"
#include "trim_single.h"

#ifdef EMBED_HIP
extern "C" int hipMemGetInfo(size_t *free_bytes, size_t *total_bytes); // libamdhip64 (hipError_t is an int-sized enum, 0 = success)
static long long device_free()
{
    size_t fr = 0, total = 0;
    if (hipMemGetInfo(&fr, &total) != 0) return -2;
    return (long long)fr;
}
#else
static long long device_free() { return -1; }
#endif

static long rss_kb()
{
    long pages = 0, rss = 0;
    FILE *f = fopen("/proc/self/statm", "r");
    if (!f) return -1;
    if (fscanf(f, "%ld %ld", &pages, &rss) != 2) rss = -1;
    fclose(f);
    return rss * 4;
}

template <typename Trimmer> static int one_run(int argc, char **argv)
{
    Trimmer trimmer; // on the stack, like the reference's main
    int retval = trimmer.parse_args(argc, argv);
    if (retval != 0) return retval;
    return trimmer.trim_main();
}

int main(int argc, char *argv[])
{
    std::vector<std::vector<char *>> runs(1);
    for (int i = 1; i < argc; ++i) {
        if (strcmp(argv[i], "--") == 0) runs.emplace_back();
        else runs.back().push_back(argv[i]);
    }
    fprintf(stderr, "[embed] start device_free_bytes %lld rss_kb %ld\n", device_free(), rss_kb());
    int k = 0, worst = 0;
    for (std::vector<char *> &r : runs) {
        if (r.empty()) continue;
        std::vector<char *> av;
        av.push_back(argv[0]);
        for (char *a : r) av.push_back(a);
        av.push_back(nullptr);
        optind = 0; // getopt_long keeps its cursor in globals: an embedder that parses twice resets it (glibc: 0 = re-initialise)
        const int ac = (int)av.size() - 1;
        const int rc = strcmp(r[0], "pe") == 0 ? one_run<Trim_Paired>(ac, av.data()) : one_run<Trim_Single>(ac, av.data());
        fflush(stdout);
        fprintf(stderr, "[embed] run %d rc %d device_free_bytes %lld rss_kb %ld\n", k++, rc, device_free(), rss_kb());
        if (rc > worst) worst = rc;
    }
    return worst;
}
