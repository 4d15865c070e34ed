// sickle_main.cpp -- `sickle {se,pe,--help,--version}`: the dispatch of reference src/sickle.cpp:40-86.
#include <malloc.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "sickle.h"
#include "trim_paired.h"
#include "trim_single.h"

static void main_usage(int status)
{
    // text of reference src/sickle.cpp:26-38 (on stdout there too)
    fprintf(stdout, "\nUsage: %s <command> [options]\n\
\n\
Command:\n\
pe\tpaired-end sequence trimming\n\
se\tsingle-end sequence trimming\n\
\n\
--help, display this help and exit\n\
--version, output version information and exit\n\n", PROGRAM_NAME);
    exit(status);
}

int main(int argc, char *argv[])
{
    // Batches are tens to hundreds of megabytes and are allocated and freed once per batch.
    // Keep such blocks inside the heap instead of mmap/munmap-ing (and page-faulting) them
    // every time: measured, this alone removes most of the system time of a run.
    sickle_wallclock_mark("main");
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_TOP_PAD, 64 << 20);
    int retval = 0;
    if (argc < 2 || (strcmp(argv[1], "pe") != 0 && strcmp(argv[1], "se") != 0 &&
                     strcmp(argv[1], "--version") != 0 && strcmp(argv[1], "--help") != 0)) {
        main_usage(EXIT_FAILURE);
    }
    if (strcmp(argv[1], "--version") == 0) {
        // reference src/sickle.cpp:52-54: the line continuations inside that string literal put
        // two tabs and one tab into the text; kept, since this is what `sickle --version` prints
        fprintf(stdout, "%s version %0.2f\nCopyright (c) 2011 The Regents of University of California, \
		Davis Campus.\n%s is free software and comes with ABSOLUTELY NO WARRANTY.\nDistributed under the\
		 MIT License.\n\nWritten by %s\n", PROGRAM_NAME, VERSION, PROGRAM_NAME, AUTHORS);
        exit(EXIT_SUCCESS);
    } else if (strcmp(argv[1], "--help") == 0) {
        main_usage(EXIT_SUCCESS);
    } else if (strcmp(argv[1], "pe") == 0) {
        sickle_front_process();
        sickle_leave_fast = true;
        static Trim_Paired trimmer; // static: not destroyed on the way out (fatal_exit does not unwind)
        retval = trimmer.parse_args(argc, argv);
        if (retval != 0) return retval;
        retval = trimmer.trim_main();
    } else {
        sickle_front_process();
        sickle_leave_fast = true;
        static Trim_Single trimmer;
        retval = trimmer.parse_args(argc, argv);
        if (retval != 0) return retval;
        retval = trimmer.trim_main();
    }
    fatal_exit(retval); // outputs are closed; everything else goes with the process (sickle.h)
}
