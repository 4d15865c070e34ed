// sickle.h -- program-wide constants of the drop-in CLI.
// Role of reference src/sickle.h (names kept so that code written against the reference's
// headers still reads naturally); quality tables live behind the C ABI (sk_quality_constants).
#ifndef SICKLE_HOST_H
#define SICKLE_HOST_H

#include <string>

#ifndef PROGRAM_NAME
#define PROGRAM_NAME "sickle"
#endif
#ifndef AUTHORS
#define AUTHORS "Nikhil Joshi, UC Davis Bioinformatics Core\n"
#endif
#ifndef VERSION
#define VERSION 1.33 /* reference Makefile:2 */
#endif
#ifndef DEFAULT_BATCH_LEN
#define DEFAULT_BATCH_LEN 512 /* MiB, reference src/sickle.h:27-29 */
#endif

// quality_type, reference src/sickle.h:61-66 (values == SK_PHRED.. of the C ABI)
typedef enum { PHRED, SANGER, SOLEXA, ILLUMINA } quality_type;

// == reference cutsites (src/sickle.h:93-96) == sk_cut of the C ABI
typedef struct __cutsites_ {
    int five_prime_cut;
    int three_prime_cut;
} cutsites;

// "[ERROR] ..." on stderr, reference src/sickle.h:113-120
void error(const char *content);
void error(const std::string &content);
// The reference's always-on "[DEBUGGING] ..." stdout chatter (src/sickle.h:102-111) is not
// reproduced unless SICKLE_DEBUG_CHATTER=1 is set in the environment.
void msg(const char *content);
void msg(const std::string &content);

// How the pipeline leaves on a fatal error (malformed record, range error, device failure).  These
// fire on the reader, pool, flusher or main thread while other threads are inside HIP calls; exit()
// would run the static destructors -- HIP's among them -- under live GPU work.  So: flush the two
// stdio streams (the output files are raw descriptors, nothing of them is buffered here) and leave
// without unwinding, like the reference's exit(1) leaves its detached threads behind.
[[noreturn]] void fatal_exit(int status);

// Set by main(): the process ends right after trim_main().  The trimmers then close their OUTPUT
// files and leave the rest -- unmapping gigabytes of input, unpinning the staging buffers, destroying
// the HIP streams and contexts one call at a time -- to the kernel's process teardown, which does it
// in bulk (measured: 0.19 s of a 1.1 s run were spent after the last batch had been written).  A
// caller that embeds Trim_Single / Trim_Paired leaves it false and gets everything released.
extern bool sickle_leave_fast;

// The front process.  What the kernel does with a process that mapped gigabytes of input, pinned staging buffers
// and held a HIP context takes 0.2-0.3 s after the last byte of output has been written and every output file
// closed -- a fifth of a 20 M-read run, and nobody needs to wait for it.  So `sickle se|pe` forks before anything is
// allocated or the GPU runtime touched: the child does the whole run; fatal_exit() in the child reports the exit
// status through a pipe once the outputs are closed and stdout / stderr flushed and closed; the parent, which holds
// nothing, leaves at once with that status, and the child's address space is torn down behind it.  A child that
// ends any other way (a signal, a plain exit) is waited for and its status passed on; a parent that is killed
// takes the child with it (PR_SET_PDEATHSIG).  SICKLE_NO_FRONT=1 runs everything in the one process, as before.
// The worker ends as an orphan and is reaped by the system's init (or a sub-reaper), like any process that outlives
// its parent; measured on the GPU boxes and in the build container: none left a few seconds after bursts of runs.
// (What `time` and getrusage(RUSAGE_CHILDREN) show for the front process is its own, next to nothing: the worker is
// not waited for.  SICKLE_STAGE_TIMES=1 prints the worker's own CPU time.)
// Returns in the process that is to do the work.
void sickle_front_process();
void sickle_wallclock_mark(const char *what);
extern int sickle_done_fd;

#endif
