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

#endif
