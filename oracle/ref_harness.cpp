/*
 * ref_harness.cpp -- exposes the REFERENCE's own Abstract_Trimmer::sliding_window
 * (compiled from /root/reference/src where it lies, never copied) behind a C ABI,
 * so that oracle/sk_oracle.c can be validated against it and golden vectors can be
 * generated from it (tests/golden/make_golden.py).
 *
 * TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/ (git-ignored)
 * and only when /root/reference is present (the build container).  The .so travels to
 * the GPU box, the reference sources do not.
 *
 * sliding_window/get_quality_num are protected members (reference src/trim.h:14-15);
 * a subclass is the only way to call them without touching the reference.
 */
#include <cstdint>
#include <cstring>
#include <string_view>
#include <thread>
#include <vector>

#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#include "trim.h" /* -I/root/reference/src */

namespace {

class RefTrimmer : public Abstract_Trimmer {
public:
    explicit RefTrimmer(const int32_t prm[5])
    {
        qualtype = prm[0];
        qual_threshold = prm[1];
        length_threshold = prm[2];
        no_fiveprime = prm[3];
        trunc_n = prm[4];
        debug = 0;
    }
    int parse_args(int, char **) override { return 0; }
    int trim_main() override { return 0; }
    void usage(int, char const *) override {}

    void scan(const char *name, size_t name_len, const uint8_t *seq, const uint8_t *qual, int len,
              int32_t out[2])
    {
        FQEntry e;
        e.name = std::string_view(name, name_len);
        e.seq = std::string_view(reinterpret_cast<const char *>(seq), (size_t)len);
        e.qual = std::string_view(reinterpret_cast<const char *>(qual), (size_t)len);
        e.comment = "+";
        cutsites *c = sliding_window(e);
        out[0] = c->five_prime_cut;
        out[1] = c->three_prime_cut;
        free(c); /* the caller frees, as reference src/trim_single.cpp:401 */
    }
};

} // namespace

extern "C" {

/* Direct call.  On an out-of-range quality the reference prints to stderr and exit(1)s
 * the whole process -- use ref_sliding_window_forked where that may happen. */
void ref_sliding_window(const int32_t prm[5], const char *name, const uint8_t *seq,
                        const uint8_t *qual, int len, int32_t out[2])
{
    RefTrimmer t(prm);
    t.scan(name, strlen(name), seq, qual, len, out);
}

/* Same in a forked child.  Returns 0 and fills out, or the child's exit status
 * (1 for the range error) with its stderr text in errbuf. */
int ref_sliding_window_forked(const int32_t prm[5], const char *name, const uint8_t *seq,
                              const uint8_t *qual, int len, int32_t out[2], char *errbuf,
                              int errbuf_len)
{
    int res_pipe[2], err_pipe[2];
    if (pipe(res_pipe) || pipe(err_pipe)) return -1;
    pid_t pid = fork();
    if (pid < 0) return -1;
    if (pid == 0) {
        close(res_pipe[0]);
        close(err_pipe[0]);
        dup2(err_pipe[1], 2);
        int32_t o[2];
        RefTrimmer t(prm);
        t.scan(name, strlen(name), seq, qual, len, o);
        ssize_t w = write(res_pipe[1], o, sizeof o);
        _exit(w == (ssize_t)sizeof o ? 0 : 3);
    }
    close(res_pipe[1]);
    close(err_pipe[1]);
    int got = 0;
    if (errbuf && errbuf_len > 0) {
        ssize_t n;
        while (got < errbuf_len - 1 && (n = read(err_pipe[0], errbuf + got, (size_t)(errbuf_len - 1 - got))) > 0)
            got += (int)n;
        errbuf[got] = '\0';
    }
    int32_t o[2] = {0, 0};
    ssize_t n = read(res_pipe[0], o, sizeof o);
    close(res_pipe[0]);
    close(err_pipe[0]);
    int status = 0;
    waitpid(pid, &status, 0);
    if (WIFEXITED(status) && WEXITSTATUS(status) == 0 && n == (ssize_t)sizeof o) {
        out[0] = o[0];
        out[1] = o[1];
        return 0;
    }
    return WIFEXITED(status) ? WEXITSTATUS(status) : -2;
}

/* A fixed-stride batch split over `threads` std::threads in contiguous ranges: the
 * fork-join of reference src/trim_single.cpp:323-333 around the reference's own scan.
 * Used to generate golden cut vectors and as bench.py's cpu_baseline kind "reference".
 * Inputs must be range-clean (an error exit(1)s the process, as in the reference). */
void ref_trim_batch(const int32_t prm[5], const uint8_t *qual, const uint8_t *seq,
                    const uint64_t *offsets, uint32_t stride, uint32_t read_len,
                    const uint32_t *lengths, uint64_t n_reads, int32_t *out, int threads)
{
    if (threads < 1) threads = 1;
    auto work = [&](uint64_t begin, uint64_t end) {
        RefTrimmer t(prm);
        for (uint64_t r = begin; r < end; r++) {
            uint64_t off = offsets ? offsets[r] : r * (uint64_t)stride;
            int len = offsets ? (int)(offsets[r + 1] - offsets[r]) : (int)(lengths ? lengths[r] : read_len);
            t.scan("@r", 2, (seq ? seq : qual) + off, qual + off, len, out + 2 * r);
        }
    };
    std::vector<std::thread> pool;
    for (int k = 0; k < threads; k++)
        pool.emplace_back(work, n_reads * (uint64_t)k / (uint64_t)threads,
                          n_reads * (uint64_t)(k + 1) / (uint64_t)threads);
    for (auto &th : pool) th.join();
}

} /* extern "C" */
