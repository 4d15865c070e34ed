/*
 * sk_oracle.c -- CPU restatement of the reference's sliding-window scan.
 *
 * TEST INFRASTRUCTURE ONLY (see sk_oracle.h).  Parity status: PINNED against the
 * compiled reference (oracle/_ref) and tests/golden/.
 *
 * The control flow deliberately mirrors the reference statement by statement
 * (including its floating-point window average and its N-handling bug) instead of
 * the closed forms the HIP kernel uses, so that the two are independent
 * derivations of the same behaviour.
 */
#include "sk_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* reference src/sickle.h:85-91 */
const int sko_quality_constants[4][3] = {
    {0, 4, 60},    /* PHRED (not reachable from the CLI) */
    {33, 33, 126}, /* SANGER */
    {64, 58, 112}, /* SOLEXA */
    {64, 64, 110}, /* ILLUMINA */
};

/* reference src/sickle.h:68-73 */
const char sko_typenames[4][10] = {{"Phred"}, {"Sanger"}, {"Solexa"}, {"Illumina"}};

enum { Q_OFFSET = 0, Q_MIN = 1, Q_MAX = 2 };

/* Scan state shared by the quality lookups of one read. */
typedef struct {
    const sko_params *p;
    const uint8_t *qual;
    int failed;
    sko_err *err;
} scan_ctx;

/*
 * reference src/trim.cpp:118-140 (get_quality_num).  `char` is signed on the
 * x86-64 build of the reference, so bytes >= 0x80 are negative and always fail
 * the range check.  Instead of exit(1) the failure is latched in the context and
 * the caller unwinds at once.
 */
static int quality_num(scan_ctx *c, int pos)
{
    int qual_value = (int)(signed char)c->qual[pos];
    const int *k = sko_quality_constants[c->p->qualtype];
    if (qual_value < k[Q_MIN] || qual_value > k[Q_MAX]) {
        if (!c->failed) {
            c->failed = 1;
            if (c->err) {
                c->err->pos = (uint32_t)pos;
                c->err->ch = qual_value;
            }
        }
        return 0;
    }
    return qual_value - k[Q_OFFSET];
}

/* first occurrence of byte `what` in seq[0..len), or (size_t)-1: std::string_view::find */
static size_t find_byte(const uint8_t *seq, size_t len, uint8_t what)
{
    const uint8_t *hit = seq ? (const uint8_t *)memchr(seq, what, len) : NULL;
    return hit ? (size_t)(hit - seq) : (size_t)-1;
}

/* reference src/trim.cpp:3-116 (Abstract_Trimmer::sliding_window) */
int sko_sliding_window(const sko_params *p, const uint8_t *seq, const uint8_t *qual,
                       int32_t len, sko_cut *out, sko_err *err)
{
    scan_ctx c = {p, qual, 0, err};
    size_t length = (size_t)len;
    int window_size = (int)(0.1 * (double)length); /* :8 */
    int i, j;
    int window_start = 0;
    int window_total = 0;
    int three_prime_cut = (int)length; /* :13 */
    int five_prime_cut = 0;
    int found_five_prime = 0;
    double window_avg;
    size_t npos = 0;

    /* :21-26 discard if shorter than the length threshold (before any quality is read) */
    if (length < (size_t)p->length_threshold) {
        out->three = -1;
        out->five = -1;
        return 0;
    }
    /* The reference throws std::out_of_range on an empty read that gets here
     * (only with -l 0; FQEntry::validate rejects empty reads first, src/FQEntry.cpp:76).
     * Defined here as "discard". */
    if (length == 0) {
        out->three = -1;
        out->five = -1;
        return 0;
    }

    if (window_size == 0) window_size = (int)length; /* :30 */
    for (i = 0; i < window_size; i++) {              /* :31-33 */
        window_total += quality_num(&c, i);
        if (c.failed) return 1;
    }
    for (i = 0; (size_t)i <= length - (size_t)window_size; i++) { /* :34 */
        window_avg = (double)window_total / (double)window_size;  /* :36 */

        /* :42-56 the 5' cut: first window whose average reaches the threshold */
        if (p->no_fiveprime == 0 && found_five_prime == 0 && window_avg >= p->qual_threshold) {
            for (j = window_start; j < window_start + window_size; j++) {
                int q = quality_num(&c, j);
                if (c.failed) return 1;
                if (q >= p->qual_threshold) {
                    five_prime_cut = j;
                    break;
                }
            }
            found_five_prime = 1;
        }

        /* :61-73 the 3' cut: first later window whose average falls below it */
        if ((window_avg < p->qual_threshold || (size_t)(window_start + window_size) > length) &&
            (found_five_prime == 1 || p->no_fiveprime)) {
            for (j = window_start; j < window_start + window_size; j++) {
                int q = quality_num(&c, j);
                if (c.failed) return 1;
                if (q < p->qual_threshold) {
                    three_prime_cut = j;
                    break;
                }
            }
            break;
        }

        /* :76-80 slide: drop the first quality, add the next */
        window_total -= quality_num(&c, window_start);
        if (c.failed) return 1;
        if ((size_t)(window_start + window_size) < length) {
            window_total += quality_num(&c, window_start + window_size);
            if (c.failed) return 1;
        }
        window_start++;
    }

    /* :86-98 the N rule.  Bug-compatible: in the uppercase branch the reference
     * assigns npos = nIndex (the failed lowercase search, string::npos), so the
     * cut becomes (int)(npos - 1) = -2 and the read is discarded below. */
    {
        size_t nIndex = find_byte(seq, length, 'n');
        size_t NIndex = find_byte(seq, length, 'N');
        int hasN = 0;
        if (nIndex != (size_t)-1) {
            npos = nIndex;
            hasN = 1;
        } else if (NIndex != (size_t)-1) {
            npos = nIndex;
            hasN = 1;
        }
        if (p->trunc_n && hasN) three_prime_cut = (int)(npos - 1);
    }

    /* :103-108 */
    if ((found_five_prime == 0 && !p->no_fiveprime) ||
        (three_prime_cut - five_prime_cut < p->length_threshold)) {
        three_prime_cut = -1;
        five_prime_cut = -1;
    }

    out->three = three_prime_cut; /* :112-115 */
    out->five = five_prime_cut;
    return 0;
}

static void locate(const uint64_t *offsets, uint32_t stride, uint32_t read_len,
                   const uint32_t *lengths, uint64_t r, uint64_t *off, int32_t *len)
{
    if (offsets) {
        *off = offsets[r];
        *len = (int32_t)(offsets[r + 1] - offsets[r]);
    } else {
        *off = r * (uint64_t)stride;
        *len = (int32_t)(lengths ? lengths[r] : read_len);
    }
}

/* what Trim_Single::processing_thread does for one queue, reference
 * src/trim_single.cpp:357-372, over the packed batch layout */
int sko_trim_batch(const sko_params *p, const uint8_t *qual, const uint8_t *seq,
                   const uint64_t *offsets, uint32_t stride, uint32_t read_len,
                   const uint32_t *lengths, uint64_t n_reads, sko_cut *out, sko_err *err)
{
    uint64_t r;
    for (r = 0; r < n_reads; r++) {
        uint64_t off;
        int32_t len;
        sko_err e = {0, 0, 0};
        locate(offsets, stride, read_len, lengths, r, &off, &len);
        if (sko_sliding_window(p, seq ? seq + off : NULL, qual + off, len, &out[r], &e)) {
            if (err) {
                *err = e;
                err->read = (uint32_t)r;
            }
            return 1;
        }
    }
    return 0;
}

typedef struct {
    const sko_params *p;
    const uint8_t *qual, *seq;
    const uint64_t *offsets;
    uint32_t stride, read_len;
    const uint32_t *lengths;
    uint64_t begin, end;
    sko_cut *out;
    sko_err err;
    int failed;
} mt_job;

static void *mt_worker(void *arg)
{
    mt_job *j = (mt_job *)arg;
    uint64_t r;
    for (r = j->begin; r < j->end; r++) {
        uint64_t off;
        int32_t len;
        sko_err e = {0, 0, 0};
        locate(j->offsets, j->stride, j->read_len, j->lengths, r, &off, &len);
        if (sko_sliding_window(j->p, j->seq ? j->seq + off : NULL, j->qual + off, len, &j->out[r], &e)) {
            j->err = e;
            j->err.read = (uint32_t)r;
            j->failed = 1;
            break;
        }
    }
    return NULL;
}

/* the fork-join of reference src/trim_single.cpp:323-333, contiguous ranges */
int sko_trim_batch_mt(const sko_params *p, const uint8_t *qual, const uint8_t *seq,
                      const uint64_t *offsets, uint32_t stride, uint32_t read_len,
                      const uint32_t *lengths, uint64_t n_reads, sko_cut *out, sko_err *err,
                      int threads)
{
    int t, rc = 0;
    pthread_t *tid;
    mt_job *jobs;
    if (threads < 1) threads = 1;
    tid = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    jobs = (mt_job *)calloc((size_t)threads, sizeof(mt_job));
    for (t = 0; t < threads; t++) {
        mt_job *j = &jobs[t];
        j->p = p; j->qual = qual; j->seq = seq; j->offsets = offsets;
        j->stride = stride; j->read_len = read_len; j->lengths = lengths;
        j->begin = n_reads * (uint64_t)t / (uint64_t)threads;
        j->end = n_reads * (uint64_t)(t + 1) / (uint64_t)threads;
        j->out = out;
        pthread_create(&tid[t], NULL, mt_worker, j);
    }
    for (t = 0; t < threads; t++) pthread_join(tid[t], NULL);
    for (t = 0; t < threads; t++) {
        if (jobs[t].failed) {
            if (err) *err = jobs[t].err;
            rc = 1;
            break; /* ranges are in index order: the first failing range holds the lowest read */
        }
    }
    free(jobs);
    free(tid);
    return rc;
}

/* reference src/trim.cpp:130-135 */
int sko_format_error(const sko_params *p, const char *name, size_t name_len,
                     const uint8_t *qual, size_t qual_len, const sko_err *err,
                     char *buf, size_t buf_len)
{
    const int *k = sko_quality_constants[p->qualtype];
    const char *tn = sko_typenames[p->qualtype];
    return snprintf(buf, buf_len,
                    "ERROR: Quality value (%d) does not fall within correct range for %s encoding.\n"
                    "Range for %s encoding: %d-%d\n"
                    "FastQ record: %.*s\n"
                    "Quality string: %.*s\n"
                    "Quality char: '%c'\n"
                    "Quality position: %d\n",
                    err->ch, tn, tn, k[Q_MIN], k[Q_MAX], (int)name_len, name, (int)qual_len,
                    (const char *)qual, (char)err->ch, (int)err->pos + 1);
}
