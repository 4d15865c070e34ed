/*
 * sk_oracle.h -- CPU restatement of the reference's per-read quality scan.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (sickle_amd/, the `sickle`
 * binary, libsickle_amd.so) may include, link or call this.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * Parity status: PINNED.  Checked against the reference itself compiled in the
 * build container (oracle/_ref, see oracle/Makefile and oracle/ref_harness.cpp)
 * and against the golden vectors in tests/golden/ that were generated from it
 * (tests/golden/make_golden.py).
 *
 * Follows /root/reference/src/trim.cpp:3-140 and the tables of
 * /root/reference/src/sickle.h:61-96 (citations at each function).
 */
#ifndef SK_ORACLE_H
#define SK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* quality_type, reference src/sickle.h:61-66 */
enum { SKO_PHRED = 0, SKO_SANGER = 1, SKO_SOLEXA = 2, SKO_ILLUMINA = 3 };

/* the config ints of Abstract_Trimmer, reference src/trim.h:16-20 */
typedef struct {
    int32_t qualtype;
    int32_t qual_threshold;
    int32_t length_threshold;
    int32_t no_fiveprime;
    int32_t trunc_n;
} sko_params;

/* == reference `cutsites`, src/sickle.h:93-96 */
typedef struct {
    int32_t five;
    int32_t three;
} sko_cut;

/* what the reference prints before exit(1), src/trim.cpp:129-137 */
typedef struct {
    uint32_t read; /* index of the read inside the batch          */
    uint32_t pos;  /* 0-based position of the offending character */
    int32_t ch;    /* the character as (int)(char), i.e. signed    */
} sko_err;

/* {offset,min,max} per quality type, reference src/sickle.h:85-91 */
extern const int sko_quality_constants[4][3];
extern const char sko_typenames[4][10];

/*
 * One read.  Returns 0 and fills *out, or returns 1 and fills err->pos/err->ch
 * (err->read untouched) where the reference would print its range error and
 * exit(1).  seq may be NULL when p->trunc_n == 0.
 */
int sko_sliding_window(const sko_params *p, const uint8_t *seq, const uint8_t *qual,
                       int32_t len, sko_cut *out, sko_err *err);

/*
 * A batch, in the layout of the C-ABI (include/sickle_amd.h): read r occupies
 * bytes [offsets[r], offsets[r+1]) when offsets != NULL, else
 * [r*stride, r*stride + (lengths ? lengths[r] : read_len)).
 * Stops at the first read (in index order) whose scan errors: returns 1 with *err
 * filled (like the reference at -a 1, which exits on the first one it meets).
 */
int sko_trim_batch(const sko_params *p, const uint8_t *qual, const uint8_t *seq,
                   const uint64_t *offsets, uint32_t stride, uint32_t read_len,
                   const uint32_t *lengths, uint64_t n_reads, sko_cut *out, sko_err *err);

/* Same, split over `threads` pthreads in contiguous ranges (for the CPU baseline).
 * On error reports the lowest erroring read index. */
int sko_trim_batch_mt(const sko_params *p, const uint8_t *qual, const uint8_t *seq,
                      const uint64_t *offsets, uint32_t stride, uint32_t read_len,
                      const uint32_t *lengths, uint64_t n_reads, sko_cut *out, sko_err *err,
                      int threads);

/* The six stderr lines of reference src/trim.cpp:130-135 into buf (NUL-terminated). */
int sko_format_error(const sko_params *p, const char *name, size_t name_len,
                     const uint8_t *qual, size_t qual_len, const sko_err *err,
                     char *buf, size_t buf_len);

#ifdef __cplusplus
}
#endif
#endif
