/*
 * sk_shim.c -- TEST INFRASTRUCTURE ONLY.  Stands in for libsickle_amd.so when the host
 * pipeline (sickle_amd/csrc/host/) is exercised on a machine WITHOUT a GPU: it implements
 * the handful of C-ABI entry points the host code calls on top of the test oracle
 * (oracle/sk_oracle.c).  It exists so that `pytest -m "not gpu"` can check ingest, record
 * framing, batch-cut emulation, pair classification, output assembly, counters and messages
 * against the reference's golden outputs.  It is linked ONLY into tests/cpu_shim/sickle_hostcheck;
 * the product binary (sickle_amd/sickle) links the HIP library and has no CPU path.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sickle_amd.h"
#include "sk_oracle.h"

struct sk_ctx {
    int rc[16];
    sk_err err[16];
    int busy[16];
};

const int32_t *sk_quality_constants(int32_t qualtype)
{
    return (qualtype < 0 || qualtype > 3) ? NULL : (const int32_t *)sko_quality_constants[qualtype];
}
const char *sk_typename(int32_t qualtype) { return (qualtype < 0 || qualtype > 3) ? NULL : sko_typenames[qualtype]; }

int sk_create(int device, int slots, sk_ctx **out)
{
    (void)device;
    (void)slots;
    *out = (sk_ctx *)calloc(1, sizeof(sk_ctx));
    return *out ? SK_OK : SK_EINVAL;
}
void sk_destroy(sk_ctx *ctx) { free(ctx); }
const char *sk_last_error(const sk_ctx *ctx)
{
    (void)ctx;
    return "cpu shim";
}
void *sk_host_alloc(sk_ctx *ctx, size_t bytes)
{
    (void)ctx;
    return malloc(bytes ? bytes : 1);
}
void sk_host_free(sk_ctx *ctx, void *p)
{
    (void)ctx;
    free(p);
}

int sk_submit(sk_ctx *ctx, int slot, const sk_params *p, const sk_batch *b, sk_cut *out)
{
    sko_params op = {p->qualtype, p->qual_threshold, p->length_threshold, p->no_fiveprime, p->trunc_n};
    sko_err e = {0, 0, 0};
    if (ctx->busy[slot]) return SK_EBUSY;
    if (b->tiles) { /* segmented layout: read by read through the oracle, lowest output index wins on error */
        uint32_t t, i, best = 0xffffffffu;
        ctx->rc[slot] = 0;
        for (t = 0; t < b->n_tiles; ++t) {
            const sk_tile *d = &b->tiles[t];
            for (i = 0; i < d->rows; ++i) {
                const uint64_t off = d->byte_off + (uint64_t)i * d->stride;
                const uint32_t dst = b->out_index[d->slot0 + i];
                sko_err e1 = {0, 0, 0};
                /* cuts_in_slot_order: the cut goes to the read's slot; the error still names the caller's read */
                if (sko_sliding_window(&op, b->seq ? b->seq + off : NULL, b->qual + off, d->read_len,
                                       (sko_cut *)&out[b->cuts_in_slot_order ? d->slot0 + i : dst], &e1)) {
                    ctx->rc[slot] = 1;
                    if (dst < best) {
                        best = dst;
                        e = e1;
                        e.read = dst;
                    }
                }
            }
        }
    } else
    ctx->rc[slot] = sko_trim_batch(&op, b->qual, b->seq, b->offsets, b->stride, b->read_len, b->lengths, b->n_reads,
                                   (sko_cut *)out, &e);
    ctx->err[slot].read = e.read;
    ctx->err[slot].pos = e.pos;
    ctx->err[slot].ch = e.ch;
    ctx->busy[slot] = 1;
    return SK_OK;
}

int sk_wait(sk_ctx *ctx, int slot, sk_err *err)
{
    if (!ctx->busy[slot]) return SK_EINVAL;
    ctx->busy[slot] = 0;
    if (ctx->rc[slot]) {
        if (err) *err = ctx->err[slot];
        return SK_ERANGE;
    }
    return SK_OK;
}
