#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (uses the oracle / the compiled reference, like everything under tests/).  Soak: random short-read batches (1 .. 504 bp) through every layout of the lane-per-read kernels -- fixed stride
(aligned with an odd / even number of 8-byte units, packed back to back, with and without per-read lengths), ragged
offsets (one length / mixed), segmented (cuts in read order and in slot order) -- against the oracle; every encoding,
thresholds 0 .. 41, -l, -x, -n, chars out of range.  usage: soak_tiles.py [iterations] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_bind as ob
from fastq_util import segment_by_length
from sickle_amd import capi


def run(iters=50, seed=1, verbose=True):
    rng = np.random.default_rng(seed)
    ctx = capi.Context(0, 2)
    t0 = time.time()
    checked = 0
    for it in range(iters):
        qt = ["sanger", "solexa", "illumina"][it % 3]
        lo, hi = {"sanger": (33, 126), "solexa": (59, 112), "illumina": (64, 110)}[qt]
        n = int(rng.choice([1, 7, 63, 64, 65, 200, 1000, 5000]))
        thr = int(rng.choice([0, 2, 15, 20, 25, 30, 41]))
        uniform = it % 2 == 0
        if uniform:
            L = int(rng.choice([1, 9, 10, 19, 20, 36, 50, 65, 72, 75, 100, 101, 125, 150, 151, 152, 160, 161, 200, 250, 251, 301, 340, 341, 400, 504]))
            lens = np.full(n, L, dtype=np.int64)
        else:
            top = int(rng.choice([40, 160, 301, 504]))
            lens = rng.integers(1, top + 1, size=n).astype(np.int64)
        offs = np.zeros(n + 1, dtype=np.uint64)
        offs[1:] = np.cumsum(lens)
        tot = int(offs[-1])
        mid = min(hi - 3, max(lo + 3, lo + thr + int(rng.integers(-4, 12))))
        mode = (it // 2) % 5
        if mode == 0:
            qual = np.clip(rng.normal(mid, 6, tot).astype(int), lo, hi)
        elif mode == 1:
            qual = np.clip(mid + rng.integers(-2, 3, size=tot), lo, hi)
        elif mode == 2:
            level = np.repeat(rng.integers(lo, hi, size=tot // 40 + 2), 40)[:tot]
            qual = np.clip(level + rng.integers(-3, 4, size=tot), lo, hi)
        elif mode == 3:
            qual = np.where(rng.random(tot) < 0.5, lo, hi)
        else:  # good with a bad start or a bad end, read by read
            qual = np.clip(rng.normal(mid + 8, 4, tot).astype(int), lo, hi)
            for i in range(min(n, 300)):
                a, b = int(offs[i]), int(offs[i + 1])
                c = a + int(rng.integers(0, b - a))
                if i % 2:
                    qual[c:b] = lo + 2
                else:
                    qual[a:c] = lo + 2
        qual = qual.astype(np.uint8)
        if it % 5 == 4:
            qual[int(rng.integers(0, tot))] = int(rng.choice([lo - 1, hi + 1, 200, 10]))
        seq = rng.choice(np.frombuffer(b"ACGT" * 300 + b"Nn", dtype=np.uint8), size=tot)
        l = int(rng.choice([0, 20, 100]))
        x, tn = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        p, po = capi.make_params(qt, thr, l, x, tn), ob.make_params(qt, thr, l, x, tn)
        want, err = ob.oracle_trim_batch(po, qual, seq, offsets=offs, threads=4)

        def padded(stride):
            qs = np.full((n, stride), lo, dtype=np.uint8)
            ss = np.full((n, stride), 65, dtype=np.uint8)
            for i in range(n):
                a, b = int(offs[i]), int(offs[i + 1])
                qs[i, :b - a] = qual[a:b]
                ss[i, :b - a] = seq[a:b]
            return qs.reshape(-1), ss.reshape(-1)

        runs = [("ragged", lambda: ctx.trim_batch(p, qual, seq, offsets=offs))]
        lmax = int(lens.max())
        odd = ((lmax + 7) // 8 | 1) * 8
        even = odd + 8
        for name, stride in (("stride_odd", odd), ("stride_even", even), ("stride_packed", lmax), ("stride_unaligned", lmax + 3)):
            if n * stride > 8_000_000:
                continue
            qs, ss = padded(stride)
            if uniform:
                runs.append((name, lambda qs=qs, ss=ss, stride=stride: ctx.trim_batch(p, qs, ss, stride=stride, read_len=lmax, n_reads=n)))
            else:
                ln = lens.astype(np.uint32)
                runs.append((name + "+lengths", lambda qs=qs, ss=ss, stride=stride, ln=ln: ctx.trim_batch(p, qs, ss, stride=stride, lengths=ln)))
        if lmax <= 504:
            sseq, squal, tiles, order, max_stride = segment_by_length(seq, qual, offs)
            runs.append(("segmented", lambda: ctx.trim_segmented(p, squal, tiles, order, max_stride, seq=sseq)))
            inv = order.astype(np.int64)

            def slot_run():
                g = ctx.trim_segmented(p, squal, tiles, order, max_stride, seq=sseq, slot_order=True)
                back = np.empty_like(g)
                back[inv] = g
                return back
            runs.append(("segmented_slot_order", slot_run))
        for name, fn in runs:
            try:
                got = fn()
                assert err is None, ("device missed the error", it, name, err)
                bad = np.nonzero((got != want).any(axis=1))[0]
                if bad.size:
                    b0 = int(bad[0])
                    raise AssertionError("tile kernels differ from the oracle: %r" % ((it, name, qt, thr, l, x, tn, n, bad[:5], got[bad[:5]], want[bad[:5]], lens[bad[:5]],
                                                                                     qual[int(offs[b0]):int(offs[b0 + 1])].tobytes().hex()),))
            except capi.RangeError as e:
                assert err is not None and (e.read, e.pos, e.ch) == tuple(err), (it, name, err, (e.read, e.pos, e.ch))
            checked += 1
        if verbose and it % 20 == 19:
            print("iteration %d, %d comparisons, %.0f s" % (it + 1, checked, time.time() - t0), flush=True)
    if verbose:
        print("soak ok: %d iterations, %d comparisons, seed %d" % (iters, checked, seed))
    return checked


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 50, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
