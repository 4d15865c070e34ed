#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (uses the oracle / the compiled reference, like everything under tests/).  Soak: random long-read batches through both general kernels (SK_GENERAL=team|stream), host and device entry
points, with and without longest-read hints, against the oracle.  usage: soak_general.py [iterations] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle_bind as ob
from sickle_amd import capi

def run(iters=50, seed=1, verbose=True):
    rng = np.random.default_rng(seed)
    ctx = capi.Context(0, 2)
    t0 = time.time()
    checked = 0
    for it in range(iters):
        qt = ["sanger", "solexa", "illumina"][it % 3]
        lo, hi = {"sanger": (33, 126), "solexa": (59, 112), "illumina": (64, 110)}[qt]
        n = int(rng.integers(1, 400))
        top = float(rng.choice([300, 3000, 12_000, 70_000, 200_000]))
        lens = np.maximum(1, np.exp(rng.uniform(0, np.log(top), size=n))).astype(np.uint32)
        if it % 7 == 0:
            lens[rng.integers(0, n)] = int(rng.choice([1024, 2048, 16_384, 30_720, 65_536]))
        offs = np.zeros(n + 1, dtype=np.uint64)
        offs[1:] = np.cumsum(lens)
        tot = int(offs[-1])
        thr = int(rng.choice([0, 2, 15, 20, 25, 30, 41]))
        mid = min(hi - 3, max(lo + 3, lo + (0 if qt == "sanger" else 0) + thr + int(rng.integers(-4, 12))))
        mode = it % 5
        if mode == 0:
            qual = np.clip(rng.normal(mid, 6, tot).astype(int), lo, hi)
        elif mode == 1:
            qual = np.clip(mid + rng.integers(-2, 3, size=tot), lo, hi)
        elif mode == 2:
            level = np.repeat(rng.integers(lo, hi, size=tot // 700 + 2), 700)[:tot]
            qual = np.clip(level + rng.integers(-3, 4, size=tot), lo, hi)
        elif mode == 3:
            qual = np.where(rng.random(tot) < 0.5, lo, hi)
        else:
            qual = np.clip(rng.normal(mid + 8, 4, tot).astype(int), lo, hi)
            for i in range(n):
                a, b = int(offs[i]), int(offs[i + 1])
                c = a + int(rng.integers(0, b - a))
                qual[c:b] = np.clip(rng.normal(lo + 5, 3, b - c).astype(int), lo, hi)
        qual = qual.astype(np.uint8)
        if it % 4 == 3:  # a char out of range somewhere
            qual[int(rng.integers(0, tot))] = int(rng.choice([lo - 1, hi + 1, 200, 10]))
        seq = rng.choice(np.frombuffer(b"ACGT" * 2000 + b"Nn", dtype=np.uint8), size=tot)
        l = int(rng.choice([0, 20, 300, 5000]))
        x, tn = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        p, po = capi.make_params(qt, thr, l, x, tn), ob.make_params(qt, thr, l, x, tn)
        want, err = ob.oracle_trim_batch(po, qual, seq, offsets=offs, threads=8)
        dq, ds, do = torch.from_numpy(qual).cuda(), torch.from_numpy(seq).cuda(), torch.from_numpy(offs.view(np.int64)).cuda()
        out = torch.empty((n, 2), dtype=torch.int32, device="cuda")
        for which in ("band", "team", "stream"):
            os.environ["SK_GENERAL"] = which
            runs = [("submit", None)] + [("device", h) for h in (0, int(lens.max()), 2000)]
            for kind, hint in runs:
                try:
                    if kind == "submit":
                        got = ctx.trim_batch(p, qual, seq, offsets=offs)
                    else:
                        out.fill_(-7)
                        ctx.scan_device_async(p, dq.data_ptr(), out.data_ptr(), n, offsets_ptr=do.data_ptr(), stride=hint,
                                              seq_ptr=ds.data_ptr() if tn else None)
                        ctx.scan_device_finish()
                        got = out.cpu().numpy()
                    assert err is None, ("device missed the error", it, which, kind, hint, err)
                    bad = np.nonzero((got != want).any(axis=1))[0]
                    if bad.size:
                        b0 = int(bad[0])
                        a_, b_ = int(offs[b0]), int(offs[b0 + 1])
                        print("MISMATCH", (it, which, kind, hint, qt, thr, l, x, tn, bad[:5], got[bad[:5]], want[bad[:5]], lens[bad[:5]]))
                        print("read", b0, "of", n, "qual hex", qual[a_:b_].tobytes().hex())
                        # the read alone, and with its neighbours
                        for lo_, hi_ in ((b0, b0 + 1), (max(0, b0 - 2), min(n, b0 + 3))):
                            o2 = (offs[lo_:hi_ + 1] - offs[lo_]).astype(np.uint64)
                            q2 = qual[int(offs[lo_]):int(offs[hi_])]
                            s2 = seq[int(offs[lo_]):int(offs[hi_])]
                            w2, _ = ob.oracle_trim_batch(po, q2, s2, offsets=o2, threads=1)
                            try:
                                g2 = ctx.trim_batch(p, q2, s2, offsets=o2)
                            except capi.RangeError as e:
                                g2 = "error %s" % ((e.read, e.pos, e.ch),)
                            print("reads", lo_, hi_, "lens", np.diff(o2.astype(np.int64)), "got", g2, "want", w2)
                        raise AssertionError("general kernel differs from the oracle: iteration %d, %s, %s" % (it, which, kind))
                except capi.RangeError as e:
                    assert err is not None and (e.read, e.pos, e.ch) == tuple(err), (it, which, kind, hint, err, (e.read, e.pos, e.ch))
                checked += 1
        if it % 10 == 9:
            if verbose:
                print("iteration %d, %d comparisons, %.0f s" % (it + 1, checked, time.time() - t0), flush=True)
    os.environ.pop("SK_GENERAL", None)
    if verbose:
        print("soak ok: %d iterations, %d comparisons, seed %d" % (iters, checked, seed))
    return checked


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 50, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
