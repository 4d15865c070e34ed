#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (uses the oracle, like everything under tests/).  Soak: random UNIFORM batches of medium reads
(320 .. 2600 bases at any stride, 1 .. 300 reads) through the kernel the library selects for them (the 32-read tiles of
sk_kernels.hip, WIDE; beyond its range the general kernels), host and device entry points, against the oracle.
usage: soak_wide.py [iterations] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle_bind as ob
from sickle_amd import capi

EDGE = [320, 321, 329, 330, 351, 352, 505, 512, 600, 639, 640, 641, 650, 959, 960, 1000, 1023, 1024, 1025, 1279, 1280, 1500, 2000, 2047, 2048, 2520, 2528, 2529, 2600]


def run(iters=60, seed=1, verbose=True):
    rng = np.random.default_rng(seed)
    ctx = capi.Context(0, 2)
    t0 = time.time()
    checked = 0
    kernels = set()
    for it in range(iters):
        qt = ["sanger", "solexa", "illumina"][it % 3]
        lo, hi = {"sanger": (33, 126), "solexa": (59, 112), "illumina": (64, 110)}[qt]
        L = int(rng.choice(EDGE)) if it % 2 == 0 else int(rng.integers(320, 2601))
        n = int(rng.choice([1, 31, 32, 33, 64, 65])) if it % 5 == 0 else int(rng.integers(1, 300))
        stride = L + int(rng.choice([0, 0, 1, 3, 8, 16, 40]))
        tot = n * L
        thr = int(rng.choice([0, 2, 15, 20, 25, 30, 41]))
        mid = min(hi - 3, max(lo + 3, lo + thr + int(rng.integers(-4, 12))))
        mode = it % 5
        if mode == 0:
            qual = np.clip(rng.normal(mid, 6, tot).astype(int), lo, hi)
        elif mode == 1:
            qual = np.clip(mid + rng.integers(-2, 3, size=tot), lo, hi)  # hovering at the threshold
        elif mode == 2:
            level = np.repeat(rng.integers(lo, hi, size=tot // 300 + 2), 300)[:tot]
            qual = np.clip(level + rng.integers(-3, 4, size=tot), lo, hi)
        elif mode == 3:
            qual = np.where(rng.random(tot) < 0.5, lo, hi)
        else:
            qual = np.clip(rng.normal(mid + 8, 4, tot).astype(int), lo, hi).reshape(n, L)
            for i in range(n):
                c = int(rng.integers(0, L))
                qual[i, c:] = np.clip(rng.normal(lo + 5, 3, L - c).astype(int), lo, hi)
                if i % 3 == 0:
                    h = int(rng.integers(0, 40))
                    qual[i, :h] = lo + 1
            qual = qual.reshape(-1)
        qual = qual.astype(np.uint8)
        if it % 4 == 3:  # a char out of range somewhere
            qual[int(rng.integers(0, tot))] = int(rng.choice([lo - 1, hi + 1, 200, 10]))
        seq = rng.choice(np.frombuffer(b"ACGT" * 3000 + b"Nn", dtype=np.uint8), size=tot)
        l = int(rng.choice([0, 20, 300, 1500]))
        x, tn = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        p, po = capi.make_params(qt, thr, l, x, tn), ob.make_params(qt, thr, l, x, tn)
        offs = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
        want, err = ob.oracle_trim_batch(po, qual, seq, offsets=offs, threads=8)
        qs = np.full((n, stride), lo, dtype=np.uint8)
        ss = np.full((n, stride), ord("A"), dtype=np.uint8)
        qs[:, :L] = qual.reshape(n, L)
        ss[:, :L] = seq.reshape(n, L)
        qs, ss = qs.reshape(-1), ss.reshape(-1)
        kernels.add(capi.lib().sk_kernel_for(capi.Batch(qs.ctypes.data, None, None, stride, L, None, n)))
        dq, ds = torch.from_numpy(qs).cuda(), torch.from_numpy(ss).cuda()
        out = torch.empty((n, 2), dtype=torch.int32, device="cuda")
        for kind in ("submit", "device"):
            try:
                if kind == "submit":
                    got = ctx.trim_batch(p, qs, ss if tn else None, stride=stride, read_len=L, n_reads=n)
                else:
                    out.fill_(-7)
                    ctx.scan_device_async(p, dq.data_ptr(), out.data_ptr(), n, stride=stride, read_len=L, seq_ptr=ds.data_ptr() if tn else None)
                    ctx.scan_device_finish()
                    got = out.cpu().numpy()
                assert err is None, ("device missed the error", it, kind, L, n, stride, err)
                bad = np.nonzero((got != want).any(axis=1))[0]
                if bad.size:
                    b0 = int(bad[0])
                    raise AssertionError("medium-read kernel differs from the oracle: %r" % ((it, kind, qt, thr, l, x, tn, L, n, stride, bad[:5], got[bad[:5]], want[bad[:5]],
                                                                                            qual[b0 * L:(b0 + 1) * L].tobytes().hex()),))
            except capi.RangeError as e:
                assert err is not None and (e.read, e.pos, e.ch) == tuple(err), (it, kind, L, n, stride, err, (e.read, e.pos, e.ch))
            checked += 1
        if verbose and it % 20 == 19:
            print("iteration %d, %d comparisons, %.0f s, kernels %s" % (it + 1, checked, time.time() - t0, sorted(kernels)), flush=True)
    if verbose:
        print("soak ok: %d iterations, %d comparisons, seed %d, kernels used %s" % (iters, checked, seed, sorted(kernels)))
    return checked, kernels


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
