#!/usr/bin/env python3
"""Uniform medium reads with -n (the sequence tile rides the same LDS buffer after the quality scan): the kernel the
library selects against the teams of 16 forced, 600 / 1000 / 1500 bases back to back."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sickle_amd import capi
dev = torch.device("cuda", 0)
ctx = capi.Context(0, 1)
s = torch.cuda.Stream(dev)
total = 600_000_000
g = torch.Generator(device=dev); g.manual_seed(5)
q = torch.randint(60, 74, (total + 65536,), dtype=torch.uint8, device=dev, generator=g)
q.view(-1)[: total].view(-1, 1000)[:, 800:] -= 25
sq = torch.full((total + 65536,), 65, dtype=torch.uint8, device=dev)
sq[torch.randint(0, total, (total // 400,), device=dev, generator=g)] = ord("N")
torch.cuda.synchronize()
def timeit(fn, reps=10):
    for _ in range(60): fn()
    s.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in evs:
        e0.record(s); fn(); e1.record(s)
    s.synchronize()
    return sum(e0.elapsed_time(e1) for e0, e1 in evs) / reps
for L in (600, 1000, 1500):
    n = total // L
    out = torch.empty((n, 2), dtype=torch.int32, device=dev)
    res = {}
    for tn in (0, 1):
        p = capi.make_params("sanger", 20, 20, False, bool(tn))
        for which in ("default", "team"):
            if which == "default": os.environ.pop("SK_GENERAL", None)
            else: os.environ["SK_GENERAL"] = which
            ms = timeit(lambda: ctx.scan_device_async(p, q.data_ptr(), out.data_ptr(), n, stride=L, read_len=L, seq_ptr=sq.data_ptr() if tn else None, stream=s.cuda_stream))
            ctx.scan_device_finish(s.cuda_stream)
            res[(tn, which)] = n * ((2 if tn else 1) * L + 8) / ms / 1e6
    print("L %5d  selected %5.0f GB/s (teams %5.0f)   with -n: selected %5.0f GB/s (teams %5.0f)" % (L, res[(0, "default")], res[(0, "team")], res[(1, "default")], res[(1, "team")]), flush=True)
