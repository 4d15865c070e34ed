#!/usr/bin/env python3
"""Uniform medium reads, 520 ... 4090 bases back to back: the kernel the library selects (the 32-read tiles of
sk_kernels.hip up to SK_WIDE_MAX, the general kernels beyond) next to each general kernel forced (SK_GENERAL)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sickle_amd import capi
dev = torch.device("cuda", 0)
ctx = capi.Context(0, 1)
s = torch.cuda.Stream(dev)
p = capi.make_params("sanger", 20, 20)
total = 1_000_000_000
g = torch.Generator(device=dev); g.manual_seed(5)
q = torch.randint(60, 74, (total + 65536,), dtype=torch.uint8, device=dev, generator=g)
q.view(-1)[: total].view(-1, 1000)[:, 800:] -= 25
torch.cuda.synchronize()
def timeit(fn, reps=10):
    for _ in range(60): fn()
    s.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in evs:
        e0.record(s); fn(); e1.record(s)
    s.synchronize()
    return sum(e0.elapsed_time(e1) for e0, e1 in evs) / reps
for L in (520, 600, 640, 800, 1000, 1024, 1280, 1500, 2000, 2500, 3000, 4000, 4090):
    n = total // L
    out = torch.empty((n, 2), dtype=torch.int32, device=dev)
    res = {}
    os.environ.pop("SK_GENERAL", None)
    kern = capi.lib().sk_kernel_for(capi.Batch(q.data_ptr(), None, None, L, L, None, n))
    for which in ("default", "band", "team", "stream"):
        if which == "default":
            os.environ.pop("SK_GENERAL", None)
        else:
            os.environ["SK_GENERAL"] = which
        ms = timeit(lambda: ctx.scan_device_async(p, q.data_ptr(), out.data_ptr(), n, stride=L, read_len=L, stream=s.cuda_stream))
        ctx.scan_device_finish(s.cuda_stream)
        res[which] = n * (L + 8) / ms / 1e6
    print("L %5d  selected (kernel %d) %5.0f GB/s  band %5.0f GB/s  team16 %5.0f GB/s  stream %5.0f GB/s" % (L, kern, res["default"], res["band"], res["team"], res["stream"]), flush=True)
