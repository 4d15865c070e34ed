#!/usr/bin/env python3
"""Diagnostic: HBM-resident rates of the general (team) kernel on long reads."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sickle_amd import capi

dev = torch.device("cuda", 0)
ctx = capi.Context(0, 1)
s = torch.cuda.Stream(dev)
_settled = False


def timeit(fn, reps=10):
    global _settled
    for _ in range(3 if _settled else 40):
        fn()
    s.synchronize()
    _settled = True
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in evs:
        e0.record(s); fn(); e1.record(s)
    s.synchronize()
    return sum(e0.elapsed_time(e1) for e0, e1 in evs) / reps


def report(name, ms, n, algo_bytes):
    print("%-52s %8.3f ms  %9.4f G reads/s  %6.0f GB/s algorithmic" % (name, ms, n / ms / 1e6, algo_bytes / ms / 1e6), flush=True)


p = capi.make_params("sanger", 20, 20)
pn = capi.make_params("sanger", 20, 20, False, True)
total = 1_000_000_000
g = torch.Generator(device=dev); g.manual_seed(5)
q = torch.randint(60, 74, (total + 65536,), dtype=torch.uint8, device=dev, generator=g)
# a quality collapse in the last fifth of every kilobase so that cuts land inside reads
q.view(-1)[: total].view(-1, 1000)[:, 800:] -= 25
sq = torch.full((total + 65536,), 65, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for L in (600, 1000, 2000, 2048, 2049, 5000, 10_000, 10_240, 30_000, 30_720, 100_000):
    n = total // L
    out = torch.empty((n, 2), dtype=torch.int32, device=dev)
    report("uniform %d (stride %d)" % (L, L), timeit(lambda: ctx.scan_device_async(p, q.data_ptr(), out.data_ptr(), n, stride=L, read_len=L, stream=s.cuda_stream)), n, n * (L + 8))
    if L in (1000, 10_000):
        report("uniform %d, -n" % L, timeit(lambda: ctx.scan_device_async(pn, q.data_ptr(), out.data_ptr(), n, stride=L, read_len=L, seq_ptr=sq.data_ptr(), stream=s.cuda_stream)), n, n * (2 * L + 8))
    ctx.scan_device_finish(s.cuda_stream)
# ragged, lengths 1000..30000
lens = torch.randint(1000, 30_001, (60_000,), device=dev, dtype=torch.int64, generator=g)
off = torch.zeros(len(lens) + 1, dtype=torch.int64, device=dev)
off[1:] = torch.cumsum(lens, 0)
n = int((off <= total).sum().item()) - 1
out = torch.empty((n, 2), dtype=torch.int32, device=dev)
tot = int(off[n].item())
torch.cuda.synchronize()
report("ragged 1-30 kb, hint 30000", timeit(lambda: ctx.scan_device_async(p, q.data_ptr(), out.data_ptr(), n, offsets_ptr=off.data_ptr(), stride=30_000, stream=s.cuda_stream)), n, tot + 8 * n)
report("ragged 1-30 kb, no hint", timeit(lambda: ctx.scan_device_async(p, q.data_ptr(), out.data_ptr(), n, offsets_ptr=off.data_ptr(), stream=s.cuda_stream)), n, tot + 8 * n)
ctx.scan_device_finish(s.cuda_stream)
