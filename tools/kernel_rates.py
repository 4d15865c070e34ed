#!/usr/bin/env python3
"""Diagnostic: HBM-resident rates of every kernel variant (not the bench metric)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from sickle_amd import capi

dev = torch.device("cuda", 0)
ctx = capi.Context(0, 1)
s = torch.cuda.Stream(dev)


_settled = False


def timeit(fn, reps=30):
    """Average of `reps` launches queued back to back (like bench.py), after the device's start-up
    clock ramp (the first ~100 launches of a process: tools/probes/bench_times.py) and 5 warm-ups."""
    global _settled
    for _ in range(5 if _settled else 150):
        fn()
    s.synchronize()
    _settled = True
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in evs:
        e0.record(s); fn(); e1.record(s)
    s.synchronize()
    return sum(e0.elapsed_time(e1) for e0, e1 in evs) / reps


def report(name, ms, n, algo_bytes):
    print("%-46s %8.3f ms  %7.2f G reads/s  %6.0f GB/s algorithmic" % (name, ms, n / ms / 1e6, algo_bytes / ms / 1e6), flush=True)


n, L = 10_000_000, 150
q = bench.synth_quals_device(torch, n, L, 152, 1, dev)
seq = torch.full((n, 152), 65, dtype=torch.uint8, device=dev)
seq[torch.rand(n, device=dev) < 0.25, 77] = ord("N")
out = torch.empty((n, 2), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
p = capi.make_params("sanger", 20, 20)
pn = capi.make_params("sanger", 20, 20, False, True)
report("tile+mfma uniform 150 (stride 152)", timeit(lambda: ctx.scan_device_async(p, q.data_ptr(), out.data_ptr(), n, stride=152, read_len=L, stream=s.cuda_stream)), n, n * 158)
report("tile+mfma uniform 150, -n", timeit(lambda: ctx.scan_device_async(pn, q.data_ptr(), out.data_ptr(), n, stride=152, read_len=L, seq_ptr=seq.data_ptr(), stream=s.cuda_stream)), n, n * 308)
lens = torch.full((n,), L, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
report("tile valu, per-read lengths (all 150)", timeit(lambda: ctx.scan_device_async(p, q.data_ptr(), out.data_ptr(), n, stride=152, lengths_ptr=lens.data_ptr(), stream=s.cuda_stream)), n, n * 158)
pk = q[:, :L].contiguous()
off = torch.arange(n + 1, device=dev, dtype=torch.int64) * L
torch.cuda.synchronize()
report("ragged offsets (all 150), hint 150", timeit(lambda: ctx.scan_device_async(p, pk.data_ptr(), out.data_ptr(), n, offsets_ptr=off.data_ptr(), stride=150, stream=s.cuda_stream), 10), n, n * 158)
report("ragged offsets (all 150), no hint", timeit(lambda: ctx.scan_device_async(p, pk.data_ptr(), out.data_ptr(), n, offsets_ptr=off.data_ptr(), stream=s.cuda_stream), 10), n, n * 158)
report("packed uniform 150 (stride 150)", timeit(lambda: ctx.scan_device_async(p, pk.data_ptr(), out.data_ptr(), n, stride=150, read_len=150, stream=s.cuda_stream), 10), n, n * 158)
del pk, off, seq
# mixed lengths 75..301
m = 4_000_000
q3 = bench.synth_quals_device(torch, m, 301, 304, 2, dev)
lens3 = torch.randint(75, 302, (m,), device=dev, dtype=torch.int32)
out3 = torch.empty((m, 2), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
tot = int(lens3.sum().item())
report("tile valu, mixed 75-301 at stride 304", timeit(lambda: ctx.scan_device_async(p, q3.data_ptr(), out3.data_ptr(), m, stride=304, lengths_ptr=lens3.data_ptr(), stream=s.cuda_stream)), m, tot + 8 * m)
q250 = bench.synth_quals_device(torch, m, 250, 264, 3, dev)
torch.cuda.synchronize()
report("tile+mfma uniform 250 (stride 264)", timeit(lambda: ctx.scan_device_async(p, q250.data_ptr(), out3.data_ptr(), m, stride=264, read_len=250, stream=s.cuda_stream)), m, m * 258)
q100 = bench.synth_quals_device(torch, n, 100, 104, 3, dev)
torch.cuda.synchronize()
report("tile+mfma uniform 100 (stride 104)", timeit(lambda: ctx.scan_device_async(p, q100.data_ptr(), out.data_ptr(), n, stride=104, read_len=100, stream=s.cuda_stream)), n, n * 108)
ctx.scan_device_finish(s.cuda_stream)
# segmented batch: lengths 75..301, ~17.6 K reads each, one descriptor per 64-read tile
import numpy as np
from sickle_amd.capi import TILE_DTYPE
rows_per_len = 17_664  # 276 tiles
tl = []
at = 0
slot = 0
for Ls in range(75, 302):
    st = ((Ls + 7) // 8 | 1) * 8
    for a0 in range(0, rows_per_len, 64):
        tl.append((at + a0 * st, slot + a0, st, 64, Ls, 0))
    at += rows_per_len * st
    slot += rows_per_len
tiles_np = np.array(tl, dtype=TILE_DTYPE)
nseg = slot
qseg = torch.randint(40, 74, (at,), dtype=torch.uint8, device=dev)
tiles_t = torch.from_numpy(tiles_np.view(np.uint8)).to(dev)
oi = torch.arange(nseg, dtype=torch.int32, device=dev)
outs = torch.empty((nseg, 2), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
import ctypes as C
def seg():
    b = capi.Batch(qseg.data_ptr(), None, None, 304, 0, None, nseg, tiles_t.data_ptr(), len(tiles_np), oi.data_ptr())
    rc = capi.lib().sk_scan_device_async(ctx._h, C.byref(p), C.byref(b), outs.data_ptr(), s.cuda_stream)
    assert rc == 0, rc
totL = sum(rows_per_len * Ls for Ls in range(75, 302))
report("segmented, lengths 75-301 (per-tile stride)", timeit(seg), nseg, totL + 8 * nseg)
ctx.scan_device_finish(s.cuda_stream)
