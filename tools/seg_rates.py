#!/usr/bin/env python3
"""Diagnostic: what the segmented kernel costs next to the uniform kernels at the same shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
from sickle_amd import capi
from sickle_amd.capi import TILE_DTYPE

dev = torch.device("cuda", 0)
ctx = capi.Context(0, 1)
s = torch.cuda.Stream(dev)
_settled = False


def timeit(fn, reps=30):
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
    print("%-58s %8.3f ms  %7.2f G reads/s  %6.0f GB/s algorithmic" % (name, ms, n / ms / 1e6, algo_bytes / ms / 1e6), flush=True)


def seg_case(name, lens_rows, p, with_seq=False, shuffle_out=False, use_classes=True):
    """lens_rows: list of (length, rows); tiles of 64 rows, each length at stride 8*odd."""
    tl, at, slot = [], 0, 0
    for Ls, rows in lens_rows:
        st = ((Ls + 7) // 8 | 1) * 8
        for a0 in range(0, rows, 64):
            tl.append((at + a0 * st, slot + a0, st, min(64, rows - a0), Ls, 0))
        at += rows * st
        at = (at + 15) & ~15
        slot += rows
    tiles_np = np.array(tl, dtype=TILE_DTYPE)
    n = slot
    q = torch.randint(40, 74, (at + 4096,), dtype=torch.uint8, device=dev)
    sq = torch.full((at + 4096,), 65, dtype=torch.uint8, device=dev) if with_seq else None
    tiles_t = torch.from_numpy(tiles_np.view(np.uint8)).to(dev)
    oi = (torch.randperm(n, device=dev) if shuffle_out else torch.arange(n, device=dev)).to(torch.int32)
    out = torch.empty((n, 2), dtype=torch.int32, device=dev)
    max_stride = int(tiles_np["stride"].max())
    torch.cuda.synchronize()

    cls, ncls = capi.seg_classes(tiles_np) if use_classes else (None, 0)

    def go():
        b = capi.Batch(q.data_ptr(), sq.data_ptr() if with_seq else None, None, max_stride, 0, None, n, tiles_t.data_ptr(),
                       len(tiles_np), oi.data_ptr(), C.cast(cls, C.c_void_p) if ncls else None, ncls)
        rc = capi.lib().sk_scan_device_async(ctx._h, C.byref(p), C.byref(b), out.data_ptr(), s.cuda_stream)
        assert rc == 0, rc
    totL = sum(L * r for L, r in lens_rows)
    report(name + (" [%d classes]" % ncls if use_classes else " [one launch]"), timeit(go), n, (2 if with_seq else 1) * totL + 8 * n)
    ctx.scan_device_finish(s.cuda_stream)


p = capi.make_params("sanger", 20, 20)
pn = capi.make_params("sanger", 20, 20, False, True)
n = 10_000_000
q = torch.randint(40, 74, (n, 152), dtype=torch.uint8, device=dev)
out = torch.empty((n, 2), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
report("uniform 150 staged (random quals)", timeit(lambda: ctx.scan_device_async(p, q.data_ptr(), out.data_ptr(), n, stride=152, read_len=150, stream=s.cuda_stream)), n, n * 158)
del q, out
m = 4_000_000
for L, st in ((250, 264), (301, 312), (75, 88)):  # the uniform LDS-DMA kernel on the same kind of data as the segmented cases below
    q = torch.randint(40, 74, (m, st), dtype=torch.uint8, device=dev)
    out = torch.empty((m, 2), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    report("uniform %d (stride %d, random quals)" % (L, st), timeit(lambda: ctx.scan_device_async(p, q.data_ptr(), out.data_ptr(), m, stride=st, read_len=L, stream=s.cuda_stream)), m, m * (L + 8))
    del q, out
seg_case("segmented: all 150 (one class), 10 M", [(150, n)], p)
seg_case("segmented: all 150, shuffled out_index", [(150, n)], p, shuffle_out=True)
seg_case("segmented: all 250, 4 M", [(250, 4_000_000)], p)
seg_case("segmented: all 301, 4 M", [(301, 4_000_000)], p)
seg_case("segmented: all 75, 10 M", [(75, n)], p)
mix = [(L, 17_664) for L in range(75, 302)]
seg_case("segmented: 75-301 x 17.6 K each", mix, p)
seg_case("segmented: 75-301 x 17.6 K each", mix, p, use_classes=False)
seg_case("segmented: 75-301, shuffled out_index", mix, p, shuffle_out=True)
seg_case("segmented: 75-301, -n", mix, pn, with_seq=True)
seg_case("segmented: 75-150 x 52 K each", [(L, 52_992) for L in range(75, 151)], p)
seg_case("segmented: 151-301 x 26 K each", [(L, 26_496) for L in range(151, 302)], p)
