#!/usr/bin/env python3
"""Diagnostic: the PCIe-inclusive rate of the host-buffer path (sk_submit / sk_wait, two slots:
H2D of batch i+1 overlaps the scan of batch i), from pinned host memory.  Not the bench metric."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sickle_amd import capi, synth

n, L, stride = 4_000_000, 150, 152
ctx = capi.Context(0, 2)
lib = capi.lib()
_, qual = synth.make_reads(1, 200_000, L)
tile = synth.pack_fixed(qual, stride)
bufs, outs = [], []
for i in range(2):
    p = lib.sk_host_alloc(ctx._h, n * stride)
    q = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n * stride,))
    for a in range(0, n * stride, tile.size):
        m = min(tile.size, n * stride - a)
        q[a:a + m] = tile[:m]
    po = lib.sk_host_alloc(ctx._h, n * 8)
    o = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_int32)), shape=(n, 2))
    bufs.append(q)
    outs.append(o)
params = capi.make_params("sanger", 20, 20)
for rounds in (2, 12):
    t0 = time.perf_counter()
    for i in range(rounds):
        s = i % 2
        if i >= 2:
            ctx.wait(s)
        ctx.submit(s, params, bufs[s], outs[s], stride=stride, read_len=L, n_reads=n)
    ctx.wait(0)
    ctx.wait(1)
    dt = time.perf_counter() - t0
    if rounds > 2:
        print("%d batches x %d reads: %.3f s  %.1f M reads/s  H2D %.1f GB/s + D2H %.1f GB/s" %
              (rounds, n, dt, rounds * n / dt / 1e6, rounds * n * stride / dt / 1e9, rounds * n * 8 / dt / 1e9))
