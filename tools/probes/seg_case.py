#!/usr/bin/env python3
"""The uniform LDS-DMA tile kernel and the segmented kernel on THE SAME buffer (one length, rows at the segmented
layout's stride), a few launches each (for rocprofv3 --pmc / --kernel-trace): L from argv, reads = 1e9 / stride."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
from sickle_amd import capi
from sickle_amd.capi import TILE_DTYPE

L = int(sys.argv[1]) if len(sys.argv) > 1 else 250
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
settle = int(sys.argv[3]) if len(sys.argv) > 3 else 0
slot_order = int(os.environ.get("SEG_SLOT_ORDER", "1"))  # 1: cuts in slot order (what the CLI takes); 0: scattered through out_index
st = ((L + 7) // 8 | 1) * 8
n = 1_000_000_000 // st // 64 * 64
dev = torch.device("cuda", 0)
ctx = capi.Context(0, 1)
s = torch.cuda.Stream(dev)
p = capi.make_params("sanger", 20, 20)
g = torch.Generator(device=dev); g.manual_seed(5)
q = torch.randint(40, 74, (n * st + 4096,), dtype=torch.uint8, device=dev, generator=g)
out = torch.empty((n, 2), dtype=torch.int32, device=dev)
out2 = torch.empty((n, 2), dtype=torch.int32, device=dev)
a0 = np.arange(0, n, 64, dtype=np.int64)
t = np.zeros(len(a0), dtype=TILE_DTYPE)
t["byte_off"] = a0 * st; t["slot0"] = a0; t["stride"] = st; t["rows"] = 64; t["read_len"] = L
tiles_t = torch.from_numpy(t.view(np.uint8)).to(dev)
oi = torch.arange(n, device=dev).to(torch.int32)
torch.cuda.synchronize()
cls, ncls = capi.seg_classes(t)
def uniform():
    ctx.scan_device_async(p, q.data_ptr(), out.data_ptr(), n, stride=st, read_len=L, stream=s.cuda_stream)


def segmented():
    b = capi.Batch(q.data_ptr(), None, None, st, 0, None, n, tiles_t.data_ptr(), len(t), oi.data_ptr(), C.cast(cls, C.c_void_p) if ncls else None, ncls, slot_order)
    rc = capi.lib().sk_scan_device_async(ctx._h, C.byref(p), C.byref(b), out2.data_ptr(), s.cuda_stream)
    assert rc == 0, rc


# the clocks of the device settle over the first ~100 launches; the two kernels alternate so that neither gets
# the better half of whatever drift is left (the summary takes the LAST `reps` launches of each from the trace)
for _ in range(settle + reps):
    uniform()
    segmented()
ctx.scan_device_finish(s.cuda_stream)
assert torch.equal(out, out2)
print("reads", n, "tiles", n // 64, "stride", st)
