#!/usr/bin/env python3
"""The mixed-length batch of BASELINE configs[4] (U{75..301}) three ways, alternating, for a kernel trace: segmented
(slot order, as the CLI hands it over), ragged offsets in input order (regrouped on the device), and uniform 150 bp as
the yardstick of the box.  argv: reads (default 4 M), timed launches, settle launches.  -n with SEQ=1."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ctypes as C
import numpy as np
import torch
import workloads as wl
from sickle_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
settle = int(sys.argv[3]) if len(sys.argv) > 3 else 120
with_seq = os.environ.get("SEQ", "0") == "1"
dev = torch.device("cuda", 0)
ctx = capi.Context(0, 1)
s = torch.cuda.Stream(dev)
p = capi.make_params("illumina", 20, 20, False, with_seq)
lens, qual, seq = wl.mixed_shard(torch, dev, 1234, 0, n)
seg = wl.segment(torch, lens, qual, seq if with_seq else None)
tiles_t = torch.from_numpy(seg["tiles"].view(np.uint8)).to(dev)
cls, ncls = capi.seg_classes(seg["tiles"])
out_seg = torch.empty((n, 2), dtype=torch.int32, device=dev)
# ragged, input order
offs = torch.zeros(n + 1, dtype=torch.int64, device=dev)
offs[1:] = torch.cumsum(lens, 0)
total = int(offs[-1].item())
mask = torch.arange(wl.MIX_HI, device=dev)[None, :] < lens[:, None]
rq = torch.zeros(total + 4096, dtype=torch.uint8, device=dev)
rq[:total] = qual[mask]
rs = None
if with_seq:
    rs = torch.zeros(total + 4096, dtype=torch.uint8, device=dev)
    rs[:total] = seq[mask]
del mask
offs_u = offs.to(torch.uint64) if hasattr(torch, "uint64") else offs
out_rag = torch.empty((n, 2), dtype=torch.int32, device=dev)
# the yardstick
m = 6_578_944
uq = wl.se_shard(torch, dev, 7, 0, m, 150, 152)
out_u = torch.empty((m, 2), dtype=torch.int32, device=dev)
pu = capi.make_params("sanger", 20, 20)
torch.cuda.synchronize()


def segmented():
    b = capi.Batch(seg["q"].data_ptr(), seg["seq"].data_ptr() if with_seq else None, None, seg["max_stride"], 0, None, n, tiles_t.data_ptr(),
                   len(seg["tiles"]), seg["out_index"].data_ptr(), C.cast(cls, C.c_void_p) if ncls else None, ncls, 1)
    rc = capi.lib().sk_scan_device_async(ctx._h, C.byref(p), C.byref(b), out_seg.data_ptr(), s.cuda_stream)
    assert rc == 0, rc


def ragged():
    b = capi.Batch(rq.data_ptr(), rs.data_ptr() if with_seq else None, offs.data_ptr(), wl.MIX_HI, 0, None, n)
    rc = capi.lib().sk_scan_device_async(ctx._h, C.byref(p), C.byref(b), out_rag.data_ptr(), s.cuda_stream)
    assert rc == 0, rc


def uniform():
    ctx.scan_device_async(pu, uq.data_ptr(), out_u.data_ptr(), m, stride=152, read_len=150, stream=s.cuda_stream)


for _ in range(settle + reps):
    uniform()
    segmented()
    ragged()
ctx.scan_device_finish(s.cuda_stream)
back = torch.empty_like(out_seg)
back[seg["out_index"].long()] = out_seg
assert torch.equal(back, out_rag), "segmented and ragged disagree"
bases = int(lens.sum().item())
print("reads", n, "bases", bases, "algorithmic_bytes", (2 if with_seq else 1) * bases + 8 * n, "classes", ncls, "tiles", len(seg["tiles"]))
