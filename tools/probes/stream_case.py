#!/usr/bin/env python3
"""One uniform long-read case of the general kernel, a few launches (for rocprofv3 --pmc): L from argv."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sickle_amd import capi

L = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
dev = torch.device("cuda", 0)
ctx = capi.Context(0, 1)
s = torch.cuda.Stream(dev)
p = capi.make_params("sanger", 20, 20)
total = 1_000_000_000
g = torch.Generator(device=dev); g.manual_seed(5)
q = torch.randint(60, 74, (total + 65536,), dtype=torch.uint8, device=dev, generator=g)
q.view(-1)[: total].view(-1, 1000)[:, 800:] -= 25
n = total // L
out = torch.empty((n, 2), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for _ in range(10):
    ctx.scan_device_async(p, q.data_ptr(), out.data_ptr(), n, stride=L, read_len=L, stream=s.cuda_stream)
ctx.scan_device_finish(s.cuda_stream)
print("reads", n, "blocks", n * (L // 1024 + 1))
