"""probe: 1500 launches queued back to back; per-launch times averaged in groups of 50"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from sickle_amd import capi
dev = torch.device("cuda", 0)
n = 10_000_000
qual = bench.synth_quals_device(torch, n, 150, 152, 1234, dev)
out = torch.empty((n, 2), dtype=torch.int32, device=dev)
ctx = capi.Context(device=0, slots=1)
params = capi.make_params("sanger", 20, 20)
stream = torch.cuda.Stream(dev)
torch.cuda.synchronize(dev)
for rep in range(2):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(1500)]
    for a, b in evs:
        a.record(stream)
        ctx.scan_device_async(params, qual.data_ptr(), out.data_ptr(), n, stride=152, read_len=150, stream=stream.cuda_stream)
        b.record(stream)
    torch.cuda.synchronize(dev)
    ctx.scan_device_finish(stream.cuda_stream)
    ts = [a.elapsed_time(b) for a, b in evs]
    print("rep %d groups of 50:" % rep, " ".join("%.3f" % (sum(ts[i:i + 50]) / 50) for i in range(0, 1500, 50)), flush=True)
