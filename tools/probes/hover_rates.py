#!/usr/bin/env python3
"""Long reads whose window averages HOVER at the threshold (every 16-window cell of the streaming kernel flagged):
uniform reads of a few lengths; chars = threshold + U{-1, 0, 1} (`hover`: the cuts are found in the first windows) and
chars alternating threshold / threshold + 1 (`sustain`: Q20.5 against -q 20 over the whole read, nothing ever found),
against the usual synthetic pattern (which itself hovers at 4 kb: its 400-base windows average Q21 where they cover a low fifth).  Kernel as the
library selects it, and the streaming kernel forced (SK_GENERAL=stream) below 4096."""
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
usual = torch.randint(60, 74, (total + 65536,), dtype=torch.uint8, device=dev, generator=g)
usual.view(-1)[: total].view(-1, 1000)[:, 800:] -= 25
hover = (53 + torch.randint(-1, 2, (total + 65536,), device=dev, generator=g)).to(torch.uint8)  # crosses the threshold early: both windows found at once
sustain = (53 + (torch.arange(total + 65536, device=dev) & 1)).to(torch.uint8)  # averages Q20.5 against -q 20 from end to end: never below, every cell flagged
torch.cuda.synchronize()
def timeit(fn, reps=10):
    for _ in range(60): fn()
    s.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in evs:
        e0.record(s); fn(); e1.record(s)
    s.synchronize()
    return sum(e0.elapsed_time(e1) for e0, e1 in evs) / reps
LENS = tuple(int(x) for x in sys.argv[1:]) or (4000, 5000, 10_000, 30_000)  # (600 1000 1500: the medium-read tiles)
for L in LENS:
    n = total // L
    out = torch.empty((n, 2), dtype=torch.int32, device=dev)
    res = {}
    for name, q in (("usual", usual), ("hover", hover), ("sustain", sustain)):
        for which in (("default", "stream") if L <= 4096 else ("default",)):
            if which == "stream":
                os.environ["SK_GENERAL"] = "stream"
            else:
                os.environ.pop("SK_GENERAL", None)
            ms = timeit(lambda: ctx.scan_device_async(p, q.data_ptr(), out.data_ptr(), n, stride=L, read_len=L, stream=s.cuda_stream))
            ctx.scan_device_finish(s.cuda_stream)
            res[name + "/" + which] = n * (L + 8) / ms / 1e6
    print("L %6d  " % L + "  ".join("%s %5.0f GB/s" % kv for kv in res.items()), flush=True)
