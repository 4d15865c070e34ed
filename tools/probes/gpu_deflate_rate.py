"""probe: throughput of sk_bgzf_deflate (H2D + kernel + D2H, pinned host buffers) on synthetic FASTQ"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from sickle_amd import capi, synth
s, q = synth.make_reads(1000, 250000, 150, "sanger")
one = np.frombuffer(synth.fastq_bytes_fast(s, q, start=0, suffix="/1"), dtype=np.uint8)
data = np.tile(one, 6)  # ~480 MB
n_blocks = (data.size + capi.BGZF_INPUT - 1) // capi.BGZF_INPUT
text = torch.zeros(n_blocks * capi.BGZF_INPUT, dtype=torch.uint8).pin_memory()
text[:data.size] = torch.from_numpy(data)
sizes = np.array([min(capi.BGZF_INPUT, data.size - b * capi.BGZF_INPUT) for b in range(n_blocks)], dtype=np.uint32)
out = torch.zeros(n_blocks * capi.BGZF_SLOT, dtype=torch.uint8).pin_memory()
out_sizes = np.zeros(n_blocks, dtype=np.uint32)
L = capi.lib()
for rep in range(4):
    t0 = time.perf_counter()
    rc = L.sk_bgzf_deflate(0, text.data_ptr(), sizes.ctypes.data, n_blocks, out.data_ptr(), out_sizes.ctypes.data)
    dt = time.perf_counter() - t0
    assert rc == 0, L.sk_bgzf_last_error()
    print("%d blocks, %.1f MB: %.3f s  %.2f GB/s of text; compressed to %.1f%%; %d blocks not compressed" %
          (n_blocks, data.size / 1e6, dt, data.size / dt / 1e9, 100.0 * out_sizes.sum() / data.size, int((out_sizes == 0).sum())), flush=True)
