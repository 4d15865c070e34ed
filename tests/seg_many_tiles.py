#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (uses the oracle).  A segmented batch with more tiles than the launch has waves -- every wave
goes round its loop several times: full tile after full tile, the last tile of a length (fewer than 64 rows) and the
full tile after it, a new window width every few turns -- which tests/soak_tiles.py's small batches do not do.
300 000 reads of 75..301 bases (BASELINE configs[4]'s model), cuts in slot order and scattered, with and without -n,
against the oracle.  SK_SEG_STAGE=1 in the environment runs the register-staged variant of the kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ctypes as C
import numpy as np
import torch
import oracle_bind as ob
import workloads as wl
from sickle_amd import capi


def run(n=300_000, seed=5):
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0, 1)
    lens, qual, seq = wl.mixed_shard(torch, dev, seed, 0, n)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens.cpu().numpy())
    mask = torch.arange(wl.MIX_HI, device=dev)[None, :] < lens[:, None]
    rq, rs = qual[mask].cpu().numpy(), seq[mask].cpu().numpy()
    checked = 0
    for tn in (False, True):
        p, po = capi.make_params("illumina", 20, 20, False, tn), ob.make_params("illumina", 20, 20, False, tn)
        want, err = ob.oracle_trim_batch(po, rq, rs, offsets=offs, threads=8)
        assert err is None
        seg = wl.segment(torch, lens, qual, seq if tn else None)
        tiles_t = torch.from_numpy(seg["tiles"].view(np.uint8)).to(dev)
        cls, ncls = capi.seg_classes(seg["tiles"])
        assert len(seg["tiles"]) > 4096  # more tiles than waves
        for slot_order in (1, 0):
            out = torch.full((n, 2), -7, dtype=torch.int32, device=dev)
            b = capi.Batch(seg["q"].data_ptr(), seg["seq"].data_ptr() if tn else None, None, seg["max_stride"], 0, None, n, tiles_t.data_ptr(),
                           len(seg["tiles"]), seg["out_index"].data_ptr(), C.cast(cls, C.c_void_p) if ncls else None, ncls, slot_order)
            rc = capi.lib().sk_scan_device_async(ctx._h, C.byref(p), C.byref(b), out.data_ptr(), None)
            assert rc == 0, rc
            ctx.scan_device_finish()
            got = out
            if slot_order:
                got = torch.empty_like(out)
                got[seg["out_index"].long()] = out
            got = got.cpu().numpy()
            bad = np.nonzero((got != want).any(axis=1))[0]
            assert bad.size == 0, (tn, slot_order, bad[:5], got[bad[:5]], want[bad[:5]])
            checked += 1
    print("seg many tiles ok: %d scans of %d reads in %d tiles" % (checked, n, len(seg["tiles"])))
    return checked


if __name__ == "__main__":
    run()
