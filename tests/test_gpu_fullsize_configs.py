"""GPU, the other BASELINE configurations at their full single-GPU sizes:
  configs[2]/[3]  100 M x 150 bp resident in HBM (15.2 GB), one launch: slices agree with separate scans
                  (position independence), the packed / ragged kernels agree with the tiled one on a
                  10 M slice, the oracle on a 1 M sample, pair classification counts from the cuts;
  configs[3]      one GPU's 12.5 M-read shard through sk_submit / sk_wait in 8 batches over 2 slots;
  configs[4]      12 M reads of 75..301 bp, illumina, -n: segmented against ragged on every read, the
                  oracle on a sample, a planted range error reported in the caller's numbering."""
import ctypes as C

import numpy as np
import pytest

import oracle_bind as ob
from sickle_amd import capi

pytestmark = pytest.mark.gpu


def dev_scan(ctx, torch, dev, params, qual_t, n, seq_t=None, **kw):
    out = torch.empty((n, 2), dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    ptrs = {k: (v.data_ptr() if hasattr(v, "data_ptr") else v) for k, v in kw.items()}
    ctx.scan_device_async(params, qual_t.data_ptr(), out.data_ptr(), n, seq_ptr=None if seq_t is None else seq_t.data_ptr(), **ptrs)
    ctx.scan_device_finish()
    return out


def pair_classes(cuts):
    """reference src/trim_paired.cpp:543-567 on mates at 2k / 2k+1: (pairs kept, singles from 1, singles from 2, pairs discarded)"""
    k1, k2 = cuts[0::2, 1] >= 0, cuts[1::2, 1] >= 0
    return int((k1 & k2).sum()), int((k1 & ~k2).sum()), int((~k1 & k2).sum()), int((~k1 & ~k2).sum())


def test_100m_reads_resident(sk_ctx):
    import torch
    import bench
    dev = torch.device("cuda", 0)
    N, L, S = 100_000_000, 150, 152
    qual = bench.synth_quals_device(torch, N, L, S, 2024, dev)
    p, po = capi.make_params("sanger", 20, 20), ob.make_params("sanger", 20, 20)
    a = dev_scan(sk_ctx, torch, dev, p, qual, N, stride=S, read_len=L)
    # a slice scanned by itself gives the slice of the cuts (start not on a tile boundary, ragged end)
    lo, m = 61_234_567, 10_000_001
    b = dev_scan(sk_ctx, torch, dev, p, qual[lo:lo + m], m, stride=S, read_len=L)
    assert bool((b == a[lo:lo + m]).all())
    # the same 10 M reads packed back to back: re-strided tiles (matrix path), then as a ragged batch
    packed = qual[lo:lo + m, :L].contiguous()
    c = dev_scan(sk_ctx, torch, dev, p, packed, m, stride=L, read_len=L)
    assert bool((c == b).all()), "packed (stride 150) batch disagrees with the strided one"
    off = torch.arange(m + 1, device=dev, dtype=torch.int64) * L
    d = dev_scan(sk_ctx, torch, dev, p, packed, m, offsets_ptr=off, stride=L)
    assert bool((d == b).all()), "ragged batch disagrees with the strided one"
    del packed, off, b, c, d
    # the oracle on a 1 M-read sample, and the PE classification of that sample taken as 500 k pairs
    s0, sm = 87_000_000, 1_000_000
    want, err = ob.oracle_trim_batch(po, qual[s0:s0 + sm].cpu().numpy().reshape(-1), stride=S, read_len=L, n_reads=sm, threads=8)
    got = a[s0:s0 + sm].cpu().numpy()
    assert err is None and (got == want).all()
    assert pair_classes(got) == pair_classes(want)
    # whole batch as 50 M pairs: the four classes partition it; the device's own classification
    # (sk_count_pairs_device_*: src/trim_paired.cpp:543-567) gives the same counts and per-pair classes
    k1, k2 = a[0::2, 1] >= 0, a[1::2, 1] >= 0
    classes = [int((k1 & k2).sum()), int((k1 & ~k2).sum()), int((~k1 & k2).sum()), int((~k1 & ~k2).sum())]
    assert sum(classes) == N // 2 and classes[0] > 0.99 * (N // 2)
    cls_dev = torch.empty((N // 2,), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    assert list(sk_ctx.count_pairs_device(a.data_ptr(), N // 2, cls_dev.data_ptr())) == classes
    want_cls = torch.where(k1, torch.where(k2, 0, 1), torch.where(k2, 2, 3)).to(torch.uint8)
    assert bool((cls_dev == want_cls).all())
    assert list(sk_ctx.count_pairs_device(a[2 * (s0 // 2):].data_ptr(), sm // 2)) == list(pair_classes(want))  # the oracle's sample


def test_shard_of_12_5m_reads_through_submit_wait(sk_ctx):
    import torch
    import bench
    dev = torch.device("cuda", 0)
    N, L, S, B = 12_500_000, 150, 152, 8
    qual = bench.synth_quals_device(torch, N, L, S, 77, dev)
    p, po = capi.make_params("sanger", 20, 20), ob.make_params("sanger", 20, 20)
    resident = dev_scan(sk_ctx, torch, dev, p, qual, N, stride=S, read_len=L).cpu().numpy()
    host = qual.cpu().numpy().reshape(-1)
    del qual
    out = np.full((N, 2), -9, dtype=np.int32)
    per = N // B
    spans = [(i * per, N if i == B - 1 else (i + 1) * per) for i in range(B)]
    for i, (a, b) in enumerate(spans):
        if i >= 2:
            sk_ctx.wait(i % 2)
        sk_ctx.submit(i % 2, p, host[a * S:b * S], out[a:b], stride=S, read_len=L, n_reads=b - a)
    sk_ctx.wait(0)
    sk_ctx.wait(1)
    assert (out == resident).all()
    lo, m = 9_000_000, 300_000
    want, err = ob.oracle_trim_batch(po, host[lo * S:(lo + m) * S], stride=S, read_len=L, n_reads=m, threads=8)
    assert err is None and (out[lo:lo + m] == want).all()


def test_mixed_lengths_illumina_trunc_n_12m(sk_ctx):
    import torch
    from sickle_amd.capi import TILE_DTYPE
    dev = torch.device("cuda", 0)
    n = 12_000_000
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    lens = torch.randint(75, 302, (n,), device=dev, dtype=torch.int64, generator=g)
    off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(lens, 0)
    tot = int(off[n].item())
    # illumina chars 66..105 with a collapse in the last stretch of every 512 bytes; bases with N and n
    qual = torch.randint(80, 106, (tot,), dtype=torch.uint8, device=dev, generator=g)
    qual[: tot - tot % 512].view(-1, 512)[:, 400:] -= 14
    seq = torch.full((tot,), 65, dtype=torch.uint8, device=dev)
    r = torch.rand((tot,), device=dev, generator=g)
    seq[r < 0.003] = ord("N")
    seq[r > 0.9998] = ord("n")
    del r
    p, po = capi.make_params("illumina", 20, 20, False, True), ob.make_params("illumina", 20, 20, False, True)
    ragged = dev_scan(sk_ctx, torch, dev, p, qual, n, seq_t=seq, offsets_ptr=off, stride=301)

    # ---- the segmented image of the same reads: grouped by length (stable), each length at its own stride
    lens_h = lens.cpu().numpy()
    order = np.argsort(lens_h, kind="stable")                 # slot -> read
    counts = np.bincount(lens_h, minlength=302)
    first_slot = np.concatenate([[0], np.cumsum(counts)])[:-1]
    strides = (((np.arange(302) + 7) // 8) | 1) * 8
    group_off = np.zeros(302, dtype=np.int64)
    tl, at = [], 0
    for Lx in range(75, 302):
        cnt = int(counts[Lx])
        if not cnt:
            continue
        at = (at + 15) & ~15
        group_off[Lx] = at
        a0 = np.arange(0, cnt, 64, dtype=np.int64)
        t = np.zeros(len(a0), dtype=TILE_DTYPE)
        t["byte_off"], t["slot0"], t["stride"] = at + a0 * strides[Lx], first_slot[Lx] + a0, strides[Lx]
        t["rows"], t["read_len"] = np.minimum(64, cnt - a0), Lx
        tl.append(t)
        at += cnt * int(strides[Lx])
    tiles = np.concatenate(tl)
    segq = torch.zeros((at + 4096,), dtype=torch.uint8, device=dev)
    segs = torch.zeros((at + 4096,), dtype=torch.uint8, device=dev)
    # destination of every read, then the bytes, a million reads at a time
    slot_of = np.empty(n, dtype=np.int64)
    slot_of[order] = np.arange(n)
    dst_h = group_off[lens_h] + (slot_of - first_slot[lens_h]) * strides[lens_h]
    dst = torch.from_numpy(dst_h).to(dev)
    for a0 in range(0, n, 1_000_000):
        b0 = min(n, a0 + 1_000_000)
        ln = lens[a0:b0]
        rel = torch.arange(int(ln.sum().item()), device=dev) - torch.repeat_interleave(off[a0:b0] - off[a0], ln)
        src_idx = torch.repeat_interleave(off[a0:b0], ln) + rel
        dst_idx = torch.repeat_interleave(dst[a0:b0], ln) + rel
        segq[dst_idx] = qual[src_idx]
        segs[dst_idx] = seq[src_idx]
        del rel, src_idx, dst_idx
    tiles_t = torch.from_numpy(tiles.view(np.uint8).copy()).to(dev)
    oi = torch.from_numpy(order.astype(np.uint32).view(np.int32)).to(dev)
    cls, ncls = capi.seg_classes(tiles)

    def seg_scan(q_t, slot_order=0):
        out = torch.empty((n, 2), dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        b = capi.Batch(q_t.data_ptr(), segs.data_ptr(), None, int(tiles["stride"].max()), 0, None, n, tiles_t.data_ptr(), len(tiles),
                       oi.data_ptr(), C.cast(cls, C.c_void_p) if ncls else None, ncls, slot_order)
        rc = capi.lib().sk_scan_device_async(sk_ctx._h, C.byref(p), C.byref(b), out.data_ptr(), None)
        assert rc == 0
        return out

    seg = seg_scan(segq)
    sk_ctx.scan_device_finish()
    assert bool((seg == ragged).all()), "segmented and ragged kernels disagree"
    # cuts left in slot order (what the CLI asks for): slot k holds the cut of read order[k]
    seg_slots = seg_scan(segq, 1)
    sk_ctx.scan_device_finish()
    assert bool((seg_slots == ragged[torch.from_numpy(order.astype(np.int64)).to(dev)]).all())
    del seg_slots
    kept = int((seg[:, 1] >= 0).sum())
    assert 0.5 * n < kept < n

    # ---- the oracle on a 200 k-read slice of the ragged batch
    lo, m = 7_000_000, 200_000
    o_h = off[lo:lo + m + 1].cpu().numpy()
    q_h = qual[int(o_h[0]):int(o_h[-1])].cpu().numpy()
    s_h = seq[int(o_h[0]):int(o_h[-1])].cpu().numpy()
    want, err = ob.oracle_trim_batch(po, q_h, s_h, offsets=(o_h - o_h[0]).astype(np.uint64), threads=8)
    assert err is None and (seg[lo:lo + m].cpu().numpy() == want).all()

    # ---- a planted out-of-range char: two victims, the lower ORIGINAL index must be reported
    victims = [9_876_543, 2_345_678]
    for v in victims:
        segq[int(dst_h[v]) + 5] = 20
    for slot_order in (0, 1):
        seg_scan(segq, slot_order)
        with pytest.raises(capi.RangeError) as ei:
            sk_ctx.scan_device_finish()
        assert (ei.value.read, ei.value.pos, ei.value.ch) == (2_345_678, 5, 20)
