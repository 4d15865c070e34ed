"""GPU, BASELINE.json's full single-GPU size (10 M x 150 bp): properties that do not need the
oracle to run over the whole batch --
  * the two independent kernels (lane-per-read tiled + MFMA, wave-per-read general) agree on every read,
  * cuts do not depend on where a read sits in the batch (reversed batch -> reversed cuts),
  * a prefix of the batch gives the prefix of the cuts (partial last tile),
and the oracle on a 1 M-read sample of the same batch."""
import numpy as np
import pytest

import oracle_bind as ob
from sickle_amd import capi

pytestmark = pytest.mark.gpu
N, L, STRIDE = 10_000_000, 150, 152


@pytest.fixture(scope="module")
def batch():
    import torch
    import bench
    dev = torch.device("cuda", 0)
    qual = bench.synth_quals_device(torch, N, L, STRIDE, 4321, dev)
    torch.cuda.synchronize()
    return torch, dev, qual


def scan(ctx, torch, dev, params, qual_t, n, stride=0, read_len=0, offsets_t=None):
    out = torch.empty((n, 2), dtype=torch.int32, device=dev)
    # the scan runs on the context's own stream: everything torch queued for its inputs must be done
    torch.cuda.synchronize(dev)
    ctx.scan_device_async(params, qual_t.data_ptr(), out.data_ptr(), n, stride=stride, read_len=read_len,
                          offsets_ptr=None if offsets_t is None else offsets_t.data_ptr())
    ctx.scan_device_finish()
    return out


def test_full_size_properties(sk_ctx, batch):
    torch, dev, qual = batch
    for q, l, x in ((20, 20, 0), (30, 60, 1)):
        p = capi.make_params("sanger", q, l, x, 0)
        a = scan(sk_ctx, torch, dev, p, qual, N, stride=STRIDE, read_len=L)
        # 1. the general kernel on the same reads packed back to back
        packed = qual[:, :L].contiguous()
        offsets = (torch.arange(N + 1, device=dev, dtype=torch.int64) * L)
        b = scan(sk_ctx, torch, dev, p, packed, N, offsets_t=offsets)
        assert bool((a == b).all()), "tiled and general kernels disagree"
        del packed, offsets, b
        # 2. position independence
        rev = torch.flip(qual, dims=[0]).contiguous()
        c = scan(sk_ctx, torch, dev, p, rev, N, stride=STRIDE, read_len=L)
        assert bool((torch.flip(c, dims=[0]) == a).all())
        del rev, c
        # 3. prefix (n not a multiple of 64)
        k = 7_654_321
        d = scan(sk_ctx, torch, dev, p, qual, k, stride=STRIDE, read_len=L)
        assert bool((d == a[:k]).all())
        # 4. the oracle on a slice from the middle
        lo, m = 3_333_333, 1_000_000
        host = qual[lo:lo + m].cpu().numpy().reshape(-1)
        want, err = ob.oracle_trim_batch(ob.make_params("sanger", q, l, x, 0), host, stride=STRIDE, read_len=L,
                                         n_reads=m, threads=8)
        assert err is None
        assert (a[lo:lo + m].cpu().numpy() == want).all()
        kept = int((a[:, 1] >= 0).sum())
        assert 0 < kept <= N


def test_full_size_trunc_n(sk_ctx, batch):
    """-n at full size: sequence tile through the same pipeline; tiled vs general kernel + oracle sample."""
    torch, dev, qual = batch
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    seq = torch.zeros((N, STRIDE), dtype=torch.uint8, device=dev)
    acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    for a0 in range(0, N, 2_000_000):
        m = min(2_000_000, N - a0)
        s = acgt[torch.randint(0, 4, (m, L), generator=g, device=dev)]
        r = torch.rand((m, L), generator=g, device=dev)
        s = torch.where(r < 0.002, torch.full_like(s, ord("N")), s)
        s = torch.where(r > 0.9997, torch.full_like(s, ord("n")), s)
        seq[a0:a0 + m, :L] = s
    p = capi.make_params("sanger", 20, 20, 0, 1)
    out = torch.empty((N, 2), dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    sk_ctx.scan_device_async(p, qual.data_ptr(), out.data_ptr(), N, stride=STRIDE, read_len=L, seq_ptr=seq.data_ptr())
    sk_ctx.scan_device_finish()
    pq, ps = qual[:, :L].contiguous(), seq[:, :L].contiguous()
    offsets = (torch.arange(N + 1, device=dev, dtype=torch.int64) * L)
    out2 = torch.empty((N, 2), dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    sk_ctx.scan_device_async(p, pq.data_ptr(), out2.data_ptr(), N, seq_ptr=ps.data_ptr(), offsets_ptr=offsets.data_ptr())
    sk_ctx.scan_device_finish()
    assert bool((out == out2).all())
    lo, m = 6_000_000, 500_000
    want, err = ob.oracle_trim_batch(ob.make_params("sanger", 20, 20, 0, 1), qual[lo:lo + m].cpu().numpy().reshape(-1),
                                     seq[lo:lo + m].cpu().numpy().reshape(-1), stride=STRIDE, read_len=L, n_reads=m, threads=8)
    assert err is None and (out[lo:lo + m].cpu().numpy() == want).all()
