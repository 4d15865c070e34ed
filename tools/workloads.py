"""Synthetic workloads of bench.py, seeded per GLOBAL READ BLOCK: read k of a job is the same bytes whichever
rank generates it, so an N-shard run of a job scans exactly the reads of the 1-GPU run (same kept / discarded).

  se    BASELINE configs[1] / [3]: fixed-length Sanger reads, the quality model of sickle_amd/synth.py
        (base Q ~ U[30,40], linear decay U[0,0.25] per base, N(0,4) noise, clip [2,41], a low 5' head of 0-7
        bases, 0.2 % of the bases N: quality 2)
  mixed BASELINE configs[4]: lengths U{75..301}, phred+64 (chars 66..105), 0.3 % N, 5 % of the reads with one
        lowercase n (scanned with -n, so the sequence bytes are part of the batch)

Everything is torch on the device handed in ("cuda" in bench.py; "cpu" in tests/test_shard_gloo.py -- the two
give different random streams, each consistent with itself).  Also here: the segmented layout of
include/sickle_amd.h built on the device (reads grouped by length into tiles of <= 64 equal-length rows)."""
import numpy as np

BLOCK = 1 << 16  # reads per seed block
MIX_LO, MIX_HI = 75, 301


def _gen(torch, device, seed, blk):
    g = torch.Generator(device=device)
    g.manual_seed(int(seed) * 1_000_003 + int(blk))
    return g


def _quality(torch, g, device, m, length, offset):
    pos = torch.arange(length, device=device, dtype=torch.float32)[None, :]
    base = 30.0 + 10.0 * torch.rand((m, 1), generator=g, device=device)
    decay = 0.25 * torch.rand((m, 1), generator=g, device=device)
    q = base - decay * pos + 4.0 * torch.randn((m, length), generator=g, device=device)
    q = q.round_().clamp_(2, 41)
    head = torch.randint(0, 8, (m, 1), generator=g, device=device)
    q = torch.where(pos < head, torch.full_like(q, 2.0), q)
    return q, pos


def se_block(torch, device, seed, blk, length, offset=33):
    """Block `blk` of a fixed-length job: (BLOCK, length) uint8 quality chars."""
    g = _gen(torch, device, seed, blk)
    q, _ = _quality(torch, g, device, BLOCK, length, offset)
    isn = torch.rand((BLOCK, length), generator=g, device=device) < 0.002
    q = torch.where(isn, torch.full_like(q, 2.0), q)
    return (q + float(offset)).to(torch.uint8)


def _blocks(lo, n):
    """(block, first row taken, one past the last row taken, destination row) for reads [lo, lo + n)."""
    out = []
    for blk in range(lo // BLOCK, (lo + n + BLOCK - 1) // BLOCK if n else lo // BLOCK):
        a, b = max(lo, blk * BLOCK), min(lo + n, (blk + 1) * BLOCK)
        if b > a:
            out.append((blk, a - blk * BLOCK, b - blk * BLOCK, a - lo))
    return out


def se_shard(torch, device, seed, lo, n, length, stride):
    """Reads [lo, lo + n) of the job as an (n, stride) uint8 matrix (rows zero-padded to the stride)."""
    out = torch.zeros((n, stride), dtype=torch.uint8, device=device)
    for blk, a, b, at in _blocks(lo, n):
        out[at:at + (b - a), :length] = se_block(torch, device, seed, blk, length)[a:b]
    return out


def mixed_block(torch, device, seed, blk):
    """Block `blk` of a mixed-length job: lens (BLOCK,) int64, qual and seq (BLOCK, MIX_HI) uint8; the bytes
    beyond a read's length are not part of it."""
    g = _gen(torch, device, seed, blk)
    m = BLOCK
    lens = torch.randint(MIX_LO, MIX_HI + 1, (m,), generator=g, device=device)
    q, pos = _quality(torch, g, device, m, MIX_HI, 64)
    seq = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)[torch.randint(0, 4, (m, MIX_HI), generator=g, device=device)]
    isn = torch.rand((m, MIX_HI), generator=g, device=device) < 0.003
    seq = torch.where(isn, torch.full_like(seq, ord("N")), seq)
    low = torch.rand((m,), generator=g, device=device) < 0.05
    at = (torch.rand((m,), generator=g, device=device) * lens).long().clamp_(max=MIX_HI - 1)
    lown = low[:, None] & (pos.long() == at[:, None])
    seq = torch.where(lown, torch.full_like(seq, ord("n")), seq)
    q = torch.where(isn | lown, torch.full_like(q, 2.0), q)
    return lens, (q + 64.0).to(torch.uint8), seq


def mixed_shard(torch, device, seed, lo, n):
    """Reads [lo, lo + n) of the mixed job: lens (n,), qual (n, MIX_HI), seq (n, MIX_HI)."""
    lens = torch.zeros((n,), dtype=torch.int64, device=device)
    qual = torch.zeros((n, MIX_HI), dtype=torch.uint8, device=device)
    seq = torch.zeros((n, MIX_HI), dtype=torch.uint8, device=device)
    for blk, a, b, at in _blocks(lo, n):
        l, q, s = mixed_block(torch, device, seed, blk)
        lens[at:at + (b - a)] = l[a:b]
        qual[at:at + (b - a)] = q[a:b]
        seq[at:at + (b - a)] = s[a:b]
    return lens, qual, seq


def tile_stride(length):
    """Row stride of a tile of `length`-byte reads: a multiple of 8 with an odd number of 8-byte units."""
    return ((int(length) + 7) // 8 | 1) * 8


def seg_layout(lens_counts):
    """[(length, count), ...] in tile order -> (tiles as numpy TILE_DTYPE, total bytes, reads): every length
    class at its own bank-friendly stride, classes 16-byte aligned, tiles of <= 64 rows."""
    from sickle_amd.capi import TILE_DTYPE
    tl, at, slot = [], 0, 0
    for L, cnt in lens_counts:
        if cnt == 0:
            continue
        st = tile_stride(L)
        at = (at + 15) & ~15
        a0 = np.arange(0, cnt, 64, dtype=np.int64)
        t = np.zeros(len(a0), dtype=TILE_DTYPE)
        t["byte_off"] = at + a0 * st
        t["slot0"] = slot + a0
        t["stride"] = st
        t["rows"] = np.minimum(64, cnt - a0)
        t["read_len"] = L
        tl.append(t)
        at += cnt * st
        slot += cnt
    if not tl:
        return np.zeros(0, dtype=TILE_DTYPE), 0, 0
    return np.concatenate(tl), at, slot


def segment(torch, lens, qual, seq=None):
    """The segmented layout of a batch given as row matrices (n, >= longest) + lens (n,): reads grouped by length
    (stable, so out_index is the counting sort the CLI's packer does) -> dict(q, seq, tiles, out_index, max_stride,
    bases); q / seq are flat uint8 device tensors with 4 KiB of slack behind the last tile."""
    dev = lens.device
    n = int(lens.numel())
    order = torch.argsort(lens, stable=True)
    counts = torch.bincount(lens, minlength=1).cpu().numpy()
    present = [(L, int(c)) for L, c in enumerate(counts) if c]
    tiles, nbytes, nreads = seg_layout(present)
    assert nreads == n
    q = torch.zeros((nbytes + 4096,), dtype=torch.uint8, device=dev)
    s = torch.zeros((nbytes + 4096,), dtype=torch.uint8, device=dev) if seq is not None else None
    at, first = 0, 0
    for L, cnt in present:
        st = tile_stride(L)
        at = (at + 15) & ~15
        rows = order[first:first + cnt]
        q[at:at + cnt * st].view(cnt, st)[:, :L] = qual[rows, :L]
        if s is not None:
            s[at:at + cnt * st].view(cnt, st)[:, :L] = seq[rows, :L]
        at += cnt * st
        first += cnt
    return {"q": q, "seq": s, "tiles": tiles, "out_index": order.to(torch.int32), "n": n,
            "max_stride": int(tiles["stride"].max()) if n else 8, "bases": int(lens.sum().item()), "bytes": nbytes}
