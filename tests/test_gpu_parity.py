"""GPU parity: the HIP kernels, called through the C ABI (libsickle_amd.so), against the
oracle on the same inputs -- bit-exact (integer work)."""
import json
import os

import numpy as np
import pytest

import oracle_bind as ob
from sickle_amd import capi, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def key(qt, q, l, x, n):
    return "%s_q%d_l%d_x%d_n%d" % (qt, q, l, int(x), int(n))


def parse_key(k):
    qt, q, l, x, n = k.split("_")
    return qt, int(q[1:]), int(l[1:]), int(x[1:]), int(n[1:])


def both_params(qt, q, l, x, n):
    return capi.make_params(qt, q, l, x, n), ob.make_params(qt, q, l, x, n)


def pad_rows(seq, qual, offsets, stride):
    """ragged -> fixed stride (+ lengths)"""
    n = len(offsets) - 1
    lens = np.diff(offsets).astype(np.uint32)
    qs = np.zeros((n, stride), dtype=np.uint8)
    ss = np.zeros((n, stride), dtype=np.uint8)
    for i in range(n):
        a, b = int(offsets[i]), int(offsets[i + 1])
        qs[i, :b - a] = qual[a:b]
        ss[i, :b - a] = seq[a:b]
    return ss.reshape(-1), qs.reshape(-1), lens


def test_kernel_selection():
    q = np.zeros(4096, dtype=np.uint8)  # numpy data is 16/64-byte aligned
    assert q.ctypes.data % 16 == 0
    b = capi.Batch(q.ctypes.data, None, None, 152, 150, None, 10)
    assert capi.lib().sk_kernel_for(b) == 4  # equal lengths, no -n, rows of 72..160 bytes: register-staged tiles
    b = capi.Batch(q.ctypes.data, q.ctypes.data, None, 152, 150, None, 10)
    assert capi.lib().sk_kernel_for(b) == 1  # with a sequence buffer (-n): LDS-DMA tiles
    b = capi.Batch(q.ctypes.data, None, None, 264, 250, None, 10)
    assert capi.lib().sk_kernel_for(b) == 4  # rows up to 320 bytes too (20 pieces)
    b = capi.Batch(q.ctypes.data, None, None, 328, 325, None, 10)
    assert capi.lib().sk_kernel_for(b) == 1  # longer rows: LDS-DMA tiles
    assert capi.lib().sk_kernel_name(4) == b"sk_scan_tile_staged_kernel"
    b = capi.Batch(q.ctypes.data, None, None, 150, 150, None, 10)
    assert capi.lib().sk_kernel_for(b) == 5  # packed rows (stride not a multiple of 8): tiles with rows at any address
    off = np.zeros(2, dtype=np.uint64)
    b = capi.Batch(q.ctypes.data, None, off.ctypes.data, 0, 0, None, 1)
    assert capi.lib().sk_kernel_for(b) == 5  # ragged: the same kernel, per-lane lengths
    b = capi.Batch(q.ctypes.data, None, None, 600, 600, None, 10)
    assert capi.lib().sk_kernel_for(b) == 8  # uniform medium reads: tiles of 32 / 16 reads, windows of any width on the matrix path
    assert capi.lib().sk_kernel_name(8) == b"sk_scan_tile_wide_kernel"
    b = capi.Batch(q.ctypes.data, None, None, 2600, 2600, None, 10)
    assert capi.lib().sk_kernel_for(b) == 2  # beyond its range: general kernel (teams of 16 lanes)
    b = capi.Batch(q.ctypes.data, None, None, 1 << 24, 1000, None, 2)
    assert capi.lib().sk_kernel_for(b) == 2  # rows 16 MiB apart: beyond the loader's 24-bit row offsets
    assert capi.lib().sk_kernel_name(7) == b"sk_scan_band_kernel"
    assert capi.lib().sk_kernel_name(5) == b"sk_scan_tile_any_kernel"
    b = capi.Batch(q.ctypes.data, None, None, 5000, 5000, None, 10)
    assert capi.lib().sk_kernel_for(b) == 6  # longer than 4096: the streaming general kernel
    b = capi.Batch(q.ctypes.data, None, off.ctypes.data, 30_000, 0, None, 1)
    assert capi.lib().sk_kernel_for(b) == 6  # ragged with a longest-read hint beyond 4096: a long-read batch
    b = capi.Batch(q.ctypes.data, None, off.ctypes.data, 301, 0, None, 1)
    assert capi.lib().sk_kernel_for(b) == 5
    assert capi.lib().sk_kernel_name(6) == b"sk_scan_stream_kernel"


@pytest.mark.gpu
def test_segmented_tiles_staged_through_registers(sk_ctx):
    """SK_SEG_STAGE=1 (read once per process, so a child): the register-staged variant of the segmented kernel --
    built, not the default -- through tests/soak_tiles.py."""
    import subprocess, sys as _sys
    env = dict(os.environ, SK_SEG_STAGE="1")
    r = subprocess.run([_sys.executable, os.path.join(os.path.dirname(__file__), "soak_tiles.py"), "150", "31"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "soak ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([_sys.executable, os.path.join(os.path.dirname(__file__), "seg_many_tiles.py")], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "seg many tiles ok: 4 scans" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_segmented_batch_many_tiles_per_wave(sk_ctx):
    """tests/seg_many_tiles.py with the default (LDS-DMA) segmented kernel: 300 000 reads of 75..301 bases in > 4 096 tiles."""
    import seg_many_tiles
    assert seg_many_tiles.run() == 4


@pytest.mark.gpu
def test_uniform_medium_reads_soak(sk_ctx):
    """tests/soak_wide.py: random uniform batches of 320 .. 2600 bases (edge lengths: multiples of 32 and of 10, the
    switch from 32- to 16-read tiles, the end of the kernel's range), any stride, 1 .. 300 reads, every parameter, a
    planted range error in every fourth -- host and device entry points against the oracle.  The kernel the library
    selects must have been the medium-read tile kernel (8) among them."""
    import soak_wide
    checked, kernels = soak_wide.run(150, 2027, verbose=False)
    assert checked == 300 and 8 in kernels


@pytest.mark.gpu
@pytest.mark.parametrize("L,n", [(600, 150_000), (1000, 120_000), (1201, 100_000), (1500, 80_000)])
def test_uniform_medium_reads_many_tiles_per_wave(sk_ctx, L, n):
    """More tiles than the launch has waves (2 304 .. 4 096): every wave goes round its loop several times -- refill
    after the scan, the next tile's first step behind the last one's state -- which tests/soak_wide.py's small batches
    never do.  Device-resident, with and without -n, 3' declines at random places, some reads bad from the start, some
    good to the end; against the oracle."""
    import torch
    rng = np.random.default_rng(L)
    q = rng.integers(60, 74, size=(n, L), dtype=np.uint8)
    cut = rng.integers(0, L + L // 4, size=n)
    cols = np.arange(L)[None, :]
    q[cols >= cut[:, None]] -= 25
    head = rng.integers(0, 60, size=n)
    q[(cols < head[:, None]) & (rng.random(n) < 0.3)[:, None]] = 35
    seq = rng.choice(np.frombuffer(b"ACGT" * 2000 + b"Nn", dtype=np.uint8), size=(n, L))
    offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    dq, ds = torch.from_numpy(q.reshape(-1)).cuda(), torch.from_numpy(seq.reshape(-1)).cuda()
    out = torch.empty((n, 2), dtype=torch.int32, device="cuda")
    assert capi.lib().sk_kernel_for(capi.Batch(dq.data_ptr(), None, None, L, L, None, n)) == 8
    for tn in (False, True):
        p, po = both_params("sanger", 20, 20, False, tn)
        want, err = ob.oracle_trim_batch(po, q.reshape(-1), seq.reshape(-1), offsets=offs, threads=8)
        assert err is None
        out.fill_(-7)
        sk_ctx.scan_device_async(p, dq.data_ptr(), out.data_ptr(), n, stride=L, read_len=L, seq_ptr=ds.data_ptr() if tn else None)
        sk_ctx.scan_device_finish()
        got = out.cpu().numpy()
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, (L, tn, bad[:5], got[bad[:5]], want[bad[:5]])


@pytest.mark.gpu
@pytest.mark.parametrize("rows,stage", [("16", "1"), ("16", "0"), ("32", "1"), ("32", "0")])
def test_uniform_medium_reads_both_tile_heights(sk_ctx, rows, stage, monkeypatch):
    """SK_WIDE_ROWS forces 16- or 32-read tiles for every length the medium-read kernel takes; with 16, the next tile
    waits in registers wherever it can (SK_WIDE_STAGE=0: nowhere)."""
    monkeypatch.setenv("SK_WIDE_ROWS", rows)
    monkeypatch.setenv("SK_WIDE_STAGE", stage)
    import subprocess, sys as _sys
    r = subprocess.run([_sys.executable, os.path.join(os.path.dirname(__file__), "soak_wide.py"), "80", "77"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "soak ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["band", "team", "stream"])
def test_both_general_kernels_on_every_length(sk_ctx, which, monkeypatch):
    """SK_GENERAL forces one of the general kernels (a wave per read with the read resident and its window sums from
    the matrix pipe / round 2's teams of 16 lanes / a wave per read, the
    read streamed through a ring of 1 KiB blocks) for every batch that takes the general path: all must agree
    with the oracle on lengths from 1 to 70 kb -- around the block size (1023..1025, 2047..2049), multiples of
    1024 (the empty last block), 16 k + 0..15 (the partial last chunk), windows of every residue modulo 16 --
    with cuts at the very start, in the middle and in the last windows, device-resident ragged (all reads, spans of
    equal cost), through sk_submit (the host counts the tiles that fit), as left-overs behind the tile kernel, and
    as uniform fixed-stride batches; with -x, -n and a range error."""
    monkeypatch.setenv("SK_GENERAL", which)
    rng = np.random.default_rng(4242)
    lens = [1, 2, 9, 10, 15, 16, 17, 19, 20, 31, 100, 159, 160, 161, 1023, 1024, 1025, 2047, 2048, 2049, 3072, 4096, 5000,
            10_240, 10_250, 16_384, 16_385, 16_399, 20_480, 30_000, 30_720, 69_999]
    lens = np.array(lens + list(rng.integers(1, 12_000, size=96)), dtype=np.uint32)
    rng.shuffle(lens)
    n = len(lens)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    tot = int(offs[-1])
    qual = np.clip(rng.normal(62, 6, tot).astype(int), 33, 74).astype(np.uint8)
    for i in range(n):
        a, b = int(offs[i]), int(offs[i + 1])
        mode = i % 5
        if mode == 0:    # bad start, good rest: the 5' cut moves in
            qual[a:a + (b - a) // 3] = np.clip(rng.normal(40, 4, (b - a) // 3).astype(int), 33, 74)
        elif mode == 1:  # collapse somewhere: the 3' cut
            c = a + int(rng.integers(0, b - a))
            qual[c:b] = np.clip(rng.normal(40, 4, b - c).astype(int), 33, 74)
        elif mode == 2:  # collapse only in the last few bases: the windows after the last aligned one
            c = max(a, b - int(rng.integers(1, 40)))
            qual[c:b] = 35
        elif mode == 3:  # hovering at the threshold
            qual[a:b] = 53 + rng.integers(-1, 2, size=b - a)
    seq = rng.choice(np.frombuffer(b"ACGT" * 400 + b"Nn", dtype=np.uint8), size=tot)
    for q, l, x, tn in ((20, 20, 0, 0), (20, 0, 1, 0), (22, 100, 0, 1)):
        p, po = both_params("sanger", q, l, x, tn)
        want, err = ob.oracle_trim_batch(po, qual, seq, offsets=offs, threads=4)
        assert err is None
        got = sk_ctx.trim_batch(p, qual, seq, offsets=offs)  # sk_submit: host offsets
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, ("submit", q, l, x, tn, bad[:5], got[bad[:5]], want[bad[:5]], lens[bad[:5]])
        import torch
        dq, ds, do = (torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v).cuda() for v in (qual, seq, offs))
        out = torch.empty((n, 2), dtype=torch.int32, device="cuda")
        for hint in (0, 70_000, 2000):  # left-overs behind the tile kernel / all reads by the general kernel / a hint too small
            out.fill_(-7)
            sk_ctx.scan_device_async(p, dq.data_ptr(), out.data_ptr(), n, offsets_ptr=do.data_ptr(), stride=hint,
                                     seq_ptr=ds.data_ptr() if tn else None)
            sk_ctx.scan_device_finish()
            got = out.cpu().numpy()
            bad = np.nonzero((got != want).any(axis=1))[0]
            assert bad.size == 0, ("device", hint, q, l, x, tn, bad[:5], got[bad[:5]], want[bad[:5]], lens[bad[:5]])
    # a char out of range: reported iff the reference would have read it (before the 3' break + window), lowest read first
    p, po = both_params("sanger", 20, 20, 0, 0)
    for trial in range(12):
        q2 = qual.copy()
        r = int(rng.integers(0, n))
        pos = int(offs[r]) + int(rng.integers(0, lens[r]))
        q2[pos] = rng.choice([10, 32, 127, 200])
        want, err = ob.oracle_trim_batch(po, q2, seq, offsets=offs, threads=1)
        try:
            got = sk_ctx.trim_batch(p, q2, seq, offsets=offs)
            assert err is None, (trial, r, err, "device missed the error")
            assert (got == want).all()
        except capi.RangeError as e:
            assert err is not None and (e.read, e.pos, e.ch) == tuple(err), (trial, r, err, (e.read, e.pos, e.ch))
        # the same through the device-resident entry points: all reads by the general kernel / left-overs of the tile kernel
        dq2 = torch.from_numpy(q2).cuda()
        for hint in (70_000, 0):
            out.fill_(-7)
            sk_ctx.scan_device_async(p, dq2.data_ptr(), out.data_ptr(), n, offsets_ptr=do.data_ptr(), stride=hint)
            try:
                sk_ctx.scan_device_finish()
                assert err is None, (trial, hint, r, err, "device missed the error")
                assert (out.cpu().numpy() == want).all()
            except capi.RangeError as e:
                assert err is not None and (e.read, e.pos, e.ch) == tuple(err), (trial, hint, err, (e.read, e.pos, e.ch))
    # a long-read batch that comes WITHOUT a hint: the tile kernel leaves every tile and counts them, the general
    # kernel then cuts the batch into spans of equal cost as if the hint had been there
    lens2 = rng.integers(2100, 9000, size=300).astype(np.uint64)
    offs2 = np.zeros(301, dtype=np.uint64)
    offs2[1:] = np.cumsum(lens2)
    q3 = np.clip(rng.normal(60, 7, int(offs2[-1])).astype(int), 33, 74).astype(np.uint8)
    for i in range(300):
        q3[int(offs2[i]) + int(rng.integers(0, lens2[i])):int(offs2[i + 1])] = 38
    want3, _ = ob.oracle_trim_batch(po, q3, None, offsets=offs2, threads=4)
    dq3, do3 = torch.from_numpy(q3).cuda(), torch.from_numpy(offs2.view(np.int64)).cuda()
    out3 = torch.empty((300, 2), dtype=torch.int32, device="cuda")
    for hint in (0, 9000):
        out3.fill_(-7)
        sk_ctx.scan_device_async(p, dq3.data_ptr(), out3.data_ptr(), 300, offsets_ptr=do3.data_ptr(), stride=hint)
        sk_ctx.scan_device_finish()
        assert (out3.cpu().numpy() == want3).all(), hint
    # uniform fixed-stride batches (all reads, equal numbers of reads per wave)
    for L in (1024, 3000, 8192, 12_345):
        m = 50
        qm = np.clip(rng.normal(58, 8, size=(m, L)).astype(int), 33, 74).astype(np.uint8)
        qm[:, L - L // 4:] = 36
        want, _ = ob.oracle_trim_batch(po, qm.reshape(-1), stride=L, read_len=L, n_reads=m)
        got = sk_ctx.trim_batch(p, qm.reshape(-1), stride=L, read_len=L, n_reads=m)
        assert (got == want).all(), L


@pytest.mark.parametrize("layout", ["tile", "wave_ragged", "wave_stride"])
def test_golden_edge_set(sk_ctx, layout):
    """Every golden grid point of the edge-case reads (lengths 1..31, 99-101, 149-151, 250, 301)."""
    inp = np.load(os.path.join(GOLD, "edge_inputs.npz"))
    seq, qual, offsets = inp["seq"], inp["qual"], inp["offsets"]
    cuts = np.load(os.path.join(GOLD, "cuts_edge.npz"))
    stride = 304 if layout == "tile" else 301
    ss, qs, lens = pad_rows(seq, qual, offsets, stride)
    for k in cuts.files:
        p, _ = both_params(*parse_key(k))
        if layout == "wave_ragged":
            got = sk_ctx.trim_batch(p, qual, seq, offsets=offsets)
        else:
            got = sk_ctx.trim_batch(p, qs, ss, stride=stride, lengths=lens)
        want = cuts[k].astype(np.int32)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, (layout, k, bad[:5], got[bad[:5]], want[bad[:5]], lens[bad[:5]])


@pytest.mark.parametrize("layout", ["tile", "wave_ragged"])
def test_golden_bundled_file(sk_ctx, layout):
    """The reference's own test/test.fastq over the golden flag grid."""
    from fastq_util import parse_fastq, pack_records
    recs = parse_fastq(open(os.path.join(GOLD, "inputs", "test.fastq"), "rb").read())
    seq, qual, offsets = pack_records(recs)
    cuts = np.load(os.path.join(GOLD, "cuts_bundled.npz"))
    ss, qs, lens = pad_rows(seq, qual, offsets, 152)
    assert (lens == 150).all()
    for k in cuts.files:
        p, _ = both_params(*parse_key(k))
        if layout == "tile":
            got = sk_ctx.trim_batch(p, qs, ss, stride=152, read_len=150, n_reads=len(recs))  # uniform variant
        else:
            got = sk_ctx.trim_batch(p, qual, seq, offsets=offsets)
        assert (got == cuts[k].astype(np.int32)).all(), (layout, k)


@pytest.mark.parametrize("qt,trunc_n", [("sanger", False), ("sanger", True), ("illumina", True)])
def test_synthetic_fixed_vs_oracle(sk_ctx, qt, trunc_n):
    """The bench workload's shape (150 bp, stride 152) at a size the oracle does in a second."""
    n = 200_003  # not a multiple of 64: exercises the partial last tile
    seq, qual = synth.make_reads(77, n, 150, qt, lower_n_frac=0.01 if trunc_n else 0.0)
    qs, ss = synth.pack_fixed(qual, 152), synth.pack_fixed(seq, 152)
    for q, l, x in ((20, 20, 0), (30, 50, 1), (2, 0, 0)):
        p, po = both_params(qt, q, l, x, trunc_n)
        want, err = ob.oracle_trim_batch(po, qs, ss, stride=152, read_len=150, n_reads=n, threads=8)
        assert err is None
        got = sk_ctx.trim_batch(p, qs, ss, stride=152, read_len=150, n_reads=n)
        assert (got == want).all(), (qt, q, l, x, np.nonzero((got != want).any(axis=1))[0][:5])
        # the general kernel on the unpadded layout must agree too
        got2 = sk_ctx.trim_batch(p, qual.reshape(-1), seq.reshape(-1), stride=150, read_len=150, n_reads=n)
        assert (got2 == want).all()


def test_synthetic_ragged_vs_oracle(sk_ctx):
    seq, qual, offsets = synth.make_ragged_reads(5, 50_000, 75, 301, "illumina")
    for q, l, x, n in ((20, 20, 0, 1), (25, 75, 0, 0), (20, 20, 1, 1)):
        p, po = both_params("illumina", q, l, x, n)
        want, err = ob.oracle_trim_batch(po, qual, seq, offsets=offsets, threads=8)
        assert err is None
        got = sk_ctx.trim_batch(p, qual, seq, offsets=offsets)
        assert (got == want).all()
        ss, qs, lens = pad_rows(seq, qual, offsets, 304)
        got = sk_ctx.trim_batch(p, qs, ss, stride=304, lengths=lens)
        assert (got == want).all()


def test_random_fuzz_all_layouts(sk_ctx):
    """Adversarial quality patterns, all encodings, thresholds 0..45, lengths 1..320."""
    rng = np.random.default_rng(11)
    for trial in range(60):
        qt = ["sanger", "solexa", "illumina"][trial % 3]
        lo, hi = {"sanger": (33, 74), "solexa": (58, 105), "illumina": (64, 105)}[qt]
        n = 3000
        lens = rng.integers(1, 321, size=n).astype(np.uint32)
        if trial % 5 == 0:
            lens[:] = rng.integers(1, 40)
        offs = np.zeros(n + 1, dtype=np.uint64)
        offs[1:] = np.cumsum(lens)
        tot = int(offs[-1])
        mode = trial % 4
        if mode == 0:
            qual = rng.integers(lo, hi, size=tot)
        elif mode == 1:
            qual = np.where(rng.random(tot) < 0.5, lo, hi - 1)
        elif mode == 2:
            qual = np.clip(rng.normal((lo + hi) / 2 + rng.integers(-10, 10), 8, tot).astype(int), lo, hi - 1)
        else:
            off = {"sanger": 33, "solexa": 64, "illumina": 64}[qt]
            qual = np.clip(off + rng.integers(0, 45) + rng.integers(-3, 4, size=tot), lo, hi - 1)
        qual = qual.astype(np.uint8)
        seq = rng.choice(np.frombuffer(b"ACGTACGTACGTACGTNn" if trial % 2 else b"ACGT" * 50 + b"N", dtype=np.uint8), size=tot)
        args = (qt, int(rng.integers(0, 45)), int(rng.integers(0, 120)) if trial % 3 else 20, trial % 2, (trial // 2) % 2)
        p, po = both_params(*args)
        want, err = ob.oracle_trim_batch(po, qual, seq, offsets=offs, threads=4)
        assert err is None
        got = sk_ctx.trim_batch(p, qual, seq, offsets=offs)
        assert (got == want).all(), ("ragged", trial, args)
        ss, qs, l32 = pad_rows(seq, qual, offs, 320)
        got = sk_ctx.trim_batch(p, qs, ss, stride=320, lengths=l32)
        assert (got == want).all(), ("tile", trial, args, np.nonzero((got != want).any(axis=1))[0][:5])


def test_range_errors_match_reference(sk_ctx):
    """errors.json: for each case the device either returns the reference's cut or reports the
    same (position, char) the reference printed before exit(1)."""
    cases = json.load(open(os.path.join(GOLD, "errors.json")))
    for c in cases:
        pd = c["params"]
        p = capi.make_params(pd["qualtype"], pd["q"], pd["l"], pd["no5"], pd["trunc_n"])
        po = ob.make_params(pd["qualtype"], pd["q"], pd["l"], pd["no5"], pd["trunc_n"])
        qual = np.frombuffer(bytes.fromhex(c["qual_hex"]), dtype=np.uint8)
        seq = np.frombuffer(c["seq"].encode("latin-1"), dtype=np.uint8)
        L = len(qual)
        for layout in ("tile", "wave"):
            # put the read in the middle of a batch of clean reads
            stride = 152 if layout == "tile" else 150
            n = 130
            qs = np.full((n, stride), qual[0] if c["rc"] == 0 else 75, dtype=np.uint8)
            qs[:] = 75  # 'K': legal in all three encodings
            ss = np.full((n, stride), ord("A"), dtype=np.uint8)
            lens = np.full(n, 150, dtype=np.uint32)
            qs[70, :L] = qual
            ss[70, :L] = seq
            lens[70] = L
            try:
                got = sk_ctx.trim_batch(p, qs.reshape(-1), ss.reshape(-1), stride=stride, lengths=lens)
                assert c["rc"] == 0, (c["desc"], layout, "device missed the error")
                assert list(got[70]) == c["cut"], (c["desc"], layout, got[70])
            except capi.RangeError as e:
                assert c["rc"] == 1, (c["desc"], layout, "device raised a spurious error")
                assert e.read == 70
                text = ob.oracle_format_error(po, c["name"].encode("latin-1"), qual.tobytes(), (e.read, e.pos, e.ch))
                assert text.decode("latin-1") == c["stderr"], (c["desc"], layout)


def test_error_reports_lowest_read(sk_ctx):
    n = 5000
    qs = np.full((n, 152), 75, dtype=np.uint8)
    qs[4000, 10] = 10
    qs[1234, 100] = 200
    qs[1234, 120] = 7
    qs[3000, 0] = 0
    p = capi.make_params("sanger")
    with pytest.raises(capi.RangeError) as ei:
        sk_ctx.trim_batch(p, qs.reshape(-1), stride=152, read_len=150, n_reads=n)
    assert (ei.value.read, ei.value.pos, ei.value.ch) == (1234, 100, 200 - 256)


def test_empty_and_tiny_batches(sk_ctx):
    p = capi.make_params("sanger")
    assert sk_ctx.trim_batch(p, np.zeros(0, dtype=np.uint8), stride=152, read_len=150, n_reads=0).shape == (0, 2)
    for n in (1, 63, 64, 65, 255, 257):
        seq, qual = synth.make_reads(n, n, 150)
        want, _ = ob.oracle_trim_batch(ob.make_params("sanger"), synth.pack_fixed(qual, 152), stride=152, read_len=150, n_reads=n)
        got = sk_ctx.trim_batch(p, synth.pack_fixed(qual, 152), stride=152, read_len=150, n_reads=n)
        assert (got == want).all()
        # the same reads packed back to back and as a ragged batch (re-strided tiles: partial last tile,
        # the last chunks of the buffer copied byte by byte)
        got = sk_ctx.trim_batch(p, qual.reshape(-1), stride=150, read_len=150, n_reads=n)
        assert (got == want).all(), ("packed", n)
        offs = (np.arange(n + 1, dtype=np.uint64) * 150)
        got = sk_ctx.trim_batch(p, qual.reshape(-1), offsets=offs)
        assert (got == want).all(), ("ragged", n)
    # ragged batches of empty and one-byte reads
    offs = np.array([0, 0, 1, 1, 2, 5, 5], dtype=np.uint64)
    q = np.array([70, 40, 70, 70, 70], dtype=np.uint8)
    want, _ = ob.oracle_trim_batch(ob.make_params("sanger", 20, 0), q, offsets=offs)
    got = sk_ctx.trim_batch(capi.make_params("sanger", 20, 0), q, offsets=offs)
    assert (got == want).all(), (got, want)


def test_async_slots_overlap(sk_ctx):
    """sk_submit / sk_wait with two slots in flight give the same cuts as the synchronous call."""
    p, po = both_params("sanger", 20, 20, 0, 0)
    batches = []
    for i in range(4):
        _, qual = synth.make_reads(100 + i, 40_000, 150)
        qs = synth.pack_fixed(qual, 152)
        batches.append((qs, np.zeros((40_000, 2), dtype=np.int32)))
    for i, (qs, out) in enumerate(batches):
        if i >= 2:
            sk_ctx.wait(i % 2)
        sk_ctx.submit(i % 2, p, qs, out, stride=152, read_len=150, n_reads=40_000)
    sk_ctx.wait(0)
    sk_ctx.wait(1)
    for qs, out in batches:
        want, _ = ob.oracle_trim_batch(po, qs, stride=152, read_len=150, n_reads=40_000, threads=4)
        assert (out == want).all()


def test_trunc_n_adversarial_sequences(sk_ctx):
    """Sequence bytes chosen to break byte-parallel n/N searches: o/O and m/M next to n/N, n and N in
    every order and alignment, bytes >= 0x80."""
    rng = np.random.default_rng(5)
    n, L = 20_000, 150
    _, qual = synth.make_reads(8, n, L, "sanger")
    alphabet = np.frombuffer(b"ACGTACGTACGTNnoOmMpP\x6f\xee\xce", dtype=np.uint8)
    seq = alphabet[rng.integers(0, len(alphabet), size=(n, L))]
    seq[rng.random((n, L)) < 0.93] = ord("A")
    for trial, (stride, lens) in enumerate(((152, None), (160, rng.integers(1, 151, size=n).astype(np.uint32)))):
        qs, ss = synth.pack_fixed(qual, stride), synth.pack_fixed(seq, stride)
        p, po = both_params("sanger", 20, 20, 0, 1)
        kw = dict(stride=stride, n_reads=n)
        if lens is None:
            kw["read_len"] = L
        else:
            kw["lengths"] = lens
        want, err = ob.oracle_trim_batch(po, qs, ss, threads=4, **kw)
        assert err is None
        got = sk_ctx.trim_batch(p, qs, ss, **kw)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, (trial, bad[:5], got[bad[:5]], want[bad[:5]], bytes(seq[bad[0]]))


def test_long_reads_and_tile_boundary(sk_ctx):
    """Lengths around the tiled kernel's limit (stride 504 / 512) and long reads (up to 30 kb)
    through the general kernel, all against the oracle."""
    rng = np.random.default_rng(21)
    lens = np.array([329, 339, 340, 341, 400, 496, 503, 504, 505, 511, 512, 513, 1000, 4097, 30_000] * 8, dtype=np.uint32)
    rng.shuffle(lens)
    n = len(lens)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    tot = int(offs[-1])
    qual = np.clip(rng.normal(60, 9, tot).astype(int), 33, 74).astype(np.uint8)
    # long stretches of low quality so that 3' cuts land deep inside long reads
    for i in range(n):
        a, b = int(offs[i]), int(offs[i + 1])
        cut = a + int(rng.integers(0, b - a))
        qual[cut:b] = np.clip(rng.normal(40, 6, b - cut).astype(int), 33, 74)
    seq = rng.choice(np.frombuffer(b"ACGT" * 200 + b"Nn", dtype=np.uint8), size=tot)
    for q, l, x, tn in ((20, 20, 0, 0), (25, 100, 1, 1), (18, 0, 0, 1)):
        p, po = both_params("sanger", q, l, x, tn)
        want, err = ob.oracle_trim_batch(po, qual, seq, offsets=offs, threads=4)
        assert err is None
        got = sk_ctx.trim_batch(p, qual, seq, offsets=offs)
        assert (got == want).all(), (q, l, x, tn, np.nonzero((got != want).any(axis=1))[0][:5])
    # uniform batches right at the tiled limit: 504 (tiled, stride 504) and 505 (general kernel)
    for L in (339, 340, 504, 505):
        m = 1000
        _, qm = synth.make_reads(L, m, L, "sanger")
        stride = L if L == 505 else ((L + 7) // 8 | 1) * 8
        qs = synth.pack_fixed(qm, stride)
        p, po = both_params("sanger", 20, 20, 0, 0)
        want, _ = ob.oracle_trim_batch(po, qs, stride=stride, read_len=L, n_reads=m)
        got = sk_ctx.trim_batch(p, qs, stride=stride, read_len=L, n_reads=m)
        assert (got == want).all(), L


def test_uniform_length_sweep(sk_ctx):
    """Every uniform read length 1..345 and every 7th up to the tiled kernel's limit of 504 (two
    32-position blocks per MFMA chain up to w = 33 / L = 339, three beyond), at the stride the host
    packer would pick, three encodings, with and without -x / -n, against the oracle."""
    rng = np.random.default_rng(314)
    for L in list(range(1, 346)) + list(range(346, 505, 7)) + [500, 503, 504]:
        n = 193  # three tiles + a partial one
        qt = ("sanger", "illumina", "solexa")[L % 3]
        lo, hi = {"sanger": (33, 74), "solexa": (59, 105), "illumina": (64, 105)}[qt]
        mid = int(rng.integers(lo + 5, hi - 5))
        qual = np.clip(mid + rng.integers(-14, 15, size=(n, L)), lo, hi).astype(np.uint8)
        drop = rng.integers(0, L + 1, size=n)
        qual[np.arange(L)[None, :] >= drop[:, None]] = lo + 1  # a 3' collapse somewhere
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, L))].copy()
        seq[rng.random((n, L)) < 0.01] = ord("N")
        seq[rng.random((n, L)) < 0.003] = ord("n")
        stride = ((L + 7) // 8 | 1) * 8
        qs, ss = synth.pack_fixed(qual, stride), synth.pack_fixed(seq, stride)
        q = int(rng.integers(0, 42))
        l = int(rng.integers(0, max(1, L)))
        x, tn = L % 2, (L // 2) % 2
        p, po = both_params(qt, q, l, x, tn)
        want, err = ob.oracle_trim_batch(po, qs, ss, stride=stride, read_len=L, n_reads=n)
        assert err is None
        got = sk_ctx.trim_batch(p, qs, ss, stride=stride, read_len=L, n_reads=n)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, (L, qt, q, l, x, tn, bad[:4], got[bad[:4]], want[bad[:4]])


def test_segmented_batches(sk_ctx):
    """Mixed lengths grouped by length into per-tile descriptors (the layout the CLI uses for
    mixed-length files): the tiled kernel's matrix path tile by tile, cuts scattered back to the
    caller's order, against the oracle on the ragged original.  Includes -n, -x, lengths below
    -l, and an out-of-range char whose read index must come back in the caller's numbering."""
    from fastq_util import segment_by_length
    seq, qual, offsets = synth.make_ragged_reads(17, 30_000, 1, 330, "illumina", lower_n_frac=0.02)
    ss, qs, tiles, out_index, max_stride = segment_by_length(seq, qual, offsets)
    assert len(out_index) == 30_000 and tiles["rows"].sum() == 30_000
    for q, l, x, tn in ((20, 20, 0, 0), (22, 50, 0, 1), (20, 0, 1, 1)):
        p, po = both_params("illumina", q, l, x, tn)
        want, err = ob.oracle_trim_batch(po, qual, seq, offsets=offsets, threads=4)
        assert err is None
        got = sk_ctx.trim_segmented(p, qs, tiles, out_index, max_stride, seq=ss)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, (q, l, x, tn, bad[:5], got[bad[:5]], want[bad[:5]], np.diff(offsets)[bad[:5]])
        # the same with the cuts left in slot order (what the CLI asks for): slot k holds read out_index[k]
        got = sk_ctx.trim_segmented(p, qs, tiles, out_index, max_stride, seq=ss, slot_order=True)
        assert (got == want[out_index]).all(), (q, l, x, tn, "slot order")
    # range error: reported with the ORIGINAL read index, lowest original index first
    q2 = qual.copy()
    victims = [25_000, 7_777]
    for v in victims:
        q2[int(offsets[v]) + 3] = 20
    ss2, qs2, tiles2, oi2, ms2 = segment_by_length(seq, q2, offsets)
    p, _ = both_params("illumina", 20, 0, 0, 0)
    for slot_order in (False, True):  # the error names the caller's read either way
        with pytest.raises(capi.RangeError) as ei:
            sk_ctx.trim_segmented(p, qs2, tiles2, oi2, ms2, seq=ss2, slot_order=slot_order)
        assert (ei.value.read, ei.value.pos, ei.value.ch) == (7_777, 3, 20)


def test_register_staged_tiles(sk_ctx):
    """The register-staged tile kernel (equal lengths, rows of 72..160 bytes): every stage size
    (5..10 KiB tiles), batches of one read, just under / on / over a tile boundary, and one large
    enough that every wave takes several tiles in turn (12 waves x 256 CUs x 64 reads = 196 608
    per round) ending in a ragged tile; plus a range error deep inside such a batch, which must be
    the reference's (read, position, char)."""
    rng = np.random.default_rng(2718)
    for stride in range(72, 161, 8):
        L = stride - int(rng.integers(0, 8)) if stride > 72 else 70
        b = capi.Batch(0, None, None, stride, L, None, 10)
        assert capi.lib().sk_kernel_for(b) == 4, stride
        for n in (1, 63, 64, 65, 129, 1000, 3 * 196608 + 77 if stride in (104, 152) else 5000):
            qt = ("sanger", "illumina", "solexa")[(stride // 8 + n) % 3]
            lo, hi = {"sanger": (33, 74), "solexa": (59, 105), "illumina": (64, 105)}[qt]
            mid = int(rng.integers(lo + 6, hi - 6))
            qual = np.clip(mid + rng.integers(-16, 17, size=(n, L), dtype=np.int16), lo, hi).astype(np.uint8)
            drop = rng.integers(0, L + 1, size=n)
            qual[np.arange(L)[None, :] >= drop[:, None]] = lo + 1
            qs = synth.pack_fixed(qual, stride)
            q, l, x = int(rng.integers(0, 42)), int(rng.integers(0, L)), int(rng.integers(0, 2))
            p, po = both_params(qt, q, l, x, 0)
            want, err = ob.oracle_trim_batch(po, qs, None, stride=stride, read_len=L, n_reads=n)
            assert err is None
            got = sk_ctx.trim_batch(p, qs, stride=stride, read_len=L, n_reads=n)
            bad = np.nonzero((got != want).any(axis=1))[0]
            assert bad.size == 0, (stride, L, n, qt, q, l, x, bad[:4], got[bad[:4]], want[bad[:4]])
    # a bad char in the last full tile of a long batch and one in its ragged tail
    n, stride, L = 2 * 196608 + 40, 152, 150
    qs = np.full((n, stride), 75, dtype=np.uint8)
    qs[n - 3, 5] = 20
    qs[n - 70, 149] = 31
    p, po = both_params("sanger", 20, 20, 0, 0)
    with pytest.raises(capi.RangeError) as ei:
        sk_ctx.trim_batch(p, qs.reshape(-1), stride=stride, read_len=L, n_reads=n)
    assert (ei.value.read, ei.value.pos, ei.value.ch) == (n - 70, 149, 31)
    qs[n - 70, 149] = 75
    with pytest.raises(capi.RangeError) as ei:
        sk_ctx.trim_batch(p, qs.reshape(-1), stride=stride, read_len=L, n_reads=n)
    assert (ei.value.read, ei.value.pos, ei.value.ch) == (n - 3, 5, 20)


def test_streams_keep_their_own_range_errors(sk_ctx):
    """Device-resident scans enqueued on two streams of one context: each sk_scan_device_finish reports what the
    scans of ITS stream found, whatever finishes first."""
    import torch
    dev = torch.device("cuda", 0)
    n = 4096
    p = capi.make_params("sanger")
    bufs = []
    for bad_read in (10, 700, None):
        q = torch.full((n, 152), 75, dtype=torch.uint8, device=dev)
        if bad_read is not None:
            q[bad_read, 7] = 20
        bufs.append((q, torch.empty((n, 2), dtype=torch.int32, device=dev)))
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize(dev)
    sk_ctx.scan_device_async(p, bufs[0][0].data_ptr(), bufs[0][1].data_ptr(), n, stride=152, read_len=150, stream=s1.cuda_stream)
    sk_ctx.scan_device_async(p, bufs[1][0].data_ptr(), bufs[1][1].data_ptr(), n, stride=152, read_len=150, stream=s2.cuda_stream)
    with pytest.raises(capi.RangeError) as e2:
        sk_ctx.scan_device_finish(s2.cuda_stream)
    assert (e2.value.read, e2.value.pos, e2.value.ch) == (700, 7, 20)
    with pytest.raises(capi.RangeError) as e1:
        sk_ctx.scan_device_finish(s1.cuda_stream)
    assert (e1.value.read, e1.value.pos, e1.value.ch) == (10, 7, 20)
    sk_ctx.scan_device_async(p, bufs[2][0].data_ptr(), bufs[2][1].data_ptr(), n, stride=152, read_len=150, stream=s2.cuda_stream)
    sk_ctx.scan_device_finish(s2.cuda_stream)  # clean: the earlier error was consumed
    assert bool((bufs[2][1][:, 1] == 150).all())


def test_pair_classification_on_device(sk_ctx):
    """sk_count_pairs_device_*: the four pair classes of reference src/trim_paired.cpp:543-567 from cuts on the
    device, against numpy on the same cuts (all class mixes, odd sizes, an empty batch, counts cleared per finish)."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(8)
    for n_pairs in (0, 1, 63, 64, 65, 1000, 300_001):
        cuts = np.zeros((2 * n_pairs, 2), dtype=np.int32)
        cuts[:, 1] = np.where(rng.random(2 * n_pairs) < 0.6, rng.integers(0, 150, 2 * n_pairs), rng.choice([-1, -2], 2 * n_pairs))
        cuts[:, 0] = np.where(cuts[:, 1] >= 0, 0, -1)
        k1, k2 = cuts[0::2, 1] >= 0, cuts[1::2, 1] >= 0
        want = (int((k1 & k2).sum()), int((k1 & ~k2).sum()), int((~k1 & k2).sum()), int((~k1 & ~k2).sum()))
        t = torch.from_numpy(cuts).to(dev)
        cls = torch.full((max(1, n_pairs),), 9, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        assert sk_ctx.count_pairs_device(t.data_ptr(), n_pairs, cls.data_ptr()) == want
        assert sk_ctx.count_pairs_device(t.data_ptr(), n_pairs) == want  # counters were cleared by the finish
        if n_pairs:
            want_cls = np.where(k1, np.where(k2, 0, 1), np.where(k2, 2, 3)).astype(np.uint8)
            assert (cls.cpu().numpy()[:n_pairs] == want_cls).all()


def test_misaligned_base_pointers_and_odd_strides(sk_ctx):
    """Device buffers that do not start on a 16-byte boundary, strides that are not multiples of 8, with and without
    -n: every combination takes the re-striding loader (or the aligned kernels when it can) and must give the
    oracle's cuts; the last reads of the buffer sit right at its end (nothing may be read past it)."""
    import torch
    dev = torch.device("cuda", 0)
    n, L = 5_003, 150
    seq, qual = synth.make_reads(31, n, L, "sanger", lower_n_frac=0.01)
    for stride in (150, 151, 152, 157):
        qs, ss = synth.pack_fixed(qual, stride), synth.pack_fixed(seq, stride)
        for shift_q, shift_s in ((0, 0), (3, 0), (5, 9), (16, 1)):
            # exact-size device buffers: the batch ends where the allocation's used part ends
            tq = torch.zeros(shift_q + n * stride, dtype=torch.uint8, device=dev)
            ts = torch.zeros(shift_s + n * stride, dtype=torch.uint8, device=dev)
            tq[shift_q:] = torch.from_numpy(qs.reshape(-1)).to(dev)
            ts[shift_s:] = torch.from_numpy(ss.reshape(-1)).to(dev)
            for tn in (False, True):
                p, po = both_params("sanger", 20, 20, 0, tn)
                want, err = ob.oracle_trim_batch(po, qs, ss, stride=stride, read_len=L, n_reads=n, threads=4)
                assert err is None
                out = torch.full((n, 2), -7, dtype=torch.int32, device=dev)
                torch.cuda.synchronize(dev)
                sk_ctx.scan_device_async(p, tq.data_ptr() + shift_q, out.data_ptr(), n, stride=stride, read_len=L,
                                         seq_ptr=ts.data_ptr() + shift_s)
                sk_ctx.scan_device_finish()
                got = out.cpu().numpy()
                bad = np.nonzero((got != want).any(axis=1))[0]
                assert bad.size == 0, (stride, shift_q, shift_s, tn, bad[:5], got[bad[:5]], want[bad[:5]])


def test_long_reads_hovering_at_the_threshold(sk_ctx):
    """The skip-ahead window search of the whole-wave general kernel (prefix table, jumps of |S - T| / 255 windows)
    on reads built to defeat it: window averages that sit ON the threshold for thousands of windows (chars
    alternating around it, slow sawtooth drifts, single-char spikes of 126 and 33 that move a window sum by the
    largest legal step), lengths across the team-of-16 / whole-wave boundary and up to 40 kb, with -x and -n."""
    rng = np.random.default_rng(77)
    lens = np.array([4095, 4096, 4097, 5000, 8191, 12_345, 20_000, 40_000] * 3, dtype=np.uint32)
    n = len(lens)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    tot = int(offs[-1])
    thr = 33 + 20
    qual = np.empty(tot, dtype=np.uint8)
    for i in range(n):
        a, b = int(offs[i]), int(offs[i + 1])
        L = b - a
        k = np.arange(L)
        mode = i % 3
        if mode == 0:    # alternate thr-1 / thr+1 (average exactly thr over even windows), random flips
            q = thr + np.where(k % 2 == 0, -1, 1) + (rng.random(L) < 0.02) * rng.choice([-1, 1], L)
        elif mode == 1:  # sawtooth of the window average through the threshold, period ~ 3 windows
            w = max(1, L // 10)
            q = thr + np.round(2.5 * np.sin(2 * np.pi * k / (3.1 * w))).astype(int)
        else:            # flat at the threshold with rare extreme chars
            q = np.full(L, thr)
            hits = rng.random(L) < 0.001
            q = np.where(hits, rng.choice([33, 126], L), q)
        qual[a:b] = np.clip(q, 33, 126)
    seq = rng.choice(np.frombuffer(b"ACGT" * 500 + b"Nn", dtype=np.uint8), size=tot)
    for q, l, x, tn in ((20, 20, 0, 0), (20, 1000, 1, 0), (21, 0, 0, 1), (19, 20, 0, 0)):
        p, po = both_params("sanger", q, l, x, tn)
        want, err = ob.oracle_trim_batch(po, qual, seq, offsets=offs, threads=4)
        assert err is None
        got = sk_ctx.trim_batch(p, qual, seq, offsets=offs)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, (q, l, x, tn, bad[:5], got[bad[:5]], want[bad[:5]], lens[bad[:5]])
        # and each length as a uniform fixed-stride batch of its own (the all-reads mode of the kernel)
    for L in (4097, 12_345):
        m = 40
        qm = np.clip(thr + rng.integers(-1, 2, size=(m, L)), 33, 126).astype(np.uint8)
        p, po = both_params("sanger", 20, 20, 0, 0)
        want, _ = ob.oracle_trim_batch(po, qm.reshape(-1), stride=L, read_len=L, n_reads=m)
        got = sk_ctx.trim_batch(p, qm.reshape(-1), stride=L, read_len=L, n_reads=m)
        assert (got == want).all(), L


def test_random_fuzz_medium_and_long_reads(sk_ctx):
    """Random batches for the general kernel (teams of 16, whole-wave teams with the skip-ahead search, the
    in-kernel global-memory fallback): lengths 330 .. 20 000 log-uniform with a few beyond the LDS buffer, every
    encoding, thresholds from 0 to beyond the scale, -x, -n, -l up to beyond the read; ragged and stride + lengths."""
    rng = np.random.default_rng(2027)
    for trial in range(24):
        qt = ["sanger", "solexa", "illumina"][trial % 3]
        lo, hi = {"sanger": (33, 74), "solexa": (59, 105), "illumina": (64, 105)}[qt]
        n = 150
        lens = np.exp(rng.uniform(np.log(330), np.log(20_000), size=n)).astype(np.uint32)
        if trial % 6 == 0:
            lens[:3] = [90_000, 120_000, 100_001]  # beyond any LDS buffer: the byte-by-byte fallback
        if trial % 4 == 1:
            lens[:] = lens[0]  # one length: the all-reads mode via a fixed stride below
        offs = np.zeros(n + 1, dtype=np.uint64)
        offs[1:] = np.cumsum(lens)
        tot = int(offs[-1])
        mid = rng.integers(lo + 8, hi - 8)
        mode = trial % 4
        if mode == 0:
            qual = np.clip(rng.normal(mid, 7, tot).astype(int), lo, hi)
        elif mode == 1:
            qual = np.clip(mid + rng.integers(-2, 3, size=tot), lo, hi)
        elif mode == 2:  # long plateaus: every few thousand bases the level changes
            level = np.repeat(rng.integers(lo, hi, size=tot // 1500 + 2), 1500)[:tot]
            qual = np.clip(level + rng.integers(-3, 4, size=tot), lo, hi)
        else:
            qual = np.where(rng.random(tot) < 0.5, lo, hi)
        qual = qual.astype(np.uint8)
        seq = rng.choice(np.frombuffer(b"ACGT" * 300 + b"Nn", dtype=np.uint8), size=tot)
        q = int(rng.choice([0, 2, 15, 20, 25, 30, 41, 60]))
        l = int(rng.choice([0, 20, 300, 5000, 30_000]))
        x, tn = trial % 2, (trial // 2) % 2
        p, po = both_params(qt, q, l, x, tn)
        want, err = ob.oracle_trim_batch(po, qual, seq, offsets=offs, threads=4)
        assert err is None
        got = sk_ctx.trim_batch(p, qual, seq, offsets=offs)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, ("ragged", trial, qt, q, l, x, tn, bad[:5], got[bad[:5]], want[bad[:5]], lens[bad[:5]])
        if trial % 4 == 1:  # the same reads as a uniform fixed-stride batch
            L = int(lens[0])
            got = sk_ctx.trim_batch(p, qual, seq, stride=L, read_len=L, n_reads=n)
            assert (got == want).all(), ("uniform", trial, L)


def test_general_kernels_soak_and_late_five_prime_cut(sk_ctx, monkeypatch):
    """(1) The read that a soak run caught: 137 bases, bad until the last 14 -- the first window at the threshold is
    one of those after the last aligned one, which a 64-window cell of the streaming kernel had already run through
    when the tail step looked at them again (phase "looking for the first S < T") and cut at a window BEFORE the 5'
    one.  Next to a read long enough to send the batch to the general kernels.  (2) tests/soak_general.py,
    300 random batches: every encoding, thresholds 0..41, -l, -x, -n, lengths 1..200 kb, chars out of range, both
    general kernels forced in turn, sk_submit and the device entry points with and without hints."""
    h = ("2f2d322f2d2c2c32312d312d322e2e302e31302f31322e2c2e322d2d2f322c312d2f2d2e302d2c2c2f32322f2f3230312c312e2c2d2e2c2f"
         "3130302f2d2c2d3030322d2c2e2e2c2d2c32302e32322d2f322e2f2e2f2e2e2f302e2c2f3130322d312e302f2c2d312c302f31302c2d3032"
         "2c2d312e322e2d322e30305b5a5f5c5d5f5c5a5f5e5c5e5a59")
    r = np.frombuffer(bytes.fromhex(h), dtype=np.uint8)
    p, po = both_params("sanger", 30, 0, 0, 0)
    for which in ("band", "team", "stream"):
        monkeypatch.setenv("SK_GENERAL", which)
        for before, after in (([], [2387]), ([53, 103], [2387, 1]), ([1500], []), ([100] * 4, [100, 100, 5000])):
            parts = [np.full(l, 45, dtype=np.uint8) for l in before] + [r] + [np.full(l, 70, dtype=np.uint8) for l in after]
            offs = np.zeros(len(parts) + 1, dtype=np.uint64)
            offs[1:] = np.cumsum([len(x) for x in parts])
            q = np.concatenate(parts)
            want, _ = ob.oracle_trim_batch(po, q, None, offsets=offs, threads=1)
            assert list(want[len(before)]) == [123, 137]
            got = sk_ctx.trim_batch(p, q, None, offsets=offs)
            assert (got == want).all(), (which, before, after, got, want)
    monkeypatch.delenv("SK_GENERAL")
    import soak_general
    assert soak_general.run(300, 2026, verbose=False) == 300 * 12


def test_tile_kernels_soak(sk_ctx):
    """tests/soak_tiles.py, 400 random short-read batches through every layout of the lane-per-read kernels
    (fixed stride with an odd / even number of 8-byte units, packed, unaligned, with per-read lengths; ragged;
    segmented in read order and in slot order), every encoding, thresholds 0..41, -l, -x, -n, chars out of range.
    (18 000 such batches ran clean when the soak was written.)"""
    import soak_tiles
    assert soak_tiles.run(400, 2027, verbose=False) > 400 * 5


def test_segmented_long_rows_and_partial_tiles(sk_ctx):
    """Segmented batches without -n, rows up to 312 bytes (8 waves per CU), length groups of every size (1 read ...
    several full tiles + a partial one) so that full and partial tiles alternate in every order; cuts in read order
    and in slot order; a planted range error.  (Written for a variant that staged the next tile in registers -- 20
    KiB through 80 VGPRs at two waves per SIMD; it measured 11 % slower than LDS-DMA, 3.8 against 4.2 TB/s: the
    80 ds_write_b128 per tile cost more than the prefetch gains -- and kept for its coverage.)"""
    from fastq_util import segment_by_length
    rng = np.random.default_rng(99)
    lens = np.concatenate([rng.integers(230, 302, size=60_000), rng.integers(75, 302, size=40_000),
                           np.full(64 * 7, 250), np.full(1, 301), np.full(65, 77)]).astype(np.uint32)
    rng.shuffle(lens)
    n = len(lens)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    tot = int(offsets[-1])
    qual = np.clip(rng.normal(58, 9, tot).astype(int), 33, 74).astype(np.uint8)
    drop = rng.random(tot) < 0.0004
    qual[np.nonzero(drop)[0]] = 34
    seq = np.full(tot, 65, dtype=np.uint8)
    ss, qs, tiles, out_index, max_stride = segment_by_length(seq, qual, offsets)
    assert max_stride == 312
    for q, l, x in ((20, 20, 0), (24, 100, 1)):
        p, po = both_params("sanger", q, l, x, 0)
        want, err = ob.oracle_trim_batch(po, qual, seq, offsets=offsets, threads=8)
        assert err is None
        got = sk_ctx.trim_segmented(p, qs, tiles, out_index, max_stride)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, (q, l, x, bad[:5], got[bad[:5]], want[bad[:5]], lens[bad[:5]])
        got = sk_ctx.trim_segmented(p, qs, tiles, out_index, max_stride, slot_order=True)
        assert (got == want[out_index]).all()
    q2 = qual.copy()
    q2[int(offsets[77_777]) + 9] = 12
    _, qs2, tiles2, oi2, ms2 = segment_by_length(seq, q2, offsets)
    with pytest.raises(capi.RangeError) as ei:
        sk_ctx.trim_segmented(capi.make_params("sanger", 20, 0), qs2, tiles2, oi2, ms2)
    assert (ei.value.read, ei.value.pos, ei.value.ch) == (77_777, 9, 12)
