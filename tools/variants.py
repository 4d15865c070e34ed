"""The kernel variants bench.py reports beside its headline (and tools/profile.sh profiles one by one):
each builds an HBM-resident synthetic batch of one layout and returns a launcher for one scan of it
plus its ALGORITHMIC bytes per launch (L quality bytes + 8 bytes of cut per read, + L
sequence bytes with -n; offsets / descriptors / padding are layout overhead and not counted)."""
import ctypes as C

import numpy as np

NAMES = ("n150", "u250", "u100", "seg", "seg_n", "seg_scatter", "packed150", "ragged150", "ragged_mix", "u600", "u1000", "long", "long30k")


def _quals(torch, shape, dev, seed):
    """Valid Sanger chars with a quality collapse in the last fifth of every 1000 bytes, so that cuts land
    inside reads of any length."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    n = int(np.prod(shape))
    q = torch.randint(60, 74, (n + 4096,), dtype=torch.uint8, device=dev, generator=g)
    body = q[: n - n % 1000].view(-1, 1000)
    body[:, 800:] -= 25
    return q


def _seg_layout(lens_sorted_counts):
    """lens_sorted_counts: list of (length, count).  -> tiles (numpy TILE_DTYPE), total bytes, reads"""
    from sickle_amd.capi import TILE_DTYPE
    tl, at, slot = [], 0, 0
    for L, cnt in lens_sorted_counts:
        if cnt == 0:
            continue
        st = ((L + 7) // 8 | 1) * 8
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
    return np.concatenate(tl), at, slot


def build(name, torch, capi, ctx, dev, stream, bench):
    """-> dict(launch, n_reads, algo_bytes, kernel, workload)"""
    lib = capi.lib()
    p = capi.make_params("sanger", 20, 20)
    pn = capi.make_params("sanger", 20, 20, False, True)
    sp = stream.cuda_stream
    keep = []  # tensors the launcher needs alive

    def uniform(L, n, with_seq, stride=None):
        stride = stride or ((L + 7) // 8 | 1) * 8
        q = bench.synth_quals_device(torch, n, L, stride, 7 + L, dev)
        out = torch.empty((n, 2), dtype=torch.int32, device=dev)
        seq = None
        if with_seq:
            seq = torch.full((n, stride), 65, dtype=torch.uint8, device=dev)
            g = torch.Generator(device=dev)
            g.manual_seed(3)
            seq[torch.rand(n, device=dev, generator=g) < 0.25, L // 2] = ord("N")
        keep.extend([q, out, seq])
        par = pn if with_seq else p
        kid = lib.sk_kernel_for(C.byref(capi.Batch(q.data_ptr(), seq.data_ptr() if with_seq else None, None, stride, L, None, n)))
        return dict(launch=lambda: ctx.scan_device_async(par, q.data_ptr(), out.data_ptr(), n, stride=stride, read_len=L,
                                                         seq_ptr=seq.data_ptr() if with_seq else None, stream=sp),
                    n_reads=n, algo_bytes=n * ((2 if with_seq else 1) * L + 8), kernel=lib.sk_kernel_name(kid).decode(),
                    workload="uniform %d bp at stride %d%s, %d reads" % (L, stride, ", -n" if with_seq else "", n))

    if name == "n150":
        return dict(uniform(150, 10_000_000, True), keep=keep)
    if name == "u250":
        return dict(uniform(250, 4_000_000, False), keep=keep)
    if name == "u150":  # diagnostic (with SK_TILE_STAGE=0: the LDS-DMA kernel the segmented one is built on)
        return dict(uniform(150, 10_000_000, False), keep=keep)
    if name == "u100":
        return dict(uniform(100, 10_000_000, False), keep=keep)
    if name == "long30k":  # uniform long reads back to back: the streaming general kernel without per-read variety
        L, n = 30_000, 33_333
        q = _quals(torch, (n * L,), dev, 35)  # (the 150 bp quality model decays to nothing within a kilobase)
        out = torch.empty((n, 2), dtype=torch.int32, device=dev)
        keep.extend([q, out])
        return dict(launch=lambda: ctx.scan_device_async(p, q.data_ptr(), out.data_ptr(), n, stride=L, read_len=L, stream=sp),
                    n_reads=n, algo_bytes=n * (L + 8), kernel="sk_scan_stream_kernel",
                    workload="30 kb reads back to back (stride 30000), %d reads: a wave per read, the read streamed through LDS" % n, keep=keep)
    if name in ("u600", "u1000"):  # uniform medium reads back to back: rows beyond the 64-read tiles
        L = int(name[1:])
        n = 1_000_000_000 // L
        q = _quals(torch, (n * L,), dev, 36)
        out = torch.empty((n, 2), dtype=torch.int32, device=dev)
        keep.extend([q, out])
        kid = lib.sk_kernel_for(C.byref(capi.Batch(q.data_ptr(), None, None, L, L, None, n)))
        return dict(launch=lambda: ctx.scan_device_async(p, q.data_ptr(), out.data_ptr(), n, stride=L, read_len=L, stream=sp),
                    n_reads=n, algo_bytes=n * (L + 8), kernel=lib.sk_kernel_name(kid).decode(),
                    workload="%d bp reads back to back (stride %d), %d reads: tiles of %s reads, windows %d wide on the matrix path" % (L, L, n, "32" if L < 940 else "16", L // 10), keep=keep)
    if name == "packed150":
        d = uniform(150, 10_000_000, False, stride=150)
        d["workload"] = "150 bp reads packed back to back (stride 150): tiles re-strided into LDS, " + d["workload"]
        return dict(d, keep=keep)
    if name in ("seg", "seg_n", "seg_scatter", "seg150"):
        with_seq = name == "seg_n"
        slot_order = name != "seg_scatter"  # the CLI takes its cuts in slot order and un-permutes on the host
        m = 4_000_000
        rng = np.random.default_rng(11)
        lens = rng.integers(75, 302, size=m)
        if name == "seg150":  # diagnostic: one length, the segmented kernel against the uniform ones
            m = 10_000_000
            lens = np.full(m, 150)
        order = np.argsort(lens, kind="stable").astype(np.uint32)  # slot -> the read's place in the caller's order
        counts = np.bincount(lens, minlength=302)
        tiles, nbytes, nreads = _seg_layout([(L, int(counts[L])) for L in range(75, 302)])  # noqa
        assert nreads == m
        q = _quals(torch, (nbytes,), dev, 21)
        seq = torch.full((nbytes + 4096,), 65, dtype=torch.uint8, device=dev) if with_seq else None
        tiles_t = torch.from_numpy(tiles.view(np.uint8).copy()).to(dev)
        oi = torch.from_numpy(order.view(np.int32).copy()).to(dev)
        out = torch.empty((m, 2), dtype=torch.int32, device=dev)
        cls, ncls = capi.seg_classes(tiles)
        max_stride = int(tiles["stride"].max())
        keep.extend([q, seq, tiles_t, oi, out, cls])
        par = pn if with_seq else p

        def go():
            b = capi.Batch(q.data_ptr(), seq.data_ptr() if with_seq else None, None, max_stride, 0, None, m, tiles_t.data_ptr(),
                           len(tiles), oi.data_ptr(), C.cast(cls, C.c_void_p) if ncls else None, ncls, 1 if slot_order else 0)
            rc = lib.sk_scan_device_async(ctx._h, C.byref(par), C.byref(b), out.data_ptr(), sp)
            if rc != 0:
                raise capi.SickleError("sk_scan_device_async(segmented) -> %d" % rc)
        tot = int(lens.sum())
        return dict(launch=go, n_reads=m, algo_bytes=(2 if with_seq else 1) * tot + 8 * m, kernel="sk_scan_tile_kernel",
                    workload="segmented: %d reads of U{75..301} bp grouped by length into %d tiles, cuts %s%s"
                             % (m, len(tiles), "in slot order (the caller un-permutes, as the CLI does)" if slot_order
                                else "scattered back to input order on the device", ", -n" if with_seq else ""), keep=keep)
    if name in ("ragged150", "ragged_mix", "long"):
        g = torch.Generator(device=dev)
        g.manual_seed(17)
        if name == "ragged150":
            n = 10_000_000
            lens = torch.full((n,), 150, dtype=torch.int64, device=dev)
            hint = 150
        elif name == "ragged_mix":
            n = 4_000_000
            lens = torch.randint(75, 302, (n,), device=dev, dtype=torch.int64, generator=g)
            hint = 301
        else:
            lens = torch.randint(1000, 30_001, (70_000,), device=dev, dtype=torch.int64, generator=g)
            n = int((torch.cumsum(lens, 0) <= 1_000_000_000).sum().item())
            lens = lens[:n]
            hint = 30_000
        off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        off[1:] = torch.cumsum(lens, 0)
        tot = int(off[n].item())
        q = _quals(torch, (tot,), dev, 33)
        out = torch.empty((n, 2), dtype=torch.int32, device=dev)
        keep.extend([q, off, out])
        kern = "sk_scan_stream_kernel" if name == "long" else "sk_scan_tile_sorted_kernel" if name == "ragged_mix" else "sk_scan_tile_any_kernel"
        what = {"ragged150": "ragged offsets, every read 150 bp", "ragged_mix": "ragged offsets, U{75..301} bp in input order (regrouped on the device: sort + sorted scan)",
                "long": "ragged offsets, U{1000..30000} bp"}[name]
        return dict(launch=lambda: ctx.scan_device_async(p, q.data_ptr(), out.data_ptr(), n, offsets_ptr=off.data_ptr(), stride=hint,
                                                         stream=sp),
                    n_reads=n, algo_bytes=tot + 8 * n, kernel=kern, workload="%s, %d reads" % (what, n), keep=keep)
    raise KeyError(name)
