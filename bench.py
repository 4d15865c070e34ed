#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json: reads/s (+ Gbases/s) trimmed, 150 bp SE
Sanger, q=20 l=20, on N MI355X, with the quality bytes already resident in HBM.

One process per GPU, reads independent, nothing exchanged in a timed region, only kept/discarded counters
summed afterwards (sickle_amd/shard.py).  Every read of a job is seeded by its GLOBAL block (tools/workloads.py),
so the N-shard run of a job scans the very reads of its 1-GPU run.

  --workload se     (default) fixed-length Sanger reads: BASELINE configs[1] (weak: --reads per GPU) or, with
                    --total-reads 100000000, configs[3] (strong: contiguous shards)
  --workload mixed  BASELINE configs[4]'s batch: U{75..301} bp, phred+64, -n, grouped by length (segmented layout,
                    cuts in slot order); strong scaling over --total-reads (default 8 M)
  --mode resident   (default) a step = one scan (sk_scan_device_async) of this rank's batch, resident in HBM
  --mode pipeline   a step = this rank's shard pushed through sk_submit / sk_wait from pinned host memory in
                    --batches batches over 2 slots (H2D of batch i+1 overlaps the scan of batch i): configs[3]'s
                    "async batch pipeline"; reports the H2D rate per rank and in total

The default run also carries, measured OUTSIDE the timed region and compactly (the driver's record keeps ~6 kB):
  at every N   legs      `pipe` (se, pipeline), `mix` (mixed, resident), `mixpipe` (mixed, pipeline): aggregates
                         over the ranks, each leg between barriers, max-over-ranks time
  at N = 1     roofline.peak_measured, cpu_baseline, variants (tools/variants.py names the shapes), e2e (this
               CLI against the compiled reference CLI: 10 M pairs compared byte for byte, 50 M pairs = 100 M reads
               = configs[3] with the reference extrapolated per read), e2e_mixed (configs[4] through both CLIs)

Launch: `python bench.py` (N=1) or
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import workloads as wl  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
SEED = 1234


def r4(x):
    """Four significant digits: the JSON line has to fit the driver's record."""
    return None if x is None else float("%.4g" % x)


def r6(x):
    return None if x is None else float("%.6g" % x)


def synth_quals_device(torch, n, length, stride, seed, device, lo=0):
    """(n, stride) uint8: reads [lo, lo + n) of the fixed-length job `seed` (tools/workloads.py)."""
    return wl.se_shard(torch, device, seed, lo, n, length, stride)


def host_cores():
    """CPU threads this process may really use: the affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // period))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def profile_traffic(n, length):
    """HBM bytes per launch of the headline kernel, scaled per read from the newest COMMITTED rocprofv3 PMC
    summary (tools/profile.sh -> profiles/rNN/pmc_latest.json: FETCH_SIZE x 2 (gfx950) + WRITE_SIZE, separate
    passes).  Profile-derived, not a counter of this run; None for shapes the profile does not cover."""
    if length != 150:
        return None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_latest.json")), reverse=True):
        try:
            d = json.load(open(path))
            per_read = d["hbm_bytes_per_launch"]["total"] / float(d.get("reads_per_launch", 10_000_000))
            return per_read * n, os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def cpu_baseline(qual_host, n, stride, length, threads):
    """The CPU path timed beside the GPU: the reference's own sliding_window (oracle/_ref, kind
    "reference") when that prebuilt library travelled with the repo, else the oracle port."""
    import oracle_bind as ob
    p = ob.make_params("sanger", 20, 20)
    if ob.have_ref():
        kind = "reference"
        run = lambda th, m: ob.ref_trim_batch(p, qual_host[:m * stride], stride=stride, read_len=length, n_reads=m, threads=th)
    else:
        kind = "port"
        run = lambda th, m: ob.oracle_trim_batch(p, qual_host[:m * stride], stride=stride, read_len=length, n_reads=m, threads=th)[0]
    run(threads, min(n, 200_000))  # warm the pages
    m1 = min(n, 2_000_000)
    t0 = time.perf_counter()
    run(1, m1)
    t1 = time.perf_counter() - t0
    reps = 2
    t0 = time.perf_counter()
    for _ in range(reps):
        cuts = run(threads, n)
    tn = (time.perf_counter() - t0) / reps
    return {"value": r4(n / tn), "unit": "reads/s", "cores": threads, "kind": kind,
            "sample": "%dx%d reads on %d threads (%.1f s); 1 thread: %d reads" % (reps, n, threads, tn * reps, m1),
            "value_1thread": r4(m1 / t1)}, cuts


def time_launches(torch, stream, launch, finish, reps, settle):
    """Average duration of `reps` launches queued back to back, by HIP events on the launch stream."""
    for _ in range(settle):
        launch()
    finish()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(stream)
        launch()
        b.record(stream)
    finish()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    return sum(ms) / len(ms), ms[0], ms[-1]


def short_kernel(name):
    return name.replace("sk_scan_", "").replace("_kernel", "")


def run_variants(torch, capi, ctx, device, stream, names, reps, full=False):
    """Every other kernel, HBM-resident: {variant: {k: kernel, ms, GBps: algorithmic GB/s, frac: of 8 TB/s}}."""
    import variants
    me = sys.modules[__name__]
    out = {}
    for name in names:
        try:
            v = variants.build(name, torch, capi, ctx, device, stream, me)
            torch.cuda.synchronize(device)
            # (building a variant's batch leaves the device idle long enough for its clocks to drop: 100 untimed launches first
            # -- with 10, `seg` read 0.46 of peak where 300 launches in a row say 0.55-0.61)
            avg, lo, hi = time_launches(torch, stream, v["launch"], lambda: ctx.scan_device_finish(stream.cuda_stream), reps, 100)
            gbs = v["algo_bytes"] / (avg * 1e-3) / 1e9
            out[name] = {"k": short_kernel(v["kernel"]), "ms": r4(avg), "GBps": r4(gbs), "frac": r4(gbs / HBM_PEAK_GBS)}
            if full:
                out[name].update({"workload": v["workload"], "kernel": v["kernel"], "reads": v["n_reads"], "kernel_ms_avg": avg, "kernel_ms_min": lo, "kernel_ms_max": hi,
                                  "achieved": gbs, "unit": "GB/s",
                                  "algorithmic_bytes_per_launch": v["algo_bytes"], "scans_in_run": reps + 100})
            del v
            torch.cuda.empty_cache()
        except Exception as e:  # a failing variant must not cost the headline line
            out[name] = {"error": ("%s: %s" % (type(e).__name__, e))[:80]}
    return out


# ---------------------------------------------------------------------------------------------------------
# what a rank scans


def job_plan(args, rank, world):
    """What this rank scans: (first read of the job it owns, number of reads, row stride, strong?).  Weak
    scaling: --reads per GPU, rank r owns [r * reads, (r + 1) * reads).  Strong scaling (--total-reads; always for
    the mixed workload): contiguous shards, sizes differ by at most one."""
    from sickle_amd.shard import shard_range
    total = args.total_reads
    if args.workload == "mixed" and total <= 0:
        total = args.mixed_reads
    strong = total > 0
    if strong:
        lo, hi = shard_range(total, rank, world)
        n = hi - lo
    else:
        lo, n = rank * args.reads, args.reads
    return {"lo": lo, "n": n, "stride": wl.tile_stride(args.len), "strong": strong, "total": total if strong else args.reads * world}


class ResidentSE:
    """This rank's fixed-length reads resident in HBM at the bank-friendly stride; a step = one scan."""
    has_events = True

    def __init__(self, torch, capi, ctx, device, stream, args, plan):
        self.torch, self.ctx, self.stream = torch, ctx, stream
        self.n, self.stride, self.length = plan["n"], plan["stride"], args.len
        self.qual = wl.se_shard(torch, device, SEED, plan["lo"], self.n, self.length, self.stride)
        self.out = torch.empty((self.n, 2), dtype=torch.int32, device=device)
        self.params = capi.make_params("sanger", 20, 20)
        self.algo_bytes = (self.length + 8) * self.n  # L quality bytes read + one 8-byte cut pair written, per read
        self.bases = self.n * self.length
        self.h2d_bytes = 0
        kid = capi.lib().sk_kernel_for(ctypes.byref(capi.Batch(self.qual.data_ptr(), None, None, self.stride, self.length, None, self.n)))
        self.kernel = capi.lib().sk_kernel_name(kid).decode()
        torch.cuda.synchronize(device) if device.type == "cuda" else None

    def step(self):
        self.ctx.scan_device_async(self.params, self.qual.data_ptr(), self.out.data_ptr(), self.n, stride=self.stride,
                                   read_len=self.length, stream=self.stream.cuda_stream)

    def finish(self):
        self.ctx.scan_device_finish(self.stream.cuda_stream)  # raises on a range error

    def counts(self):
        kept = int((self.out[:, 1] >= 0).sum().item())
        bases_kept = int((self.out[:, 1] - self.out[:, 0]).clamp_(min=0).sum().item())
        return [kept, self.n - kept, bases_kept, self.n]


class ResidentMixed:
    """This rank's mixed-length reads (configs[4]) resident in HBM in the segmented layout, scanned with -n."""
    has_events = True

    def __init__(self, torch, capi, ctx, device, stream, args, plan):
        self.torch, self.capi, self.ctx, self.stream = torch, capi, ctx, stream
        self.n = plan["n"]
        lens, qual, seq = wl.mixed_shard(torch, device, SEED + 4, plan["lo"], self.n)
        self.seg = wl.segment(torch, lens, qual, seq)
        del qual, seq
        self.tiles = torch.from_numpy(self.seg["tiles"].view("u1").copy()).to(device)
        self.out = torch.empty((self.n, 2), dtype=torch.int32, device=device)
        self.params = capi.make_params("illumina", 20, 20, False, True)
        self.bases = self.seg["bases"]
        self.algo_bytes = 2 * self.bases + 8 * self.n  # SURVEY 8d with -n: quality + sequence bytes + the cut pair
        self.h2d_bytes = 0
        self.cls, self.ncls = capi.seg_classes(self.seg["tiles"])
        self.kernel = "sk_scan_tile_kernel"
        torch.cuda.synchronize(device)

    def batch(self):
        s = self.seg
        return self.capi.Batch(s["q"].data_ptr(), s["seq"].data_ptr(), None, s["max_stride"], 0, None, self.n, self.tiles.data_ptr(),
                               len(s["tiles"]), s["out_index"].data_ptr(), ctypes.cast(self.cls, ctypes.c_void_p) if self.ncls else None,
                               self.ncls, 1)

    def step(self):
        b = self.batch()
        rc = self.capi.lib().sk_scan_device_async(self.ctx._h, ctypes.byref(self.params), ctypes.byref(b), self.out.data_ptr(),
                                                  self.stream.cuda_stream)
        if rc != 0:
            raise self.capi.SickleError("sk_scan_device_async(segmented) -> %d" % rc)

    def finish(self):
        self.ctx.scan_device_finish(self.stream.cuda_stream)

    def counts(self):
        kept = int((self.out[:, 1] >= 0).sum().item())
        bases_kept = int((self.out[:, 1] - self.out[:, 0]).clamp_(min=0).sum().item())
        return [kept, self.n - kept, bases_kept, self.n]


class Pipeline:
    """This rank's shard in `batches` pieces in PINNED HOST memory; a step = every piece through sk_submit /
    sk_wait over two slots, cuts back in host memory: what a caller with host data gets (PCIe-bound)."""
    has_events = False
    kernel = "sk_submit/sk_wait"

    def __init__(self, torch, capi, device, args, plan, workload):
        import numpy as np
        from sickle_amd.shard import shard_range
        self.capi, self.np = capi, np
        self.ctx = capi.Context(device.index, 2)
        self.lib = capi.lib()
        self.n, self.workload = plan["n"], workload
        self.raw, self.items, self.keep = [], [], []
        self.algo_bytes = self.bases = self.h2d_bytes = 0
        nb = max(1, min(args.batches, self.n)) if self.n else 0
        for k in range(nb):
            a, b = shard_range(self.n, k, nb)
            m = b - a
            outp, out = self.pinned(m * 8, np.int32, (m, 2))
            if workload == "se":
                q = wl.se_shard(torch, device, SEED, plan["lo"] + a, m, args.len, plan["stride"])
                qp, _ = self.pinned_from(q.reshape(-1))
                bt = capi.Batch(qp, None, None, plan["stride"], args.len, None, m)
                self.algo_bytes += (args.len + 8) * m
                self.bases += args.len * m
                self.h2d_bytes += m * plan["stride"]
            else:
                lens, qual, seq = wl.mixed_shard(torch, device, SEED + 4, plan["lo"] + a, m)
                s = wl.segment(torch, lens, qual, seq)
                qp, _ = self.pinned_from(s["q"])
                sp, _ = self.pinned_from(s["seq"])
                tp, _ = self.pinned_from(torch.from_numpy(s["tiles"].view("u1").copy()))
                ip, _ = self.pinned_from(s["out_index"].view(torch.uint8) if s["out_index"].numel() else torch.zeros(0, dtype=torch.uint8))
                bt = capi.Batch(qp, sp, None, s["max_stride"], 0, None, m, tp, len(s["tiles"]), ip, None, 0, 1)
                self.algo_bytes += 2 * s["bases"] + 8 * m
                self.bases += s["bases"]
                self.h2d_bytes += 2 * s["bytes"] + 24 * len(s["tiles"]) + 4 * m
            self.items.append((bt, outp, out, m))
        self.params = capi.make_params("sanger", 20, 20) if workload == "se" else capi.make_params("illumina", 20, 20, False, True)
        if device.type == "cuda":
            torch.cuda.empty_cache()

    def pinned(self, nbytes, dtype, shape):
        p = self.lib.sk_host_alloc(self.ctx._h, max(nbytes, 1))
        if not p:
            raise self.capi.SickleError("sk_host_alloc(%d) failed" % nbytes)
        self.raw.append(p)
        if nbytes == 0:
            return p, self.np.zeros(shape, dtype=dtype)
        arr = self.np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes,)).view(dtype).reshape(shape)
        return p, arr

    def pinned_from(self, t):
        """A pinned host copy of the flat uint8 tensor t."""
        nbytes = int(t.numel())
        p, arr = self.pinned(nbytes, self.np.uint8, (nbytes,))
        if nbytes:
            arr[:] = t.cpu().numpy()
        return p, arr

    def step(self):
        for i, (bt, outp, _, _) in enumerate(self.items):
            s = i % 2
            if i >= 2:
                self.ctx.wait(s)
            self.ctx._check(self.lib.sk_submit(self.ctx._h, s, ctypes.byref(self.params), ctypes.byref(bt), outp))
        for s in range(min(2, len(self.items))):
            self.ctx.wait(s)

    def finish(self):
        pass

    def counts(self):
        kept = bases_kept = 0
        for _, _, out, m in self.items:
            if m:
                kept += int((out[:, 1] >= 0).sum())
                bases_kept += int(self.np.clip(out[:, 1] - out[:, 0], 0, None).sum())
        return [kept, self.n - kept, bases_kept, self.n]

    def close(self):
        for p in self.raw:
            self.lib.sk_host_free(self.ctx._h, p)
        self.raw = []
        self.ctx.close()


def timed_steps(torch, work, stream, steps, warmup, settle, barrier):
    """`settle` + `warmup` untimed steps, then exactly `steps` timed ones between two barriers (each a
    dist.barrier + device synchronize).  -> (elapsed s on this rank, per-step kernel ms by HIP events on the
    launch stream, or [] for a work that launches on streams of its own)."""
    for _ in range(settle):
        work.step()
    work.finish()
    for _ in range(warmup):
        work.step()
    work.finish()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)] if work.has_events else []
    barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        if evs:
            evs[k][0].record(stream)
        work.step()
        if evs:
            evs[k][1].record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    work.finish()
    return elapsed, [a.elapsed_time(b) for a, b in evs]


WORKLOADS = {
    # (workload, strong, mode) -> text naming the BASELINE config the line is measured on
    "se_weak": "sickle se, %(n)d synthetic %(len)d bp Sanger reads per GPU, q=20 l=20 (BASELINE configs[1])",
    "se_strong": "sickle se, %(total)d synthetic %(len)d bp Sanger reads sharded across %(world)d GPU(s), q=20 l=20 (BASELINE configs[3]%(shape)s)",
    "mixed": "sickle pe batch, %(total)d mixed 75-301 bp Illumina reads, -n, grouped by length, sharded across %(world)d GPU(s) (BASELINE configs[4] shape; gzip ingest is the CLI's: e2e_mixed)",
}


def headline(args, plan, world, counts, elapsed, kern_ms, kernel_name, algo_bytes=None, bases=None, h2d=None):
    """Rank 0's JSON line from the job-wide counters [kept, discarded, bases kept, reads] and the max-over-ranks
    elapsed time of the K timed steps; kern_ms = this rank's per-launch kernel durations (HIP events; [] in
    pipeline mode, where the step is timed as a whole)."""
    length, n, stride, strong = args.len, plan["n"], plan["stride"], plan["strong"]
    job_reads = counts[3]
    total_reads = job_reads * args.steps
    if algo_bytes is None:
        algo_bytes = (length + 8) * n
    job_bases = bases if bases is not None else job_reads * length
    fmt = {"n": n, "len": length, "total": plan.get("total", job_reads), "world": world,
           "shape": "" if plan.get("total") == 100_000_000 else " shape"}
    workload = WORKLOADS["mixed" if args.workload == "mixed" else ("se_strong" if strong else "se_weak")] % fmt
    if args.mode == "pipeline":
        workload += "; async batch pipeline: %d batches per rank through sk_submit/sk_wait from pinned host memory, 2 slots" % args.batches
    res = {
        "metric": "reads/sec trimmed (+ Gbases/sec), 150 bp SE Sanger q20 l20, inputs resident in HBM",
        "value": r6(total_reads / elapsed), "unit": "reads/s",
        "gbases_per_s": r4(job_bases * args.steps / elapsed / 1e9),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_launches": args.settle,
        "ms_per_step": r6(elapsed / args.steps * 1e3), "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": workload, "mode": args.mode, "reads_per_gpu": n, "reads_in_job": job_reads, "read_len": length if args.workload == "se" else "75-301",
                   "stride": stride if args.workload == "se" else "8*odd per length", "kernel": kernel_name,
                   "sharding": "reads split across ranks, no collective"},
        "kept": counts[0], "discarded": counts[1],
        "mean_bases_kept": r4(counts[2] / max(1, counts[0])),
    }
    if args.mode == "pipeline":
        res["metric"] = "reads/sec trimmed (+ Gbases/sec), host buffers through sk_submit/sk_wait (PCIe-inclusive)"
    if kern_ms:
        kern_ms = sorted(kern_ms)
        kern_avg_ms = sum(kern_ms) / len(kern_ms)
        achieved = algo_bytes / (kern_avg_ms * 1e-3) / 1e9
        traffic_bytes, traffic_file = profile_traffic(n, length) if args.workload == "se" else (None, None)
        res["roofline"] = {"bound": "hbm", "achieved": r4(achieved), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": r4(achieved / HBM_PEAK_GBS),
                           # HBM bytes per launch from the committed PMC profile of this kernel, per read x this
                           # run's reads, over this run's kernel time: profile-derived, not a same-run counter
                           "traffic": r4(traffic_bytes / (kern_avg_ms * 1e-3) / 1e9) if traffic_bytes else None,
                           "traffic_bytes_per_launch": int(traffic_bytes) if traffic_bytes else None,
                           "traffic_is": "profile-derived: %s" % traffic_file if traffic_file else None,
                           "kernel_ms_avg": r4(kern_avg_ms), "kernel_ms_min": r4(kern_ms[0]), "kernel_ms_max": r4(kern_ms[-1]),
                           "algorithmic_bytes_per_launch": algo_bytes}
    else:  # pipeline mode: the step is PCIe-bound; what the HBM roofline sees of it
        achieved = algo_bytes * args.steps / elapsed / 1e9 if world == 1 else None
        res["roofline"] = {"bound": "hbm", "achieved": r4(achieved), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": r4(achieved / HBM_PEAK_GBS) if achieved else None, "traffic": None,
                           "note": "whole step incl. H2D/D2H over PCIe; the kernel alone: --mode resident"}
    if h2d is not None:
        res["h2d_GBps_per_rank"] = [r4(x) for x in h2d]
        res["h2d_GBps_total"] = r4(sum(h2d))
    return res


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--settle", type=int, default=-1,
                    help="untimed steps before the warm-up steps, to get past the device's clock ramp (default: 150 resident, 1 pipeline)")
    ap.add_argument("--mode", choices=("resident", "pipeline"), default="resident")
    ap.add_argument("--workload", choices=("se", "mixed"), default="se")
    ap.add_argument("--batches", type=int, default=8, help="pipeline mode: batches per rank and step")
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU (weak scaling)")
    ap.add_argument("--total-reads", type=int, default=0,
                    help="strong scaling: this many reads in all, split into contiguous shards over the ranks (100000000 = BASELINE configs[3])")
    ap.add_argument("--mixed-reads", type=int, default=8_000_000, help="reads of the mixed job (always strong scaling)")
    ap.add_argument("--len", type=int, default=150)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline (+ cpu_baseline) only: no legs, variants, e2e")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--e2e-pairs", type=int, default=10_000_000, help="pairs both CLIs run, outputs compared")
    ap.add_argument("--e2e-big-pairs", type=int, default=50_000_000, help="pairs this CLI alone runs on (configs[3] size); 0 = skip")
    ap.add_argument("--e2e-mixed-pairs", type=int, default=4_000_000, help="pairs of the configs[4] run; 0 = skip")
    ap.add_argument("--full", action="store_true", help="long form of the variants (workload text, min/max): for profiles/, not for the driver")
    ap.add_argument("--variant", default="", help="time ONE kernel variant (tools/variants.py) instead of the headline: for rocprofv3 runs")
    args = ap.parse_args(argv)
    if args.settle < 0:
        args.settle = 150 if args.mode == "resident" else 1
    return args


def make_work(torch, capi, ctx, device, stream, args, plan, workload=None, mode=None):
    workload, mode = workload or args.workload, mode or args.mode
    if mode == "pipeline":
        return Pipeline(torch, capi, device, args, plan, workload)
    return (ResidentMixed if workload == "mixed" else ResidentSE)(torch, capi, ctx, device, stream, args, plan)


def run_leg(torch, capi, ctx, device, stream, args, rank, world, dist, reduce_device, barrier, workload, mode, steps):
    """One extra leg (outside the headline's timed region): the job of `workload` sharded over the ranks in `mode`,
    `steps` timed steps between barriers; rank 0 gets the aggregate."""
    from sickle_amd.shard import reduce_counters
    sub = argparse.Namespace(**vars(args))
    sub.workload, sub.mode, sub.steps = workload, mode, steps
    if workload == "mixed":
        sub.total_reads = 0
    plan = job_plan(sub, rank, world)
    work = make_work(torch, capi, ctx, device, stream, sub, plan)
    elapsed, kern_ms = timed_steps(torch, work, stream, steps, 1, 10 if mode == "resident" else 1, barrier)
    counts, tmax = reduce_counters(dist, work.counts() + [work.bases, work.algo_bytes], elapsed, reduce_device)
    h2d = gather_floats(torch, dist, work.h2d_bytes * steps / elapsed / 1e9, reduce_device) if mode == "pipeline" else None
    res = {"reads": counts[3], "Mreads_s": r4(counts[3] * steps / tmax / 1e6), "ms_per_step": r4(tmax / steps * 1e3),
           "kept": counts[0], "scaling": "strong" if plan["strong"] else "weak"}
    if kern_ms:
        avg = sum(kern_ms) / len(kern_ms)
        res.update({"k": short_kernel(work.kernel), "kernel_ms": r4(avg), "GBps": r4(work.algo_bytes / (avg * 1e-3) / 1e9),
                    "frac": r4(work.algo_bytes / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS)})
    if h2d is not None:
        res.update({"batches": args.batches, "h2d_GBps": [r4(x) for x in h2d], "h2d_GBps_total": r4(sum(h2d))})
    if hasattr(work, "close"):
        work.close()
    del work
    if device.type == "cuda":
        torch.cuda.empty_cache()
    return res


def gather_floats(torch, dist, x, device):
    """x of every rank, in rank order, on every rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(x)]
    t = torch.zeros(dist.get_world_size(), dtype=torch.float64, device=device)
    t[dist.get_rank()] = float(x)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]


def main():
    args = parse_args()
    import torch
    from sickle_amd import capi
    from sickle_amd.shard import reduce_counters

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the scan has no CPU path")
    # Rehearsal knobs (never set by the driver): BENCH_FORCE_DEVICE puts every rank on one GPU and
    # BENCH_DIST_BACKEND=gloo swaps RCCL for gloo, so that the N>1 code path can be exercised on a
    # one-GPU box.
    dev_index = int(os.environ.get("BENCH_FORCE_DEVICE", local_rank))
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # backend "nccl" IS RCCL on ROCm
        else:
            dist.init_process_group(backend)
    reduce_device = device if backend == "nccl" else None

    ctx = capi.Context(device=dev_index, slots=1)
    # A dedicated (non-default) stream: the C ABI takes the hipStream_t the kernel is launched on,
    # and the HIP events below are recorded on that same stream.
    stream = torch.cuda.Stream(device)

    if args.variant:  # one variant, many launches, one small JSON line: what tools/profile.sh traces
        res = run_variants(torch, capi, ctx, device, stream, [args.variant], args.steps, full=True)
        print(json.dumps({"variant": args.variant, **res[args.variant]}))
        ctx.close()
        return

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    plan = job_plan(args, rank, world)
    work = make_work(torch, capi, ctx, device, stream, args, plan)

    # Device clocks first: for its first ~50 launches after start-up the device ramps, overshoots and
    # settles (launch 2-9: 0.27 ms, 10-25: 0.31-0.34 ms, settled: 0.27 ms; tools/probes/bench_times.py),
    # which is about the length of a default run.  A fixed number of untimed launches carries the
    # measurement past that; they are outside the W warm-up steps and the K timed steps, and the JSON
    # line says so (settle_launches).
    elapsed, kern_ms = timed_steps(torch, work, stream, args.steps, args.warmup, args.settle, barrier)

    # the only exchange: kept / discarded counters, reads per rank and the max elapsed, outside the timed region
    counts, tmax = reduce_counters(dist, work.counts() + [work.bases], elapsed, reduce_device)
    h2d = gather_floats(torch, dist, work.h2d_bytes * args.steps / elapsed / 1e9, reduce_device) if args.mode == "pipeline" else None

    res = None
    if rank == 0:
        res = headline(args, plan, world, counts, tmax, kern_ms, work.kernel, work.algo_bytes, counts[4], h2d)
    default_run = args.mode == "resident" and args.workload == "se"
    if rank == 0 and world == 1 and default_run:
        achieved = res["roofline"]["achieved"]
        # the second denominator: what a read-only stream of this very buffer gets on this device
        try:
            peak = ctx.probe_read_bandwidth(work.qual.data_ptr(), work.n * work.stride, 30, stream.cuda_stream)
            res["roofline"]["peak_measured"] = r4(peak)
            res["roofline"]["frac_of_measured"] = r4(achieved / peak)
        except Exception as e:
            res["roofline"]["peak_measured_error"] = str(e)[:80]
        if not args.no_cpu_baseline:
            threads = host_cores()
            qh = work.qual.cpu().numpy().reshape(-1)
            base, cuts = cpu_baseline(qh, work.n, work.stride, work.length, threads)
            res["cpu_baseline"] = base
            # and the checker: the GPU cuts of the whole batch against the CPU path's
            res["parity_vs_cpu_baseline"] = bool((work.out.cpu().numpy() == cuts).all())
            res["speedup_vs_cpu_baseline"] = r4(res["value"] / base["value"])
            del qh, cuts
            if not res["parity_vs_cpu_baseline"]:
                print(json.dumps(res))
                raise SystemExit("GPU cuts differ from the CPU baseline's")
    if hasattr(work, "close"):
        work.close()
    del work
    torch.cuda.empty_cache()

    if default_run and not args.no_extras:
        # the legs every rank takes part in: configs[3]'s async batch pipeline and configs[4]'s mixed batch
        legs = {}
        for key, wk, md, st in (("pipe", "se", "pipeline", 3), ("mix", "mixed", "resident", 30), ("mixpipe", "mixed", "pipeline", 3)):
            try:
                legs[key] = run_leg(torch, capi, ctx, device, stream, args, rank, world, dist, reduce_device, barrier, wk, md, st)
            except Exception as e:
                if world > 1:
                    raise  # a rank that leaves a collective leg alone would hang the others
                legs[key] = {"error": ("%s: %s" % (type(e).__name__, e))[:80]}
        if rank == 0:
            res["legs"] = legs
            res["legs_are"] = "pipe: se via sk_submit/sk_wait (host buffers); mix: configs[4] batch resident, -n; mixpipe: the same via sk_submit/sk_wait"
    if rank == 0 and world == 1 and default_run and not args.no_extras:
        import variants
        res["variants"] = run_variants(torch, capi, ctx, device, stream, variants.NAMES, 30, full=args.full)
        res["variants_are"] = "tools/variants.py; k kernel, ms per scan, GBps algorithmic, frac of 8 TB/s"
        if not args.no_e2e:
            import e2e_bench
            threads = min(host_cores(), 16)
            try:
                res["e2e"] = e2e_bench.pe_against_reference(args.e2e_pairs, threads, big_pairs=args.e2e_big_pairs)
            except Exception as e:
                res["e2e"] = {"error": ("%s: %s" % (type(e).__name__, e))[:160]}
            if args.e2e_mixed_pairs:
                try:
                    res["e2e_mixed"] = e2e_bench.mixed_against_reference(args.e2e_mixed_pairs, threads)
                except Exception as e:
                    res["e2e_mixed"] = {"error": ("%s: %s" % (type(e).__name__, e))[:160]}
    if rank == 0:
        print(json.dumps(res))
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
