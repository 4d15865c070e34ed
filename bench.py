#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json: reads/s (+ Gbases/s) trimmed, 150 bp SE
Sanger, q=20 l=20, on N MI355X, with the quality bytes already resident in HBM.

A step = one pass of the scan (sk_scan_device_async, include/sickle_amd.h) over this rank's
whole batch of synthetic reads.  Two modes, one process per GPU in both, reads independent, nothing
exchanged in the timed region, only kept/discarded counters summed afterwards:
  weak   (default) every GPU holds --reads reads (10 M, BASELINE configs[1]);
  strong (--total-reads T, e.g. 100000000 = BASELINE configs[3]) the T reads are split into
         contiguous shards, rank r scans shard_range(T, r, world).

At N=1 the same JSON line also carries, measured OUTSIDE the timed region (each can be switched off):
  roofline.peak_measured  a read-only stream of the same buffer on the same device
  cpu_baseline            the reference's own sliding_window on the host cores
  variants                the other kernels (-n, 250 bp, segmented, packed, ragged, long reads)
  pipeline                sk_submit / sk_wait from pinned host memory (PCIe-inclusive)
  e2e                     the `sickle pe` binary against the compiled reference CLI, outputs compared

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

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def synth_quals_device(torch, n, length, stride, seed, device, chunk=1 << 20):
    """The quality model of sickle_amd/synth.py, generated on the GPU: (n, stride) uint8."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.zeros((n, stride), dtype=torch.uint8, device=device)
    pos = torch.arange(length, device=device, dtype=torch.float32)[None, :]
    for a in range(0, n, chunk):
        m = min(chunk, n - a)
        base = 30.0 + 10.0 * torch.rand((m, 1), generator=g, device=device)
        decay = 0.25 * torch.rand((m, 1), generator=g, device=device)
        q = base - decay * pos + 4.0 * torch.randn((m, length), generator=g, device=device)
        q = q.round_().clamp_(2, 41)
        head = torch.randint(0, 8, (m, 1), generator=g, device=device)
        q = torch.where(pos < head, torch.full_like(q, 2.0), q)
        isn = torch.rand((m, length), generator=g, device=device) < 0.002  # 0.2 % N bases: quality 2
        q = torch.where(isn, torch.full_like(q, 2.0), q)
        out[a:a + m, :length] = (q + 33.0).to(torch.uint8)
    return out


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


def pmc_traffic(n, length, kern_ms):
    """HBM bytes per launch from the newest committed rocprofv3 PMC summary (tools/profile.sh ->
    profiles/rNN/pmc_latest.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes),
    scaled to this run's read count; as GB/s over this run's kernel time.  None if absent."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_latest.json")), reverse=True):
        try:
            d = json.load(open(path))
            per_read = d["hbm_bytes_per_launch"]["total"] / float(d.get("reads_per_launch", 10_000_000))
            if length != 150:
                return None, None, None
            total = per_read * n
            return total / (kern_ms * 1e-3) / 1e9, total, os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return None, None, None


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
    return {"value": n / tn, "unit": "reads/s", "cores": threads, "kind": kind,
            "sample": "%d x %d reads of the same batch on %d threads (%.1f s wall); 1 thread: %.0f reads/s on %d reads"
                      % (reps, n, threads, tn * reps, m1 / t1, m1),
            "value_1thread": m1 / t1}, cuts


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


def run_variants(torch, capi, ctx, device, stream, names, reps):
    import variants
    me = sys.modules[__name__]
    out = {}
    for name in names:
        try:
            v = variants.build(name, torch, capi, ctx, device, stream, me)
            torch.cuda.synchronize(device)
            avg, lo, hi = time_launches(torch, stream, v["launch"], lambda: ctx.scan_device_finish(stream.cuda_stream), reps, 10)
            gbs = v["algo_bytes"] / (avg * 1e-3) / 1e9
            out[name] = {"workload": v["workload"], "kernel": v["kernel"], "reads": v["n_reads"], "kernel_ms_avg": avg,
                         "kernel_ms_min": lo, "kernel_ms_max": hi, "algorithmic_bytes_per_launch": v["algo_bytes"],
                         "achieved": gbs, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "reads_per_s": v["n_reads"] / (avg * 1e-3),
                         "scans_in_run": reps + 10}
            del v
            torch.cuda.empty_cache()
        except Exception as e:  # a failing variant must not cost the headline line
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def pipeline_rate(capi, n=4_000_000, length=150, stride=152, rounds=12):
    """sk_submit / sk_wait from pinned host memory, two slots: the H2D copy of batch i+1 overlaps the scan
    of batch i; the cuts come back by D2H.  What the C ABI costs when the caller's data is on the host."""
    import numpy as np
    from sickle_amd import synth
    ctx = capi.Context(0, 2)
    lib = capi.lib()
    _, qual = synth.make_reads(1, 200_000, length)
    tile = synth.pack_fixed(qual, stride)
    bufs, outs, raw = [], [], []
    for _ in range(2):
        p = lib.sk_host_alloc(ctx._h, n * stride)
        q = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(n * stride,))
        for a in range(0, n * stride, tile.size):
            m = min(tile.size, n * stride - a)
            q[a:a + m] = tile[:m]
        po = lib.sk_host_alloc(ctx._h, n * 8)
        o = np.ctypeslib.as_array(ctypes.cast(po, ctypes.POINTER(ctypes.c_int32)), shape=(n, 2))
        bufs.append(q)
        outs.append(o)
        raw.extend([p, po])
    params = capi.make_params("sanger", 20, 20)
    dt = None
    for r in (2, rounds):  # the first pass grows the device slots
        t0 = time.perf_counter()
        for i in range(r):
            s = i % 2
            if i >= 2:
                ctx.wait(s)
            ctx.submit(s, params, bufs[s], outs[s], stride=stride, read_len=length, n_reads=n)
        ctx.wait(0)
        ctx.wait(1)
        dt = time.perf_counter() - t0
    res = {"reads_per_s": rounds * n / dt, "h2d_gb_per_s": rounds * n * stride / dt / 1e9, "d2h_gb_per_s": rounds * n * 8 / dt / 1e9,
           "batches": rounds, "reads_per_batch": n, "slots": 2, "wall_s": dt,
           "what": "sk_submit/sk_wait, pinned host buffers, %d bp at stride %d: H2D + scan + D2H, copies overlapped with scans" % (length, stride)}
    for p in raw:
        lib.sk_host_free(ctx._h, p)
    ctx.close()
    return res


def job_plan(args, rank, world):
    """What this rank scans: (first read of the job it owns, number of reads, row stride, strong?).  Weak
    scaling: --reads per GPU.  Strong scaling (--total-reads): contiguous shards, sizes differ by at most one."""
    from sickle_amd.shard import shard_range
    strong = args.total_reads > 0
    if strong:
        lo, hi = shard_range(args.total_reads, rank, world)
        n = hi - lo
    else:
        lo, n = rank * args.reads, args.reads
    stride = (args.len + 7) // 8  # multiple of 8 with an odd number of 8-byte units: LDS-bank friendly rows
    stride = (stride + (1 - stride % 2)) * 8
    return {"lo": lo, "n": n, "stride": stride, "strong": strong}


def headline(args, plan, world, counts, elapsed, kern_ms, kernel_name):
    """Rank 0's JSON line from the job-wide counters [kept, discarded, bases kept, reads] and the max-over-ranks
    elapsed time of the K timed steps; kern_ms = this rank's per-launch kernel durations (HIP events)."""
    length, n, stride, strong = args.len, plan["n"], plan["stride"], plan["strong"]
    kern_ms = sorted(kern_ms)
    kern_avg_ms = sum(kern_ms) / len(kern_ms)
    job_reads = counts[3]
    total_reads = job_reads * args.steps
    algo_bytes = (length + 8) * n  # L quality bytes read + one 8-byte cut pair written, per read
    achieved = algo_bytes / (kern_avg_ms * 1e-3) / 1e9
    traffic, traffic_bytes, traffic_file = pmc_traffic(n, length, kern_avg_ms)
    if strong:
        workload = ("sickle se, %d synthetic %d bp Sanger reads sharded across %d GPU(s), q=20 l=20 (BASELINE configs[3]%s)"
                    % (args.total_reads, length, world, "" if args.total_reads == 100_000_000 else " shape"))
    else:
        workload = "sickle se, %d synthetic %d bp Sanger reads per GPU, q=20 l=20 (BASELINE configs[1])" % (n, length)
    return {
        "metric": "reads/sec trimmed (+ Gbases/sec), 150 bp SE Sanger q20 l20, inputs resident in HBM",
        "value": total_reads / elapsed, "unit": "reads/s",
        "gbases_per_s": total_reads * length / elapsed / 1e9,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_launches": args.settle,
        "untimed_launches_before_timing": args.warmup + args.settle,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": workload, "reads_per_gpu": n, "reads_in_job": job_reads, "read_len": length, "stride": stride,
                   "kernel": kernel_name, "sharding": "reads split across ranks, no collective"},
        "kept": counts[0], "discarded": counts[1],
        "mean_bases_kept": counts[2] / max(1, counts[0]),
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_bytes_per_launch": traffic_bytes,
                     "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, %s" % traffic_file,
                     "kernel_ms_avg": kern_avg_ms, "kernel_ms_min": kern_ms[0], "kernel_ms_max": kern_ms[-1],
                     "algorithmic_bytes_per_launch": algo_bytes},
    }


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--settle", type=int, default=150,
                    help="untimed launches before the warm-up steps, to get past the device's clock ramp (0 = none)")
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU (weak scaling)")
    ap.add_argument("--total-reads", type=int, default=0,
                    help="strong scaling: this many reads in all, split into contiguous shards over the ranks (100000000 = BASELINE configs[3])")
    ap.add_argument("--len", type=int, default=150)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline (+ cpu_baseline) only: no variants, pipeline, e2e")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--e2e-pairs", type=int, default=10_000_000)
    ap.add_argument("--variant", default="", help="time ONE kernel variant (tools/variants.py) instead of the headline: for rocprofv3 runs")
    return ap.parse_args(argv)


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
        res = run_variants(torch, capi, ctx, device, stream, [args.variant], args.steps)
        print(json.dumps({"variant": args.variant, **res[args.variant]}))
        ctx.close()
        return

    length = args.len
    plan = job_plan(args, rank, world)
    n, stride = plan["n"], plan["stride"]
    # every read of the job has its own seed block: the shards of a strong-scaling run are different data
    qual = synth_quals_device(torch, n, length, stride, 1234 + rank, device)
    out = torch.empty((n, 2), dtype=torch.int32, device=device)
    params = capi.make_params("sanger", 20, 20)
    torch.cuda.synchronize(device)  # the synthetic batch is complete before anything is launched

    kernel_id = capi.lib().sk_kernel_for(ctypes.byref(capi.Batch(qual.data_ptr(), None, None, stride, length, None, n)))

    def step():
        ctx.scan_device_async(params, qual.data_ptr(), out.data_ptr(), n, stride=stride, read_len=length,
                              stream=stream.cuda_stream)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    # Device clocks first: for its first ~50 launches after start-up the device ramps, overshoots and
    # settles (launch 2-9: 0.27 ms, 10-25: 0.31-0.34 ms, settled: 0.27 ms; tools/probes/bench_times.py),
    # which is about the length of a default run.  A fixed number of untimed launches carries the
    # measurement past that; they are outside the W warm-up steps and the K timed steps, and the JSON
    # line says so (settle_launches, untimed_launches_before_timing).
    for _ in range(args.settle):
        step()
    ctx.scan_device_finish(stream.cuda_stream)
    for _ in range(args.warmup):
        step()
    ctx.scan_device_finish(stream.cuda_stream)  # raises on a range error

    # per-launch kernel durations: HIP events on the launch stream
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for a, b in evs:
        a.record(stream)
        step()
        b.record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.scan_device_finish(stream.cuda_stream)
    kern_ms = [a.elapsed_time(b) for a, b in evs]

    kept = int((out[:, 1] >= 0).sum().item())
    bases_kept = int((out[:, 1] - out[:, 0]).clamp_(min=0).sum().item())
    # the only exchange: kept / discarded counters, reads per rank and the max elapsed, outside the timed region
    counts, elapsed = reduce_counters(dist, [kept, n - kept, bases_kept, n], elapsed, reduce_device)

    res = None
    if rank == 0:
        res = headline(args, plan, world, counts, elapsed, kern_ms, capi.lib().sk_kernel_name(kernel_id).decode())
        achieved = res["roofline"]["achieved"]
        if world == 1:
            # the second denominator: what a read-only stream of this very buffer gets on this device
            try:
                peak = ctx.probe_read_bandwidth(qual.data_ptr(), n * stride, 30, stream.cuda_stream)
                res["roofline"]["peak_measured"] = peak
                res["roofline"]["frac_of_measured"] = achieved / peak
                res["roofline"]["peak_measured_what"] = "read-only kernel (16-byte nt loads, no stores) over the same %d-byte buffer, 30 launches" % (n * stride)
            except Exception as e:
                res["roofline"]["peak_measured_error"] = str(e)
        if world == 1 and not args.no_cpu_baseline:
            threads = host_cores()
            qh = qual.cpu().numpy().reshape(-1)
            base, cuts = cpu_baseline(qh, n, stride, length, threads)
            res["cpu_baseline"] = base
            # and the checker: the GPU cuts of the whole batch against the CPU path's
            res["parity_vs_cpu_baseline"] = bool((out.cpu().numpy() == cuts).all())
            res["speedup_vs_cpu_baseline"] = res["value"] / base["value"]
            del qh, cuts
            if not res["parity_vs_cpu_baseline"]:
                print(json.dumps(res))
                raise SystemExit("GPU cuts differ from the CPU baseline's")
        if world == 1 and not args.no_extras:
            del qual, out
            torch.cuda.empty_cache()
            import variants
            res["variants"] = run_variants(torch, capi, ctx, device, stream, variants.NAMES, 30)
            try:
                res["pipeline"] = pipeline_rate(capi)
            except Exception as e:
                res["pipeline"] = {"error": "%s: %s" % (type(e).__name__, e)}
            if not args.no_e2e:
                try:
                    import e2e_bench
                    res["e2e"] = e2e_bench.pe_against_reference(args.e2e_pairs, min(host_cores(), 16))
                except Exception as e:
                    res["e2e"] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(res))
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
