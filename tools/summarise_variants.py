#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/<variant>/ (tools/profile_variants.sh) into profiles/<round>/:
<variant>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `bench.py --variant V --steps 30`) and
variants_pmc.json: per variant the scan kernel's average duration from the trace, the bench's own HIP-event
average in that run, the counters per launch, the HBM bytes per launch corrected as MI355X_MICROARCH.md
prescribes (FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024, separate passes) and their ratio to the
algorithmic bytes."""
import collections, csv, glob, json, os, shutil, sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
summary = {}
for vdir in sorted(glob.glob(os.path.join(src, "*/"))):
    v = os.path.basename(vdir.rstrip("/"))
    try:
        line = json.loads(open(os.path.join(vdir, "trace_bench.json")).read().strip().splitlines()[-1])
    except (OSError, ValueError, IndexError):
        continue
    if "error" in line:
        summary[v] = {"error": line["error"]}
        continue
    want = line["kernel"]
    stats = glob.glob(os.path.join(vdir, "trace/*/*kernel_stats.csv"))
    ent = {"workload": line["workload"], "kernel": want, "bench_events_avg_ms": line["kernel_ms_avg"],
           "algorithmic_bytes_per_launch": line["algorithmic_bytes_per_launch"], "algorithmic_GBps_by_events": line["achieved"]}
    if stats:
        shutil.copyfile(stats[0], os.path.join(dst, "%s_kernel_stats.csv" % v))
        rows = [r for r in csv.DictReader(open(stats[0])) if "sk_scan" in r["Name"] or "sk_sort" in r["Name"]]
        ent["trace"] = [{"name": r["Name"][:100], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"])} for r in rows]
        tot = sum(float(r["AverageNs"]) * int(r["Calls"]) for r in rows)
        # a variant may take two kernels per scan (ragged: tile + general): time per scan = total / scans
        scans = int(line.get("scans_in_run", 0)) or (max(int(r["Calls"]) for r in rows) if rows else 0)
        if scans:
            ent["trace_avg_ms_per_scan"] = tot / scans / 1e6
            ent["algorithmic_GBps_by_trace"] = line["algorithmic_bytes_per_launch"] / (tot / scans) if tot else None
    counters = {}
    for d in sorted(glob.glob(os.path.join(vdir, "pmc*/"))):
        f = glob.glob(d + "*/*counter_collection.csv")
        if not f:
            continue
        try:  # scans in THIS counter run (settle + steps), from the run's own JSON line
            pl = json.loads(open(d.rstrip("/") + "_bench.json").read().strip().splitlines()[-1])
            nscans = int(pl["scans_in_run"])
        except (OSError, ValueError, KeyError, IndexError):
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for row in csv.DictReader(open(f[0])):
            if "sk_scan" in row["Kernel_Name"] or "sk_sort" in row["Kernel_Name"]:  # every kernel of the scan (a ragged scan launches several)
                agg[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for k, per in agg.items():
            counters[k] = {"dispatches": len(per), "scans": nscans, "per_scan": sum(per.values()) / nscans}
    ent["counters_per_scan"] = counters
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        fetch = counters["FETCH_SIZE"]["per_scan"] * 1024 * 2
        write = counters["WRITE_SIZE"]["per_scan"] * 1024
        ent["hbm_bytes_per_scan"] = {"read": fetch, "write": write, "total": fetch + write,
                                     "over_algorithmic": (fetch + write) / line["algorithmic_bytes_per_launch"],
                                     "note": "FETCH_SIZE (KB) x 1024 x 2 (gfx950 correction) + WRITE_SIZE (KB) x 1024; separate --pmc passes"}
    summary[v] = ent
    print("%-11s events %.3f ms  trace %s ms  traffic/algorithmic %s" % (
        v, ent["bench_events_avg_ms"], "%.3f" % ent["trace_avg_ms_per_scan"] if "trace_avg_ms_per_scan" in ent else "-",
        "%.3f" % ent["hbm_bytes_per_scan"]["over_algorithmic"] if "hbm_bytes_per_scan" in ent else "-"))
json.dump(summary, open(os.path.join(dst, "variants_pmc.json"), "w"), indent=1)
