#!/usr/bin/env python3
"""Condenses a tools/profile.sh output directory (gpurun_out/prof_<tag>) into the two files kept
under profiles/<round>/: the kernel-stats CSV of the trace pass and pmc_latest.json (per-launch
averages of every counter for the scan kernel + the corrected HBM byte counts)."""
import collections, csv, glob, json, os, shutil, sys

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)
out, name = {}, None
for d in sorted(glob.glob(os.path.join(src, "pmc*/"))):
    f = glob.glob(d + "*/*counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if "sk_scan" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
            name = row["Kernel_Name"]
    for k, v in agg.items():
        out[k] = {"launches": len(v), "avg_per_launch": sum(v) / len(v)}
fetch = out["FETCH_SIZE"]["avg_per_launch"] * 1024 * 2  # KB -> B; x2: gfx950 counts half of a wide coalesced stream
write = out["WRITE_SIZE"]["avg_per_launch"] * 1024
tiles = 10_000_000 / 64
summary = {"kernel": name, "workload": "10M x 150bp, stride 152 (bench.py --no-cpu-baseline --steps 10 --warmup 2)",
           "counters": out,
           "hbm_bytes_per_launch": {"read": fetch, "write": write, "total": fetch + write,
                                    "note": "FETCH_SIZE (KB) x 1024 x 2 (gfx950 half-count correction for 16 B/lane streams, "
                                            "MI355X_MICROARCH.md HBM section) + WRITE_SIZE (KB) x 1024; separate --pmc passes"},
           "per_tile": {k: v["avg_per_launch"] / tiles for k, v in out.items() if k.startswith("SQ_")}}
json.dump(summary, open(os.path.join(dst, "pmc_latest.json"), "w"), indent=1)
stats = glob.glob(os.path.join(src, "trace/*/*kernel_stats.csv"))[0]
shutil.copyfile(stats, os.path.join(dst, "%s_kernel_stats.csv" % tag))
shutil.copyfile(os.path.join(src, "trace_bench.json"), os.path.join(dst, "%s_bench_under_trace.json" % tag))
# the stats file averages every launch of the run, the settle and warm-up launches included (and the
# pipeline leg's launches of the same kernel on smaller batches); the timed launches are the `steps`
# that follow the settle + warm-up launches at the head of the trace
trace = glob.glob(os.path.join(src, "trace/*/*kernel_trace.csv"))
if trace:
    line = json.loads(open(os.path.join(src, "trace_bench.json")).read().strip().splitlines()[-1])
    # the headline kernel only (the default bench run also launches every variant's kernels): the name the
    # counter passes saw most often
    head = name.split("(")[0] if name else "sk_scan"
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(trace[0])) if r["Kernel_Name"].startswith(head))
    d = [(e - s) / 1e6 for s, e in rows]
    k = line["steps"]
    first = line.get("settle_launches", 0) + line.get("warmup", 0)
    timed = d[first:first + k]
    json.dump({"kernel": head, "kernel_launches_in_trace": len(d), "avg_ms_all_launches": sum(d) / len(d), "timed_steps": k,
               "avg_ms_timed_launches": sum(timed) / k, "bench_events_avg_ms_same_run": line["roofline"]["kernel_ms_avg"]},
              open(os.path.join(dst, "%s_timed_launches.json" % tag), "w"), indent=1)
    print("trace: all %d launches avg %.4f ms, the %d timed ones %.4f ms, bench events %.4f ms" %
          (len(d), sum(d) / len(d), k, sum(timed) / k, line["roofline"]["kernel_ms_avg"]))
for row in csv.DictReader(open(stats)):
    if "sk_scan" in row["Name"]:
        print(row["Name"][:60], "calls", row["Calls"], "avg ns", row["AverageNs"])
print("VALU/tile %.0f  SALU/tile %.0f  traffic %.4f GB" % (summary["per_tile"]["SQ_INSTS_VALU"], summary["per_tile"]["SQ_INSTS_SALU"], (fetch + write) / 1e9))
