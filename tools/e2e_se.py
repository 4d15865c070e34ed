#!/usr/bin/env python3
"""Diagnostic: `sickle se` end to end on one synthetic file (the reference's SE driver crashes,
so the comparison is its `pe` on the file paired with a copy of itself, which writes the
SE-equivalent output to file 1 while doing twice the work)."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import e2e_bench as eb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/dev/shm")) as d:
    p1, p2 = eb.write_pair(d, n)
    os.remove(p2)
    out = os.path.join(d, "se.out")
    for i in range(2):
        t0 = time.perf_counter()
        pr = subprocess.run([eb.NEW, "se", "-f", p1, "-t", "sanger", "-o", out, "-a", "1"], capture_output=True)
        dt = time.perf_counter() - t0
        assert pr.returncode == 0, pr.stderr
    res = {"reads": n, "se_s": dt, "se_reads_per_s": n / dt, "md5": eb.md5(out)}
    if os.path.exists(eb.REF) and n <= 20_000_000:
        cp = os.path.join(d, "copy.fastq")
        subprocess.run(["cp", p1, cp], check=True)
        t0 = time.perf_counter()
        pr = subprocess.run([eb.REF, "pe", "-f", p1, "-r", cp, "-t", "sanger", "-o", d + "/r1", "-p", d + "/r2", "-s", d + "/rs", "-a", "1"], capture_output=True)
        res["ref_selfpair_s"] = time.perf_counter() - t0
        res["identical_to_ref_file1"] = eb.md5(d + "/r1") == res["md5"]
    print(json.dumps(res))
