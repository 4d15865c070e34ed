#!/usr/bin/env python3
"""End-to-end CLI timing on the GPU box: this repo's `sickle pe` (HIP) against the compiled
reference CLI (oracle/_ref/sickle, when it travelled with the repo) on the same synthetic
two-file paired input, outputs compared byte for byte.  Not part of bench.py's metric (that is
the HBM-resident scan); this is the PCIe- and parse-inclusive number DESIGN.md quotes."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sickle_amd import synth  # noqa: E402

NEW = os.path.join(ROOT, "sickle_amd", "sickle")
REF = os.path.join(ROOT, "oracle", "_ref", "sickle")


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 22), b""):
            h.update(chunk)
    return h.hexdigest()


def write_pair(d, n, chunk=250_000):
    """Two FASTQ files of n reads each.  The read content cycles through one seeded block of
    `chunk` reads per file (rotated per block), the headers count up: cheap to generate at tens
    of millions of reads, and still a different byte stream in every block."""
    p1, p2 = os.path.join(d, "R1.fastq"), os.path.join(d, "R2.fastq")
    s1, q1 = synth.make_reads(1000, chunk, 150, "sanger")
    s2, q2 = synth.make_reads(5000, chunk, 150, "sanger")
    with open(p1, "wb") as f1, open(p2, "wb") as f2:
        for k, a in enumerate(range(0, n, chunk)):
            m = min(chunk, n - a)
            r = (k * 7919) % chunk
            idx = (np.arange(m) + r) % chunk
            f1.write(synth.fastq_bytes_fast(s1[idx], q1[idx], start=a, suffix="/1"))
            f2.write(synth.fastq_bytes_fast(s2[idx], q2[idx], start=a, suffix="/2"))
    return p1, p2


def run(binary, d, tag, p1, p2, threads):
    outs = [os.path.join(d, "%s_%s.fastq" % (tag, k)) for k in ("o1", "o2", "os")]
    t0 = time.perf_counter()
    pr = subprocess.run([binary, "pe", "-f", p1, "-r", p2, "-t", "sanger", "-o", outs[0], "-p", outs[1], "-s", outs[2],
                         "-a", str(threads)], capture_output=True)
    dt = time.perf_counter() - t0
    assert pr.returncode == 0, pr.stderr.decode()[-500:]
    return dt, [md5(o) for o in outs]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    gz = len(sys.argv) > 2 and sys.argv[2] == "gz"
    res = {"pairs": n, "reads": 2 * n, "gzip_input": gz}
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
        p1, p2 = write_pair(d, n)
        if gz:  # both tools inflate with zlib, one stream per file
            subprocess.run(["gzip", "-1", p1, p2], check=True)
            p1, p2 = p1 + ".gz", p2 + ".gz"
        res["input_bytes"] = os.path.getsize(p1) + os.path.getsize(p2)
        run(NEW, d, "warm", p1, p2, 1)  # first touch of the GPU runtime and the page cache
        t_new, m_new = run(NEW, d, "new", p1, p2, 1)
        res["new_s"] = t_new
        res["new_reads_per_s"] = 2 * n / t_new
        if os.path.exists(REF):
            t_ref1, m_ref1 = run(REF, d, "ref1", p1, p2, 1)
            res["ref_a1_s"] = t_ref1
            res["ref_a1_reads_per_s"] = 2 * n / t_ref1
            res["identical_to_ref_a1"] = m_ref1 == m_new
            if m_ref1 != m_new:
                # the reference's per-batch output threads race each other for the file
                # (src/trim_paired.cpp:445-458), so on many-batch inputs its batch ORDER can differ
                # from run to run: compare the records themselves, order-independently
                fp = os.path.join(ROOT, "tools", "probes", "fq_fingerprint.bin")
                if not os.path.exists(fp):
                    subprocess.run(["g++", "-O2", "-o", fp, fp[:-4] + ".cpp"], check=True)
                same = []
                for k in ("o1", "o2", "os"):
                    a = subprocess.run([fp, os.path.join(d, "new_%s.fastq" % k)], capture_output=True).stdout
                    b = subprocess.run([fp, os.path.join(d, "ref1_%s.fastq" % k)], capture_output=True).stdout
                    same.append(a == b and len(a) > 0)
                res["same_records_as_ref_a1"] = all(same)
            cores = len(os.sched_getaffinity(0))
            cores = min(cores, 16)
            t_refn, _ = run(REF, d, "refn", p1, p2, cores)
            res["ref_aN_s"] = t_refn
            res["ref_aN_threads"] = cores
            res["ref_aN_reads_per_s"] = 2 * n / t_refn
            res["speedup_vs_ref_a1"] = t_ref1 / t_new
            res["speedup_vs_ref_aN"] = t_refn / t_new
    print(json.dumps(res))


if __name__ == "__main__":
    main()
