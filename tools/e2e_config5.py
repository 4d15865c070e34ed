#!/usr/bin/env python3
"""End-to-end run of BASELINE config 5 (not the bench metric): paired reads of mixed length
L in U{75..301} (same L for both mates), phred+64 (`-t illumina`), 0.3 % N, 5 % of reads with a
lowercase n, `-n`, gzip input -- the product CLI against the compiled reference CLI, outputs compared.
Exercises the segmented kernel with the sequence tile, the parallel gzip decoder and the N rule at
scale.  Usage: e2e_config5.py [pairs]"""
import gzip, hashlib, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sickle_amd import synth
NEW = os.path.join(ROOT, "sickle_amd", "sickle")
REF = os.path.join(ROOT, "oracle", "_ref", "sickle")


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 22), b""):
            h.update(chunk)
    return h.hexdigest()


def write_pair(d, n, chunk=50_000):
    p1, p2 = os.path.join(d, "R1.fastq"), os.path.join(d, "R2.fastq")
    with open(p1, "wb") as f1, open(p2, "wb") as f2:
        for k, a in enumerate(range(0, n, chunk)):
            m = min(chunk, n - a)
            s1, q1, off = synth.make_ragged_reads(9000 + k, m, 75, 301, "illumina")
            # mate 2: same lengths, other content
            s2, q2, off2 = synth.make_ragged_reads(9000 + k, m, 75, 301, "illumina")
            rng = np.random.default_rng(77 + k)
            perm = rng.permutation(len(q2))
            lens = np.diff(off).astype(np.int64)
            # shuffle the quality/sequence bytes of mate 2 inside each read (keeps N/n frequencies, changes the cuts)
            idx = np.concatenate([int(off[i]) + rng.permutation(int(lens[i])) for i in range(m)]) if m else perm
            s2, q2 = s2[idx], q2[idx]
            f1.write(synth.fastq_bytes_ragged(s1, q1, off, start=a, suffix="/1"))
            f2.write(synth.fastq_bytes_ragged(s2, q2, off, start=a, suffix="/2"))
    return p1, p2


def run(binary, d, tag, p1, p2):
    outs = [os.path.join(d, "%s_%s.fastq" % (tag, k)) for k in ("o1", "o2", "os")]
    t0 = time.perf_counter()
    pr = subprocess.run([binary, "pe", "-f", p1, "-r", p2, "-t", "illumina", "-n", "-o", outs[0], "-p", outs[1], "-s", outs[2], "-a", "1"],
                        capture_output=True)
    dt = time.perf_counter() - t0
    assert pr.returncode == 0, pr.stderr.decode()[-500:]
    text = [l for l in pr.stdout.decode().split("\n") if l.startswith("FastQ")]
    return dt, [md5(o) for o in outs], text


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
    res = {"pairs": n, "reads": 2 * n, "config": "PE, L in U{75..301}, illumina, -n, gzip input"}
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
        p1, p2 = write_pair(d, n)
        subprocess.run(["gzip", "-1", p1, p2], check=True)
        p1, p2 = p1 + ".gz", p2 + ".gz"
        res["input_bytes"] = os.path.getsize(p1) + os.path.getsize(p2)
        run(NEW, d, "warm", p1, p2)
        t_new, m_new, s_new = run(NEW, d, "new", p1, p2)
        res["new_s"], res["new_reads_per_s"], res["summary"] = t_new, 2 * n / t_new, s_new
        if os.path.exists(REF):
            t_ref, m_ref, s_ref = run(REF, d, "ref", p1, p2)
            res["ref_a1_s"], res["identical_to_ref_a1"], res["speedup_vs_ref_a1"] = t_ref, m_ref == m_new, t_ref / t_new
    print(json.dumps(res))


if __name__ == "__main__":
    main()
