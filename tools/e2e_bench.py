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


def run(binary, d, tag, p1, p2, threads, gzout=False):
    outs = [os.path.join(d, "%s_%s.fastq" % (tag, k)) for k in ("o1", "o2", "os")]
    t0 = time.perf_counter()
    pr = subprocess.run([binary, "pe", "-f", p1, "-r", p2, "-t", "sanger", "-o", outs[0], "-p", outs[1], "-s", outs[2],
                         "-a", str(threads)] + (["-g"] if gzout else []), capture_output=True)
    dt = time.perf_counter() - t0
    assert pr.returncode == 0, pr.stderr.decode()[-500:]
    if gzout:  # compare what the files inflate to (any gzip reader: here zcat)
        sums = []
        for o in outs:
            h = hashlib.md5()
            z = subprocess.Popen(["zcat", o], stdout=subprocess.PIPE)
            for blk in iter(lambda: z.stdout.read(1 << 24), b""):
                h.update(blk)
            assert z.wait() == 0
            sums.append(h.hexdigest())
        return dt, sums, sum(os.path.getsize(o) for o in outs)
    return dt, [md5(o) for o in outs]


def pick_tmp(need_bytes):
    """A directory with room for the run: tmpfs first (the comparison is about the programs, not a disk)."""
    import shutil
    for d in (os.environ.get("SICKLE_E2E_TMP"), "/dev/shm", os.environ.get("TMPDIR"), "/tmp"):
        if d and os.path.isdir(d) and os.access(d, os.W_OK) and shutil.disk_usage(d).free > need_bytes:
            return d
    return None


def same_files(a, b):
    return os.path.getsize(a) == os.path.getsize(b) and subprocess.run(["cmp", "-s", a, b]).returncode == 0


def pe_against_reference(n_pairs, threads=None):
    """`sickle pe` of this repo (HIP) and the compiled reference CLI (oracle/_ref/sickle, -a 1 and -a threads)
    on the same synthetic two-file input of n_pairs 150 bp pairs in tmpfs: wall times of the whole
    processes, outputs compared byte for byte (-a 1; the reference's -a T order differs by design).
    Reference path timed: /root/reference/src/trim_paired.cpp:265-476."""
    res = {"pairs": n_pairs, "reads": 2 * n_pairs, "read_len": 150, "command": "sickle pe -f R1 -r R2 -t sanger -o O1 -p O2 -s OS -a T"}
    need = n_pairs * 330 * 2 * 3 + (1 << 30)  # input + two sets of outputs
    root = pick_tmp(need)
    if root is None:
        res["error"] = "no temporary directory with %d bytes free" % need
        return res
    if threads is None:
        threads = min(len(os.sched_getaffinity(0)), 16)
    with tempfile.TemporaryDirectory(dir=root) as d:
        res["tmp"] = root
        t0 = time.perf_counter()
        p1, p2 = write_pair(d, n_pairs)
        res["input_bytes"] = os.path.getsize(p1) + os.path.getsize(p2)
        res["generate_s"] = time.perf_counter() - t0

        def go(binary, tag, a, env=None):
            outs = [os.path.join(d, "%s_%s.fastq" % (tag, k)) for k in ("o1", "o2", "os")]
            t0 = time.perf_counter()
            pr = subprocess.run([binary, "pe", "-f", p1, "-r", p2, "-t", "sanger", "-o", outs[0], "-p", outs[1], "-s", outs[2],
                                 "-a", str(a)], capture_output=True, env=env)
            dt = time.perf_counter() - t0
            if pr.returncode != 0:
                raise RuntimeError("%s exited %d: %s" % (binary, pr.returncode, pr.stderr.decode("latin-1")[-300:]))
            return dt, outs

        go(NEW, "warm", 1)  # first touch of the GPU runtime and of the input's pages
        t_new, o_new = go(NEW, "new", 1)
        res["new_s"], res["new_reads_per_s"] = t_new, 2 * n_pairs / t_new
        res["output_bytes"] = sum(os.path.getsize(o) for o in o_new)
        t_newn, o_newn = go(NEW, "newn", threads)
        res["new_aN_s"] = t_newn
        for o in o_newn:
            os.unlink(o)
        # new_s is what the caller waits for: the CLI's front process leaves once the outputs are closed, the worker's
        # address space (mapped input, pinned staging, HIP context) is torn down behind it.  The same run in ONE
        # process, teardown included:
        time.sleep(1.0)  # (the teardown of the run before is still going on)
        t_single, o_single = go(NEW, "single", 1, env=dict(os.environ, SICKLE_NO_FRONT="1"))
        res["new_single_process_s"] = t_single
        res["new_s_is"] = "wall clock until the launching process gets control back, outputs complete and closed (front process, host/sickle.h)"
        for o in o_single:
            os.unlink(o)
        time.sleep(1.0)
        if os.path.exists(REF):
            t_ref1, o_ref1 = go(REF, "ref1", 1)
            res["ref_a1_s"], res["ref_a1_reads_per_s"] = t_ref1, 2 * n_pairs / t_ref1
            res["byte_identical"] = all(same_files(a, b) for a, b in zip(o_new, o_ref1))
            for o in o_ref1:
                os.unlink(o)
            t_refn, o_refn = go(REF, "refn", threads)
            res["ref_aN_s"], res["ref_aN_threads"], res["ref_aN_reads_per_s"] = t_refn, threads, 2 * n_pairs / t_refn
            res["speedup_vs_ref_a1"], res["speedup_vs_ref_aN"] = t_ref1 / t_new, t_refn / t_new
        else:
            res["reference"] = "oracle/_ref/sickle did not travel with the repo: reference not timed"
    return res


def to_bgzf(d, path):
    """A BGZF copy of a FASTQ file, made by this CLI itself: -q 0 -l 0 keeps every read whole."""
    out = path + ".bgzf.gz"
    pr = subprocess.run([NEW, "se", "-f", path, "-t", "sanger", "-o", out, "-q", "0", "-l", "0", "-g", "-a", "1"], capture_output=True)
    assert pr.returncode == 0, pr.stderr.decode()[-500:]
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    mode = sys.argv[2] if len(sys.argv) > 2 else ""
    gz = mode == "gz"
    res = {"pairs": n, "reads": 2 * n, "gzip_input": gz, "mode": mode}
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
        p1, p2 = write_pair(d, n)
        if mode in ("bgzf", "gzout"):
            # BGZF input and/or -g output of this CLI against its own plain run (the reference's -g
            # hands the records to gzprintf as a format string and cannot be compared)
            run(NEW, d, "warm", p1, p2, 1)
            t_plain, m_plain = run(NEW, d, "plain", p1, p2, 1)
            res["plain_s"] = t_plain
            if mode == "bgzf":
                b1, b2 = to_bgzf(d, p1), to_bgzf(d, p2)
                assert subprocess.run("zcat %s | cmp - %s" % (b1, p1), shell=True).returncode == 0
                res["bgzf_input_bytes"] = os.path.getsize(b1) + os.path.getsize(b2)
                t, m = run(NEW, d, "bgzf", b1, b2, 1)
                res["bgzf_in_s"], res["bgzf_in_reads_per_s"], res["bgzf_in_identical"] = t, 2 * n / t, m == m_plain
                env = dict(os.environ, SICKLE_NO_BGZF="1")
                t0 = time.perf_counter()
                pr = subprocess.run([NEW, "pe", "-f", b1, "-r", b2, "-t", "sanger", "-o", d + "/s1", "-p", d + "/s2", "-s",
                                     d + "/s3"], capture_output=True, env=env)
                res["same_file_streamed_s"] = time.perf_counter() - t0
                assert pr.returncode == 0
            else:
                t, m, size = run(NEW, d, "gzout", p1, p2, 1, gzout=True)
                res["gz_out_s"], res["gz_out_reads_per_s"], res["gz_out_identical"] = t, 2 * n / t, m == m_plain
                res["gz_out_bytes"] = size
                for setting in ("gpu", "fast"):
                    os.environ["SICKLE_GZ_LEVEL"] = setting
                    run(NEW, d, "gzwarm", p1, p2, 1, gzout=True)
                    t, m, size = run(NEW, d, "gz_" + setting, p1, p2, 1, gzout=True)
                    del os.environ["SICKLE_GZ_LEVEL"]
                    res["gz_out_%s_s" % setting], res["gz_out_%s_identical" % setting], res["gz_out_%s_bytes" % setting] = t, m == m_plain, size
            print(json.dumps(res))
            return
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
