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
sys.path.insert(0, os.path.join(ROOT, "tools"))
from sickle_amd import synth  # noqa: E402

NEW = os.path.join(ROOT, "sickle_amd", "sickle")
REF = os.path.join(ROOT, "oracle", "_ref", "sickle")


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 22), b""):
            h.update(chunk)
    return h.hexdigest()


RECORD = 321  # bytes of one synthetic record: "@SYN:%09d/1\n" + 150 + "\n+\n" + 150 + "\n"
CHUNK = 250_000


def write_part(d, first_chunk, n_chunks, n):
    """Chunks [first_chunk, first_chunk + n_chunks) of the two files, written in place (records have a fixed
    width, so every chunk knows its offset): one worker process of write_pair()."""
    s1, q1, s2, q2 = np.load(os.path.join(d, "base.npy"), mmap_mode="r")
    f1 = os.open(os.path.join(d, "R1.fastq"), os.O_WRONLY)
    f2 = os.open(os.path.join(d, "R2.fastq"), os.O_WRONLY)
    for k in range(first_chunk, first_chunk + n_chunks):
        a = k * CHUNK
        m = min(CHUNK, n - a)
        if m <= 0:
            break
        r = (k * 7919) % CHUNK
        idx = (np.arange(m) + r) % CHUNK
        os.pwrite(f1, synth.fastq_bytes_fast(s1[idx], q1[idx], start=a, suffix="/1"), a * RECORD)
        os.pwrite(f2, synth.fastq_bytes_fast(s2[idx], q2[idx], start=a, suffix="/2"), a * RECORD)
    os.close(f1)
    os.close(f2)


def write_pair(d, n, start=0, workers=None):
    """Two FASTQ files of n reads each (records [start, n) are written; the files are created or grown).  The
    read content cycles through one seeded block of CHUNK reads per file (rotated per chunk), the headers count
    up: cheap to generate at tens of millions of reads, and still a different byte stream in every chunk.
    Written by `workers` child processes side by side (children of a process that holds a GPU context never
    touch the GPU: they are fresh interpreters, not forks)."""
    assert start % CHUNK == 0
    p1, p2 = os.path.join(d, "R1.fastq"), os.path.join(d, "R2.fastq")
    for p in (p1, p2):
        with open(p, "ab") as f:
            f.truncate(n * RECORD)
    base = os.path.join(d, "base.npy")  # the two seeded blocks, made once, read by every worker
    if not os.path.exists(base):
        np.save(base, np.stack(synth.make_reads(1000, CHUNK, 150, "sanger") + synth.make_reads(5000, CHUNK, 150, "sanger")))
    chunks = (n - start + CHUNK - 1) // CHUNK
    workers = max(1, min(workers or min(len(os.sched_getaffinity(0)), 16), chunks))
    per = (chunks + workers - 1) // workers
    procs = []
    for w in range(workers):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "_part", d, str(start // CHUNK + w * per), str(per), str(n)]))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("input generator failed")
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


def timed_run(argv, env=None):
    t0 = time.perf_counter()
    pr = subprocess.run(argv, capture_output=True, env=env)
    dt = time.perf_counter() - t0
    if pr.returncode != 0:
        raise RuntimeError("%s exited %d: %s" % (argv[0], pr.returncode, pr.stderr.decode("latin-1")[-300:]))
    return dt, pr.stdout.decode("latin-1")


def r4(x):
    return float("%.4g" % x)


def pe_against_reference(n_pairs, threads=None, big_pairs=0):
    """`sickle pe` of this repo (HIP) and the compiled reference CLI (oracle/_ref/sickle, -a 1 and -a threads)
    on the same synthetic two-file input of n_pairs 150 bp pairs in tmpfs: wall times of the whole
    processes, outputs compared byte for byte (-a 1; the reference's -a T order differs by design).
    Reference path timed: /root/reference/src/trim_paired.cpp:265-476.

    new_s (and every speed-up) is ONE process from exec to exit, teardown of the GPU context, the pinned staging
    and the mapped input included (SICKLE_NO_FRONT=1): like for like with the reference's process.  front_s is the
    default invocation, whose front process (host/sickle.h) hands control back when the outputs are complete
    and closed and leaves the teardown to the worker behind it.

    big_pairs > n_pairs: the input is then grown to big_pairs (50 M pairs = 100 M reads = BASELINE configs[3]) and
    this CLI runs on it; the reference is NOT run at that size (90 s and 70 s per setting, 63 GiB resident) --
    its time is extrapolated per read from the n_pairs run and labelled so."""
    res = {"pairs": n_pairs, "cmd": "sickle pe -f R1 -r R2 -t sanger -o O1 -p O2 -s OS -a T"}
    need = max(n_pairs * 3, big_pairs * 2) * RECORD * 2 + (1 << 30)  # input + outputs (two sets at n_pairs)
    root = pick_tmp(need)
    if root is None:
        res["error"] = "no temporary directory with %d bytes free" % need
        return res
    if threads is None:
        threads = min(len(os.sched_getaffinity(0)), 16)
    no_front = dict(os.environ, SICKLE_NO_FRONT="1")
    with tempfile.TemporaryDirectory(dir=root) as d:
        t0 = time.perf_counter()
        p1, p2 = write_pair(d, n_pairs)
        res["generate_s"] = r4(time.perf_counter() - t0)

        def go(binary, tag, a, env=None):
            outs = [os.path.join(d, "%s_%s.fastq" % (tag, k)) for k in ("o1", "o2", "os")]
            dt, _ = timed_run([binary, "pe", "-f", p1, "-r", p2, "-t", "sanger", "-o", outs[0], "-p", outs[1], "-s", outs[2], "-a", str(a)], env)
            return dt, outs

        def drop(outs):
            for o in outs:
                os.unlink(o)

        drop(go(NEW, "warm", 1, no_front)[1])  # first touch of the GPU runtime and of the input's pages
        t_new, o_new = go(NEW, "new", 1, no_front)
        res["new_s"], res["new_Mreads_s"] = r4(t_new), r4(2 * n_pairs / t_new / 1e6)
        t, o = go(NEW, "newn", threads, no_front)
        res["new_aN_s"] = r4(t)
        drop(o)
        t_front, o = go(NEW, "front", 1)
        res["front_s"] = r4(t_front)
        drop(o)
        time.sleep(1.0)  # (the worker behind the front process is still tearing down)
        t_ref1 = t_refn = None
        if os.path.exists(REF):
            t_ref1, o_ref1 = go(REF, "ref1", 1)
            res["ref_a1_s"] = r4(t_ref1)
            res["identical"] = all(same_files(a, b) for a, b in zip(o_new, o_ref1))
            drop(o_ref1)
            t_refn, o = go(REF, "refn", threads)
            res["ref_aN_s"], res["threads"] = r4(t_refn), threads
            drop(o)
            res["x_a1"], res["x_aN"] = r4(t_ref1 / t_new), r4(t_refn / t_new)
        else:
            res["reference"] = "oracle/_ref/sickle did not travel with the repo: reference not timed"
        drop(o_new)
        if big_pairs > n_pairs:
            big = {"pairs": big_pairs, "reads": 2 * big_pairs}
            t0 = time.perf_counter()
            write_pair(d, big_pairs, start=n_pairs - n_pairs % CHUNK)
            big["generate_s"] = r4(time.perf_counter() - t0)
            drop(go(NEW, "warm", 1, no_front)[1])
            t_big, o = go(NEW, "big", 1, no_front)
            big["new_s"], big["new_Mreads_s"] = r4(t_big), r4(2 * big_pairs / t_big / 1e6)
            big["out_GB"] = r4(sum(os.path.getsize(x) for x in o) / 1e9)
            drop(o)
            t, o = go(NEW, "bigf", 1)
            big["front_s"] = r4(t)
            drop(o)
            if t_ref1:
                scale = big_pairs / float(n_pairs)
                big["ref_a1_s_extrapolated"], big["ref_aN_s_extrapolated"] = r4(t_ref1 * scale), r4(t_refn * scale)
                big["x_a1"], big["x_aN"] = r4(t_ref1 * scale / t_big), r4(t_refn * scale / t_big)
                big["ref_is"] = "per-read time of the %d-pair run x %g (SURVEY 8d: scaled down and extrapolated)" % (n_pairs, scale)
            res["big"] = big
    return res


def write_mixed_pair(d, n_pairs, device=None):
    """BASELINE configs[4]'s input: two FASTQ files of n_pairs reads, lengths U{75..301} (the same for both mates),
    phred+64, 0.3 % N, 5 % of the reads with one lowercase n (tools/workloads.py's model; mate 2 from another seed),
    then gzip -1 (two processes).  The records are laid out with torch on `device` (the GPU when there is one)."""
    import torch
    import workloads as wl
    if device is None:
        device = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
    paths = [os.path.join(d, "M1.fastq"), os.path.join(d, "M2.fastq")]
    files = [open(p, "wb") for p in paths]
    H = 17  # "@SYN:%09d/1\n"
    for blk, a, b, _ in wl._blocks(0, n_pairs):
        lens, q1, s1 = wl.mixed_block(torch, device, 9001, blk)
        _, q2, s2 = wl.mixed_block(torch, device, 9002, blk)
        lens = lens[a:b]
        m = b - a
        rec = H + 2 * lens + 4
        off = torch.cumsum(rec, 0) - rec
        total = int(rec.sum().item())
        ids = torch.arange(blk * wl.BLOCK + a, blk * wl.BLOCK + b, device=device)
        col = torch.arange(wl.MIX_HI, device=device)[None, :]
        inside = col < lens[:, None]
        for mate, (q, s) in enumerate(((q1, s1), (q2, s2))):
            out = torch.empty((total,), dtype=torch.uint8, device=device)
            head = torch.empty((m, H), dtype=torch.uint8, device=device)
            head[:, :5] = torch.tensor(list(b"@SYN:"), dtype=torch.uint8, device=device)
            v = ids.clone()
            for k in range(9):
                head[:, 13 - k] = (v % 10 + 48).to(torch.uint8)
                v = v // 10
            head[:, 14] = ord("/")
            head[:, 15] = ord("1") + mate
            head[:, 16] = 10
            out[(off[:, None] + torch.arange(H, device=device)[None, :]).reshape(-1)] = head.reshape(-1)
            out[(off[:, None] + H + col)[inside]] = s[a:b][inside]
            mid = off + H + lens
            out[mid] = 10
            out[mid + 1] = ord("+")
            out[mid + 2] = 10
            out[(mid[:, None] + 3 + col)[inside]] = q[a:b][inside]
            out[mid + 3 + lens] = 10
            files[mate].write(out.cpu().numpy().tobytes())
    for f in files:
        f.close()
    procs = [subprocess.Popen(["gzip", "-1", p]) for p in paths]
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("gzip failed")
    return [p + ".gz" for p in paths]


def mixed_against_reference(n_pairs, threads=None):
    """BASELINE configs[4] through both CLIs: `sickle pe -t illumina -n` on gzip input of mixed 75-301 bp pairs
    (segmented kernel with the sequence tile, parallel gzip decoder, N rule, pair classes), outputs compared byte
    for byte.  Times are whole single processes (SICKLE_NO_FRONT=1), see pe_against_reference."""
    res = {"pairs": n_pairs, "cmd": "sickle pe -f M1.gz -r M2.gz -t illumina -n -o O1 -p O2 -s OS -a T"}
    root = pick_tmp(n_pairs * 420 * 2 * 4 + (1 << 30))
    if root is None:
        res["error"] = "no temporary directory"
        return res
    if threads is None:
        threads = min(len(os.sched_getaffinity(0)), 16)
    no_front = dict(os.environ, SICKLE_NO_FRONT="1")
    with tempfile.TemporaryDirectory(dir=root) as d:
        t0 = time.perf_counter()
        p1, p2 = write_mixed_pair(d, n_pairs)
        res["generate_s"], res["gz_MB"] = r4(time.perf_counter() - t0), r4((os.path.getsize(p1) + os.path.getsize(p2)) / 1e6)

        def go(binary, tag, a, env=None):
            outs = [os.path.join(d, "%s_%s.fastq" % (tag, k)) for k in ("o1", "o2", "os")]
            dt, text = timed_run([binary, "pe", "-f", p1, "-r", p2, "-t", "illumina", "-n", "-o", outs[0], "-p", outs[1], "-s", outs[2],
                                  "-a", str(a)], env)
            return dt, outs, text

        for o in go(NEW, "warm", 1, no_front)[1]:
            os.unlink(o)
        t_new, o_new, text = go(NEW, "new", 1, no_front)
        res["new_s"], res["new_Mreads_s"] = r4(t_new), r4(2 * n_pairs / t_new / 1e6)
        kept = [l for l in text.split("\n") if l.startswith("FastQ paired records kept")]
        res["kept_pairs_line"] = kept[0][len("FastQ paired records kept: "):] if kept else ""
        if os.path.exists(REF):
            t_ref1, o_ref1, _ = go(REF, "ref1", 1)
            res["ref_a1_s"] = r4(t_ref1)
            res["identical"] = all(same_files(a, b) for a, b in zip(o_new, o_ref1))
            for o in o_ref1:
                os.unlink(o)
            t_refn, o, _ = go(REF, "refn", threads)
            res["ref_aN_s"], res["threads"] = r4(t_refn), threads
            res["x_a1"], res["x_aN"] = r4(t_ref1 / t_new), r4(t_refn / t_new)
    return res


def to_bgzf(d, path):
    """A BGZF copy of a FASTQ file, made by this CLI itself: -q 0 -l 0 keeps every read whole."""
    out = path + ".bgzf.gz"
    pr = subprocess.run([NEW, "se", "-f", path, "-t", "sanger", "-o", out, "-q", "0", "-l", "0", "-g", "-a", "1"], capture_output=True)
    assert pr.returncode == 0, pr.stderr.decode()[-500:]
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "_part":  # worker of write_pair()
        write_part(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
        return
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
