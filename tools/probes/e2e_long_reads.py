#!/usr/bin/env python3
"""End to end on LONG reads: an interleaved paired file of reads of 1 ... 30 kb (about 2 GB of text) through this CLI and
through the compiled reference (`pe -c`, -a 1 and -a N), wall clock, outputs compared as sets of per-batch chunks are
not needed here: -a 1 of both, byte for byte (few batches).  usage: e2e_long_reads.py [pairs]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

NEW = os.path.join(ROOT, "sickle_amd", "sickle")
REF = os.path.join(ROOT, "oracle", "_ref", "sickle")
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 30_000
rng = np.random.default_rng(5)
with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
    path = os.path.join(d, "long.fastq")
    t0 = time.time()
    with open(path, "wb") as f:
        for i in range(pairs):
            for mate in (1, 2):
                L = int(rng.integers(1000, 30_001))
                q = rng.integers(60, 74, size=L).astype(np.uint8)
                q[L - int(rng.integers(0, L // 3)):] -= 25  # a worse 3' end
                s = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L)
                f.write(b"@long%d/%d\n" % (i, mate) + s.tobytes() + b"\n+\n" + q.tobytes() + b"\n")
    size = os.path.getsize(path)
    print("input: %d pairs, %.2f GB, written in %.0f s" % (pairs, size / 1e9, time.time() - t0), flush=True)

    def run(binary, tag, threads):
        outs = [os.path.join(d, tag + "_om"), os.path.join(d, tag + "_os")]
        t = time.perf_counter()
        pr = subprocess.run([binary, "pe", "-c", path, "-m", outs[0], "-s", outs[1], "-t", "sanger", "-a", str(threads)], capture_output=True,
                            env=dict(os.environ, SICKLE_STAGE_TIMES="1") if tag == "new" else None)
        dt = time.perf_counter() - t
        assert pr.returncode == 0, pr.stderr[-300:]
        if tag == "new":
            print("   " + " | ".join(l.strip() for l in pr.stderr.decode().splitlines() if l.startswith("[stage]") or "device open" in l or "closed" in l)[:900])
        return dt, outs

    run(NEW, "warm", 1)
    t_new, o_new = run(NEW, "new", 1)
    print("this CLI, -a 1: %.2f s (%.2f GB/s of text, %.0f k reads/s)" % (t_new, size / t_new / 1e9, 2 * pairs / t_new / 1e3), flush=True)
    if os.path.exists(REF):
        t_ref, o_ref = run(REF, "ref", 1)
        same = all(open(a, "rb").read() == open(b, "rb").read() for a, b in zip(o_new, o_ref))
        print("reference, -a 1: %.2f s -> %.1fx; outputs byte-identical: %s" % (t_ref, t_ref / t_new, same), flush=True)
        n = os.cpu_count() or 16
        t_refn, _ = run(REF, "refn", min(n, 16))
        print("reference, -a %d: %.2f s -> %.1fx" % (min(n, 16), t_refn, t_refn / t_new), flush=True)
