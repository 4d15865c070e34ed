#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (uses the oracle / the compiled reference, like everything under tests/).  Soak: random paired FASTQ files through the product CLI and through the compiled reference CLI (oracle/_ref/sickle,
`pe -a 1`).  Two files and interleaved; equal, mixed and long read lengths (uniform / segmented / ragged batches
behind the CLI); every encoding; -q, -l, -x, -n.  The expectation is DERIVED (tests/fastq_util.py: the restated
batch-cut rule + the oracle's cuts -> the output chunk of every ingest batch): this CLI must write the chunks in
batch order, byte for byte; the reference -- whose per-batch output threads race each other for the files, and
whose main thread does not wait for the last of them -- must write a permutation of exactly those chunks in at
least one of six runs (which pins the derivation to the reference), and its summary must equal ours.
usage: soak_cli.py [iterations] [seed]"""
import gzip, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cli_util as cu
import oracle_bind as ob
from fastq_util import (expected_pe_outputs, expected_se_output, file_lines, is_permutation_of_chunks, pack_records, parse_fastq, reference_batch_len,
                        reference_batches)

NEW = os.path.join(ROOT, "sickle_amd", "sickle")
REF = os.path.join(ROOT, "oracle", "_ref", "sickle")


def records(rng, n, lens, lo, hi, mid, tag, mate):
    out = []
    for i in range(n):
        L = int(lens[i])
        q = np.clip(rng.normal(mid, 7, L).astype(int), lo, hi)
        mode = int(rng.integers(0, 5))
        if mode == 0 and L > 3:
            q[int(rng.integers(0, L)):] = lo + int(rng.integers(0, 6))
        elif mode == 1 and L > 3:
            q[:int(rng.integers(0, L))] = lo + int(rng.integers(0, 6))
        s = rng.choice(np.frombuffer(b"ACGT" * 60 + b"Nn", dtype=np.uint8), size=L)
        plus = b"+" if i % 3 else b"+" + tag + b"%d" % i
        out.append(b"@" + tag + b"%d/%d\n" % (i, mate) + s.tobytes() + b"\n" + plus + b"\n" + q.astype(np.uint8).tobytes() + b"\n")
    return out


def run(iters=20, seed=1, verbose=True):
    rng = np.random.default_rng(seed)
    t0 = time.time()
    races = 0
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
        for it in range(iters):
            qt = ["sanger", "solexa", "illumina"][it % 3]
            lo, hi = {"sanger": (33, 126), "solexa": (59, 112), "illumina": (64, 110)}[qt]
            thr = int(rng.choice([2, 15, 20, 25, 30]))
            mid = min(hi - 3, max(lo + 3, lo + thr + int(rng.integers(-3, 12))))
            n = int(rng.choice([1, 2, 63, 64, 65, 500, 3000]))
            shape = it % 4
            inter = bool(it % 2)
            if shape == 0:
                L = int(rng.choice([20, 36, 75, 100, 150, 151, 250, 301, 600, 1000, 1500, 2600]))  # (the last four: one length beyond the 64-read tiles)
                if L >= 600:
                    n = min(n, 500)
                l1 = l2 = np.full(n, L)
            elif shape == 1:
                l1, l2 = rng.integers(1, 400, size=n), rng.integers(1, 400, size=n)
            elif shape == 2:
                l1 = np.exp(rng.uniform(0, np.log(20_000), size=n)).astype(int) + 1
                l2 = np.exp(rng.uniform(0, np.log(20_000), size=n)).astype(int) + 1
                n = min(n, 200)
                l1, l2 = l1[:n], l2[:n]
            else:
                l1 = l2 = np.full(n, int(rng.integers(1, 12)))  # reads shorter than any window
            if not inter:
                l2 = l1  # two files must cut their batches at the same lines (src/trim_paired.cpp:335-338 stops otherwise)
            r1 = records(rng, n, l1, lo, hi, mid, b"r", 1)
            r2 = records(rng, n, l2, lo, hi, mid, b"r", 2)
            flags = ["-t", qt, "-q", str(thr), "-l", str(int(rng.choice([0, 20, 50])))]
            T = int(rng.choice([1, 1, 1, 2, 3, 7, 16]))  # -a T: the queue-major order inside every batch
            if rng.integers(0, 2):
                flags.append("-x")
            if rng.integers(0, 2):
                flags.append("-n")
            gz = it % 8 >= 6  # gzip input (one member, or several): the batch budget follows the size of the FILE
            ext = ".fastq.gz" if gz else ".fastq"

            def put(name, data):
                for old in (name + ".fastq", name + ".fastq.gz"):
                    if os.path.exists(os.path.join(d, old)):
                        os.remove(os.path.join(d, old))
                if gz:
                    level = int(rng.integers(1, 10))
                    cutp = int(rng.integers(0, len(data) + 1)) if it % 16 >= 14 else len(data)
                    data = gzip.compress(data[:cutp], level, mtime=0) + (gzip.compress(data[cutp:], level, mtime=0) if cutp < len(data) else b"")
                open(os.path.join(d, name + ext), "wb").write(data)
            if inter:
                put("c", b"".join(a + b for a, b in zip(r1, r2)))
                outs = ["om", "os"]
                argv = lambda pre: ["pe", "-c", os.path.join(d, "c" + ext), "-m", os.path.join(d, pre + "om"), "-s", os.path.join(d, pre + "os"), "-a", str(T)] + flags  # noqa: E731
            else:
                put("f", b"".join(r1))
                put("r", b"".join(r2))
                outs = ["o1", "o2", "os"]
                argv = lambda pre: ["pe", "-f", os.path.join(d, "f" + ext), "-r", os.path.join(d, "r" + ext), "-o", os.path.join(d, pre + "o1"),  # noqa: E731
                                    "-p", os.path.join(d, pre + "o2"), "-s", os.path.join(d, pre + "os"), "-a", str(T)] + flags
            # the expectation: per ingest batch, what `pe -a 1` writes for it
            paths = [os.path.join(d, "c" + ext)] if inter else [os.path.join(d, "f" + ext), os.path.join(d, "r" + ext)]
            datas = [gzip.decompress(open(q, "rb").read()) if gz else open(q, "rb").read() for q in paths]
            po = ob.make_params(qt, thr, int(flags[5]), "-x" in flags, "-n" in flags)
            cuts = []
            for dat in datas:
                sq, ql, of = pack_records(parse_fastq(dat))
                c, e = ob.oracle_trim_batch(po, ql, sq, offsets=of)
                assert e is None
                cuts.append(c)
            blen = reference_batch_len(os.path.getsize(paths[0]), 512, paired=True)
            b1 = reference_batches(file_lines(datas[0]), blen, 8 if inter else 4)
            b2 = None if inter else reference_batches(file_lines(datas[1]), blen, 4)
            chunks = expected_pe_outputs(b1, b2, lambda f, r: cuts[f][r], T, interleaved=inter)
            idx = {"om": 0, "o1": 0, "o2": 1, "os": 2}
            pn = subprocess.run([NEW] + argv("new_"), capture_output=True, timeout=120)
            assert pn.returncode == 0, (it, argv(""), pn.returncode, pn.stderr[-300:])
            for o in outs:
                got = open(os.path.join(d, "new_" + o), "rb").read() if os.path.exists(os.path.join(d, "new_" + o)) else b""
                assert got == b"".join(c[idx[o]] for c in chunks), (it, o, argv(""), len(got))
            same = False
            for attempt in range(6):
                for o in outs:
                    if os.path.exists(os.path.join(d, "ref_" + o)):
                        os.remove(os.path.join(d, "ref_" + o))
                pr = subprocess.run([REF] + argv("ref_"), capture_output=True, timeout=120)
                assert pr.returncode == 0, (it, argv(""), pr.returncode, pr.stderr[-300:])
                same = True
                for o in outs:
                    b = open(os.path.join(d, "ref_" + o), "rb").read() if os.path.exists(os.path.join(d, "ref_" + o)) else b""
                    same = same and is_permutation_of_chunks(b, [c[idx[o]] for c in chunks])
                if same:
                    break
                races += 1
            assert same, (it, argv(""), "six reference runs, none a permutation of the derived chunks")
            # The summaries: equal line by line, except "Total input FastQ records", which in the reference is the size
            # of whichever batch its racing output threads finished LAST (src/trim_paired.cpp:593) -- here the last
            # batch in order.  Both must name the size of one of the batches.
            sn = cu.summary_block(pn.stdout.decode("latin-1")).replace("new_", "").split("\n")
            sr = cu.summary_block(pr.stdout.decode("latin-1")).replace("ref_", "").split("\n")
            # (chatter of the racing output threads can lose its [DEBUGGING] tag to interleaving: the summary starts
            # at its first own line)
            first = next((i for i, x in enumerate(sr) if x.startswith(("PE ", "SE ", "Total input"))), 0)
            sr = sr[first:]
            sizes = [len(b) // 4 * (1 if inter else 2) for b in b1] or [0]
            tn = [x for x in sn if x.startswith("Total input")]
            tr = [x for x in sr if x.startswith("Total input")]
            assert [x for x in sn if x not in tn] == [x for x in sr if x not in tr], (it, argv(""), sn, sr)
            assert tn == ["Total input FastQ records: %d (%d pairs)" % (sizes[-1], sizes[-1] // 2)], (it, tn, sizes)
            assert len(tr) == 1 and any(tr[0] == "Total input FastQ records: %d (%d pairs)" % (z, z // 2) for z in sizes), (it, tr, sizes)
            if it % 5 == 0:  # -g: the outputs as gzip (BGZF blocks; deflated on the GPU on every other turn), same content
                env = dict(os.environ, SICKLE_GZ_LEVEL="gpu") if it % 10 == 0 else None
                pg = subprocess.run([NEW] + argv("gz_") + ["-g"], capture_output=True, timeout=120, env=env)
                assert pg.returncode == 0, (it, pg.stderr[-300:])
                for o in outs:
                    z = open(os.path.join(d, "gz_" + o), "rb").read()
                    assert gzip.decompress(z) == b"".join(c[idx[o]] for c in chunks), (it, o, "-g")
            if not inter:  # the forward file alone through `sickle se` (the reference's SE driver crashes: derived expectation only)
                bs = reference_batches(file_lines(datas[0]), reference_batch_len(os.path.getsize(paths[0]), 512, paired=False), 4)
                want_se = b"".join(expected_se_output(bs, lambda f, r: cuts[0][r], T))
                ps = subprocess.run([NEW, "se", "-f", paths[0], "-o", os.path.join(d, "new_se"), "-a", str(T)] + flags, capture_output=True, timeout=120)
                assert ps.returncode == 0, (it, ps.stderr[-300:])
                assert open(os.path.join(d, "new_se"), "rb").read() == want_se, (it, "se", T, flags)
            if verbose and it % 10 == 9:
                print("iteration %d, %.0f s, short reference runs %d" % (it + 1, time.time() - t0, races), flush=True)
    if verbose:
        print("soak ok: %d iterations, seed %d, short reference runs met: %d" % (iters, seed, races))
    return iters


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
