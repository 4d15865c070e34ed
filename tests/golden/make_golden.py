#!/usr/bin/env python3
"""Generates the golden fixtures in this directory from the REFERENCE ITSELF, compiled in
the build container by oracle/Makefile into oracle/_ref/ (libsickle_ref.so = the reference's
own Abstract_Trimmer::sliding_window behind a C ABI; sickle = the reference CLI).

Run from the repo root in the container that has /root/reference:
    make -C oracle && python tests/golden/make_golden.py

Outputs (all data, no reference source):
    inputs/*.fastq      the reference's own test inputs (reference test/*.fastq, data files)
    cuts_bundled.npz    (five,three) per read of inputs/test.fastq over a flag grid
    edge_inputs.npz     seeded edge-case reads (lengths 1..31, 99-101, 149-151, 250, 301)
    cuts_edge.npz       their cuts over a flag grid
    errors.json         range-error cases: exit status + the reference's stderr text
    e2e.json            `sickle pe -a 1` runs: argv, exit status, md5/size of each output, stdout
"""
import gzip
import hashlib
import itertools
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import oracle_bind as ob  # noqa: E402
from fastq_util import parse_fastq, pack_records  # noqa: E402
from sickle_amd import synth  # noqa: E402

REF_TEST = "/root/reference/test"
INPUTS = os.path.join(HERE, "inputs")


def key(qt, q, l, x, n):
    return "%s_q%d_l%d_x%d_n%d" % (qt, q, l, int(x), int(n))


def grid(qts, qs, ls):
    return [(qt, q, l, x, n) for qt in qts for q in qs for l in ls for x in (0, 1) for n in (0, 1)]


def make_edge_inputs():
    """Seeded reads that exercise window-size edges and all three encodings.  Stored, not regenerated."""
    rng = np.random.default_rng(20250104)
    lengths = list(range(1, 32)) + [99, 100, 101, 149, 150, 151, 250, 301]
    seqs, quals = [], []
    for length in lengths:
        for pattern in range(40):
            lo, hi = 64, 105  # valid in illumina AND solexa AND sanger ranges (sanger: q = c-33 is then 31..72)
            kind = pattern % 8
            if kind == 0:
                q = rng.integers(lo, hi + 1, size=length)
            elif kind == 1:  # good then bad
                cut = rng.integers(0, length + 1)
                q = np.where(np.arange(length) < cut, rng.integers(90, hi + 1, size=length),
                             rng.integers(lo, 75, size=length))
            elif kind == 2:  # bad head, good middle, bad tail
                a, b = sorted(rng.integers(0, length + 1, size=2))
                q = rng.integers(lo, 72, size=length)
                q[a:b] = rng.integers(88, hi + 1, size=b - a)
            elif kind == 3:  # hovering around the q=20 threshold (char 84)
                q = 84 + rng.integers(-2, 3, size=length)
            elif kind == 4:  # all low
                q = rng.integers(lo, 70, size=length)
            elif kind == 5:  # all high
                q = rng.integers(95, hi + 1, size=length)
            elif kind == 6:  # alternating extremes
                q = np.where(np.arange(length) % 2 == 0, lo, hi)
            else:  # slow decay
                q = np.clip(hi - (np.arange(length) * rng.uniform(0, 60.0 / max(length, 1))).astype(int)
                            + rng.integers(-3, 4, size=length), lo, hi)
            s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=length)].copy()
            r = pattern % 5
            if r == 1 and length > 2:
                s[rng.integers(0, length)] = ord("N")
            elif r == 2 and length > 2:
                s[rng.integers(0, length)] = ord("n")
            elif r == 3 and length > 4:  # both: lowercase wins regardless of order
                i, j = rng.choice(length, size=2, replace=False)
                s[i] = ord("N")
                s[j] = ord("n")
            elif r == 4:
                s[0] = ord("n") if pattern % 2 else ord("N")
            seqs.append(s)
            quals.append(q.astype(np.uint8))
    lens = np.array([len(s) for s in seqs], dtype=np.uint64)
    offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    return np.concatenate(seqs), np.concatenate(quals), offsets


def error_cases():
    """(description, params, name, seq, qual) -- each run through the reference in a forked child."""
    rng = np.random.default_rng(7)
    cases = []
    good = lambda n, c: bytes([c] * n)  # noqa: E731
    seq150 = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=150)])
    for qt, lo, hi, hiq in (("sanger", 33, 126, 73), ("solexa", 58, 112, 104), ("illumina", 64, 110, 104)):
        p = dict(qualtype=qt, q=20, l=20, no5=False, trunc_n=False)
        base = bytearray(good(150, hiq))
        for pos, ch, what in ((0, lo - 1, "below min at 0"), (14, hi + 1, "above max inside first window"),
                              (15, lo - 1, "first char after the first window"),
                              (149, hi + 1, "last char, scan reaches it"), (77, 200, "byte >= 0x80 (negative char)"),
                              (60, lo, "exactly min: legal"), (60, hi, "exactly max: legal")):
            q = bytearray(base)
            q[pos] = ch
            cases.append(("%s: %s" % (qt, what), p, b"@ERR:%s:%d" % (qt.encode(), pos), seq150, bytes(q)))
        # quality collapses at 50: the 3' break fires at window 43 (8 of 15 chars low), so chars >= 43+15 = 58 are never read
        lowc = lo if qt != "solexa" else 64
        q = bytearray(good(50, hiq) + good(100, lowc))
        for pos in (51, 56, 57, 58, 59, 80, 149):
            qq = bytearray(q)
            qq[pos] = lo - 1
            cases.append(("%s: bad char at %d after a 3' break" % (qt, pos), p, b"@BRK:%d" % pos, seq150, bytes(qq)))
        # too short: discarded before any quality is looked at
        cases.append(("%s: too-short read with a bad char" % qt, p, b"@SHORT", seq150[:10], bytes([lo - 1] * 10)))
        # -x: no 5' search, break may fire at window 0
        px = dict(p, no5=True)
        qq = bytearray(good(150, lowc))
        qq[15] = lo - 1
        cases.append(("%s: -x, bad char just outside window 0" % qt, px, b"@X15", seq150, bytes(qq)))
        qq = bytearray(good(150, lowc))
        qq[14] = lo - 1
        cases.append(("%s: -x, bad char inside window 0" % qt, px, b"@X14", seq150, bytes(qq)))
        # two bad chars: the first one is reported
        qq = bytearray(base)
        qq[100] = lo - 1
        qq[30] = hi + 1
        cases.append(("%s: two bad chars" % qt, p, b"@TWO", seq150, bytes(qq)))
        # short read (window = whole read)
        qq = bytearray(good(9, hiq))
        qq[8] = lo - 1
        cases.append(("%s: L=9 (window = read), bad last char" % qt, dict(p, l=0), b"@L9", seq150[:9], bytes(qq)))
    return cases


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def run_ref_pe(tmp, argv, outputs):
    """Run the reference CLI; argv uses {tmp} / {inputs} placeholders."""
    real = [a.format(tmp=tmp, inputs=INPUTS) for a in argv]
    pr = subprocess.run([ob.REF_BIN] + real, capture_output=True, timeout=600)
    rec = {"argv": argv, "rc": pr.returncode, "stdout": pr.stdout.decode("latin-1"),
           "stderr": pr.stderr.decode("latin-1"), "outputs": {}}
    for o in outputs:
        path = os.path.join(tmp, o)
        if os.path.exists(path):
            rec["outputs"][o] = {"md5": md5(path), "size": os.path.getsize(path)}
    return rec


def write_synth_inputs(tmp):
    """Synthetic FASTQ for the e2e runs; regenerated identically by tests/ (md5 recorded)."""
    made = {}
    s1, q1 = synth.make_reads(101, 3000, 150, "sanger")
    s2, q2 = synth.make_reads(202, 3000, 150, "sanger")
    open(os.path.join(tmp, "syn_R1.fastq"), "wb").write(synth.fastq_bytes(s1, q1, suffix="/1"))
    open(os.path.join(tmp, "syn_R2.fastq"), "wb").write(synth.fastq_bytes(s2, q2, suffix="/2"))
    # mixed lengths, illumina, with N / n, interleaved (same length for both mates is not needed with -c)
    sa, qa, oa = synth.make_ragged_reads(303, 2000, 75, 301, "illumina")
    recs = parse_fastq(synth.fastq_bytes_ragged(sa, qa, oa))
    inter = b"".join(b"\n".join(r) + b"\n" for r in recs)
    open(os.path.join(tmp, "syn_mixed_inter.fastq"), "wb").write(inter)
    with gzip.GzipFile(os.path.join(tmp, "syn_mixed_inter.fastq.gz"), "wb", mtime=0) as f:
        f.write(inter)
    for name in ("syn_R1.fastq", "syn_R2.fastq", "syn_mixed_inter.fastq"):
        made[name] = md5(os.path.join(tmp, name))
    return made


def write_long_inputs(tmp):
    """Long reads (1 .. 40 kb, the mates of a pair of different lengths, a few short ones between them),
    interleaved; regenerated identically by tests/cli_util.py (md5 recorded)."""
    sa, qa, oa = synth.make_long_reads(909, 240, 1, 40_000)
    sb, qb, obb = synth.make_ragged_reads(910, 60, 20, 400, "sanger")
    recs = parse_fastq(synth.fastq_bytes_ragged(sa, qa, oa, prefix="LONG:"))
    short = parse_fastq(synth.fastq_bytes_ragged(sb, qb, obb, prefix="SHORT:"))
    for i, r in enumerate(short):  # a short read after every fourth long one
        recs.insert(5 * i + 4, r)
    assert len(recs) % 2 == 0
    data = b"".join(b"\n".join(r) + b"\n" for r in recs)
    open(os.path.join(tmp, "syn_long_inter.fastq"), "wb").write(data)
    return {"syn_long_inter.fastq": md5(os.path.join(tmp, "syn_long_inter.fastq"))}


def long_reads_goldens(tmp):
    """`sickle pe -c` of the reference on the long-read file: the general (streaming) kernel behind the CLI."""
    inter = lambda extra: (["pe", "-c", "{tmp}/syn_long_inter.fastq", "-m", "{tmp}/om.fastq", "-a", "1", "-s", "{tmp}/os.fastq"] + extra,  # noqa: E731
                           ["om.fastq", "os.fastq"])
    runs = {
        "pe_long_inter_sanger": inter(["-t", "sanger"]),
        "pe_long_inter_sanger_n_q25": inter(["-t", "sanger", "-n", "-q", "25"]),
        "pe_long_inter_sanger_x_l2000": inter(["-t", "sanger", "-x", "-l", "2000"]),
    }
    out = {}
    for name, (argv, outputs) in runs.items():
        for o in outputs:
            if os.path.exists(os.path.join(tmp, o)):
                os.remove(os.path.join(tmp, o))
        rec = run_ref_pe(tmp, argv, outputs)
        assert rec["rc"] == 0, (name, rec["stderr"][-300:])
        out[name] = rec
        print(name, {k: v["size"] for k, v in rec["outputs"].items()})
    return out


THREAD_ORDER_RUNS = {
    # name: (kind, inputs, qualtype, extra flags, threads)
    "pe_fr_illumina_a4": ("two", ("{inputs}/test.f.fastq", "{inputs}/test.r.fastq"), "illumina", [], 4),
    "pe_fr_illumina_n_a3": ("two", ("{inputs}/test.f.fastq", "{inputs}/test.r.fastq"), "illumina", ["-n"], 3),
    "pe_syn_fr_sanger_a4": ("two", ("{tmp}/syn_R1.fastq", "{tmp}/syn_R2.fastq"), "sanger", [], 4),
    "pe_syn_fr_sanger_a16": ("two", ("{tmp}/syn_R1.fastq", "{tmp}/syn_R2.fastq"), "sanger", [], 16),
    "pe_inter_illumina_a5": ("inter", ("{inputs}/test.fastq",), "illumina", [], 5),
    "pe_syn_mixed_inter_illumina_n_a7": ("inter", ("{tmp}/syn_mixed_inter.fastq",), "illumina", ["-n"], 7),
}


def thread_order_goldens(tmp):
    """`sickle pe -a T`, T > 1.  The reference's per-batch output threads race each other for the files
    (src/trim_paired.cpp:445-458), so its BATCHES land in any order -- but inside a batch the order is fixed:
    pair k in queue k mod T, queues written in turn.  For each run: the expected per-batch chunks are derived
    (fastq_util: the restated batch-cut rule + queue order, cuts from the compiled reference's sliding_window),
    every reference run's three files must be a permutation of exactly those chunks, and what is recorded is
    the md5 of the chunks in batch order -- which is what this repo's CLI writes."""
    from fastq_util import (expected_pe_outputs, file_lines, is_permutation_of_chunks, reference_batch_len,
                            reference_batches)
    out = {}
    for name, (kind, inputs, qt, extra, threads) in THREAD_ORDER_RUNS.items():
        paths = [a.format(tmp=tmp, inputs=INPUTS) for a in inputs]
        datas = [open(p, "rb").read() for p in paths]
        blen = reference_batch_len(len(datas[0]), 512, paired=True)
        inter = kind == "inter"
        b1 = reference_batches(file_lines(datas[0]), blen, 8 if inter else 4)
        b2 = None if inter else reference_batches(file_lines(datas[1]), blen, 4)
        p = ob.make_params(qt, 20, 20, False, "-n" in extra)
        cuts = []
        for d in datas:
            recs = parse_fastq(d)
            seq, qual, offsets = pack_records(recs)
            cuts.append(ob.ref_trim_batch(p, qual, seq, offsets=offsets))
        chunks = expected_pe_outputs(b1, b2, lambda f, r: cuts[f][r], threads, interleaved=inter)
        assert len(chunks) > 3, (name, "needs several batches to mean anything")
        if inter:
            argv = ["pe", "-c", inputs[0], "-m", "{tmp}/om.fastq", "-s", "{tmp}/os.fastq", "-t", qt, "-a", str(threads)] + extra
            files = {"om.fastq": 0, "os.fastq": 2}
        else:
            argv = ["pe", "-f", inputs[0], "-r", inputs[1], "-o", "{tmp}/o1.fastq", "-p", "{tmp}/o2.fastq", "-s", "{tmp}/os.fastq",
                    "-t", qt, "-a", str(threads)] + extra
            files = {"o1.fastq": 0, "o2.fastq": 1, "os.fastq": 2}
        orders = set()
        for attempt in range(4):
            for o in files:
                if os.path.exists(os.path.join(tmp, o)):
                    os.remove(os.path.join(tmp, o))
            pr = subprocess.run([ob.REF_BIN] + [a.format(tmp=tmp, inputs=INPUTS) for a in argv], capture_output=True, timeout=600)
            assert pr.returncode == 0, pr.stderr[-300:]
            for o, idx in files.items():
                data = open(os.path.join(tmp, o), "rb").read()
                assert is_permutation_of_chunks(data, [c[idx] for c in chunks]), \
                    (name, o, "the reference's output is not a permutation of the derived per-batch chunks")
            orders.add(md5(os.path.join(tmp, list(files)[0])))
        rec = {"argv": argv, "rc": 0, "batches": len(chunks), "threads": threads, "outputs": {},
               "distinct_reference_orders_seen": len(orders)}
        for o, idx in files.items():
            whole = b"".join(c[idx] for c in chunks)
            rec["outputs"][o] = {"md5": hashlib.md5(whole).hexdigest(), "size": len(whole)}
        out[name] = rec
        print(name, "batches", len(chunks), "reference orders seen", len(orders), {k: v["size"] for k, v in rec["outputs"].items()})
    return out


def main():
    if "--only-thread-order" in sys.argv:  # adds / refreshes e2e.json["thread_order"], leaves the rest alone
        assert ob.have_ref()
        e2e = json.load(open(os.path.join(HERE, "e2e.json")))
        with tempfile.TemporaryDirectory() as tmp:
            assert write_synth_inputs(tmp) == e2e["synth_inputs_md5"]
            e2e["thread_order"] = thread_order_goldens(tmp)
        json.dump(e2e, open(os.path.join(HERE, "e2e.json"), "w"), indent=1)
        return
    if "--only-long-reads" in sys.argv:  # adds / refreshes e2e.json["long_reads"], leaves the rest alone
        assert ob.have_ref()
        e2e = json.load(open(os.path.join(HERE, "e2e.json")))
        with tempfile.TemporaryDirectory() as tmp:
            e2e["long_inputs_md5"] = write_long_inputs(tmp)
            e2e["long_reads"] = long_reads_goldens(tmp)
        json.dump(e2e, open(os.path.join(HERE, "e2e.json"), "w"), indent=1)
        return
    assert ob.have_ref(), "build oracle/_ref first (make -C oracle) in the container with /root/reference"
    os.makedirs(INPUTS, exist_ok=True)
    for f in ("test.fastq", "test.f.fastq", "test.r.fastq", "problem1.fastq"):
        shutil.copyfile(os.path.join(REF_TEST, f), os.path.join(INPUTS, f))
        os.chmod(os.path.join(INPUTS, f), 0o644)

    # ---- per-read cuts of the bundled file
    recs = parse_fastq(open(os.path.join(INPUTS, "test.fastq"), "rb").read())
    seq, qual, offsets = pack_records(recs)
    cuts = {}
    for qt, q, l, x, n in grid(("illumina", "solexa"), (0, 20, 35), (0, 20, 100)) + \
            grid(("sanger",), (20, 60), (20,)):
        p = ob.make_params(qt, q, l, x, n)
        cuts[key(qt, q, l, x, n)] = ob.ref_trim_batch(p, qual, seq, offsets=offsets).astype(np.int16)
    np.savez_compressed(os.path.join(HERE, "cuts_bundled.npz"), **cuts)
    print("cuts_bundled:", len(cuts), "grid points x", len(recs), "reads")

    # ---- edge-case reads
    eseq, equal, eoff = make_edge_inputs()
    np.savez_compressed(os.path.join(HERE, "edge_inputs.npz"), seq=eseq, qual=equal, offsets=eoff)
    ecuts = {}
    for qt, q, l, x, n in grid(("illumina", "solexa", "sanger"), (0, 20, 30, 41), (0, 5, 20, 100)):
        p = ob.make_params(qt, q, l, x, n)
        ecuts[key(qt, q, l, x, n)] = ob.ref_trim_batch(p, equal, eseq, offsets=eoff).astype(np.int16)
    np.savez_compressed(os.path.join(HERE, "cuts_edge.npz"), **ecuts)
    print("cuts_edge:", len(ecuts), "grid points x", len(eoff) - 1, "reads")

    # ---- range-error cases (the reference exit(1)s: run each in a forked child)
    errs = []
    for desc, pd, name, s, q in error_cases():
        p = ob.make_params(**pd)
        rc, cut, text = ob.ref_sliding_window_forked(p, name, s, q)
        errs.append({"desc": desc, "params": pd, "name": name.decode("latin-1"), "seq": s.decode("latin-1"),
                     "qual_hex": q.hex(), "rc": rc, "cut": list(cut) if rc == 0 else None,
                     "stderr": text.decode("latin-1")})
    json.dump(errs, open(os.path.join(HERE, "errors.json"), "w"), indent=1)
    print("errors:", len(errs), "cases,", sum(1 for e in errs if e["rc"]), "erroring")

    # ---- end-to-end `sickle pe -a 1`
    with tempfile.TemporaryDirectory() as tmp:
        synth_md5 = write_synth_inputs(tmp)
        shutil.copyfile(os.path.join(INPUTS, "test.fastq"), os.path.join(tmp, "self_copy.fastq"))
        with gzip.GzipFile(os.path.join(tmp, "test.f.fastq.gz"), "wb", mtime=0) as f:
            f.write(open(os.path.join(INPUTS, "test.f.fastq"), "rb").read())
        with gzip.GzipFile(os.path.join(tmp, "test.r.fastq.gz"), "wb", mtime=0) as f:
            f.write(open(os.path.join(INPUTS, "test.r.fastq"), "rb").read())
        two = lambda f, r, extra: (["pe", "-f", f, "-r", r, "-o", "{tmp}/o1.fastq", "-p", "{tmp}/o2.fastq",  # noqa: E731
                                    "-s", "{tmp}/os.fastq", "-a", "1"] + extra,
                                   ["o1.fastq", "o2.fastq", "os.fastq"])
        inter = lambda c, extra, singles=True: (["pe", "-c", c, "-m", "{tmp}/om.fastq", "-a", "1"]  # noqa: E731
                                                + (["-s", "{tmp}/os.fastq"] if singles else []) + extra,
                                                ["om.fastq", "os.fastq"])
        runs = {
            "pe_fr_illumina": two("{inputs}/test.f.fastq", "{inputs}/test.r.fastq", ["-t", "illumina"]),
            "pe_fr_illumina_n": two("{inputs}/test.f.fastq", "{inputs}/test.r.fastq", ["-t", "illumina", "-n"]),
            "pe_fr_illumina_x_q30_l50": two("{inputs}/test.f.fastq", "{inputs}/test.r.fastq",
                                            ["-t", "illumina", "-x", "-q", "30", "-l", "50"]),
            "pe_fr_sanger_q60": two("{inputs}/test.f.fastq", "{inputs}/test.r.fastq", ["-t", "sanger", "-q", "60"]),
            "pe_fr_solexa_q25": two("{inputs}/test.f.fastq", "{inputs}/test.r.fastq", ["-t", "solexa", "-q", "25"]),
            "pe_fr_gz_illumina": two("{tmp}/test.f.fastq.gz", "{tmp}/test.r.fastq.gz", ["-t", "illumina"]),
            "pe_inter_illumina": inter("{inputs}/test.fastq", ["-t", "illumina"]),
            "pe_inter_illumina_nosingles": inter("{inputs}/test.fastq", ["-t", "illumina"], singles=False),
            "pe_inter_illumina_n_l30": inter("{inputs}/test.fastq", ["-t", "illumina", "-n", "-l", "30"]),
            "se_equiv_selfpair_illumina": two("{inputs}/test.fastq", "{tmp}/self_copy.fastq", ["-t", "illumina"]),
            "se_equiv_selfpair_sanger": two("{inputs}/test.fastq", "{tmp}/self_copy.fastq", ["-t", "sanger"]),
            "pe_problem1_inter": inter("{inputs}/problem1.fastq", ["-t", "sanger"]),
            "pe_syn_fr_sanger": two("{tmp}/syn_R1.fastq", "{tmp}/syn_R2.fastq", ["-t", "sanger"]),
            "pe_syn_fr_sanger_n": two("{tmp}/syn_R1.fastq", "{tmp}/syn_R2.fastq", ["-t", "sanger", "-n"]),
            "pe_syn_mixed_inter_illumina_n": inter("{tmp}/syn_mixed_inter.fastq", ["-t", "illumina", "-n"]),
            "pe_syn_mixed_inter_gz_illumina_n": inter("{tmp}/syn_mixed_inter.fastq.gz", ["-t", "illumina", "-n"]),
        }
        e2e = {"synth_inputs_md5": synth_md5, "runs": {}}
        for name, (argv, outs) in runs.items():
            for o in outs:
                if os.path.exists(os.path.join(tmp, o)):
                    os.remove(os.path.join(tmp, o))
            e2e["runs"][name] = run_ref_pe(tmp, argv, outs)
            print(name, "rc", e2e["runs"][name]["rc"], {k: v["size"] for k, v in e2e["runs"][name]["outputs"].items()})
        e2e["thread_order"] = thread_order_goldens(tmp)
        e2e["long_inputs_md5"] = write_long_inputs(tmp)
        e2e["long_reads"] = long_reads_goldens(tmp)
        json.dump(e2e, open(os.path.join(HERE, "e2e.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
