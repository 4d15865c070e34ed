"""Shared by the CLI tests: replays the reference `sickle pe -a 1` runs recorded in
tests/golden/e2e.json against one of this repo's binaries and compares output md5s."""
import gzip
import hashlib
import json
import os
import shutil
import subprocess

from fastq_util import parse_fastq
from sickle_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
INPUTS = os.path.join(GOLD, "inputs")
PRODUCT_BIN = os.path.join(ROOT, "sickle_amd", "sickle")              # HIP: needs the GPU
HOSTCHECK_BIN = os.path.join(ROOT, "tests", "cpu_shim", "sickle_hostcheck")  # oracle-backed, tests only


def build_hostcheck():
    # SICKLE_HOSTCHECK_BIN: a sanitizer build of the same sources, run through the same tests
    if os.environ.get("SICKLE_HOSTCHECK_BIN"):
        return os.environ["SICKLE_HOSTCHECK_BIN"]
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpu_shim")], check=True)
    return HOSTCHECK_BIN


def md5_file(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def e2e():
    return json.load(open(os.path.join(GOLD, "e2e.json")))


def prepare_inputs(tmp):
    """The files make_golden.py wrote next to the reference run (same seeds -> same bytes)."""
    tmp = str(tmp)
    s1, q1 = synth.make_reads(101, 3000, 150, "sanger")
    s2, q2 = synth.make_reads(202, 3000, 150, "sanger")
    open(os.path.join(tmp, "syn_R1.fastq"), "wb").write(synth.fastq_bytes(s1, q1, suffix="/1"))
    open(os.path.join(tmp, "syn_R2.fastq"), "wb").write(synth.fastq_bytes(s2, q2, suffix="/2"))
    sa, qa, oa = synth.make_ragged_reads(303, 2000, 75, 301, "illumina")
    recs = parse_fastq(synth.fastq_bytes_ragged(sa, qa, oa))
    inter = b"".join(b"\n".join(r) + b"\n" for r in recs)
    open(os.path.join(tmp, "syn_mixed_inter.fastq"), "wb").write(inter)
    with gzip.GzipFile(os.path.join(tmp, "syn_mixed_inter.fastq.gz"), "wb", mtime=0) as f:
        f.write(inter)
    shutil.copyfile(os.path.join(INPUTS, "test.fastq"), os.path.join(tmp, "self_copy.fastq"))
    for name in ("test.f.fastq", "test.r.fastq"):
        with gzip.GzipFile(os.path.join(tmp, name + ".gz"), "wb", mtime=0) as f:
            f.write(open(os.path.join(INPUTS, name), "rb").read())
    want = e2e()["synth_inputs_md5"]
    for name, m in want.items():
        assert md5_file(os.path.join(tmp, name)) == m, "synthetic input %s drifted from the golden run" % name


def prepare_long_inputs(tmp):
    """The long-read file of make_golden.py's write_long_inputs (same seeds -> same bytes)."""
    tmp = str(tmp)
    sa, qa, oa = synth.make_long_reads(909, 240, 1, 40_000)
    sb, qb, obb = synth.make_ragged_reads(910, 60, 20, 400, "sanger")
    recs = parse_fastq(synth.fastq_bytes_ragged(sa, qa, oa, prefix="LONG:"))
    short = parse_fastq(synth.fastq_bytes_ragged(sb, qb, obb, prefix="SHORT:"))
    for i, r in enumerate(short):
        recs.insert(5 * i + 4, r)
    open(os.path.join(tmp, "syn_long_inter.fastq"), "wb").write(b"".join(b"\n".join(r) + b"\n" for r in recs))
    for name, m in e2e()["long_inputs_md5"].items():
        assert md5_file(os.path.join(tmp, name)) == m, "synthetic input %s drifted from the golden run" % name


def run_cli(binary, tmp, argv, env=None):
    real = [a.format(tmp=str(tmp), inputs=INPUTS) for a in argv]
    e = dict(os.environ)
    if env:
        e.update(env)
    return subprocess.run([binary] + real, capture_output=True, timeout=600, env=e)


def summary_block(stdout_text):
    """The non-chatter part of stdout: drops the reference's [DEBUGGING] lines and its other
    progress prints, keeps the summary block (with blank lines)."""
    keep = []
    for line in stdout_text.split("\n"):
        if line.startswith("[DEBUGGING]") or line.startswith("Building reader for") or \
                line in ("Setting se trimming params", "trim_main()"):
            continue
        keep.append(line)
    return "\n".join(keep).strip("\n")


def check_run(binary, tmp, name, rec, summary=True, env=None):
    """Replays one golden run; returns nothing, asserts."""
    for o in rec["outputs"]:
        p = os.path.join(str(tmp), o)
        if os.path.exists(p):
            os.remove(p)
    pr = run_cli(binary, tmp, rec["argv"], env=env)
    assert pr.returncode == rec["rc"], (name, pr.returncode, pr.stderr[-500:])
    for o, meta in rec["outputs"].items():
        p = os.path.join(str(tmp), o)
        assert os.path.exists(p), (name, o)
        assert os.path.getsize(p) == meta["size"], (name, o, os.path.getsize(p), meta["size"])
        assert md5_file(p) == meta["md5"], (name, o)
    if not summary:
        return
    want = summary_block(rec["stdout"])
    got = summary_block(pr.stdout.decode("latin-1"))
    # the golden run had its own repo root and temp dir: compare with paths normalised
    import re

    def norm(s):
        s = re.sub(r"[^\s]*/tests/golden/inputs", "{inputs}", s.replace(str(tmp) + "/", "{tmp}/"))
        return re.sub(r"/tmp/tmp[^/\s]+/", "{tmp}/", s)

    assert norm(got) == norm(want), (name, got, want)


EMBED_HOST = os.path.join(ROOT, "tests", "embed", "embed_host")  # oracle-backed shim, tests only
EMBED_GPU = os.path.join(ROOT, "tests", "embed", "embed_gpu")    # the product library


def build_embed(target):
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "embed"), os.path.join(ROOT, "tests", "embed", target)], check=True)
    return os.path.join(ROOT, "tests", "embed", target)


def check_embedded(binary, tmp, gpu):
    """Trim_Paired / Trim_Single as a library (tests/embed/embed_main.cpp, the shape of reference src/sickle.cpp:61-80):
    a PE golden, an SE run, another PE golden (mixed lengths, gzip input, -n) and the first one again, in ONE
    process with everything released between them.  Every run's files must be the reference's; the device memory
    and the resident set must not grow from run to run."""
    import re
    runs = e2e()["runs"]
    plan = [("pe_syn_fr_sanger_n", runs["pe_syn_fr_sanger_n"]["argv"]),
            ("se", ["se", "-f", "{inputs}/test.fastq", "-t", "illumina", "-o", "{tmp}/o1.fastq", "-a", "1"]),
            ("pe_syn_mixed_inter_gz_illumina_n", runs["pe_syn_mixed_inter_gz_illumina_n"]["argv"]),
            ("pe_syn_fr_sanger_n", runs["pe_syn_fr_sanger_n"]["argv"])]
    argv = []
    for k, (_, av) in enumerate(plan):
        if k:
            argv.append("--")
        argv += [a.replace("{tmp}/o", "{tmp}/e%d_o" % k) for a in av]
    pr = run_cli(binary, tmp, argv, env={"SICKLE_NO_FRONT": "1"})
    text = pr.stderr.decode("latin-1")
    assert pr.returncode == 0, text[-800:]
    marks = re.findall(r"\[embed\] run (\d+) rc (-?\d+) device_free_bytes (-?\d+) rss_kb (-?\d+)", text)
    assert [int(m[0]) for m in marks] == [0, 1, 2, 3] and all(int(m[1]) == 0 for m in marks), text[-800:]
    for k, (name, _) in enumerate(plan):
        want = runs["se_equiv_selfpair_illumina"]["outputs"] if name == "se" else runs[name]["outputs"]
        for o, meta in want.items():
            if name == "se" and o != "o1.fastq":
                continue  # file 1 of the self-paired reference run is what `se` writes
            p = os.path.join(str(tmp), "e%d_%s" % (k, o))
            assert os.path.getsize(p) == meta["size"] and md5_file(p) == meta["md5"], (k, name, o)
    free = [int(m[2]) for m in marks]
    rss = [int(m[3]) for m in marks]
    if gpu:
        assert all(f > 0 for f in free), free
        # everything a run allocated on the device is back after it: the runtime's own pools may keep a few MiB
        assert min(free[1:]) >= free[0] - (32 << 20), free
        assert free[3] >= free[0] - (8 << 20), free  # the same run again: no growth
    assert rss[3] <= rss[0] + 96 * 1024, rss  # the same run again: the resident set does not creep (kB)
    return marks
