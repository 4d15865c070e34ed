"""GPU: the PRODUCT binary (sickle_amd/sickle, linked against the HIP library) replays the
reference runs of tests/golden/e2e.json: byte-identical output files, same summary block."""
import os

import pytest

import cli_util as cu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    d = tmp_path_factory.mktemp("cli_gpu")
    cu.prepare_inputs(d)
    return d


def test_product_binary_is_the_hip_build():
    import subprocess
    if not os.path.exists(cu.PRODUCT_BIN):  # a fresh checkout: built artefacts are not in history
        subprocess.run(["make", "-s", "-C", os.path.join(cu.ROOT, "sickle_amd", "csrc"), "all"], check=True)
    assert os.path.exists(cu.PRODUCT_BIN), "build sickle_amd/sickle first (__graft_entry__.build())"
    out = subprocess.run(["ldd", cu.PRODUCT_BIN], capture_output=True).stdout.decode()
    assert "libsickle_amd.so" in out and "libamdhip64" in out


@pytest.mark.parametrize("name", sorted(cu.e2e()["runs"].keys()))
def test_reference_runs_byte_identical_on_gpu(workdir, name):
    cu.check_run(cu.PRODUCT_BIN, workdir, name, cu.e2e()["runs"][name])


@pytest.mark.parametrize("name", sorted(cu.e2e()["runs"].keys()))
def test_reference_runs_in_pieces_on_gpu(workdir, name):
    """An ingest batch cut into pieces of 7 records (3 pairs) for the device and the writers: same bytes."""
    cu.check_run(cu.PRODUCT_BIN, workdir, name, cu.e2e()["runs"][name], env={"SICKLE_SUBBATCH_READS": "7"})


@pytest.mark.parametrize("general", ["default", "band", "team", "stream"])
@pytest.mark.parametrize("name", sorted(cu.e2e()["long_reads"].keys()))
def test_long_reads_byte_identical_on_gpu(workdir, name, general):
    """Reads of 1 .. 40 kb with short ones between them: ragged batches through sk_submit, the general kernels
    (both forced in turn) behind the CLI, against the reference's files."""
    cu.prepare_long_inputs(workdir)
    cu.check_run(cu.PRODUCT_BIN, workdir, name, cu.e2e()["long_reads"][name],
                 env=None if general == "default" else {"SK_GENERAL": general})


@pytest.mark.parametrize("name", ["pe_fr_illumina", "pe_syn_mixed_inter_gz_illumina_n"])
def test_reference_runs_in_one_process_on_gpu(workdir, name):
    """SICKLE_NO_FRONT=1: the run and its teardown in the one process the caller started, same bytes."""
    cu.check_run(cu.PRODUCT_BIN, workdir, name, cu.e2e()["runs"][name], env={"SICKLE_NO_FRONT": "1"})


def test_cli_soak_against_reference_on_gpu():
    """tests/soak_cli.py with the product binary: 40 random paired inputs against the chunks derived from the
    oracle and against the compiled reference (which travels to the box as oracle/_ref)."""
    import oracle_bind as ob
    if not ob.have_ref():
        pytest.skip("needs the compiled reference (oracle/_ref)")
    import soak_cli
    soak_cli.NEW = cu.PRODUCT_BIN
    assert soak_cli.run(40, 505, verbose=False) == 40


def test_se_on_gpu_equals_selfpaired_reference(workdir):
    rec = cu.e2e()["runs"]["se_equiv_selfpair_illumina"]
    out = os.path.join(str(workdir), "se_self.fastq")
    pr = cu.run_cli(cu.PRODUCT_BIN, workdir, ["se", "-f", "{inputs}/test.fastq", "-t", "illumina", "-o", out, "-a", "1"])
    assert pr.returncode == 0, pr.stderr
    assert cu.md5_file(out) == rec["outputs"]["o1.fastq"]["md5"]


def test_range_error_exit_on_gpu(workdir):
    import json
    from sickle_amd import synth
    c = [c for c in json.load(open(os.path.join(cu.GOLD, "errors.json"))) if c["rc"] == 1 and len(c["seq"]) == 150][0]
    seq, qual = synth.make_reads(5, 40, 150, "sanger")
    bad = c["name"].encode("latin-1") + b"\n" + c["seq"].encode("latin-1") + b"\n+\n" + bytes.fromhex(c["qual_hex"]) + b"\n"
    path = os.path.join(str(workdir), "bad.fastq")
    open(path, "wb").write(synth.fastq_bytes(seq[:20], qual[:20]) + bad + synth.fastq_bytes(seq[20:], qual[20:], start=20))
    pr = cu.run_cli(cu.PRODUCT_BIN, workdir, ["se", "-f", path, "-t", c["params"]["qualtype"], "-o", "{tmp}/bad_out.fastq", "-a", "1"])
    assert pr.returncode == 1
    assert pr.stderr.decode("latin-1") == c["stderr"]


def test_multi_device_round_robin_same_output(workdir):
    """SICKLE_DEVICES spreads batches over several GPUs; on this one-GPU box the list names GPU 0
    three times, which exercises the multi-context path: output must not change."""
    rec = cu.e2e()["runs"]["pe_fr_illumina"]
    for o in rec["outputs"]:
        p = os.path.join(str(workdir), o)
        if os.path.exists(p):
            os.remove(p)
    pr = cu.run_cli(cu.PRODUCT_BIN, workdir, rec["argv"], env={"SICKLE_DEVICES": "0,0,0"})
    assert pr.returncode == 0, pr.stderr
    for o, meta in rec["outputs"].items():
        assert cu.md5_file(os.path.join(str(workdir), o)) == meta["md5"], o


def test_gzip_output_deflated_on_the_gpu(workdir):
    """-g with SICKLE_GZ_LEVEL=gpu: the BGZF blocks come from sk_bgzf_deflate; the file inflates to
    exactly the plain output and equals, byte for byte, what the zlib setting's file inflates to."""
    import gzip
    d = str(workdir)
    big = os.path.join(d, "gpu_gz_src.fastq")
    open(big, "wb").write(open(os.path.join(cu.INPUTS, "test.fastq"), "rb").read() * 30)
    outs = {}
    for tag, extra, env in (("plain", [], None), ("gpu", ["-g"], {"SICKLE_GZ_LEVEL": "gpu"}), ("zlib", ["-g"], None)):
        o = os.path.join(d, "gpu_gz_%s.out" % tag)
        pr = cu.run_cli(cu.PRODUCT_BIN, workdir, ["se", "-f", big, "-t", "illumina", "-o", o, "-a", "2"] + extra, env=env)
        assert pr.returncode == 0, pr.stderr
        outs[tag] = open(o, "rb").read()
    assert gzip.decompress(outs["gpu"]) == outs["plain"] == gzip.decompress(outs["zlib"])
    assert len(outs["gpu"]) < 0.6 * len(outs["plain"]) and outs["gpu"][-28:] == outs["zlib"][-28:]  # both end with the BGZF marker block


def test_config0_se_sanger_on_bundled_file(workdir):
    """BASELINE configs[0] verbatim: `sickle se` on the reference's test/test.fastq, Sanger q=20 l=20.  Expected
    output = file 1 of the reference's self-paired `pe` run with the same flags (its own `se` crashes)."""
    rec = cu.e2e()["runs"]["se_equiv_selfpair_sanger"]
    argv = rec["argv"]
    assert argv[argv.index("-t") + 1] == "sanger" and "-q" not in argv and "-l" not in argv  # the defaults are 20 / 20
    out = os.path.join(str(workdir), "config0.fastq")
    pr = cu.run_cli(cu.PRODUCT_BIN, workdir, ["se", "-f", "{inputs}/test.fastq", "-t", "sanger", "-q", "20", "-l", "20", "-o", out])
    assert pr.returncode == 0, pr.stderr
    assert os.path.getsize(out) == rec["outputs"]["o1.fastq"]["size"]
    # default -a (all host threads): the SE deal puts read k in queue (k+1) mod T, so compare as record sets ...
    from fastq_util import parse_fastq
    got = parse_fastq(open(out, "rb").read())
    assert len(got) == 2500 and "FastQ records kept: 2500" in pr.stdout.decode()
    # ... and byte for byte at -a 1
    pr = cu.run_cli(cu.PRODUCT_BIN, workdir, ["se", "-f", "{inputs}/test.fastq", "-t", "sanger", "-q", "20", "-l", "20", "-o", out, "-a", "1"])
    assert pr.returncode == 0, pr.stderr
    assert cu.md5_file(out) == rec["outputs"]["o1.fastq"]["md5"]


@pytest.mark.parametrize("name", sorted(cu.e2e()["thread_order"].keys()))
def test_thread_order_goldens_on_gpu(workdir, name):
    """`sickle pe -a T` with T > 1 through the product binary: the reference's per-batch chunks in batch order,
    byte for byte (tests/golden/make_golden.py: thread_order_goldens)."""
    cu.check_run(cu.PRODUCT_BIN, workdir, name, dict(cu.e2e()["thread_order"][name], stdout=""), summary=False)


def test_se_thread_order_on_gpu(workdir):
    """`sickle se -a 4` and the default -a (every host thread) through the product binary: read k of a batch in
    queue (k+1) mod T, derived from the oracle's cuts and the restated batch-cut rule."""
    import os as _os
    from test_cli_host import derived_expectation
    src = _os.path.join(cu.INPUTS, "test.fastq")
    out = _os.path.join(str(workdir), "se_gpu_T.fastq")
    for threads in (4, None):
        argv = ["se", "-f", src, "-t", "illumina", "-o", out] + (["-a", str(threads)] if threads else [])
        pr = cu.run_cli(cu.PRODUCT_BIN, workdir, argv)
        assert pr.returncode == 0, pr.stderr
        T = threads or max(1, _os.cpu_count() or 1)  # default: std::thread::hardware_concurrency()
        assert open(out, "rb").read() == derived_expectation([src], "illumina", T, single=True), threads


def test_malformed_record_in_a_later_batch_exits_1_on_gpu(workdir):
    """A malformed record far into the file: validate() fires on the reader thread while the main thread is in
    sk_submit / sk_wait for earlier batches -- the process must leave with the reference's message and status 1
    (fatal_exit: no static destructors under live GPU work)."""
    from sickle_amd import synth
    seq, qual = synth.make_reads(9, 60_000, 150, "sanger")
    good = synth.fastq_bytes(seq, qual)
    path = os.path.join(str(workdir), "late_bad.fastq")
    open(path, "wb").write(good + b"@broken\nACGT\n+\nIII\n" + synth.fastq_bytes(seq[:1000], qual[:1000], start=70_000))
    for threads in ("1", "8"):
        pr = cu.run_cli(cu.PRODUCT_BIN, workdir, ["se", "-f", path, "-t", "sanger", "-o", "{tmp}/late_bad_out.fastq", "-a", threads])
        assert pr.returncode == 1, (pr.returncode, pr.stderr[-300:])
        assert b"[ERROR] Sequence and quality lines have different lengths:" in pr.stderr


def test_drivers_embedded_as_a_library_on_gpu(workdir):
    """Trim_Paired / Trim_Single of the PRODUCT build used the way reference src/sickle.cpp:61-80 uses them, four runs in
    one process (PE, SE, PE with gzip input and -n, the first PE again), no front process, sickle_leave_fast false:
    every run's files are the reference's, hipMemGetInfo after each run shows the device memory given back
    (close_device / sk_destroy), and the resident set does not grow."""
    import subprocess
    binary = cu.build_embed("embed_gpu")
    out = subprocess.run(["ldd", binary], capture_output=True).stdout.decode()
    assert "libsickle_amd.so" in out and "libamdhip64" in out
    marks = cu.check_embedded(binary, workdir, gpu=True)
    print("embedded runs: device free / rss after each:", [(m[2], m[3]) for m in marks])
