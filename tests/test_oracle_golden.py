"""CPU: the oracle (oracle/sk_oracle.c) against the golden vectors generated from the compiled
reference (tests/golden/make_golden.py), and -- where oracle/_ref exists -- against the
reference itself on fresh random inputs.  This is what pins the oracle."""
import json
import os

import numpy as np
import pytest

import oracle_bind as ob
from fastq_util import pack_records, parse_fastq

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def parse_key(k):
    qt, q, l, x, n = k.split("_")
    return qt, int(q[1:]), int(l[1:]), int(x[1:]), int(n[1:])


def test_bundled_file_grid():
    recs = parse_fastq(open(os.path.join(GOLD, "inputs", "test.fastq"), "rb").read())
    assert len(recs) == 2500
    seq, qual, offsets = pack_records(recs)
    cuts = np.load(os.path.join(GOLD, "cuts_bundled.npz"))
    assert len(cuts.files) == 80
    for k in cuts.files:
        got, err = ob.oracle_trim_batch(ob.make_params(*parse_key(k)), qual, seq, offsets=offsets)
        assert err is None and (got == cuts[k].astype(np.int32)).all(), k
    # the counts SURVEY.md 8c quotes from the reference
    c = cuts["illumina_q20_l20_x0_n0"]
    assert int((c[:, 1] >= 0).sum()) == 2483
    c = cuts["sanger_q20_l20_x0_n0"]
    assert (c[:, 0] == 0).all() and (c[:, 1] == 150).all()


def test_edge_set_grid():
    inp = np.load(os.path.join(GOLD, "edge_inputs.npz"))
    cuts = np.load(os.path.join(GOLD, "cuts_edge.npz"))
    assert len(cuts.files) == 192
    for k in cuts.files:
        got, err = ob.oracle_trim_batch(ob.make_params(*parse_key(k)), inp["qual"], inp["seq"], offsets=inp["offsets"])
        assert err is None and (got == cuts[k].astype(np.int32)).all(), k


def test_error_cases():
    for c in json.load(open(os.path.join(GOLD, "errors.json"))):
        pd = c["params"]
        p = ob.make_params(pd["qualtype"], pd["q"], pd["l"], pd["no5"], pd["trunc_n"])
        qual = np.frombuffer(bytes.fromhex(c["qual_hex"]), dtype=np.uint8)
        seq = np.frombuffer(c["seq"].encode("latin-1"), dtype=np.uint8)
        off = np.array([0, len(qual)], dtype=np.uint64)
        got, err = ob.oracle_trim_batch(p, qual, seq, offsets=off)
        if c["rc"] == 0:
            assert err is None and list(got[0]) == c["cut"], c["desc"]
        else:
            assert err is not None, c["desc"]
            text = ob.oracle_format_error(p, c["name"].encode("latin-1"), qual.tobytes(), err)
            assert text.decode("latin-1") == c["stderr"], c["desc"]


def test_window_size_is_integer_division():
    # reference src/trim.cpp:8 computes (int)(0.1 * L) in double; the kernels use L / 10
    L = np.arange(0, 2_000_001, dtype=np.int64)
    assert ((0.1 * L.astype(np.float64)).astype(np.int64) == L // 10).all()


def test_fixed_stride_and_ragged_layouts_agree():
    from sickle_amd import synth
    seq, qual = synth.make_reads(9, 5000, 150, "sanger", lower_n_frac=0.02)
    p = ob.make_params("sanger", 20, 20, False, True)
    a, _ = ob.oracle_trim_batch(p, synth.pack_fixed(qual, 152), synth.pack_fixed(seq, 152), stride=152, read_len=150, n_reads=5000)
    off = (np.arange(5001) * 150).astype(np.uint64)
    b, _ = ob.oracle_trim_batch(p, qual.reshape(-1), seq.reshape(-1), offsets=off)
    c, _ = ob.oracle_trim_batch(p, qual.reshape(-1), seq.reshape(-1), offsets=off, threads=3)
    assert (a == b).all() and (a == c).all()


@pytest.mark.skipif(not ob.have_ref(), reason="compiled reference (oracle/_ref) not present")
def test_fuzz_against_compiled_reference():
    rng = np.random.default_rng(2024)
    for trial in range(40):
        qt = ["sanger", "solexa", "illumina"][trial % 3]
        lo, hi = {"sanger": (33, 127), "solexa": (58, 113), "illumina": (64, 111)}[qt]
        n = 1500
        lens = rng.integers(1, 400, size=n).astype(np.uint32)
        offs = np.zeros(n + 1, dtype=np.uint64)
        offs[1:] = np.cumsum(lens)
        tot = int(offs[-1])
        center = rng.integers(lo, hi)
        qual = np.clip(center + rng.integers(-12, 13, size=tot), lo, hi - 1).astype(np.uint8)
        seq = rng.choice(np.frombuffer(b"ACGTACGTACGTNn", dtype=np.uint8), size=tot)
        p = ob.make_params(qt, int(rng.integers(0, 50)), int(rng.integers(0, 150)), trial % 2, (trial // 2) % 2)
        a, err = ob.oracle_trim_batch(p, qual, seq, offsets=offs)
        assert err is None
        b = ob.ref_trim_batch(p, qual, seq, offsets=offs, threads=2)
        assert (a == b).all(), trial


@pytest.mark.skipif(not ob.have_ref(), reason="compiled reference (oracle/_ref) not present")
def test_error_fuzz_against_compiled_reference():
    """Random reads with one illegal char at a random place: the oracle raises exactly when the
    reference exits, with the same message."""
    rng = np.random.default_rng(77)
    for trial in range(60):
        L = int(rng.integers(1, 200))
        qual = rng.integers(40, 75, size=L).astype(np.uint8)
        if trial % 2:
            cut = int(rng.integers(0, L))
            qual[cut:] = 34
        qual[int(rng.integers(0, L))] = int(rng.choice([10, 32, 127, 200]))
        seq = np.frombuffer(b"A" * L, dtype=np.uint8)
        p = ob.make_params("sanger", 20, int(rng.choice([0, 20, 60])), trial % 3 == 0, False)
        rc, cut, text = ob.ref_sliding_window_forked(p, b"@fz", seq.tobytes(), qual.tobytes())
        got, err = ob.oracle_trim_batch(p, qual, seq, offsets=np.array([0, L], dtype=np.uint64))
        if rc == 0:
            assert err is None and tuple(got[0]) == cut, trial
        else:
            assert rc == 1 and err is not None, trial
            assert ob.oracle_format_error(p, b"@fz", qual.tobytes(), err) == text, trial
