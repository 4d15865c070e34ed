"""CPU: the C-ABI library loads and exports every symbol include/sickle_amd.h declares (no
compute calls -- those need the GPU), and refuses to pretend when there is no device."""
import ctypes as C
import os
import re

import pytest

from sickle_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sickle_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sk_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(capi.EXPORTS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(capi.LIB_PATH), "libsickle_amd.so not built: run __graft_entry__.build()"
    lib = C.CDLL(capi.LIB_PATH)
    for sym in declared_symbols():
        assert hasattr(lib, sym), sym


def test_tables_match_reference_constants():
    lib = capi.lib()
    assert lib.sk_abi_version() == 2
    want = {0: (0, 4, 60), 1: (33, 33, 126), 2: (64, 58, 112), 3: (64, 64, 110)}  # reference src/sickle.h:85-91
    for qt, k in want.items():
        p = lib.sk_quality_constants(qt)
        assert (p[0], p[1], p[2]) == k
    assert [lib.sk_typename(i) for i in range(4)] == [b"Phred", b"Sanger", b"Solexa", b"Illumina"]
    assert not lib.sk_quality_constants(7)
    assert lib.sk_kernel_name(1) == b"sk_scan_tile_kernel" and lib.sk_kernel_name(2) == b"sk_scan_team_kernel" and lib.sk_kernel_name(5) == b"sk_scan_tile_any_kernel" and lib.sk_kernel_name(6) == b"sk_scan_stream_kernel"


def test_no_device_means_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.SickleError):
        capi.Context(device=0)


def test_product_binary_fails_loudly_without_gpu(tmp_path):
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    binary = os.path.join(ROOT, "sickle_amd", "sickle")
    assert os.path.exists(binary)
    src = os.path.join(ROOT, "tests", "golden", "inputs", "test.fastq")
    pr = subprocess.run([binary, "se", "-f", src, "-t", "sanger", "-o", str(tmp_path / "o.fastq")], capture_output=True)
    assert pr.returncode == 1 and b"no usable MI355X" in pr.stderr
    assert not (tmp_path / "o.fastq").exists() or os.path.getsize(tmp_path / "o.fastq") == 0
