"""ctypes bindings for the TEST oracle (oracle/libsk_oracle.so) and, where it was
built, the compiled reference (oracle/_ref/libsickle_ref.so, oracle/_ref/sickle).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libsk_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libsickle_ref.so")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "sickle")

QUALTYPES = {"phred": 0, "sanger": 1, "solexa": 2, "illumina": 3}


class Params(C.Structure):
    _fields_ = [("qualtype", C.c_int32), ("qual_threshold", C.c_int32),
                ("length_threshold", C.c_int32), ("no_fiveprime", C.c_int32),
                ("trunc_n", C.c_int32)]


class Err(C.Structure):
    _fields_ = [("read", C.c_uint32), ("pos", C.c_uint32), ("ch", C.c_int32)]


def make_params(qualtype="sanger", q=20, l=20, no5=False, trunc_n=False):
    qt = QUALTYPES[qualtype] if isinstance(qualtype, str) else int(qualtype)
    return Params(qt, int(q), int(l), int(bool(no5)), int(bool(trunc_n)))


def build_oracle():
    """(Re)build the oracle with make; a no-op when up to date."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        lib = C.CDLL(ORACLE_SO)
        u8p = C.c_void_p
        lib.sko_sliding_window.restype = C.c_int
        lib.sko_sliding_window.argtypes = [C.POINTER(Params), u8p, u8p, C.c_int32,
                                           C.c_void_p, C.POINTER(Err)]
        for name in ("sko_trim_batch", "sko_trim_batch_mt"):
            fn = getattr(lib, name)
            fn.restype = C.c_int
            fn.argtypes = [C.POINTER(Params), u8p, u8p, C.c_void_p, C.c_uint32, C.c_uint32,
                           C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(Err)]
        lib.sko_trim_batch_mt.argtypes = lib.sko_trim_batch_mt.argtypes + [C.c_int]
        lib.sko_format_error.restype = C.c_int
        lib.sko_format_error.argtypes = [C.POINTER(Params), C.c_char_p, C.c_size_t, u8p, C.c_size_t,
                                         C.POINTER(Err), C.c_char_p, C.c_size_t]
        _oracle = lib
    return _oracle


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def oracle_trim_batch(params, qual, seq=None, offsets=None, stride=0, read_len=0, lengths=None,
                      n_reads=None, threads=1):
    """Returns (cuts[n,2] int32, err) with err = None or (read, pos, ch)."""
    qual = np.ascontiguousarray(qual, dtype=np.uint8)
    if seq is not None:
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
    if offsets is not None:
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
    else:
        n = n_reads if n_reads is not None else (len(lengths) if lengths is not None else qual.size // stride)
    if lengths is not None:
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
    out = np.full((n, 2), -7, dtype=np.int32)
    err = Err()
    lib = oracle()
    if threads > 1:
        rc = lib.sko_trim_batch_mt(C.byref(params), _ptr(qual), _ptr(seq), _ptr(offsets), stride, read_len,
                                   _ptr(lengths), n, _ptr(out), C.byref(err), threads)
    else:
        rc = lib.sko_trim_batch(C.byref(params), _ptr(qual), _ptr(seq), _ptr(offsets), stride, read_len,
                                _ptr(lengths), n, _ptr(out), C.byref(err))
    return out, ((err.read, err.pos, err.ch) if rc else None)


def oracle_format_error(params, name, qual_bytes, err):
    buf = C.create_string_buffer(4096 + len(qual_bytes) + len(name))
    e = Err(*err)
    q = np.frombuffer(bytes(qual_bytes), dtype=np.uint8)
    oracle().sko_format_error(C.byref(params), name, len(name), _ptr(q), len(qual_bytes), C.byref(e),
                              buf, len(buf))
    return buf.value


# ---------------------------------------------------------------- compiled reference
_ref = None


def have_ref():
    return os.path.exists(REF_SO) and os.path.exists(REF_BIN)


def ref():
    global _ref
    if _ref is None:
        lib = C.CDLL(REF_SO)
        i5 = C.c_int32 * 5
        lib.ref_sliding_window_forked.restype = C.c_int
        lib.ref_sliding_window_forked.argtypes = [i5, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int,
                                                  C.c_void_p, C.c_char_p, C.c_int]
        lib.ref_trim_batch.restype = None
        lib.ref_trim_batch.argtypes = [i5, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                       C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
        _ref = lib
    return _ref


def _i5(params):
    return (C.c_int32 * 5)(params.qualtype, params.qual_threshold, params.length_threshold,
                           params.no_fiveprime, params.trunc_n)


def ref_trim_batch(params, qual, seq=None, offsets=None, stride=0, read_len=0, lengths=None,
                   n_reads=None, threads=1):
    """The reference's own sliding_window over a packed batch (inputs must be range-clean)."""
    qual = np.ascontiguousarray(qual, dtype=np.uint8)
    if seq is not None:
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
    if offsets is not None:
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
    else:
        n = n_reads if n_reads is not None else (len(lengths) if lengths is not None else qual.size // stride)
    if lengths is not None:
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
    out = np.full((n, 2), -7, dtype=np.int32)
    ref().ref_trim_batch(_i5(params), _ptr(qual), _ptr(seq), _ptr(offsets), stride, read_len,
                         _ptr(lengths), n, _ptr(out), threads)
    return out


def ref_sliding_window_forked(params, name, seq, qual):
    """One read through the reference in a child process: (rc, (five, three), stderr_text)."""
    q = np.frombuffer(bytes(qual), dtype=np.uint8)
    s = np.frombuffer(bytes(seq), dtype=np.uint8)
    out = np.zeros(2, dtype=np.int32)
    buf = C.create_string_buffer(8192)
    rc = ref().ref_sliding_window_forked(_i5(params), name, _ptr(s), _ptr(q), len(q), _ptr(out), buf, len(buf))
    return rc, (int(out[0]), int(out[1])), buf.value
