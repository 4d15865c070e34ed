"""ctypes view of the C ABI in include/sickle_amd.h (libsickle_amd.so, built in-tree by
sickle_amd/csrc/Makefile).  No fallback of any kind: a missing library or a missing gfx950
device raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsickle_amd.so")

SK_OK, SK_ERANGE, SK_EINVAL, SK_ENODEV, SK_EHIP, SK_EBUSY = 0, 1, -1, -2, -3, -4
QUALTYPES = {"phred": 0, "sanger": 1, "solexa": 2, "illumina": 3}

# every entry point include/sickle_amd.h declares
EXPORTS = ("sk_quality_constants", "sk_typename", "sk_abi_version", "sk_device_count", "sk_create",
           "sk_destroy", "sk_last_error", "sk_device", "sk_host_alloc", "sk_host_free",
           "sk_scan_device_async", "sk_scan_device_finish", "sk_trim_batch", "sk_submit", "sk_wait",
           "sk_kernel_for", "sk_kernel_name", "sk_seg_classes", "sk_probe_read_bandwidth",
           "sk_count_pairs_device_async", "sk_count_pairs_device_finish", "sk_bgzf_deflate", "sk_bgzf_host_alloc", "sk_bgzf_host_free",
           "sk_bgzf_last_error")


class Params(C.Structure):
    _fields_ = [("qualtype", C.c_int32), ("qual_threshold", C.c_int32), ("length_threshold", C.c_int32),
                ("no_fiveprime", C.c_int32), ("trunc_n", C.c_int32)]


class Err(C.Structure):
    _fields_ = [("read", C.c_uint32), ("pos", C.c_uint32), ("ch", C.c_int32)]


class Tile(C.Structure):
    _fields_ = [("byte_off", C.c_uint64), ("slot0", C.c_uint32), ("stride", C.c_uint32), ("rows", C.c_uint16),
                ("read_len", C.c_uint16), ("reserved", C.c_uint32)]


TILE_DTYPE = np.dtype([("byte_off", "<u8"), ("slot0", "<u4"), ("stride", "<u4"), ("rows", "<u2"), ("read_len", "<u2"),
                       ("reserved", "<u4")])


class Batch(C.Structure):
    _fields_ = [("qual", C.c_void_p), ("seq", C.c_void_p), ("offsets", C.c_void_p), ("stride", C.c_uint32),
                ("read_len", C.c_uint32), ("lengths", C.c_void_p), ("n_reads", C.c_uint64),
                ("tiles", C.c_void_p), ("n_tiles", C.c_uint32), ("out_index", C.c_void_p),
                ("classes", C.c_void_p), ("n_classes", C.c_uint32), ("cuts_in_slot_order", C.c_uint32)]


class SegClass(C.Structure):
    _fields_ = [("first_tile", C.c_uint32), ("n_tiles", C.c_uint32), ("max_stride", C.c_uint32), ("wide", C.c_uint32)]


def seg_classes(tiles, max_classes=16):
    """sk_seg_classes on a numpy TILE_DTYPE array -> (ctypes array, count); count 0 = pass no table."""
    arr = (SegClass * max_classes)()
    n = lib().sk_seg_classes(tiles.ctypes.data, len(tiles), arr, max_classes)
    return arr, n


class PairCounts(C.Structure):
    _fields_ = [("both", C.c_uint64), ("only_first", C.c_uint64), ("only_second", C.c_uint64), ("none", C.c_uint64)]


class SickleError(RuntimeError):
    pass


class RangeError(SickleError):
    """A quality char outside the encoding's range (the reference's exit(1) path)."""

    def __init__(self, read, pos, ch):
        super().__init__("quality value %d out of range at read %d, position %d" % (ch, read, pos + 1))
        self.read, self.pos, self.ch = read, pos, ch


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SickleError("%s is missing: build it with `make -C sickle_amd/csrc` "
                              "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        # torch (when installed) bundles its own copy of the HIP runtime, and a process can only
        # initialise one: whichever copy is loaded first wins and the other then sees no device.
        # Load torch's first, so that a later `import torch` in the same process (bench.py, the
        # tests) still works; libsickle_amd.so binds to the copy that is already there.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.sk_quality_constants.restype = C.POINTER(C.c_int32)
        L.sk_quality_constants.argtypes = [C.c_int32]
        L.sk_typename.restype = C.c_char_p
        L.sk_typename.argtypes = [C.c_int32]
        L.sk_abi_version.restype = C.c_int
        L.sk_device_count.restype = C.c_int
        L.sk_create.restype = C.c_int
        L.sk_create.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.sk_destroy.restype = None
        L.sk_destroy.argtypes = [C.c_void_p]
        L.sk_last_error.restype = C.c_char_p
        L.sk_last_error.argtypes = [C.c_void_p]
        L.sk_device.restype = C.c_int
        L.sk_device.argtypes = [C.c_void_p]
        L.sk_host_alloc.restype = C.c_void_p
        L.sk_host_alloc.argtypes = [C.c_void_p, C.c_size_t]
        L.sk_host_free.restype = None
        L.sk_host_free.argtypes = [C.c_void_p, C.c_void_p]
        L.sk_scan_device_async.restype = C.c_int
        L.sk_scan_device_async.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Batch), C.c_void_p, C.c_void_p]
        L.sk_scan_device_finish.restype = C.c_int
        L.sk_scan_device_finish.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Err)]
        L.sk_trim_batch.restype = C.c_int
        L.sk_trim_batch.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Batch), C.c_void_p, C.POINTER(Err)]
        L.sk_submit.restype = C.c_int
        L.sk_submit.argtypes = [C.c_void_p, C.c_int, C.POINTER(Params), C.POINTER(Batch), C.c_void_p]
        L.sk_wait.restype = C.c_int
        L.sk_wait.argtypes = [C.c_void_p, C.c_int, C.POINTER(Err)]
        L.sk_kernel_for.restype = C.c_int
        L.sk_kernel_for.argtypes = [C.POINTER(Batch)]
        L.sk_kernel_name.restype = C.c_char_p
        L.sk_kernel_name.argtypes = [C.c_int]
        L.sk_probe_read_bandwidth.restype = C.c_int
        L.sk_probe_read_bandwidth.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.POINTER(C.c_double)]
        L.sk_count_pairs_device_async.restype = C.c_int
        L.sk_count_pairs_device_async.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.sk_count_pairs_device_finish.restype = C.c_int
        L.sk_count_pairs_device_finish.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(PairCounts)]
        L.sk_seg_classes.restype = C.c_uint32
        L.sk_seg_classes.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.sk_bgzf_deflate.restype = C.c_int
        L.sk_bgzf_deflate.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.sk_bgzf_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def make_params(qualtype="sanger", q=20, l=20, no5=False, trunc_n=False):
    qt = QUALTYPES[qualtype] if isinstance(qualtype, str) else int(qualtype)
    return Params(qt, int(q), int(l), int(bool(no5)), int(bool(trunc_n)))


def _np_ptr(a):
    return None if a is None else a.ctypes.data


class Context:
    """One sk_ctx.  Raises SickleError when no gfx950 device is usable."""

    def __init__(self, device=-1, slots=2):
        self._h = C.c_void_p()
        rc = lib().sk_create(device, slots, C.byref(self._h))
        if rc != SK_OK:
            raise SickleError("sk_create(device=%d) failed with %d: no usable gfx950 device "
                              "(this library has no CPU path)" % (device, rc))

    def close(self):
        if self._h:
            lib().sk_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, err=None):
        if rc == SK_OK:
            return
        if rc == SK_ERANGE:
            raise RangeError(err.read, err.pos, err.ch)
        raise SickleError("libsickle_amd call failed (%d): %s" % (rc, lib().sk_last_error(self._h).decode()))

    # ---- host buffers (numpy) -------------------------------------------------------------
    def trim_segmented(self, params, qual, tiles, out_index, max_stride, seq=None, slot_order=False):
        """sk_trim_batch on a segmented batch (tiles: numpy array of TILE_DTYPE) -> cuts[n,2] int32
        in the caller's read order (out_index), or in slot order with slot_order=True."""
        qual = np.ascontiguousarray(qual, dtype=np.uint8)
        seq = None if seq is None else np.ascontiguousarray(seq, dtype=np.uint8)
        tiles = np.ascontiguousarray(tiles, dtype=TILE_DTYPE)
        out_index = np.ascontiguousarray(out_index, dtype=np.uint32)
        n = len(out_index)
        out = np.full((n, 2), -7, dtype=np.int32)
        b = Batch(_np_ptr(qual), _np_ptr(seq), None, max_stride, 0, None, n, tiles.ctypes.data, len(tiles),
                  out_index.ctypes.data, None, 0, 1 if slot_order else 0)
        err = Err()
        rc = lib().sk_trim_batch(self._h, C.byref(params), C.byref(b), out.ctypes.data, C.byref(err))
        self._check(rc, err)
        return out

    def trim_batch(self, params, qual, seq=None, offsets=None, stride=0, read_len=0, lengths=None, n_reads=None):
        """sk_trim_batch on numpy host arrays -> cuts[n,2] int32.  Raises RangeError like the
        reference exits."""
        qual = np.ascontiguousarray(qual, dtype=np.uint8)
        seq = None if seq is None else np.ascontiguousarray(seq, dtype=np.uint8)
        if offsets is not None:
            offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
            n = len(offsets) - 1
        else:
            n = n_reads if n_reads is not None else (len(lengths) if lengths is not None else qual.size // stride)
        lengths = None if lengths is None else np.ascontiguousarray(lengths, dtype=np.uint32)
        out = np.full((n, 2), -7, dtype=np.int32)
        b = Batch(_np_ptr(qual), _np_ptr(seq), _np_ptr(offsets), stride, read_len, _np_ptr(lengths), n)
        err = Err()
        rc = lib().sk_trim_batch(self._h, C.byref(params), C.byref(b), out.ctypes.data, C.byref(err))
        self._check(rc, err)
        return out

    def submit(self, slot, params, qual, out, seq=None, offsets=None, stride=0, read_len=0, lengths=None, n_reads=0):
        b = Batch(_np_ptr(qual), _np_ptr(seq), _np_ptr(offsets), stride, read_len, _np_ptr(lengths), n_reads)
        self._check(lib().sk_submit(self._h, slot, C.byref(params), C.byref(b), out.ctypes.data))

    def wait(self, slot):
        err = Err()
        self._check(lib().sk_wait(self._h, slot, C.byref(err)), err)

    # ---- device-resident buffers (raw device pointers, e.g. torch .data_ptr()) --------------
    def scan_device_async(self, params, qual_ptr, out_ptr, n_reads, stride=0, read_len=0, seq_ptr=None,
                          offsets_ptr=None, lengths_ptr=None, stream=None):
        b = Batch(qual_ptr, seq_ptr, offsets_ptr, stride, read_len, lengths_ptr, n_reads)
        self._check(lib().sk_scan_device_async(self._h, C.byref(params), C.byref(b), out_ptr, stream))

    def probe_read_bandwidth(self, dev_ptr, nbytes, launches=20, stream=None):
        """GB/s of a read-only stream over [dev_ptr, dev_ptr + nbytes) on this device."""
        g = C.c_double()
        self._check(lib().sk_probe_read_bandwidth(self._h, dev_ptr, nbytes, launches, stream, C.byref(g)))
        return g.value

    def count_pairs_device(self, cuts_ptr, n_pairs, classes_ptr=None, stream=None):
        """Pair classes of the cuts at cuts_ptr (mates at 2k, 2k+1) -> (both, only_first, only_second, none)."""
        self._check(lib().sk_count_pairs_device_async(self._h, cuts_ptr, n_pairs, classes_ptr, stream))
        c = PairCounts()
        self._check(lib().sk_count_pairs_device_finish(self._h, stream, C.byref(c)))
        return c.both, c.only_first, c.only_second, c.none

    def scan_device_finish(self, stream=None):
        err = Err()
        self._check(lib().sk_scan_device_finish(self._h, stream, C.byref(err)), err)


BGZF_INPUT = 65280   # bytes of text per BGZF block
BGZF_SLOT = 65536    # bytes of output slot per block


def bgzf_deflate(data, device=0):
    """sk_bgzf_deflate on a bytes object -> the BGZF file image (framing done here, like the
    writer of the CLI does it).  Tests only."""
    import struct
    import zlib
    n_blocks = max(1, (len(data) + BGZF_INPUT - 1) // BGZF_INPUT)
    text = np.zeros(n_blocks * BGZF_INPUT, dtype=np.uint8)
    text[:len(data)] = np.frombuffer(data, dtype=np.uint8)
    sizes = np.array([min(BGZF_INPUT, len(data) - b * BGZF_INPUT) for b in range(n_blocks)], dtype=np.uint32)
    out = np.zeros(n_blocks * BGZF_SLOT, dtype=np.uint8)
    out_sizes = np.zeros(n_blocks, dtype=np.uint32)
    rc = lib().sk_bgzf_deflate(device, text.ctypes.data, sizes.ctypes.data, n_blocks, out.ctypes.data, out_sizes.ctypes.data)
    if rc != 0:
        raise SickleError("sk_bgzf_deflate: %d %s" % (rc, lib().sk_bgzf_last_error().decode()))
    blob = bytearray()
    for b in range(n_blocks):
        piece = data[b * BGZF_INPUT:b * BGZF_INPUT + int(sizes[b])]
        c = int(out_sizes[b])
        if c == 0 or c >= len(piece) + 5:
            n = len(piece)
            body = bytes([1, n & 0xff, n >> 8, ~n & 0xff, (~n >> 8) & 0xff]) + piece
        else:
            body = out[b * BGZF_SLOT:b * BGZF_SLOT + c].tobytes()
        total = 18 + len(body) + 8
        blob += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", total - 1) + body
        blob += struct.pack("<II", zlib.crc32(piece) & 0xffffffff, len(piece))
    return bytes(blob)
