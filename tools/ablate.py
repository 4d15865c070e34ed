#!/usr/bin/env python3
"""Diagnostic: times the uniform tile kernel whole / DMA-only / scan-only at several
occupancies (waves per block x blocks per CU).  Not part of the product or of bench.py."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

class Args(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("stride", C.c_uint32), ("read_len", C.c_uint32), ("qmin", C.c_int32),
                ("qmax", C.c_int32), ("craw", C.c_int32), ("cthr", C.c_int32), ("cthr_raw", C.c_int32),
                ("lthr", C.c_int32), ("no5", C.c_int32), ("truncn", C.c_int32), ("tile_order", C.c_int32)]

lib = C.CDLL(os.environ.get("SK_LIB", os.path.join(ROOT, "sickle_amd", "libsickle_amd.so")))
lib.sk_launch_tile_ablate.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Args), C.c_int, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
q = bench.synth_quals_device(torch, n, 150, 152, 1234, dev)
out = torch.empty((n, 2), dtype=torch.int32, device=dev)
err = torch.full((1,), -1, dtype=torch.int64, device=dev)
a = Args(n, 152, 150, 33, 126, 53, 53, 53, 20, 0, 0, int(os.environ.get('SK_TILE_ORDER', '0')))
s = torch.cuda.Stream(dev)
torch.cuda.synchronize()
NAMES = {0: "2buf mfma full", 1: "2buf dma-only", 2: "2buf mfma scan-only", 10: "2buf valu full", 12: "2buf valu scan-only",
         100: "1buf mfma full", 101: "1buf dma-only", 102: "1buf mfma scan-only", 110: "1buf valu full",
         200: "staged mfma full", 201: "staged load-only", 202: "staged scan-only"}
CONFIGS = [(8, (0, 1, 2, 10, 12)), (16, (100, 101, 102, 110)), (12, (100,)), (8, (100,)), (6, (0,))]
if os.environ.get("SK_ABLATE_SHORT"):
    CONFIGS = [(16, (100, 101, 102)), (8, (0, 1))]
if os.environ.get("SK_ABLATE_STAGED"):
    CONFIGS = [(16, (100,)), (12, (200,))] * 4 + [(16, (101,)), (12, (201,)), (16, (102,)), (12, (202,)), (16, (101,)), (12, (201,))]
    # the staged kernel must give the same cuts
    ref = torch.empty_like(out)
    for mode, dst in ((100, ref), (200, out)):
        assert lib.sk_launch_tile_ablate(mode, q.data_ptr(), dst.data_ptr(), err.data_ptr(), C.byref(a), 256, 1, 12, s.cuda_stream) == 0
    s.synchronize()
    print("staged cuts identical to the DMA kernel's:", bool(torch.equal(ref, out)), "error word", int(err.item()), flush=True)
    # ragged tail: a batch that does not end on a tile boundary
    a2 = Args(n - 37, 152, 150, 33, 126, 53, 53, 53, 20, 0, 0, 0)
    ref.zero_(); out.zero_()
    for mode, dst in ((100, ref), (200, out)):
        assert lib.sk_launch_tile_ablate(mode, q.data_ptr(), dst.data_ptr(), err.data_ptr(), C.byref(a2), 256, 1, 12, s.cuda_stream) == 0
    s.synchronize()
    print("ragged batch identical:", bool(torch.equal(ref, out)), flush=True)
if os.environ.get("SK_ABLATE_AB"):
    CONFIGS = [(16, (100,)), (8, (0,)), (16, (100,)), (8, (0,)), (16, (100,)), (8, (0,)), (16, (101,)), (8, (1,)), (16, (102,)), (8, (2,))]
B2B = bool(os.environ.get("SK_ABLATE_B2B"))  # bench style: 50 launches queued back to back, one event pair each
for per_cu, modes in CONFIGS:
    for mode in modes:
        ts = []
        if B2B:
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(55)]
            for e0, e1 in evs:
                e0.record(s)
                rc = lib.sk_launch_tile_ablate(mode, q.data_ptr(), out.data_ptr(), err.data_ptr(), C.byref(a), 256, 1, per_cu, s.cuda_stream)
                e1.record(s)
                assert rc == 0, rc
            s.synchronize()
            ts = [e0.elapsed_time(e1) for e0, e1 in evs[5:]]
            print("workgroups/CU %2d mode %3d (%-20s): avg %.4f ms over 50 queued launches  %.0f GB/s algorithmic" %
                  (per_cu, mode, NAMES[mode], sum(ts) / len(ts), 158 * n / (sum(ts) / len(ts)) / 1e6), flush=True)
            continue
        for it in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            rc = lib.sk_launch_tile_ablate(mode, q.data_ptr(), out.data_ptr(), err.data_ptr(), C.byref(a), 256, 1, per_cu, s.cuda_stream)
            e1.record(s)
            s.synchronize()
            assert rc == 0, rc
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        ms = ts[len(ts) // 2]
        print("workgroups/CU %2d mode %3d (%-20s): %.3f ms  %.0f GB/s algorithmic" %
              (per_cu, mode, NAMES[mode], ms, 158 * n / ms / 1e6), flush=True)
