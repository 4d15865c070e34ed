#!/usr/bin/env python3
"""probe: how long does the kernel take to release a process that holds (a) a HIP context, (b) + 0.8 GB of pinned
host memory, (c) + 0.8 GB of device memory, (d) + 3 GB of touched heap?  Timed from outside (fork + wait)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(mode):
    L = C.CDLL(os.path.join(ROOT, "sickle_amd", "libsickle_amd.so"))
    L.sk_host_alloc.restype = C.c_void_p
    L.sk_host_alloc.argtypes = [C.c_void_p, C.c_size_t]
    h = C.c_void_p()
    if mode >= 1:
        assert L.sk_create(0, 2, C.byref(h)) == 0
    if mode >= 2:
        for _ in range(2):
            p = L.sk_host_alloc(h, 400 << 20)
            C.memset(p, 1, 400 << 20)
    if mode >= 3:
        hip = C.CDLL("libamdhip64.so")
        for _ in range(2):
            d = C.c_void_p()
            hip.hipMalloc(C.byref(d), C.c_size_t(400 << 20))
    if mode >= 4:
        buf = (C.c_char * (3 << 30))()
        C.memset(buf, 1, 3 << 30)
    sys.stdout.write("%.6f\n" % time.time())
    sys.stdout.flush()
    os._exit(0)


if len(sys.argv) > 1:
    child(int(sys.argv[1]))
import subprocess
for mode, what in ((0, "library loaded only"), (1, "+ sk_create"), (2, "+ 0.8 GB pinned"), (3, "+ 0.8 GB device"), (4, "+ 3 GB heap")):
    pr = subprocess.Popen([sys.executable, __file__, str(mode)], stdout=subprocess.PIPE)
    out = pr.stdout.readline()
    pr.wait()
    t_end = time.time()
    print("%-22s exit took %.3f s" % (what, t_end - float(out)))
