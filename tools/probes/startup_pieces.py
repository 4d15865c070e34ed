#!/usr/bin/env python3
"""probe: what the CLI's device start-up consists of -- sk_create, pinned staging of one slot, the first launch"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.perf_counter()
L = C.CDLL(os.path.join(ROOT, "sickle_amd", "libsickle_amd.so"))
L.sk_host_alloc.restype = C.c_void_p
L.sk_host_alloc.argtypes = [C.c_void_p, C.c_size_t]
t1 = time.perf_counter()
h = C.c_void_p()
rc = L.sk_create(0, 2, C.byref(h))
t2 = time.perf_counter()
p1 = L.sk_host_alloc(h, 400 << 20)
t3 = time.perf_counter()
p2 = L.sk_host_alloc(h, 400 << 20)
t4 = time.perf_counter()
p3 = L.sk_host_alloc(h, 21 << 20)
t5 = time.perf_counter()
print("dlopen %.3f s  sk_create %.3f s (rc %d)  pinned 400 MiB %.3f s, again %.3f s, 21 MiB %.3f s" % (t1 - t0, t2 - t1, rc, t3 - t2, t4 - t3, t5 - t4))
