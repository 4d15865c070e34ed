#!/usr/bin/env python3
"""probe: does pinning 400 MiB overlap with the first stream creation when they run on two threads?"""
import ctypes as C, threading, time
hip = C.CDLL("libamdhip64.so")
t0 = time.perf_counter()
marks = {}
def a():
    hip.hipInit(0); hip.hipSetDevice(0)
    s = C.c_void_p(); hip.hipStreamCreateWithFlags(C.byref(s), 1)
    marks["stream"] = time.perf_counter() - t0
def b():
    hip.hipInit(0); hip.hipSetDevice(0)
    h = C.c_void_p(); hip.hipHostMalloc(C.byref(h), C.c_size_t(400 << 20), 0)
    marks["pinned"] = time.perf_counter() - t0
ta, tb = threading.Thread(target=a), threading.Thread(target=b)
ta.start(); tb.start(); ta.join(); tb.join()
print("two threads: stream ready at %.3f s, 400 MiB pinned at %.3f s" % (marks["stream"], marks["pinned"]))
