#!/usr/bin/env python3
"""probe: which HIP calls make up the 0.2 s of sk_create"""
import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
def t(name, fn):
    t0 = time.perf_counter(); r = fn(); dt = time.perf_counter() - t0
    print("%-34s %.3f s (rc %s)" % (name, dt, r)); return r
n = C.c_int()
t("hipInit(0)", lambda: hip.hipInit(0))
t("hipGetDeviceCount", lambda: hip.hipGetDeviceCount(C.byref(n)))
t("hipSetDevice(0)", lambda: hip.hipSetDevice(0))
buf = (C.c_char * 4096)()
t("hipGetDeviceProperties", lambda: hip.hipGetDevicePropertiesR0600(buf, 0) if hasattr(hip, "hipGetDevicePropertiesR0600") else hip.hipGetDeviceProperties(buf, 0))
s1 = C.c_void_p(); s2 = C.c_void_p()
t("hipStreamCreateWithFlags", lambda: hip.hipStreamCreateWithFlags(C.byref(s1), 1))
t("hipStreamCreateWithFlags (2nd)", lambda: hip.hipStreamCreateWithFlags(C.byref(s2), 1))
d = C.c_void_p()
t("hipMalloc(64 B)", lambda: hip.hipMalloc(C.byref(d), C.c_size_t(64)))
t("hipMemset", lambda: hip.hipMemset(d, 255, C.c_size_t(64)))
h = C.c_void_p()
t("hipHostMalloc(64 B)", lambda: hip.hipHostMalloc(C.byref(h), C.c_size_t(64), 0))
e = C.c_void_p()
t("hipEventCreateWithFlags", lambda: hip.hipEventCreateWithFlags(C.byref(e), 2))
h2 = C.c_void_p()
t("hipHostMalloc(400 MiB)", lambda: hip.hipHostMalloc(C.byref(h2), C.c_size_t(400 << 20), 0))
d2 = C.c_void_p()
t("hipMalloc(400 MiB)", lambda: hip.hipMalloc(C.byref(d2), C.c_size_t(400 << 20)))
