import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sickle_amd import capi
t0 = time.perf_counter(); lib = capi.lib(); t1 = time.perf_counter()
ctx = capi.Context(0, 2); t2 = time.perf_counter()
print("dlopen %.3f s, sk_create %.3f s" % (t1 - t0, t2 - t1))
for mb in (16, 64, 256, 256):
    t = time.perf_counter(); p = lib.sk_host_alloc(ctx._h, mb << 20); dt = time.perf_counter() - t
    print("hipHostMalloc %4d MB: %.3f s (%.2f GB/s)" % (mb, dt, (mb / 1024) / dt))
