import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_bind as ob
from sickle_amd import capi
h = "2f2d322f2d2c2c32312d312d322e2e302e31302f31322e2c2e322d2d2f322c312d2f2d2e302d2c2c2f32322f2f3230312c312e2c2d2e2c2f3130302f2d2c2d3030322d2c2e2e2c2d2c32302e32322d2f322e2f2e2f2e2e2f302e2c2f3130322d312e302f2c2d312c302f31302c2d30322c2d312e322e2d322e30305b5a5f5c5d5f5c5a5f5e5c5e5a59"
r = np.frombuffer(bytes.fromhex(h), dtype=np.uint8)
ctx = capi.Context(0, 2)
p, po = capi.make_params("sanger", 30, 0, 0, 0), ob.make_params("sanger", 30, 0, 0, 0)
os.environ["SK_GENERAL"] = "stream"
def run(lens_before, lens_after, fill=45):
    parts = [np.full(l, fill, dtype=np.uint8) for l in lens_before] + [r] + [np.full(l, 70, dtype=np.uint8) for l in lens_after]
    lens = np.array([len(x) for x in parts], dtype=np.uint64)
    offs = np.zeros(len(parts) + 1, dtype=np.uint64); offs[1:] = np.cumsum(lens)
    q = np.concatenate(parts)
    want, _ = ob.oracle_trim_batch(po, q, None, offsets=offs, threads=1)
    got = ctx.trim_batch(p, q, None, offsets=offs)
    k = len(lens_before)
    print(lens_before, lens_after, "got", got[k], "want", want[k], "OK" if (got == want).all() else "MISMATCH", flush=True)
run([], [])
run([53, 103], [2387, 1])
run([53, 103], [2387])
run([53], [2387])
run([], [2387])
run([103], [2387])
run([103], [300])
run([103], [1100])
run([5000], [2387])
run([1500], [])
run([100, 100, 100, 100], [100, 100, 2387])
