"""The case tests/soak_wide.py found when the window walk of the medium-read tiles was left early (seed 21, iteration 0):
64 reads of 600 bases hovering 3 above the threshold, no 3' window anywhere.  Prints the oracle's cuts and the library's."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_bind as ob
from sickle_amd import capi
rng = np.random.default_rng(21)
ctx = capi.Context(0, 2)
it = 0
qt = "sanger"; lo, hi = 33, 126
L = int(rng.choice([320, 321, 329, 330, 351, 352, 505, 512, 600, 639, 640, 641, 650, 959, 960, 1000, 1023, 1024, 1025, 1279, 1280, 1500, 2000, 2047, 2048, 2520, 2528, 2529, 2600]))
n = int(rng.choice([1, 31, 32, 33, 64, 65]))
stride = L + int(rng.choice([0, 0, 1, 3, 8, 16, 40]))
tot = n * L
thr = int(rng.choice([0, 2, 15, 20, 25, 30, 41]))
mid = min(hi - 3, max(lo + 3, lo + thr + int(rng.integers(-4, 12))))
qual = np.clip(rng.normal(mid, 6, tot).astype(int), lo, hi).astype(np.uint8)
seq = rng.choice(np.frombuffer(b"ACGT" * 3000 + b"Nn", dtype=np.uint8), size=tot)
l = int(rng.choice([0, 20, 300, 1500])); x, tn = int(rng.integers(0, 2)), int(rng.integers(0, 2))
print("L", L, "n", n, "stride", stride, "thr", thr, "mid", mid, "l", l, "x", x, "tn", tn)
p, po = capi.make_params(qt, thr, l, x, tn), ob.make_params(qt, thr, l, x, tn)
offs = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
want, err = ob.oracle_trim_batch(po, qual, seq, offsets=offs, threads=8)
qs = np.full((n, stride), lo, dtype=np.uint8); qs[:, :L] = qual.reshape(n, L); qs = qs.reshape(-1)
got = ctx.trim_batch(p, qs, None, stride=stride, read_len=L, n_reads=n)
print("want", want[:6].tolist()); print("got ", got[:6].tolist()); print("equal", (got == want).all())
