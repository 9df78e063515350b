"""Diagnosis: is bt_band_to_tridiagonal (fused fp64 path) deterministic?  One process, same input, several runs; run
several copies of this script at once to put the kernels of several processes on the GPU together."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dla_future_amd as d
from oracle import red2band as rb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
band, nb = 128, 256
d.initialize()
g = d.Grid.single()
a0 = rb.random_hermitian(n, np.float64, seed=3, banded=band)
dd, ee, v = d.band_to_tridiagonal(g, a0.copy(order="F"), nb, band)
rng = np.random.default_rng(5)
e0 = np.asfortranarray(rng.uniform(-1, 1, (n, k)))
hs = []
for r in range(reps):
    e = e0.copy(order="F")
    d.bt_band_to_tridiagonal(band, e, v)
    hs.append(hashlib.md5(e.tobytes()).hexdigest()[:8])
print(f"pid {os.getpid()} n={n} k={k}: {hs} -> {'DETERMINISTIC' if len(set(hs)) == 1 else 'DIFFERS'}", flush=True)
