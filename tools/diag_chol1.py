"""Diagnosis / race screen: the one-process Cholesky, same input, several runs: bitwise repeatable?  Start several copies at once
to put the kernels of several processes on the GPU together (that is what exposed the bt_apply race of round 4)."""
import hashlib
import time
T0 = time.time()
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dla_future_amd as d

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 512
t = sys.argv[3] if len(sys.argv) > 3 else "d"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
dt = {"d": np.float64, "z": np.complex128}[t]
T_IMPORT = time.time() - T0
d.initialize()
T_INIT = time.time() - T0
g = d.Grid.single()
a0 = np.zeros((n, n), dtype=dt, order="F")
d.set_random_hermitian_positive_definite(g, a0, n, nb)
hs = []
for r in range(reps):
    a = a0.copy(order="F")
    assert d.cholesky_factorization(g, "L", a, nb) == 0
    hs.append(hashlib.md5(np.tril(a).tobytes()).hexdigest()[:8])
print(f"pid {os.getpid()} {t} n={n} nb={nb}: {'DETERMINISTIC ' + hs[0] if len(set(hs)) == 1 else 'DIFFERS ' + str(hs)}", flush=True)
print(f"time pid {os.getpid()}: import {T_IMPORT:.1f} s, initialize at {T_INIT:.1f} s, all {time.time() - T0:.1f} s", flush=True)
