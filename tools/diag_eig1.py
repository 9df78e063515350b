"""Diagnosis: the whole eigensolver on ONE process, same input, several runs: bitwise repeatable?"""
import hashlib
import time
T0 = time.time()
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dla_future_amd as d
from oracle import red2band as rb
from oracle import tridiag as td

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
nb = 256
T_IMPORT = time.time() - T0
d.initialize()
T_INIT = time.time() - T0
g = d.Grid.single()
a0 = rb.random_hermitian(n, np.float64, seed=700 + n)
out = []
for r in range(reps):
    w, z = d.hermitian_eigensolver(g, "L", a0.copy(order="F"), nb)
    res = td.check_eigensolver(a0, w, z)
    out.append((hashlib.md5(z.tobytes()).hexdigest()[:8], f"{res['orth']:.1e}"))
print(f"pid {os.getpid()} n={n}: {out}", flush=True)
print(f"time pid {os.getpid()}: import {T_IMPORT:.1f} s, initialize at {T_INIT:.1f} s, all {time.time() - T0:.1f} s", flush=True)
