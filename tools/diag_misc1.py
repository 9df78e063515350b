"""Race screen of the other widenings (one process, same inputs, several runs, bitwise): triangular solver, gen_to_std,
generalized eigensolver.  Start several copies at once."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dla_future_amd as d
from oracle import oracle
from oracle import red2band as rb

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
d.initialize()
g = d.Grid.single()
n, nb = 4096, 512
b0 = oracle.set_random_hpd(n, nb, np.float64)
a0 = rb.random_hermitian(n, np.float64, seed=9)
rng = np.random.default_rng(2)
rhs = np.asfortranarray(rng.uniform(-1, 1, (n, 1500)))
out = {"solver": [], "hegst": [], "geneig": []}
for r in range(reps):
    fac = b0.copy(order="F")
    assert d.cholesky_factorization(g, "L", fac, nb) == 0
    x = rhs.copy(order="F")
    d.triangular_solver(g, "L", "L", "N", "N", 1.0, fac, x, nb)
    out["solver"].append(hashlib.md5(x.tobytes()).hexdigest()[:8])
    a = a0.copy(order="F")
    assert d.generalized_to_standard(g, "L", a, fac, nb) == 0
    out["hegst"].append(hashlib.md5(np.tril(a).tobytes()).hexdigest()[:8])
    w, z = d.hermitian_generalized_eigensolver(g, "L", a0.copy(order="F"), b0.copy(order="F"), nb)
    out["geneig"].append(hashlib.md5(z.tobytes() + w.tobytes()).hexdigest()[:8])
print(f"pid {os.getpid()}: " + "  ".join(f"{k} {'DETERMINISTIC ' + v[0] if len(set(v)) == 1 else 'DIFFERS ' + str(v)}" for k, v in out.items()),
      flush=True)
