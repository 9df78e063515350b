"""Does a resident factorization always report the negative first pivot?  (diagnosis of an intermittent failure of the
six-rank worker: factorize() returned 129 or 0 instead of 1 for n = 400, nb = 128 with a(0,0) = -1)
usage: dbg_nonspd.py [iterations]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dla_future_amd as dlaf
from oracle import oracle
it = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dlaf.initialize()
g = dlaf.Grid.single()
bad_results = {}
for n, nb, bad in [(400, 128, 0), (400, 64, 300), (200, 32, 77)]:
    a0 = oracle.set_random_hpd(n, nb, np.float64)
    a0[bad, bad] = -1.0
    for i in range(it):
        m = dlaf.DeviceMatrix(g, np.float64, "L", n, nb)
        m.upload(a0)
        r = m.factorize()
        m.close()
        if r != bad + 1:
            bad_results[(n, nb, bad, r)] = bad_results.get((n, nb, bad, r), 0) + 1
        # a host-array factorization in between, as the worker does
        if i % 3 == 0:
            x = a0.copy(order="F")
            r2 = dlaf.cholesky_factorization(g, "L", x, nb)
            if r2 != bad + 1:
                bad_results[("host", n, nb, bad, r2)] = bad_results.get(("host", n, nb, bad, r2), 0) + 1
print("YIELD", os.environ.get("DLAF_MI355X_POTRF_YIELD", "default"), "wrong results:", bad_results or "none")
