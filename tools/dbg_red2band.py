import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dla_future_amd as d
from oracle import red2band as rb
d.initialize(); g = d.Grid.single()
DT = {"s": np.float32, "d": np.float64, "c": np.complex64, "z": np.complex128}
small = [(12, 3, 3), (13, 3, 3), (24, 3, 3), (40, 5, 5), (4, 4, 2), (12, 4, 2), (42, 6, 3), (13, 6, 3), (27, 9, 3), (42, 12, 4), (29, 9, 3)]
fast = [("d", 300, 64, 32), ("d", 515, 128, 64), ("z", 260, 64, 32), ("s", 300, 64, 16), ("c", 200, 64, 64), ("d", 1100, 256, 128), ("z", 700, 256, 128)]
refs = {}
nbad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    cases = [(t, n, nb, b) for t in "dzsc" for (n, nb, b) in small] + fast
    for (t, n, nb, b) in cases:
        dt = DT[t]
        key = (t, n, nb, b)
        if key not in refs:
            a0 = rb.random_hermitian(n, dt, seed=1000 + n + nb)
            ref = a0.copy(order="F"); rt = rb.reduction_to_band(ref, nb, b)
            refs[key] = (a0, ref, rt)
        a0, ref, rt = refs[key]
        a = a0.copy(order="F")
        taus = d.reduction_to_band(g, a, nb, b)
        tol = n * n * rb.error_of(dt)
        dtau = np.abs(taus - rt)
        dm = np.abs(np.tril(a) - np.tril(ref))
        if dm.max() > tol or (len(taus) and dtau.max() > tol):
            nbad += 1
            print("BAD rep", rep, key, "max dtaus", dtau.max(), "first bad tau", np.nonzero(dtau > tol)[0][:4], "max dA", dm.max(),
                  "bad cols", np.nonzero(dm.max(axis=0) > tol)[0][:6], "bad rows", np.nonzero(dm.max(axis=1) > tol)[0][:6], flush=True)
print("done, bad =", nbad)
