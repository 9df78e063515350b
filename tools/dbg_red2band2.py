import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dla_future_amd as d
from oracle import red2band as rb
d.initialize(); g = d.Grid.single()
DT = {"s": np.float32, "d": np.float64, "c": np.complex64, "z": np.complex128}
cases = [("d", 515, 128, 64), ("d", 300, 64, 32), ("z", 260, 64, 32), ("d", 1100, 256, 128), ("z", 700, 256, 128), ("s", 300, 64, 16)]
refs = {}
nbad = 0
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for rep in range(reps):
    for (t, n, nb, b) in cases:
        dt = DT[t]
        key = (t, n, nb, b)
        if key not in refs:
            a0 = rb.random_hermitian(n, dt, seed=1000 + n + nb)
            ref = a0.copy(order="F"); rt = rb.reduction_to_band(ref, nb, b)
            refs[key] = (a0, ref, rt)
        a0, ref, rt = refs[key]
        store = np.full((n + 2, n), 5.5, dtype=dt, order="F")
        a = store[:n, :n]; a[...] = a0
        a[np.triu_indices(n, 1)] = -9.9
        taus = d.reduction_to_band(g, a, nb, b)
        tol = n * n * rb.error_of(dt)
        dtau = np.abs(taus - rt)
        dm = np.abs(np.tril(a) - np.tril(ref))
        if dm.max() > tol or dtau.max() > tol:
            nbad += 1
            print("BAD rep", rep, key, "max dtaus", dtau.max(), "first bad tau", np.nonzero(dtau > tol)[0][:4], "max dA", dm.max(),
                  "bad cols", np.nonzero(dm.max(axis=0) > tol)[0][:6], "bad rows", np.nonzero(dm.max(axis=1) > tol)[0][:6], flush=True)
print("done, bad =", nbad, "of", reps * len(cases))
