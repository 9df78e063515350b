"""band_to_tridiagonal alone: random band matrix N, band b -> device ms (stage timer), correctness by spectrum
usage: b2t_bench.py N nb band [type] [check]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dla_future_amd as d
n, nb, band = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
t = sys.argv[4] if len(sys.argv) > 4 else "d"
check = len(sys.argv) > 5
dt = {"s": np.float32, "d": np.float64, "c": np.complex64, "z": np.complex128}[t]
d.initialize()
g = d.Grid.single()
rng = np.random.default_rng(0)
a = np.zeros((n, n), dtype=dt, order="F")
for k in range(band + 1):
    v = rng.uniform(-1, 1, n - k)
    if np.dtype(dt).kind == "c" and k > 0:
        v = v + 1j * rng.uniform(-1, 1, n - k)
    a[np.arange(k, n), np.arange(0, n - k)] = v
for r in range(3):
    dd, ee, v = d.band_to_tridiagonal(g, a, nb, band)
    ms = d.eigensolver_profile()[1]
    steps = sum(-((n - s - 2) // -band) for s in range(n - 2))
    print(f"[{r}] band_to_tridiagonal N={n} band={band} type={t}: {ms:.1f} ms  ({steps} steps, {ms * 1e3 / steps * min(256, (n // band + 2) // 2 + 1):.1f} us per step per workgroup)", flush=True)
if check:
    import scipy.linalg as sl
    ab = np.zeros((band + 1, n), dtype=dt)
    for k in range(band + 1):
        ab[k, :n - k] = a[np.arange(k, n), np.arange(0, n - k)]
    ref = sl.eigvals_banded(ab, lower=True)
    got = sl.eigvalsh_tridiagonal(dd.astype(np.float64), ee.astype(np.float64))
    print("max |eig diff|", np.abs(ref - got).max(), "bar n eps |A|", n * np.finfo(dd.dtype).eps * np.abs(ref).max())
