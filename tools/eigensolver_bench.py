"""Hermitian eigensolver (BASELINE configuration 5: N=20480 nb=512) through the reference's C entry on host arrays:
per-stage device times, TFlop/s in the reference miniapp's model (miniapp_eigensolver.cpp: 4/3 n^3 + 2 n^3 ... reported
here per stage), sampled correctness (residual / orthogonality of a few eigenpairs).
usage: eigensolver_bench.py N nb [type] [runs]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dla_future_amd as d

n, nb = int(sys.argv[1]), int(sys.argv[2])
t = sys.argv[3] if len(sys.argv) > 3 else "d"
runs = int(sys.argv[4]) if len(sys.argv) > 4 else 2
dt = {"s": np.float32, "d": np.float64, "c": np.complex64, "z": np.complex128}[t]
cx = np.dtype(dt).kind == "c"
d.initialize()
g = d.Grid.single()
rng = np.random.default_rng(1)
t0 = time.time()
a0 = np.empty((n, n), dtype=dt, order="F")
for j0 in range(0, n, 1024):
    blk = rng.uniform(-1, 1, (n, min(1024, n - j0)))
    if cx:
        blk = blk + 1j * rng.uniform(-1, 1, blk.shape)
    a0[:, j0:j0 + blk.shape[1]] = blk
# Hermitian: the lower triangle is what counts
a0[np.arange(n), np.arange(n)] = a0[np.arange(n), np.arange(n)].real
print(f"generated {n}x{n} {t} in {time.time() - t0:.1f}s", flush=True)
names = ["reduction_to_band", "band_to_tridiagonal", "tridiagonal_eigensolver", "bt_band_to_tridiagonal", "bt_reduction_to_band"]
for r in range(runs):
    a = a0.copy(order="F")
    t0 = time.time()
    w, z = d.hermitian_eigensolver(g, "L", a, nb)
    wall = time.time() - t0
    ms = d.eigensolver_profile()
    tot = sum(ms)
    print(f"[{r}] eigensolver N={n} nb={nb} band={d.get_band_size(nb)} type={t}: device stages {tot:.1f} ms, wall (with PCIe staging) {wall:.2f} s", flush=True)
    for nm, m in zip(names, ms):
        print(f"     {nm:26s} {m:9.2f} ms", flush=True)
    del a
# sampled correctness: residual of a few eigenpairs with the full (Hermitian) matrix, orthogonality of a column sample
cols = np.unique(np.concatenate([np.arange(0, n, max(1, n // 24)), [n - 1]]))
lower = np.tril(a0)
zc = z[:, cols]
az = lower @ zc + np.tril(a0, -1).conj().T @ zc
res = np.abs(az - zc * w[cols][None, :]).max()
orth = np.abs(zc.conj().T @ zc - np.eye(len(cols))).max()
eps = np.finfo(np.zeros(1, dtype=dt).real.dtype).eps
err = (8 if cx else 2) * eps
print(f"RESULT eigensolver N={n} nb={nb} type={t}: sorted {bool(np.all(np.diff(w) >= 0))}  sampled |A z - w z|max {res:.3e} "
      f"(bar 2 n err |w|max = {2 * n * err * np.abs(w).max():.3e})  sampled |Z^H Z - I|max {orth:.3e} (bar {10 * n * err:.3e})", flush=True)
