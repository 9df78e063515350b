"""reduction_to_band / bt_reduction_to_band on resident operands: device time and TFlop/s (flops 4/3 n^3, 2 n^2 k).
usage: red2band_bench.py N nb [type] [runs] [bt]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dla_future_amd as d

n, nb = int(sys.argv[1]), int(sys.argv[2])
t = sys.argv[3] if len(sys.argv) > 3 else "d"
runs = int(sys.argv[4]) if len(sys.argv) > 4 else 3
do_bt = len(sys.argv) > 5 and sys.argv[5] == "bt"
dt = {"s": np.float32, "d": np.float64, "c": np.complex64, "z": np.complex128}[t]
d.initialize()
g = d.Grid.single()
band = d.get_band_size(nb)
rng = np.random.default_rng(1)
t0 = time.time()
a = np.empty((n, n), dtype=dt, order="F")
for j0 in range(0, n, 1024):  # lower triangle only is referenced: fill column blocks
    blk = rng.uniform(-1, 1, (n, min(1024, n - j0)))
    if dt in (np.complex64, np.complex128):
        blk = blk + 1j * rng.uniform(-1, 1, blk.shape)
    a[:, j0:j0 + blk.shape[1]] = blk
a[np.arange(n), np.arange(n)] = a[np.arange(n), np.arange(n)].real
print(f"generated {n}x{n} {t} in {time.time() - t0:.1f}s", flush=True)
ref = d.DeviceMatrix(g, dt, "L", n, nb)
work = d.DeviceMatrix(g, dt, "L", n, nb)
ref.upload(a)
del a
best = None
for r in range(runs + 1):
    work.copy_from(ref)
    taus = d.reduction_to_band_device(work, band)
    ms, fl = d.red2band_profile()
    tf = fl / ms / 1e9
    print(f"[{r}] reduction_to_band N={n} nb={nb} band={band} type={t}: {ms:.2f} ms  {tf:.2f} TFlop/s", flush=True)
    if r > 0:
        best = max(best or 0, tf)
print(f"RESULT red2band N={n} nb={nb} band={band} type={t} best {best:.2f} TFlop/s")
if do_bt:
    k = n
    c = d.GeneralDeviceMatrix(g, dt, n, k, nb)
    e = np.zeros((n, k), dtype=dt, order="F")
    e[np.arange(n), np.arange(n)] = 1
    for r in range(runs):
        c.upload(e)
        d.bt_reduction_to_band_device(band, c, work, taus)
        ms, fl = d.red2band_profile()
        print(f"[{r}] bt_reduction_to_band N={n} k={k} nb={nb} band={band}: {ms:.2f} ms  {fl / ms / 1e9:.2f} TFlop/s", flush=True)
