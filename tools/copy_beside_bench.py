"""copy_beside_bench.py -- the factorization beside a saturating device-to-device copy stream (the data path of the
peer-copy transport and of RCCL's broadcasts: fabric traffic that competes with the bulk update kernel's 4.4 x re-read
traffic).  Reports the factorization's TFlop/s alone and beside the copies, and the copies' GB/s alone
and beside the factorization.  Run once per DLAF_MI355X_KPHASE setting (the library reads it once).
usage: copy_beside_bench.py [N=32768] [nb=1024] [reps=3]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dla_future_amd as d

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
d.initialize()
g = d.Grid.single()
a = np.zeros((n, n), dtype=np.float64, order="F")
d.set_random_hermitian_positive_definite(g, a, n, nb)
orig = d.DeviceMatrix(g, np.float64, "L", n, nb)
orig.upload(a)
del a
work = d.DeviceMatrix(g, np.float64, "L", n, nb)
src = d.DeviceMatrix(g, np.float64, "L", n, nb)
dst = d.DeviceMatrix(g, np.float64, "L", n, nb)
src.copy_from(orig)
tile_bytes = float(n) * n * 8           # a copy moves the whole local tile array: read + write
flops = n ** 3 / 3.0


def factor_rate():
    best = 0.0
    for _ in range(reps):
        work.copy_from(orig)
        t0 = time.time()
        assert work.factorize() == 0
        best = max(best, flops / (time.time() - t0) / 1e12)
    return best


def copy_rate(seconds=1.5):
    k, t0 = 0, time.time()
    while time.time() - t0 < seconds:
        dst.copy_from(src)
        k += 1
    return k * tile_bytes / (time.time() - t0) / 1e9


stop = threading.Event()
count = [0, 0.0]


def copier():
    t0 = time.time()
    while not stop.is_set():
        dst.copy_from(src)
        count[0] += 1
    count[1] = time.time() - t0


kp = os.environ.get("DLAF_MI355X_KPHASE", "0")
alone = factor_rate()
calone = copy_rate()
th = threading.Thread(target=copier)
th.start()
time.sleep(0.2)
beside = factor_rate()
stop.set()
th.join()
cbeside = count[0] * tile_bytes / count[1] / 1e9
print(f"KPHASE={kp} N={n} nb={nb}: factorization alone {alone:.2f} TFlop/s, beside the copy stream {beside:.2f}; "
      f"copies alone {calone:.0f} GB/s (payload; x2 for read + write), beside the factorization {cbeside:.0f} GB/s", flush=True)
