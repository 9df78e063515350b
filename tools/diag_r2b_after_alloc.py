"""Diagnosis: is reduction_to_band slower on device memory that was allocated after 137 GB of other allocations were freed
(what bench.py's process does) than on fresh memory?   usage: diag_r2b_after_alloc.py [first]
first: the two 3.4 GB matrices of the measurement are allocated BEFORE the big ones (and used after their release)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dla_future_amd as dlaf

first = len(sys.argv) > 1 and sys.argv[1] == "first"
n, nb = 20480, 512
dlaf.initialize()
grid = dlaf.Grid.single()
band = dlaf.get_band_size(nb)
a = np.zeros((n, n), dtype=np.float64, order="F")
dlaf.set_random_hermitian_positive_definite(grid, a, n, nb)
a[np.arange(n), np.arange(n)] -= 2.0 * n


def measure(ref, work, tag):
    ref.upload(a)
    for r in range(3):
        work.copy_from(ref)
        dlaf.reduction_to_band_device(work, band)
        ms, flops = dlaf.red2band_profile()
        print(f"{tag} [{r}] {ms:.1f} ms {flops / ms / 1e9:.2f} TFlop/s", flush=True)


early = (dlaf.DeviceMatrix(grid, np.float64, "L", n, nb), dlaf.DeviceMatrix(grid, np.float64, "L", n, nb)) if first else None
if not first:
    r0, w0 = dlaf.DeviceMatrix(grid, np.float64, "L", n, nb), dlaf.DeviceMatrix(grid, np.float64, "L", n, nb)
    measure(r0, w0, "fresh process")
    r0.close()
    w0.close()
big = [dlaf.DeviceMatrix(grid, np.float64, "L", 65536, 1024) for _ in range(4)]
for b in big:
    b.close()
print("4 x 34 GB allocated and released", flush=True)
if first:
    measure(early[0], early[1], "allocated before the big ones")
else:
    r1, w1 = dlaf.DeviceMatrix(grid, np.float64, "L", n, nb), dlaf.DeviceMatrix(grid, np.float64, "L", n, nb)
    measure(r1, w1, "allocated after their release")
