#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry dlaf_pdpotrf (upload + relayout + factorization + download),
for DESIGN.md; the bench value is the device-resident rate.   python tools/pcie_inclusive.py [n] [nb]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dla_future_amd as dlaf  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dlaf.initialize()
g = dlaf.Grid.single()
a = np.zeros((n, n), order="F")
dlaf.set_random_hermitian_positive_definite(g, a, n, nb)
desca = [1, g.context, n, n, nb, nb, 0, 0, n]
for rep in range(3):
    w = a.copy(order="F")
    t0 = time.perf_counter()
    info = dlaf.pxpotrf("L", n, w, 1, 1, desca)
    dt = time.perf_counter() - t0
    print(f"dlaf_pdpotrf N={n} nb={nb}: {dt:.3f} s  {n ** 3 / 3 / dt / 1e12:.2f} TFlop/s PCIe-inclusive (info {info}), "
          f"{2 * n * n * 8 / 2 / dt / 1e9:.1f} GB/s of triangle traffic if it were all transfer", flush=True)
