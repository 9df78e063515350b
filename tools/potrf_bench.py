#!/usr/bin/env python3
"""Tile POTRF alone on the device: a one-tile resident matrix (n = nb) factorized in a loop; the time per
factorization is the cooperative POTRF launch plus a few microseconds of launch overhead.
   python tools/potrf_bench.py [reps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dla_future_amd as dlaf  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dlaf.initialize()
    g = dlaf.Grid.single()
    for dt, nb in ((np.float64, 512), (np.float64, 1024), (np.complex128, 512), (np.complex128, 1024), (np.float32, 1024)):
        a = np.zeros((nb, nb), dtype=dt, order="F")
        dlaf.set_random_hermitian_positive_definite(g, a, nb, nb)
        orig = dlaf.DeviceMatrix(g, dt, "L", nb, nb)
        fact = dlaf.DeviceMatrix(g, dt, "L", nb, nb)
        orig.upload(a)
        best = 1e9
        copy = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            fact.copy_from(orig)
            t1 = time.perf_counter()
            assert fact.factorize() == 0
            t2 = time.perf_counter()
            best = min(best, t2 - t1)
            copy = min(copy, t1 - t0)
        print(f"{np.dtype(dt).name:10s} nb={nb}: factorize {best * 1e6:8.1f} us  ({best * 1e6 / (nb // 64):6.1f} us per 64-column step)   copy {copy * 1e6:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
