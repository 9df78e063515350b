#!/usr/bin/env python3
"""Triangular-solver timing on one GPU: device time of the sweep (dlaf_mi355x_solver_profile) for a few
variants.   python tools/trsm_bench.py [m] [n] [nb]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dla_future_amd as dlaf  # noqa: E402


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    dlaf.initialize()
    g = dlaf.Grid.single()
    rng = np.random.default_rng(1)
    for side, uplo, op in [("R", "L", "C"), ("L", "L", "N"), ("L", "L", "C"), ("R", "U", "N")]:
        na = m if side == "L" else n
        a = np.asfortranarray(rng.uniform(-1, 1, (na, na)) / na + 2 * np.eye(na))
        b = np.asfortranarray(rng.uniform(-1, 1, (m, n)))
        for rep in range(2):
            x = b.copy(order="F")
            t0 = time.perf_counter()
            dlaf.triangular_solver(g, side, uplo, op, "N", 1.0, a, x, nb)
            wall = time.perf_counter() - t0
            ms, fl = dlaf.solver_profile()
        # residual of the last solve on a slice (full check lives in the tests)
        tri = np.tril(a) if uplo == "L" else np.triu(a)
        opa = tri if op == "N" else tri.T
        r = (opa[:256, :] @ x - b[:256, :]) if side == "L" else (x[:256, :] @ opa - b[:256, :])
        print(f"{side}{uplo}{op} m={m} n={n} nb={nb}: sweep {ms:.2f} ms  {fl / ms / 1e9:.2f} TFlop/s   "
              f"(wall incl. PCIe + relayout {wall * 1e3:.0f} ms)  resid {np.abs(r).max():.2e}", flush=True)


if __name__ == "__main__":
    main()
