#!/usr/bin/env python3
"""gen_golden.py -- TEST INFRASTRUCTURE ONLY: regenerates tests/golden/*.json.

Run in the build container (needs g++ for the <random> probe and scipy for LAPACK):
    python oracle/gen_golden.py

Fixtures written (all DATA: inputs + expected outputs, no reference source text):
  rng_stream.json        libstdc++ mt19937_64 + uniform_real_distribution(-1,1) streams for a few
                         seeds, real and complex (g++ argument order), produced by oracle/rng_probe.cpp
                         -- the generator recipe of include/dlaf/util_matrix.h:148-179.
  distribution_rows.json the 32 known-answer rows that test/unit/matrix/test_util_distribution.cpp:45-62
                         holds for util_distribution.h (numbers only).
  potrf_lapack.json      LAPACK ?potrf (scipy/OpenBLAS) factors of the oracle's random HPD matrix,
                         n=34 nb=13, s/d/c/z, L and U: an independent numeric answer for the same
                         input the HIP path and the oracle are fed.
  hpd_34_13.json         the first column and the diagonal of that input (pins the generator itself).
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def rng_stream():
    subprocess.check_call(["make", "-C", HERE, "rng_probe"], stdout=subprocess.DEVNULL)
    out = {}
    for seed in (0, 5, 13, 34 * 13 + 26, 4096 * 256 + 512):
        txt = subprocess.check_output([os.path.join(HERE, "_ref", "rng_probe"), str(seed), "16"], text=True)
        rec = {}
        for line in txt.strip().splitlines():
            key, *vals = line.split()
            rec[key] = [int(v) for v in vals] if key == "raw" else [float(v) for v in vals]
        out[str(seed)] = rec
    return out


# test/unit/matrix/test_util_distribution.cpp:45-62 -- the numeric rows only.
# columns: tile_size tiles_per_block rank grid_size src_rank global_element global_tile rank_tile
#          local_tile local_tile_next tile_element tile_offset tile_element_offset
DIST_ROWS = [
    [10, 1, 0, 1, 0, 31, 3, 0, 3, 3, 1, 0, 0], [10, 1, 0, 5, 0, 102, 10, 0, 2, 2, 2, 0, 0],
    [10, 1, 1, 5, 0, 124, 12, 2, -1, 3, 4, 0, 0], [10, 1, 4, 5, 3, 124, 12, 0, -1, 3, 4, 0, 0],
    [25, 1, 0, 1, 0, 231, 9, 0, 9, 9, 6, 0, 0], [25, 1, 0, 5, 0, 102, 4, 4, -1, 1, 2, 0, 0],
    [25, 1, 3, 5, 4, 102, 4, 3, 0, 0, 2, 0, 0], [25, 1, 4, 5, 3, 0, 0, 3, -1, 0, 0, 0, 0],
    [25, 1, 0, 5, 3, 0, 0, 3, -1, 0, 0, 0, 0], [25, 1, 3, 5, 3, 0, 0, 3, 0, 0, 0, 0, 0],
    [10, 3, 0, 1, 0, 31, 3, 0, 3, 3, 1, 0, 0], [10, 2, 0, 5, 0, 102, 10, 0, 2, 2, 2, 0, 0],
    [10, 4, 1, 5, 0, 124, 12, 3, -1, 4, 4, 0, 0], [10, 4, 4, 5, 3, 124, 12, 1, -1, 4, 4, 0, 0],
    [25, 5, 0, 1, 0, 231, 9, 0, 9, 9, 6, 0, 0], [25, 4, 0, 5, 0, 652, 26, 1, -1, 8, 2, 0, 0],
    [25, 4, 1, 5, 0, 652, 26, 1, 6, 6, 2, 0, 0], [25, 4, 2, 5, 0, 652, 26, 1, -1, 4, 2, 0, 0],
    [25, 3, 3, 5, 2, 102, 4, 3, 1, 1, 2, 0, 0], [25, 3, 4, 5, 3, 0, 0, 3, -1, 0, 0, 0, 0],
    [25, 2, 0, 5, 3, 0, 0, 3, -1, 0, 0, 0, 0], [25, 2, 3, 5, 3, 0, 0, 3, 0, 0, 0, 0, 0],
    [10, 1, 0, 1, 0, 31, 3, 0, 3, 3, 7, 0, 6], [10, 1, 0, 5, 0, 98, 10, 0, 2, 2, 1, 0, 3],
    [25, 1, 0, 1, 0, 224, 9, 0, 9, 9, 6, 0, 7], [25, 1, 0, 5, 0, 102, 4, 4, -1, 1, 24, 0, 22],
    [10, 3, 0, 1, 0, 21, 2, 0, 2, 2, 1, 1, 0], [10, 2, 0, 5, 0, 88, 9, 0, 1, 1, 2, 1, 4],
    [10, 4, 1, 5, 0, 102, 10, 3, -1, 4, 4, 2, 2], [10, 4, 4, 5, 3, 94, 9, 1, -1, 4, 4, 3, 0],
    [25, 4, 1, 5, 0, 582, 24, 1, 6, 6, 2, 2, 20], [25, 4, 2, 5, 0, 553, 23, 1, -1, 4, 2, 3, 24],
]
DIST_COLS = ["tile_size", "tiles_per_block", "rank", "grid_size", "src_rank", "global_element", "global_tile",
             "rank_tile", "local_tile", "local_tile_next", "tile_element", "tile_offset", "tile_element_offset"]


def enc(a):
    a = np.asarray(a)
    if np.iscomplexobj(a):
        return {"re": a.real.astype(np.float64).ravel(order="F").tolist(),
                "im": a.imag.astype(np.float64).ravel(order="F").tolist(), "shape": list(a.shape)}
    return {"re": a.astype(np.float64).ravel(order="F").tolist(), "shape": list(a.shape)}


def potrf_lapack():
    from scipy.linalg import lapack
    from oracle import oracle
    n, nb = 34, 13
    out, hpd = {}, {}
    for t, dt in oracle.DTYPES.items():
        a = oracle.set_random_hpd(n, nb, dt)
        hpd[t] = {"col0": enc(a[:, 0]), "diag": enc(np.diag(a)), "row_last": enc(a[n - 1, :])}
        fn = getattr(lapack, f"{t}potrf")
        for uplo in "LU":
            c, info = fn(a, lower=(uplo == "L"), clean=False, overwrite_a=False)
            assert info == 0
            out[f"{t}{uplo}"] = enc(np.tril(c) if uplo == "L" else np.triu(c))
    return out, hpd


def main():
    os.makedirs(GOLD, exist_ok=True)
    with open(os.path.join(GOLD, "rng_stream.json"), "w") as f:
        json.dump(rng_stream(), f, indent=0)
    with open(os.path.join(GOLD, "distribution_rows.json"), "w") as f:
        json.dump({"columns": DIST_COLS, "rows": DIST_ROWS,
                   "source": "test/unit/matrix/test_util_distribution.cpp:45-62"}, f)
    lap, hpd = potrf_lapack()
    with open(os.path.join(GOLD, "potrf_lapack.json"), "w") as f:
        json.dump({"n": 34, "nb": 13, "factors": lap}, f)
    with open(os.path.join(GOLD, "hpd_34_13.json"), "w") as f:
        json.dump({"n": 34, "nb": 13, "samples": hpd}, f)
    print("golden fixtures written to", GOLD)


if __name__ == "__main__":
    main()
