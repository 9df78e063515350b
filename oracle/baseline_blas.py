"""baseline_blas.py -- TEST/BENCH INFRASTRUCTURE ONLY: the reference's CPU path as it is actually
deployed, restated with vendor BLAS tile kernels.

The reference's MC backend runs the right-looking tile DAG (cholesky/impl.h:150-189) as pika tasks, each
tile task calling single-threaded BLAS/LAPACK through blaspp/lapackpp (blas/tile.h:296-366,
lapack/tile.h:362-378; scripts/miniapps.py:219 sets OMP_NUM_THREADS=1).  Here: the same DAG over
column-of-tiles storage, one tile task per host thread, tile kernels = the CPU BLAS/LAPACK that ships
with PyTorch (MKL/OpenBLAS; single-threaded per call, GIL released), phases separated by joins (no
lookahead).  Used by bench.py's cpu_baseline leg only.
"""
from __future__ import annotations

import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def tiled_cholesky_blas(a: np.ndarray, nb: int, threads: int):
    """Lower Cholesky of the SPD matrix a (numpy, any order) with nb x nb tiles.
    Returns (seconds, tiles) where tiles[i][j] (i >= j) hold the factor as torch tensors."""
    import torch
    torch.set_num_threads(1)  # single-threaded BLAS per tile task, parallelism across tiles
    n = a.shape[0]
    nt = (n + nb - 1) // nb
    t = [[torch.from_numpy(np.ascontiguousarray(a[i * nb:(i + 1) * nb, j * nb:(j + 1) * nb])) if i >= j else None
          for j in range(nt)] for i in range(nt)]

    def potrf(k):
        t[k][k] = torch.linalg.cholesky(t[k][k])

    def trsm(i, k):
        # X L^T = B  (Right, Lower, ConjTrans, NonUnit)
        t[i][k] = torch.linalg.solve_triangular(t[k][k].T, t[i][k], upper=True, left=False)

    def update(i, j, k):
        # herk on the diagonal (full-tile product, only the lower part is ever used), gemm below it
        t[i][j].addmm_(t[i][k], t[j][k].T, alpha=-1.0)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as pool:
        for k in range(nt):
            potrf(k)
            list(pool.map(lambda i: trsm(i, k), range(k + 1, nt)))
            jobs = [(i, j) for j in range(k + 1, nt) for i in range(j, nt)]
            list(pool.map(lambda ij: update(ij[0], ij[1], k), jobs))
    return time.perf_counter() - t0, t


def assemble_lower(tiles, n: int, nb: int) -> np.ndarray:
    out = np.zeros((n, n))
    nt = len(tiles)
    for i in range(nt):
        for j in range(i + 1):
            blk = tiles[i][j].numpy()
            out[i * nb:i * nb + blk.shape[0], j * nb:j * nb + blk.shape[1]] = np.tril(blk) if i == j else blk
    return out
