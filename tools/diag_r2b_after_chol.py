"""Diagnosis: why is bench.py's reduction_to_band line slower (27 TFlop/s) than tools/red2band_bench.py (32) and than the same
stage inside its eigensolver line?  The bench's measurement repeated in one process: fresh, after a Cholesky of N (argv[1], 0 =
none), with the bench's matrix.   usage: diag_r2b_after_chol.py CHOL_N [CHOL_NB]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dla_future_amd as dlaf
import bench

chol_n = int(sys.argv[1]) if len(sys.argv) > 1 else 0
chol_nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
if os.environ.get("DIAG_TORCH_CUDA") == "1":   # what bench.py's main does before it touches the library
    import torch
    torch.cuda.set_device(0)
    print("torch.cuda:", torch.cuda.mem_get_info(), flush=True)
    torch.cuda.synchronize()
dlaf.initialize()
grid = dlaf.Grid.single()
print("fresh:", {k: v for k, v in bench.red2band_line(dlaf, grid, 20480, 512).items() if k in ("value", "ms")},
      "panels (blocked, fallback)", dlaf.red2band_panel_stats(), flush=True)
if chol_n:
    a = np.zeros((chol_n, chol_n), dtype=np.float64, order="F")
    dlaf.set_random_hermitian_positive_definite(grid, a, chol_n, chol_nb)
    ref = dlaf.DeviceMatrix(grid, np.float64, "L", chol_n, chol_nb)
    work = dlaf.DeviceMatrix(grid, np.float64, "L", chol_n, chol_nb)
    ref.upload(a)
    del a
    for r in range(3):
        work.copy_from(ref)
        assert work.factorize() == 0
    print("cholesky done; update launch stats", dlaf.update_launch_stats(), flush=True)
    work.close()
    ref.close()
    for r in range(2):
        print(f"after the Cholesky [{r}]:", {k: v for k, v in bench.red2band_line(dlaf, grid, 20480, 512).items() if k in ("value", "ms")}, flush=True)
