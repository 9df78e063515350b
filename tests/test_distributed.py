"""Multi-process tests of the N > 1 path.

CPU (gloo, world_size 2 and 4): process-grid wiring, host-broadcast callback through the C ABI,
block-cyclic generation per rank.  GPU (marked gpu): up to 6 processes share the one GPU and run the
real distributed executor (panel / transposed-panel broadcasts, lookahead order) with the host-staged
transport over gloo -- the reference tests the same way with 6 oversubscribed MPI ranks on one machine
(test/include/dlaf_test/comm_grids/grids_6_ranks.h:26-71).  RCCL itself needs one GPU per rank and is
exercised by bench.py on the multi-GPU node."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(mode, nprow, npcol, order, timeout):
    n = nprow * npcol
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), mode,
           str(nprow), str(npcol), order]
    env = dict(os.environ, OMP_NUM_THREADS="1", DLAF_MI355X_DEVICE="0")
    env.pop("LOCAL_RANK", None)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0 and "DIST_WORKER_RESULT OK" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


@pytest.mark.parametrize("nprow,npcol,order", [(1, 2, "R"), (2, 1, "C"), (2, 2, "C")])
def test_grid_wiring_and_generation_gloo_cpu(nprow, npcol, order):
    launch("cpu", nprow, npcol, order, timeout=300)


@pytest.mark.gpu
@pytest.mark.parametrize("nprow,npcol,order", [(1, 2, "R"), (2, 1, "R"), (2, 2, "C"), (3, 2, "R"), (2, 3, "C")])
def test_distributed_cholesky_one_gpu_many_ranks(nprow, npcol, order):
    launch("gpu", nprow, npcol, order, timeout=600)
