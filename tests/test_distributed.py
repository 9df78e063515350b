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


def launch(mode, nprow, npcol, order, timeout, extra_env=None):
    """Start one worker process per rank directly (no launcher process: the GPU box allows at most 6
    processes on the card, and 3x2 / 2x3 grids need all of them)."""
    n = nprow * npcol
    if mode == "gpu":
        from conftest import gpu_process_budget
        gpu_process_budget(n)
    port = str(free_port())
    procs = []
    for rank in range(n):
        env = dict(os.environ, OMP_NUM_THREADS="1", DLAF_MI355X_DEVICE="0", RANK=str(rank), WORLD_SIZE=str(n),
                   LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), mode, str(nprow),
                                       str(npcol), order], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()  # exact PIDs we started
    rc = [p.returncode for p in procs]
    if os.environ.get("DIST_WORKER_TIMING") == "1":   # (diagnosis: where a worker spends its time)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"dist_timing_{mode}_{nprow}x{npcol}.log"), "w") as fh:
            fh.write("".join(ln + "\n" for ln in outs[0][0].splitlines() if "section" in ln))
    if not (all(r == 0 for r in rc) and "DIST_WORKER_RESULT OK" in outs[0][0]):
        # the whole story of a failing run goes to a file (pytest truncates long assertion messages)
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", f"dist_fail_{mode}_{nprow}x{npcol}.log"), "w") as fh:
                for r, (o, e) in enumerate(outs):
                    fh.write(f"==== rank {r} rc={rc[r]}\n-- stdout\n{o}\n-- stderr\n{e[-6000:]}\n")
        except OSError:
            pass
    assert all(r == 0 for r in rc) and "DIST_WORKER_RESULT OK" in outs[0][0], \
        (rc, outs[0][0][-2000:], "\n".join(o[1][-1500:] for o in outs))


@pytest.mark.parametrize("nprow,npcol,order", [(1, 2, "R"), (2, 1, "C"), (2, 2, "C")])
def test_grid_wiring_and_generation_gloo_cpu(nprow, npcol, order):
    launch("cpu", nprow, npcol, order, timeout=300)


SIX = pytest.mark.many_ranks


@pytest.mark.gpu
@pytest.mark.parametrize("nprow,npcol,order", [(1, 2, "R"), (2, 1, "R"), (2, 2, "C"),
                                               pytest.param(3, 2, "R", marks=SIX), pytest.param(2, 3, "C", marks=SIX)])
def test_distributed_cholesky_one_gpu_many_ranks(nprow, npcol, order):
    # every grid runs the factorization, the solver and the resident solves; the eigensolver stages and gen_to_std are
    # split between the grids with four and six ranks (3 x 2: reduction_to_band / band_to_tridiagonal,
    # 2 x 3: the eigensolvers, 2 x 2: everything but the eigensolvers) -- the two-rank grids run everything
    skip = {(3, 2): "eig,hegst", (2, 3): "red2band,b2t,hegst", (2, 2): "eig"}.get((nprow, npcol), "")
    launch("gpu", nprow, npcol, order, timeout=600, extra_env={"DIST_WORKER_SKIP": skip})


@pytest.mark.gpu
@pytest.mark.parametrize("nprow,npcol,order", [(2, 2, "R"), (1, 3, "R")])
def test_distributed_cholesky_classic_schedule(nprow, npcol, order):
    """Process grids default to the early-diagonal issue order; the classic one stays selectable."""
    launch("gpu", nprow, npcol, order, timeout=600, extra_env={"DLAF_MI355X_SCHEDULE": "classic",
                                                               "DIST_WORKER_CHOLESKY_ONLY": "1"})


@pytest.mark.gpu
@pytest.mark.parametrize("nprow,npcol,order,cholesky_only", [(1, 2, "R", "0"), (2, 2, "C", "1"), (1, 3, "R", "1")])
def test_distributed_peer_copy_transport(nprow, npcol, order, cholesky_only):
    """DLAF_MI355X_TRANSPORT=peer: every broadcast of the executor is a device-to-device copy out of the root's
    hipIpc-mapped memory, ordered by interprocess events (csrc/host/transport_peer.cpp); the control messages travel
    over the gloo callback.  Same worker, same checks as the host-staged runs; the 1 x 2 case runs the widenings too
    (solver, gen_to_std, eigensolver stages: temporaries allocated and freed per call; the whole worker on 2 x 2 passes
    as well, 63 s)."""
    launch("gpu", nprow, npcol, order, timeout=600, extra_env={"DLAF_MI355X_TRANSPORT": "peer",
                                                               "DIST_WORKER_CHOLESKY_ONLY": cholesky_only})


RCCL_SINGLE = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import dla_future_amd as dlaf
from oracle import oracle
dlaf.initialize()
g = dlaf.Grid.rccl(dlaf.Grid.rccl_unique_id(), 1, 0, 1, 1, "R")
assert g.selftest(1 << 20) == 0
g.barrier()
n, nb = 300, 64
a = oracle.set_random_hpd(n, nb, np.float64)
got = a.copy(order="F")
assert dlaf.cholesky_factorization(g, "L", got, nb) == 0
ref = a.copy(order="F")
assert oracle.cholesky_local("L", ref, nb) == 0
tol = 4 * (n + 1) * 2 * oracle.eps_of(np.float64)
assert oracle.check_near(np.tril(ref), np.tril(got), tol, tol)[0]
orig = dlaf.DeviceMatrix(g, np.float64, "L", n, nb); fact = dlaf.DeviceMatrix(g, np.float64, "L", n, nb)
orig.upload(a); fact.copy_from(orig)
assert fact.factorize() == 0
diff, norm_a = orig.residual_against(fact)      # ncclAllReduce(ncclMax) leg of the checker
assert diff / norm_a <= n * oracle.eps_of(np.float64)
orig.close(); fact.close(); g.free(); dlaf.finalize()
print("RCCL_SINGLE OK")
"""


@pytest.mark.gpu
def test_rccl_communicators_on_a_one_process_grid():
    """What a one-GPU box can exercise of the RCCL transport: unique id, ncclCommInitRank, the two
    ncclCommSplit sub-communicators, in-place / out-of-place / grouped ncclBroadcast, the all-reduce
    barrier and the ncclMax reduction, and communicator teardown (the multi-rank call sequence is the
    one the host-transport tests above verify through the same Transport interface)."""
    env = dict(os.environ, DLAF_MI355X_RCCL_SINGLE="1", DLAF_MI355X_DEVICE="0")
    r = subprocess.run([sys.executable, "-c", RCCL_SINGLE % ROOT], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "RCCL_SINGLE OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["host", "peer"])
def test_bench_line_on_a_four_rank_grid_with_the_host_transport(transport):
    """bench.py --gpus 4 as the driver launches it (torch.distributed.run, one rank per process), with the
    host-staged (or the peer-copy) transport so that the four ranks can share this box's one GPU: exercises the
    N > 1 JSON line (grid, transport, MAX-reduced time, device-side residual over the grid)."""
    import json
    from conftest import gpu_process_budget
    gpu_process_budget(4)
    env = dict(os.environ, OMP_NUM_THREADS="1", DLAF_MI355X_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1",
           "--matrix-size", "4096", "--block-size", "256", "--transport", transport, "--check", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 4 and line["config"]["grid"] == "2x2" and line["config"]["transport"] == transport
    assert line["steps"] == 2 and line["value"] > 0 and line["unit"] == "TFlop/s"
    assert line["residual"]["ok"], line["residual"]
    assert "roofline" in line and "cpu_baseline" not in line   # the CPU baseline is an N = 1 item
