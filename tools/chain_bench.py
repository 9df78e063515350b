"""chain_bench.py -- the per-step dependency chain of the grid order on ONE GPU: the tile POTRF,
the solve of the head tile A(k+1,k) and the update of the next diagonal tile / lookahead column, at stand-alone speed
(a 3 x 3-tile matrix: nothing else runs) and in situ (the same launch classes summed over a large factorization, where
they share the GPU with the bulk update).  The model of tools/scale_model.py takes its chain from these numbers.
usage: chain_bench.py [nb] [N_insitu] [type]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dla_future_amd as d

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_big = int(sys.argv[2]) if len(sys.argv) > 2 else 32 * nb
t = sys.argv[3] if len(sys.argv) > 3 else "d"
dt = {"d": np.float64, "z": np.complex128}[t]
d.initialize()
g = d.Grid.single()


def run(n, sched, reps):
    os.environ["DLAF_MI355X_SCHEDULE"] = sched
    a = np.zeros((n, n), dtype=dt, order="F")
    d.set_random_hermitian_positive_definite(g, a, n, nb)
    orig = d.DeviceMatrix(g, dt, "L", n, nb)
    orig.upload(a)
    del a
    work = d.DeviceMatrix(g, dt, "L", n, nb)
    out = None
    for r in range(reps):
        work.copy_from(orig)
        assert work.factorize() == 0
        out = {k: work.profile(k) for k in ("potrf_tile", "trsm_panel", "update_lookahead", "update_bulk")}
    orig.close()
    work.close()
    return out


print(f"# chain bench nb={nb} type={t}: per-launch ms of the launch classes on the chain", flush=True)
for label, n, sched in (("alone (3x3 tiles, grid order)", 3 * nb, "early"), (f"in situ (N={n_big}, grid order)", n_big, "early"),
                        (f"in situ (N={n_big}, one-process order)", n_big, "pairs")):
    p = run(n, sched, 3)
    row = "  ".join(f"{k} {p[k]['ms'] / max(1, p[k]['launches']):8.3f} ms x{p[k]['launches']:4d}" for k in ("potrf_tile", "trsm_panel", "update_lookahead"))
    print(f"{label:44s} {row}", flush=True)
    if n == 3 * nb:
        chain = sum(p[k]["ms"] / max(1, p[k]["launches"]) for k in ("potrf_tile", "trsm_panel", "update_lookahead"))
        print(f"{'':44s} chain POTRF + head TRSM + next-column update, stand-alone: {chain:.3f} ms", flush=True)
