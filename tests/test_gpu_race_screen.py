"""Race screen (round 4): the deterministic algorithms must repeat bit for bit from run to run ALSO while the kernels of several
processes share the GPU.  That condition exposed a latent write-after-read race on an LDS ring of the fused
bt_band_to_tridiagonal kernel (profiles/r04_bt_apply_race.txt): invisible on one process, wrong eigenvectors in ~10 % of runs
with four.  Four one-process copies of tools/diag_eig1.py / tools/diag_chol1.py at once; every run of every copy must give
the same bits (and, for the eigensolver, pass the reference's correctness bars)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.fresh_parent]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_copies(script, args, copies, timeout):
    from conftest import gpu_process_budget
    gpu_process_budget(copies)
    env = dict(os.environ, OMP_NUM_THREADS="1", DLAF_MI355X_DEVICE="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", script)] + [str(a) for a in args], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(copies)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), [o[1][-800:] for o in outs]
    return [ln for o in outs for ln in o[0].splitlines() if ln.startswith("pid ")]


def test_eigensolver_repeats_bit_for_bit_beside_three_other_processes():
    lines = run_copies("diag_eig1.py", [4096, 4], 4, 600)
    assert len(lines) == 4, lines
    hashes = set(re.findall(r"\('([0-9a-f]{8})'", "".join(lines)))
    orth = [float(x) for x in re.findall(r", '([0-9.e+-]+)'\)", "".join(lines))]
    assert len(hashes) == 1, lines
    assert len(orth) == 16 and max(orth) <= 10 * 4096 * 2 * 2.3e-16, lines


def test_cholesky_repeats_bit_for_bit_beside_three_other_processes():
    for args in ([4096, 256, "d", 4], [3072, 256, "z", 3]):
        lines = run_copies("diag_chol1.py", args, 4, 600)
        assert len(lines) == 4 and all("DETERMINISTIC" in ln for ln in lines), lines
        assert len({ln.split()[-1] for ln in lines}) == 1, lines
