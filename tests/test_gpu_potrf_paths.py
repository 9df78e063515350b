"""The two tile-POTRF paths beside a running bulk update, and the expiry path of the cooperative kernel's
bounded waits (kernels_potrf_coop.hip): DLAF_MI355X_POTRF / DLAF_MI355X_POTRF_SPIN_LIMIT are read once per
process, so every case runs in a child process (one at a time)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FACTOR = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import dla_future_amd as d
d.initialize()
g = d.Grid.single()
n, nb = %d, %d
a = np.zeros((n, n), dtype=np.float64, order="F")
d.set_random_hermitian_positive_definite(g, a, n, nb)
orig = d.DeviceMatrix(g, np.float64, "L", n, nb); orig.upload(a)
fact = d.DeviceMatrix(g, np.float64, "L", n, nb); fact.copy_from(orig)
info = fact.factorize()
diff, norm_a = orig.residual_against(fact)
print("RESULT", info, diff / norm_a, n * np.finfo(np.float64).eps, flush=True)
"""


def run_child(code, **env):
    e = dict(os.environ, **env)
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("potrf", ["chain", "coop"])
@pytest.mark.parametrize("schedule", ["pairs", "classic"])
def test_potrf_path_beside_the_bulk_update(potrf, schedule):
    # 16 x 16 tiles of 512: the trailing update of the first steps fills the GPU while POTRF(k+1) runs beside it
    r = run_child(FACTOR % (ROOT, 8192, 512), DLAF_MI355X_POTRF=potrf, DLAF_MI355X_SCHEDULE=schedule)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    _, info, ratio, bar = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1].split()
    assert int(info) == 0 and float(ratio) <= float(bar), r.stdout


def test_expired_wait_is_reported_not_hung():
    """A spin bound of zero makes the first unsatisfied wait of a strip expire: the kernel must drain (every
    workgroup leaves), flag the scheduling failure, and the host must refuse the result loudly."""
    r = run_child(FACTOR % (ROOT, 2048, 512), DLAF_MI355X_POTRF="coop", DLAF_MI355X_POTRF_SPIN_LIMIT="0")
    assert r.returncode != 0, r.stdout
    assert "bounded inter-workgroup wait expired" in r.stderr, r.stderr[-3000:]
    # the device is usable afterwards
    r = run_child(FACTOR % (ROOT, 2048, 512))
    assert r.returncode == 0 and "RESULT 0" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_kphase_alignment_opt_in():
    """DLAF_MI355X_KPHASE=1: every block of the persistent bulk launches starts its k loop at the wall-clock phase
    and wraps around (kernels_update.hip) -- another summation order, the same factor to rounding."""
    r = run_child(FACTOR % (ROOT, 8192, 512), DLAF_MI355X_KPHASE="1")
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    _, info, ratio, bar = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1].split()
    assert int(info) == 0 and float(ratio) <= float(bar), r.stdout
