"""Diagnosis run of the intermittent non-SPD failure on the 2 x 3 grid: the six-rank worker with
the POTRF strips registered on process grids, asynchronous upload copies, the kernel trace on, the non-SPD cases
repeated.  Usage: python tools/diag_nonspd.py [repeat] [extra ENV=VALUE ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import test_distributed as td  # noqa: E402

rep = sys.argv[1] if len(sys.argv) > 1 else "30"
env = {"DLAF_MI355X_POTRF_YIELD_GRIDS": "1", "DLAF_MI355X_POTRF_TRACE": "1", "DIST_WORKER_CHOLESKY_ONLY": "1",
       "DIST_WORKER_NONSPD_REPEAT": rep}
for kv in sys.argv[2:]:
    k, v = kv.split("=", 1)
    env[k] = v
try:
    td.launch("gpu", 2, 3, "C", timeout=900, extra_env=env)
    print("DIAG: worker passed", env)
except AssertionError as e:
    print("DIAG: worker FAILED", env)
    print(str(e)[-6000:])
