"""Diagnosis: the N = 4096 eigensolver on a 2 x 2 grid, several times, under different settings (env as KEY=VALUE args)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import test_distributed as td  # noqa: E402

env = {"DIST_WORKER_ONLY": "eigbig", "DIST_WORKER_REPEAT": "3", "DIST_WORKER_TIMING": "0"}
for kv in sys.argv[1:]:
    k, v = kv.split("=", 1)
    env[k] = v
import subprocess
port = str(td.free_port())
procs = []
for rank in range(4):
    e = dict(os.environ, OMP_NUM_THREADS="1", DLAF_MI355X_DEVICE="0", RANK=str(rank), WORLD_SIZE="4", LOCAL_RANK="0",
             MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    e.update(env)
    procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), "gpu", "2", "2", "C"], cwd=ROOT, env=e,
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
outs = [p.communicate(timeout=500) for p in procs]
print(" ".join(sys.argv[1:]) or "(default)")
print("".join(ln + "\n" for ln in outs[0][0].splitlines() if "eigbig" in ln), end="")
import re
for r, (o, e) in enumerate(outs):
    for ln in e.splitlines():
        if "eig debug" in ln:
            print(ln)
    if procs[r].returncode != 0:
        print("rank", r, "rc", procs[r].returncode, e[-800:])
