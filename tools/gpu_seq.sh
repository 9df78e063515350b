#!/bin/bash
# Runs the given shell steps one after another on the GPU box (one argument = one step, run under bash -c);
# a step that times out or is killed (rc 124 / 137 / 143) ends the sequence: no further GPU step is started
# after a hang.  Other failures (a failing test) are recorded and the sequence goes on.
#   tools/gpu_seq.sh 'timeout -k 10 600 python -m pytest ...' 'timeout -k 10 200 python bench.py ...'
cd ${GRAFT_REPO_ROOT:-.}
n=0
for step in "$@"; do
  n=$((n + 1))
  echo "[gpu_seq] step $n: $step"
  bash -o pipefail -c "$step"
  rc=$?
  echo "[gpu_seq] step $n rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 143 ]; then
    echo "[gpu_seq] step $n timed out or was killed: stopping here"
    exit $rc
  fi
done
exit 0
