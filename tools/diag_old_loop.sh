#!/bin/bash
# Diagnosis: N runs of the six-rank 2 x 3 worker in a given source tree with the POTRF strips
# registered on process grids; prints pass / fail per run.  usage: tools/diag_old_loop.sh <tree> <runs> <label>
tree=$1; runs=$2; label=$3
out=$GRAFT_REPO_ROOT/gpurun_out/diag_loop_$label.txt
: > $out
cd $tree
fails=0
for i in $(seq 1 $runs); do
  if DLAF_MI355X_POTRF_YIELD_GRIDS=1 DLAF_MI355X_INFO_VERBOSE=0 DIST_WORKER_CHOLESKY_ONLY=1 timeout -k 10 300 \
     python -m pytest tests/test_distributed.py -x -q -m gpu -k "test_distributed_cholesky_one_gpu_many_ranks and 2-3" > /tmp/run_$i.log 2>&1; then
    echo "run $i: pass" >> $out
  else
    fails=$((fails+1))
    echo "run $i: FAIL" >> $out
    grep -h "FAILED\|returned" /tmp/run_$i.log gpurun_out/dist_fail_gpu_2x3.log 2>/dev/null | sort | uniq -c | head -20 >> $out
  fi
done
echo "$label: $fails failures in $runs runs" >> $out
cat $out
