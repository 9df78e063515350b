#!/bin/bash
# four copies of a diag tool at once, with their timing lines (where do the race-screen tests spend their time?)
export OMP_NUM_THREADS=1 DLAF_MI355X_DEVICE=0
for spec in "diag_chol1.py 4096 256 d 4" "diag_eig1.py 4096 4"; do
  echo "== $spec"; date +%s.%N
  for i in 1 2 3 4; do timeout -k 10 500 python tools/$spec > gpurun_out/race_$i.log 2>&1 & done
  wait
  date +%s.%N
  grep -h "^time\|DETERMINISTIC\|DIFFERS" gpurun_out/race_[1-4].log | cut -c1-160
done
echo "== one copy alone"; date +%s.%N; timeout -k 10 300 python tools/diag_chol1.py 4096 256 d 4 2>&1 | tail -2; date +%s.%N
