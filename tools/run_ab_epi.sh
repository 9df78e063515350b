#!/bin/bash
# A/B: accumulator columns per load-subtract-store round trip of the bulk update kernel's epilogue (DLAF_EPI_COLS)
cd ${GRAFT_REPO_ROOT:-.}
OUT=${1:-gpurun_out/ab_epi}
mkdir -p $OUT
for n in 1 2 4 8; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -DDLAF_EPI_COLS=$n tools/update_bench.hip -o /tmp/ub_epi$n || exit 1
done
for round in 1 2; do
  for n in 1 2 4 8; do
    for args in "48 1024 3 480" "24 2048 3 480" "64 512 3 480"; do
      echo "== epi_cols=$n $args (round $round)" | tee -a $OUT/timing.txt
      /tmp/ub_epi$n $args 2>&1 | grep -i "TFlop\|error\|fault" | tail -1 | tee -a $OUT/timing.txt
    done
  done
done
