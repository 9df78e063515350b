#!/bin/bash
# A/B of the bulk update kernel with and without the C preload (DLAF_UPD_PRELOAD 0 / 1: C loaded into the
# accumulators before the K loop and subtracting MFMAs, against load-subtract-store after it), same box, alternating.
cd ${GRAFT_REPO_ROOT:-.}
OUT=${1:-gpurun_out/ab_preload}
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -DDLAF_UPD_PRELOAD=0 tools/update_bench.hip -o /tmp/ub_pre0
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -DDLAF_UPD_PRELOAD=1 tools/update_bench.hip -o /tmp/ub_pre1
for round in ${AB_ROUNDS:-1 2 3}; do
  for v in pre0 pre1; do
    for args in "48 1024 3 0" "48 1024 3 480" "24 2048 3 480" "64 512 3 480" "32 1024 3 480"; do
      echo "== $v $args (round $round)" | tee -a $OUT/timing.txt
      /tmp/ub_$v $args | grep TFlop | tail -1 | tee -a $OUT/timing.txt
    done
  done
done
