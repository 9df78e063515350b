#!/bin/bash
# A/B: address arithmetic of the in-loop direct-to-LDS loads of the bulk update kernel -- per lane (64-bit
# multiply-add + select per load, rounds 1-2: -DDLAF_GLDS_SCALAR_ADDR=0) against scalar base + lane offset.
cd ${GRAFT_REPO_ROOT:-.}
OUT=${1:-gpurun_out/ab_saddr}
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -DDLAF_GLDS_SCALAR_ADDR=0 tools/update_bench.hip -o /tmp/ub_sa0 || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -DDLAF_GLDS_SCALAR_ADDR=1 tools/update_bench.hip -o /tmp/ub_sa1 || exit 1
for round in 1 2 3; do
  for v in sa0 sa1; do
    for args in "48 1024 3 480" "24 2048 3 480" "64 512 3 480" "48 1024 3 0" "48 1024 3 256"; do
      echo "== $v $args (round $round)" | tee -a $OUT/timing.txt
      /tmp/ub_$v $args 2>&1 | grep -i "TFlop\|error\|fault" | tail -1 | tee -a $OUT/timing.txt
    done
  done
done
