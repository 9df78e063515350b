#!/bin/bash
# What the per-slab workgroup barrier of the bulk update kernel costs: the kernel as it is against a build without it
# (-DDLAF_DBG_NO_SLAB_BARRIER: timing only, the waves race on the LDS ring), with and without in-loop loads.
cd ${GRAFT_REPO_ROOT:-.}
OUT=${1:-gpurun_out/ab_barrier}
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/update_bench.hip -o /tmp/ub_b_base || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -DDLAF_DBG_NO_SLAB_BARRIER tools/update_bench.hip -o /tmp/ub_b_nobar || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -DDLAF_DBG_NO_SLAB_BARRIER -DDLAF_DBG_SKIP_GLOBAL tools/update_bench.hip -o /tmp/ub_b_nobar_noglobal || exit 1
for round in 1 2; do
  for v in b_base b_nobar b_nobar_noglobal; do
    for args in "48 1024 3 480" "24 2048 3 480" "48 1024 3 256"; do
      echo "== $v $args (round $round)" | tee -a $OUT/timing.txt
      /tmp/ub_$v $args 2>&1 | grep -i "TFlop\|error\|fault" | tail -1 | tee -a $OUT/timing.txt
    done
  done
done
