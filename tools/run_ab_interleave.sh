#!/bin/bash
# A/B of the two K-loop forms of the direct-to-LDS pipeline (DLAF_GLDS_INTERLEAVE 0 / 1) in one process per
# variant, several regimes: full grid / persistent grid, K = 1024 / 2048.
cd ${GRAFT_REPO_ROOT:-.}
OUT=${1:-gpurun_out/ab_interleave}
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -DDLAF_GLDS_INTERLEAVE=0 tools/update_bench.hip -o /tmp/ub_old
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -DDLAF_GLDS_INTERLEAVE=1 tools/update_bench.hip -o /tmp/ub_new
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -DDLAF_GLDS_INTERLEAVE=1 -DDLAF_DBG_STRIP_PACKED tools/update_bench.hip -o /tmp/ub_packed
for round in ${AB_ROUNDS:-1 2}; do
  for v in ${AB_VARIANTS:-old new packed}; do
    for args in "48 1024 3 0" "48 1024 3 480" "24 2048 3 480" "64 512 3 480"; do
      echo "== $v $args (round $round)" | tee -a $OUT/timing.txt
      /tmp/ub_$v $args | grep TFlop | tail -2 | tee -a $OUT/timing.txt
    done
  done
done
