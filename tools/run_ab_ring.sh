#!/bin/bash
# Ring depth of the bulk update kernel's direct-to-LDS pipeline (fp64): slab depth BK x stages, same box.
#   base 16x2 (64 KiB) | 8x4 (64 KiB, 3 slabs in flight) | 8x5 (80 KiB, 4 in flight) | 16x3 (96 KiB, one workgroup per CU)
cd ${GRAFT_REPO_ROOT:-.}
OUT=${1:-gpurun_out/ab_ring}
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/update_bench.hip -o /tmp/ub_r162 || exit 1
for v in "8 4" "8 5" "16 3" "8 6"; do
  set -- $v
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -DDLAF_UPD_BK=$1 -DDLAF_UPD_ST=$2 tools/update_bench.hip -o /tmp/ub_r$1$2 || exit 1
done
for round in 1 2; do
  for v in r162 r84 r85 r86 r163; do
    for args in "48 1024 3 480" "24 2048 3 480" "48 1024 3 256" "24 2048 3 256"; do
      echo "== $v $args (round $round)" | tee -a $OUT/timing.txt
      /tmp/ub_$v $args | grep -i "TFlop\|resident" | tail -2 | tr '\n' ' ' | tee -a $OUT/timing.txt; echo | tee -a $OUT/timing.txt
    done
  done
done
