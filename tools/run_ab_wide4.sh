#!/bin/bash
# Round 3 A/B of the bulk update kernel (tools/update_bench.hip): the production block (128 x 128, two workgroups per CU),
# its interior-only ("lean") instantiation, and the 256 x 128 four-wave block of DESIGN section 8.2 (one wave per SIMD)
#   tools/run_ab_wide4.sh <outdir>
set -e
cd ${GRAFT_REPO_ROOT:-.}
OUT=${1:-gpurun_out/ab_wide4}
mkdir -p $OUT
VARIANTS=${VARIANTS:-"base:-DBASE lean:-DDLAF_UPD_LEAN wide4_s2:-DDLAF_UPD_LEAN,-DDLAF_UPD_WIDE4=2 wide4_s3:-DDLAF_UPD_LEAN,-DDLAF_UPD_WIDE4=3"}
for v in $VARIANTS; do
  name=${v%%:*}
  flags=$(echo ${v#*:} | tr ',' ' ')
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include $flags tools/update_bench.hip -o /tmp/ub_$name
  for args in "48 1024 3 0" "48 1024 3 480" "48 1024 3 256" "64 512 3 480"; do
    echo "== $name ($flags) nt nb reps max_blocks = $args" | tee -a $OUT/timing.txt
    /tmp/ub_$name $args | tail -2 | tee -a $OUT/timing.txt
  done
done
