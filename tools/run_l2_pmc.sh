#!/bin/bash
# L2 behaviour of the bulk update kernel (stand-alone, tools/update_bench.hip): hit / miss requests and fabric-side
# fetch bytes, persistent grid (480 workgroups) and full grid.  Counters in their own passes (rocprofv3 --pmc).
cd ${GRAFT_REPO_ROOT:-.}
OUT=$GRAFT_REPO_ROOT/${1:-gpurun_out/l2pmc}
mkdir -p $OUT
FLAGS=${UB_FLAGS:-}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include $FLAGS tools/update_bench.hip -o /tmp/ub_l2
cd /tmp && export TMPDIR=/tmp
for mode in 480 0; do
  for ctr in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $ctr | tr ' ' '_')
    rm -rf /tmp/l2_$mode_$tag
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/l2_${mode}_$tag -- /tmp/ub_l2 48 1024 1 $mode > $OUT/run_${mode}_$tag.log 2>&1
    echo "## max_blocks=$mode counters: $ctr" | tee -a $OUT/summary.txt
    python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/l2_${mode}_$tag update_kernel | tee -a $OUT/summary.txt
  done
done
