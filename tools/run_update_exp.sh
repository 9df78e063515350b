#!/bin/bash
# Tuning aid (round 2): the standalone bulk-update kernel (tools/update_bench.hip) in several block
# configurations, timed and -- with PMC=1 -- under rocprofv3 counters (MFMA busy cycles, effective clock,
# wait cycles), to separate "the MFMA pipe waits" from "the chip holds its clock down".
#   VARIANTS="base:-DBASE big3:-DDLAF_UPD_BIG=3" PMC=1 tools/run_update_exp.sh <outdir>
set -e
cd ${GRAFT_REPO_ROOT:-.}
OUT=${1:-gpurun_out/update_exp}
mkdir -p $OUT
VARIANTS=${VARIANTS:-"base:-DBASE base_noglobal:-DDLAF_DBG_SKIP_GLOBAL big3:-DDLAF_UPD_BIG=3 big2:-DDLAF_UPD_BIG=2 big3_noglobal:-DDLAF_UPD_BIG=3,-DDLAF_DBG_SKIP_GLOBAL"}
SHAPES=${SHAPES:-"48,1024 64,512"}
for v in $VARIANTS; do
  name=${v%%:*}
  flags=$(echo ${v#*:} | tr ',' ' ')
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include $flags tools/update_bench.hip -o /tmp/ub_$name
  for sh in $SHAPES; do
    nt=${sh%%,*}; nb=${sh#*,}
    echo "== $name ($flags) nt=$nt nb=$nb" | tee -a $OUT/timing.txt
    /tmp/ub_$name $nt $nb 3 | tee -a $OUT/timing.txt
  done
done
if [ "${PMC:-0}" = "1" ]; then
  cd /tmp && export TMPDIR=/tmp
  for v in $VARIANTS; do
    name=${v%%:*}
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 \
      --kernel-trace --output-format csv -d /tmp/pmc_$name -- /tmp/ub_$name 48 1024 2 > $GRAFT_REPO_ROOT/$OUT/pmc_$name.log 2>&1
    find /tmp/pmc_$name -name "*.csv" | head -5 >> $GRAFT_REPO_ROOT/$OUT/pmc_$name.log
    echo "## $name" | tee -a $GRAFT_REPO_ROOT/$OUT/pmc_summary.txt
    python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmc_$name update_kernel | tee -a $GRAFT_REPO_ROOT/$OUT/pmc_summary.txt
  done
fi
