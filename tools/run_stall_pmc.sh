#!/bin/bash
# Where the bulk update kernel waits: SQ / TA / TCP / TCC stall counters of the kernel alone (persistent, 480
# workgroups), the normal build against the all-L2-hits probe (-DDLAF_DBG_SAME_STRIPS) as the control.
# Counters in their own passes (rocprofv3 --pmc + --kernel-trace only), at most two per TA / TCP / TCC block.   tools/run_stall_pmc.sh <outdir>
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd)
OUT=$ROOT/${1:-gpurun_out/stall_pmc}
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/update_bench.hip -o /tmp/ub_base || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -DDLAF_DBG_SAME_STRIPS tools/update_bench.hip -o /tmp/ub_same || exit 1
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_ANY"
 "SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY"
 "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum"
 "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum"
 "TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum"
 "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum"
 "TCC_TAG_STALL_sum TCC_BUSY_avr"
 "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_LATENCY_FIFO_FULL_sum"
)
for v in base same; do
  i=0
  for ctrs in "${PASSES[@]}"; do
    i=$((i+1))
    rm -rf /tmp/st_${v}_$i
    # (a counter set the hardware cannot collect makes rocprofv3 abort and linger: every pass is bounded)
    timeout -k 5 120 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/st_${v}_$i -- /tmp/ub_$v 24 2048 2 480 > $OUT/run_${v}_$i.log 2>&1 || echo "pass $i ($ctrs): rocprofv3 failed or timed out" | tee -a $OUT/summary.txt
    echo "## $v pass $i" | tee -a $OUT/summary.txt
    python3 $ROOT/tools/pmc_summary.py /tmp/st_${v}_$i update_kernel 2>&1 | tee -a $OUT/summary.txt
  done
done
