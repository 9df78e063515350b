#!/bin/bash
# A/B of the K-phase alignment of the persistent bulk update launches (DLAF_MI355X_KPHASE=0 / 1, same binary,
# same box, alternating) + the L2 hit counters of both.
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd)
OUT=$ROOT/${1:-gpurun_out/ab_kphase}
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/update_bench.hip -o /tmp/ub_kp
for round in ${AB_ROUNDS:-1 2}; do
  for v in "0" "1" "1 DLAF_MI355X_KPHASE_RATE=60" "1 DLAF_MI355X_KPHASE_RATE=72"; do
    set -- $v
    for args in "48 1024 3 480" "24 2048 3 480" "64 512 3 480" "32 1024 3 480"; do
      echo "== kphase=$v $args (round $round)" | tee -a $OUT/timing.txt
      env DLAF_MI355X_KPHASE=$1 $2 /tmp/ub_kp $args | grep TFlop | tail -1 | tee -a $OUT/timing.txt
    done
  done
done
if [ "${PMC:-1}" = "1" ]; then
  cd /tmp && export TMPDIR=/tmp
  for kp in 0 1; do
    export DLAF_MI355X_KPHASE=$kp
    rm -rf /tmp/pmc_kp$kp
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/pmc_kp$kp -- /tmp/ub_kp 24 2048 2 480 > $OUT/pmc_kp$kp.log 2>&1
    echo "## kphase=$kp (nt=24 nb=2048 persistent 480)" | tee -a $OUT/l2_summary.txt
    python3 $ROOT/tools/pmc_summary.py /tmp/pmc_kp$kp update_kernel | tee -a $OUT/l2_summary.txt
    rm -rf /tmp/pmc_kpf$kp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_kpf$kp -- /tmp/ub_kp 24 2048 2 480 >> $OUT/pmc_kp$kp.log 2>&1
    python3 $ROOT/tools/pmc_summary.py /tmp/pmc_kpf$kp update_kernel | tee -a $OUT/l2_summary.txt
  done
fi
