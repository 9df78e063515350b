#!/bin/bash
# gen_to_std: parity, then the reference's miniapp with and without the lookahead (DLAF_MI355X_HEGST_LOOKAHEAD)
out=gpurun_out/r03y; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_gen_to_std.py -x -q -m gpu > $out/parity.txt 2>&1 || { tail -40 $out/parity.txt; exit 1; }
tail -2 $out/parity.txt
for la in 1 0; do for cfg in "16384 512 d" "32768 1024 d" "32768 512 d" "16384 512 z"; do set -- $cfg
  echo "== HEGST_LOOKAHEAD=$la  N=$1 nb=$2 type=$3"
  DLAF_MI355X_HEGST_LOOKAHEAD=$la timeout -k 10 120 ./miniapp/miniapp_gen_to_std --matrix-size $1 --block-size $2 --type $3 --nruns 4 --nwarmups 1 2>&1 | grep -v amdgpu.ids | grep "GFlop" | tail -3
done; done > $out/miniapp.txt 2>&1
cat $out/miniapp.txt
