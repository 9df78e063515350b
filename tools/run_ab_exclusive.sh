#!/bin/bash
# A/B: persistent bulk update launches vacate whole compute units, one per shader engine and XCD per 64 slots
# (DLAF_MI355X_EXCLUSIVE_CUS=1), instead of leaving workgroup slots free.  C2, C1 and z N=32768 nb=512.
out=gpurun_out/r03x; mkdir -p $out; rm -f $out/ab_excl.txt
F="--no-cpu-baseline --no-red2band --no-eigensolver --no-check --no-trsm-profile"
show() {
python - $1 >> $out/ab_excl.txt <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "potrf_tile", d.get("potrf_tile"), "trsm in situ ms", d.get("trsm_panel",{}).get("in_situ_avg_launch_ms"), "roofline.achieved", d["roofline"]["achieved"])
PY
}
for a in "448 0 300" "512 64 300" "512 128 300"; do echo "== overlap: max_blocks excl_slots delay_us = $a"; timeout -k 5 60 ./tools/overlap_bench.bin 32 1024 $a 2>&1 | grep "^update\|left" ; done > $out/overlap2.txt 2>&1
cat $out/overlap2.txt
run() {  # name, env..., -- bench args
  name=$1; shift
  echo "== $name" >> $out/ab_excl.txt
  env "$@" timeout -k 10 200 python bench.py $F $BARGS > $out/$name.json 2>> $out/err.txt || exit 1
  show $out/$name.json
}
for cfg in "c2|--steps 3 --warmup 1" "c1|--n 32768 --nb 512 --steps 5 --warmup 1" "z|--type z --n 32768 --nb 512 --steps 3 --warmup 1"; do
  tag=${cfg%%|*}; BARGS=${cfg#*|}
  run ${tag}_excl0 DLAF_MI355X_EXCLUSIVE_CUS=0
  run ${tag}_excl1 DLAF_MI355X_EXCLUSIVE_CUS=1
  run ${tag}_excl1_s64 DLAF_MI355X_EXCLUSIVE_CUS=1 DLAF_MI355X_SIDECAR_SLOTS=64
  run ${tag}_excl0_s64 DLAF_MI355X_EXCLUSIVE_CUS=0 DLAF_MI355X_SIDECAR_SLOTS=64
done
cat $out/ab_excl.txt
