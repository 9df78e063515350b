#!/bin/bash
# slot reservation of the pairs order after the work-stealing change: 16 / 24 / 32 / 48 free slots at C2 and C1
out=gpurun_out/r03w; mkdir -p $out; rm -f $out/ab_slots.txt
F="--no-cpu-baseline --no-red2band --no-eigensolver --no-check --no-trsm-profile"
show() {
python - $1 >> $out/ab_slots.txt <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "potrf_tile", d.get("potrf_tile"), "trsm in situ ms", d.get("trsm_panel",{}).get("in_situ_avg_launch_ms"), "roofline.achieved", d["roofline"]["achieved"])
PY
}
for sl in 32 24 16 48 32; do
  echo "== SIDECAR_SLOTS=$sl (late boost on)  C2" >> $out/ab_slots.txt
  DLAF_MI355X_SIDECAR_DEFAULT=$sl timeout -k 10 200 python bench.py $F --steps 3 --warmup 1 > $out/c2_$sl.json 2>> $out/err.txt || exit 1
  show $out/c2_$sl.json
  echo "== SIDECAR_SLOTS=$sl (late boost on)  C1" >> $out/ab_slots.txt
  DLAF_MI355X_SIDECAR_DEFAULT=$sl timeout -k 10 200 python bench.py $F --n 32768 --nb 512 --steps 5 --warmup 1 > $out/c1_$sl.json 2>> $out/err.txt || exit 1
  show $out/c1_$sl.json
done
cat $out/ab_slots.txt
