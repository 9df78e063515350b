#!/bin/bash
# A/B: wave priority of the tile POTRF strips / side-stream panel TRSM beside the bulk update (C2 and z N=32768 nb=512)
out=gpurun_out/r03h; mkdir -p $out
F="--no-cpu-baseline --no-red2band --no-check --no-trsm-profile"
for pp in 0 1; do for tp in 0 1; do
  echo "== POTRF_PRIO=$pp TRSM_PRIO=$tp  C2" >> $out/ab_prio.txt
  DLAF_MI355X_POTRF_PRIO=$pp DLAF_MI355X_TRSM_PRIO=$tp timeout -k 10 200 python bench.py $F --steps 3 --warmup 1 > $out/b_${pp}${tp}.json 2>> $out/err.txt || exit 1
  python - $out/b_${pp}${tp}.json >> $out/ab_prio.txt <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "potrf_tile", d.get("potrf_tile"), "trsm in situ ms", d.get("trsm_panel",{}).get("in_situ_avg_launch_ms"), "roofline.achieved", d["roofline"]["achieved"])
PY
done; done
for pp in 0 1; do
  echo "== POTRF_PRIO=$pp  z N=32768 nb=512" >> $out/ab_prio.txt
  DLAF_MI355X_POTRF_PRIO=$pp timeout -k 10 200 python bench.py $F --type z --n 32768 --nb 512 --steps 3 --warmup 1 > $out/z_${pp}.json 2>> $out/err.txt || exit 1
  python - $out/z_${pp}.json >> $out/ab_prio.txt <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "potrf_tile", d.get("potrf_tile"), "roofline.achieved", d["roofline"]["achieved"])
PY
done
cat $out/ab_prio.txt
