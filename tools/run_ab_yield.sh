#!/bin/bash
# A/B: bulk-update workgroups that share a compute unit with a tile-POTRF strip sit out between work items
# (DLAF_MI355X_POTRF_YIELD).  Parity first, then C2, C1 and z N=32768 nb=512.
out=gpurun_out/r03y2; mkdir -p $out; rm -f $out/ab_yield.txt
F="--no-cpu-baseline --no-red2band --no-eigensolver --no-check --no-trsm-profile"
timeout -k 10 400 python -m pytest tests/test_gpu_cholesky.py tests/test_gpu_potrf_paths.py tests/test_gpu_tiles.py -x -q -m gpu > $out/parity.txt 2>&1 || { tail -30 $out/parity.txt; exit 1; }
tail -2 $out/parity.txt
show() {
python - $1 >> $out/ab_yield.txt <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "potrf_tile", d.get("potrf_tile"), "trsm in situ ms", d.get("trsm_panel",{}).get("in_situ_avg_launch_ms"), "roofline.achieved", d["roofline"]["achieved"])
PY
}
for y in 1 0 1 0; do
  echo "== POTRF_YIELD=$y  C2" >> $out/ab_yield.txt
  DLAF_MI355X_POTRF_YIELD=$y timeout -k 10 200 python bench.py $F --steps 3 --warmup 1 > $out/c2_$y.json 2>> $out/err.txt || exit 1
  show $out/c2_$y.json
  echo "== POTRF_YIELD=$y  C1" >> $out/ab_yield.txt
  DLAF_MI355X_POTRF_YIELD=$y timeout -k 10 200 python bench.py $F --n 32768 --nb 512 --steps 5 --warmup 1 > $out/c1_$y.json 2>> $out/err.txt || exit 1
  show $out/c1_$y.json
  echo "== POTRF_YIELD=$y  z" >> $out/ab_yield.txt
  DLAF_MI355X_POTRF_YIELD=$y timeout -k 10 200 python bench.py $F --type z --n 32768 --nb 512 --steps 3 --warmup 1 > $out/z_$y.json 2>> $out/err.txt || exit 1
  show $out/z_$y.json
done
cat $out/ab_yield.txt
