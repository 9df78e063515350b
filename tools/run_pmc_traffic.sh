#!/bin/bash
# HBM-side traffic of one factorization per kernel: PMC FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes
# (MI355X_MICROARCH.md, HBM section: FETCH_SIZE doubled on gfx950), summed per kernel by tools/pmc_summary.py, plus
# profiles/pmc_traffic.json for bench.py (bulk kernel: bytes per launch, launches per factorization).
#   tools/run_pmc_traffic.sh <outdir> [bench.py workload args]
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd)
OUT=$ROOT/${1:-gpurun_out/pmc_traffic}
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/pmc_hbm_traffic.txt
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$ctr -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check --no-trsm-profile --no-red2band --no-eigensolver "$@" > $OUT/bench_under_pmc_$ctr.log 2>&1
  python3 $ROOT/tools/pmc_summary.py /tmp/pmc_$ctr _kernel >> $OUT/pmc_hbm_traffic.txt
done
cat $OUT/pmc_hbm_traffic.txt
python3 - "$OUT" "$@" <<'PY'
import json, re, sys
out = sys.argv[1]
txt = open(out + "/pmc_hbm_traffic.txt").read()
def grab(kernel_pat, what):
    tot, n = 0.0, 0
    for m in re.finditer(r"== (.*?): (\d+) dispatches.*?\n((?:   .*\n)+)", txt):
        if re.search(kernel_pat, m.group(1)):
            g = re.search(what + r".*?= ([0-9.]+) GB", m.group(3))
            if g:
                tot += float(g.group(1)); n = int(m.group(2))
    return tot, n
fetch, n = grab(r"update_kernel<double, true, 0>", "FETCH_SIZE x2")
write, _ = grab(r"update_kernel<double, true, 0>", "WRITE_SIZE")
line = [l for l in open(out + "/bench_under_pmc_FETCH_SIZE.log") if l.startswith("{")]
wl = json.loads(line[-1])["config"] if line else {}
rec = {"workload": wl.get("workload"), "grid": wl.get("grid"), "kernel": "update_kernel<double,true,0>",
       "launches_per_factorization": n, "fetch_size_GB_corrected": fetch, "write_size_GB": write,
       "bytes_per_launch": (fetch + write) * 1e9 / max(n, 1), "source": "profiles/r04_pmc_hbm_traffic_N65536_nb1024.txt",
       "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes with --kernel-trace over one factorization "
                 "(bench.py --steps 1 --warmup 0); FETCH_SIZE x2 (gfx950); includes Infinity-Cache hits"}
json.dump(rec, open(out + "/pmc_traffic.json", "w"), indent=1)
print(json.dumps(rec))
PY
