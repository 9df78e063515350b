#!/bin/bash
# The measured artifacts of a round in one call on one box: PMC traffic of one factorization (-> pmc_traffic.json,
# which bench.py then quotes), the default bench line with the CPU baseline, the same command under rocprofv3
# --kernel-trace --stats, and the two other single-GPU configurations.   tools/final_profiles.sh <outdir>
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd)
OUT=$ROOT/${1:-gpurun_out/final}
mkdir -p $OUT
set -x
timeout -k 10 500 tools/run_pmc_traffic.sh ${1:-gpurun_out/final}/pmc > $OUT/pmc.log 2>&1 || exit 1
cp $OUT/pmc/pmc_traffic.json profiles/pmc_traffic.json
timeout -k 10 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kstats && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -- python3 $ROOT/bench.py --no-cpu-baseline --no-red2band --no-eigensolver > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err) || exit 1
cp $(find /tmp/kstats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
timeout -k 10 300 python bench.py --no-cpu-baseline --no-red2band --no-eigensolver --matrix-size 32768 --block-size 512 --steps 5 > $OUT/bench_c1.json 2> $OUT/bench_c1.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --no-red2band --no-eigensolver --type z --matrix-size 32768 --block-size 512 --steps 3 > $OUT/bench_z.json 2> $OUT/bench_z.err || exit 1
set +x
python tools/show_bench.py $OUT/bench_default.json $OUT/bench_under_rocprof.json $OUT/bench_c1.json $OUT/bench_z.json
head -8 $OUT/kernel_stats.csv | cut -c1-200
