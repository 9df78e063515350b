#!/bin/bash
# rocprofv3 kernel statistics of reduction_to_band / bt_reduction_to_band (tools/red2band_bench.py)
# usage: run_prof_red2band.sh OUTDIR N nb type runs [bt]
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $ROOT/tools/red2band_bench.py "$@" > $OUT/bench.log 2>&1
cd $ROOT
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
cut -c1-220 $OUT/kernel_stats.csv | head -30
grep -E "RESULT|\[1\]" $OUT/bench.log
