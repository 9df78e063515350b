#!/bin/bash
# rocprofv3 kernel statistics of the whole eigensolver (tools/eigensolver_bench.py)
# usage: run_prof_eigensolver.sh OUTDIR N nb type runs
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $ROOT/tools/eigensolver_bench.py "$@" > $OUT/bench.log 2>&1
cd $ROOT
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
cut -c1-200 $OUT/kernel_stats.csv | head -40
grep -E "RESULT|\[[0-9]\]|     " $OUT/bench.log
