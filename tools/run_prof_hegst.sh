#!/bin/bash
# rocprofv3 kernel statistics of miniapp_gen_to_std.  usage: run_prof_hegst.sh OUTDIR N nb type lookahead
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/$1; N=$2; NB=$3; TY=$4; LA=$5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DLAF_MI355X_HEGST_LOOKAHEAD=$LA
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- $ROOT/miniapp/miniapp_gen_to_std --matrix-size $N --block-size $NB --type $TY --nruns 3 --nwarmups 1 > $OUT/bench.log 2>&1
cd $ROOT
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
python tools/kstats.py $OUT/kernel_stats.csv | head -25
grep -E "GFlop" $OUT/bench.log
