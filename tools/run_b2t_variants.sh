#!/bin/bash
# A/B of library builds (tools/build_b2t_prof.sh NAME FLAGS): band_to_tridiagonal N=20480 band=128 d per build
for v in "$@"; do
  echo "== $v"
  DLAF_MI355X_LIB=$PWD/dla_future_amd/lib/$v/libdlaf_mi355x.so timeout -k 10 120 python tools/b2t_bench.py 4096 512 128 d check 2>&1 | tail -1 || exit 1
  DLAF_MI355X_LIB=$PWD/dla_future_amd/lib/$v/libdlaf_mi355x.so timeout -k 10 200 python tools/b2t_bench.py 20480 512 128 d 2>&1 | tail -2 || exit 1
done
