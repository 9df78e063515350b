#!/bin/bash
# Tuning aid: standalone bulk-update kernel under compile-time variants (tools/update_bench.hip).
set -e
cd ${GRAFT_REPO_ROOT:-.}
for flags in ${UB_FLAGS:-"-DBASE -DDLAF_DBG_SAME_STRIPS -DDLAF_DBG_SKIP_GLOBAL"}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include $flags tools/update_bench.hip -o /tmp/update_bench
  echo "== flags: $flags"
  /tmp/update_bench 48 1024 3
  /tmp/update_bench 48 1024 3 480
  /tmp/update_bench 64 512 3
done
