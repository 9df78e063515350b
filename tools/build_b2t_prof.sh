#!/bin/bash
# Another build of the library with compile-time switches of b2t_reg_kernel (DLAF_MI355X_B2T_PROF: phase clocks):
#   tools/build_b2t_prof.sh NAME "-DDLAF_MI355X_B2T_PROF ..."  ->  dla_future_amd/lib/NAME/libdlaf_mi355x.so,
# loaded with DLAF_MI355X_LIB=...
set -e
NAME=${1:-prof}
FLAGS=${2:--DDLAF_MI355X_B2T_PROF}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=$ROOT/dla_future_amd/csrc/build
mkdir -p $ROOT/dla_future_amd/lib/$NAME $B/$NAME
make -C $ROOT/dla_future_amd/csrc -j8 > /dev/null
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -I$ROOT/include $FLAGS \
  -c $ROOT/dla_future_amd/csrc/device/kernels_tridiag.hip -o $B/$NAME/kernels_tridiag.o
OBJS=$(ls $B/device/*.o $B/host/*.o | grep -v device/kernels_tridiag.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $ROOT/dla_future_amd/lib/$NAME/libdlaf_mi355x.so $OBJS $B/$NAME/kernels_tridiag.o \
  -L/opt/rocm/lib -lrccl -lpthread -ldl -Wl,-rpath,/opt/rocm/lib
echo built $ROOT/dla_future_amd/lib/$NAME/libdlaf_mi355x.so
