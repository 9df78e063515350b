#!/bin/bash
# runs tests/c_api/test_pdsyevd under mpiexec with progress lines (diagnosis)
export DLAF_MI355X_MPI_TRANSPORT=host DLAF_MI355X_DEVICE=0 OMP_NUM_THREADS=1 C_API_TEST_VERBOSE=1
timeout -k 5 150 /opt/conda/bin/mpiexec -n $1 ./tests/c_api/test_pdsyevd $2 $3 $4 > gpurun_out/c_eig_$1.log 2>&1
echo "rc $?" >> gpurun_out/c_eig_$1.log
tail -8 gpurun_out/c_eig_$1.log
