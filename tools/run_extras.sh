#!/bin/bash
# end-of-round extras on one box: the reference's Cholesky miniapp on the facade at C2, the triangular-solver variants,
# the large-size point N = 98304 (72 GiB matrix)
out=gpurun_out/r03e; mkdir -p $out
timeout -k 10 200 ./miniapp/miniapp_cholesky --matrix-size 65536 --block-size 1024 --nruns 2 --check-result last 2>&1 | grep -v amdgpu.ids > $out/miniapp_cholesky.txt || exit 1
tail -4 $out/miniapp_cholesky.txt
timeout -k 10 200 python tools/trsm_bench.py 16384 16384 1024 2>&1 | grep -v amdgpu.ids > $out/trsm_bench.txt || exit 1
tail -8 $out/trsm_bench.txt
timeout -k 10 300 python bench.py --matrix-size 98304 --steps 1 --warmup 1 --no-cpu-baseline --no-red2band --no-eigensolver --no-trsm-profile > $out/bench_N98304.json 2> $out/bench_N98304.err || { tail -5 $out/bench_N98304.err; exit 1; }
python tools/show_bench.py $out/bench_N98304.json | cut -c1-300
