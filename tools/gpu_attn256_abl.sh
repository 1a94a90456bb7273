#!/bin/bash
# times the ablation builds of tools/build_attn256_abl.sh against the product build (same box, same process order)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
OUT=gpurun_out/attn256_abl.txt; : > $OUT
for lib in product $(ls llamarec_amd/lib/abl/*.so 2>/dev/null); do
  if [ $lib = product ]; then unset LLAMAREC_LIB; else export LLAMAREC_LIB=$R/$lib; fi
  echo "== $lib" >> $OUT
  timeout -k 10 120 python tools/bench_attn.py 3 >> $OUT 2>&1 || { echo "bench failed for $lib" >> $OUT; tail -3 $OUT; exit 1; }
done
grep -v amdgpu.ids $OUT
