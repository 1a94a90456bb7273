#!/bin/bash
# How much of the LoRA step do the adapter-side kernels (rank-r products and token reductions, run on a side stream beside
# the qkv GEMMs) cost? Box-local build with those launches skipped when LR_SKIP_ADAPTER=1 (wrong gradients, timing only),
# against the normal step, with and without the side stream: bash tools/gpu_lora_bound.sh <tag>
set -e -o pipefail
OUT=gpurun_out/${1:-lora_bound}; mkdir -p $OUT
python3 - <<'PY'
import re
p='llamarec_amd/csrc/api_llama_train.hip'
s=open(p).read()
n=0
for pat in ['RUN(lr_launch_skinny(', 'RUN(lr_launch_lora_db(', 'RUN(lr_launch_lora_da(']:
    n+=s.count('    '+pat)
    s=s.replace('    '+pat, '    if (!getenv("LR_SKIP_ADAPTER")) '+pat)
open(p,'w').write(s)
print('patched', n, 'launch sites')
PY
make -C llamarec_amd/csrc -j16 > $OUT/make.log 2>&1 || { tail -5 $OUT/make.log; exit 1; }
for i in 1 2; do
  echo "== normal $i"; timeout -k 10 300 python tools/bench_rank_train.py --layers 8 --steps 3 2>&1 | grep "tokens/s"
  echo "== adapter kernels skipped $i"; LR_SKIP_ADAPTER=1 timeout -k 10 300 python tools/bench_rank_train.py --layers 8 --steps 3 2>&1 | grep "tokens/s"
  echo "== no side stream $i"; LR_LORA_OVERLAP=0 timeout -k 10 300 python tools/bench_rank_train.py --layers 8 --steps 3 2>&1 | grep "tokens/s"
done
