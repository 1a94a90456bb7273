#!/bin/bash
# times the DMA-placement libraries of attention variant 3 (tools/build_attn256_dma.sh) against the product; parity first
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
for n in "" dma_first8 dma_mid dma_split dma_every3 dma_late1 ""; do
  L=${n:+$R/llamarec_amd/lib/abl/libllamarec_$n.so}
  echo "== ${n:-product}"
  if [ -n "$n" ]; then LLAMAREC_LIB=$L timeout -k 10 200 python tools/try_attn256.py 2>&1 | grep "^OK\|FAIL\|Error" | head -2; fi
  LLAMAREC_LIB=$L timeout -k 10 100 python tools/bench_attn.py 3 2>&1 | grep "token-budget\|8192\|16 x 1000"
done
