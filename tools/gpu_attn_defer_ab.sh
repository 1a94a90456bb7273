#!/bin/bash
# deferred maximum in attention variant 2: accuracy against the fp32 oracle and the latency-mode gap, new library against the one
# with the previous attention kernel (llamarec_amd/lib/abl/libllamarec_oldattn.so), then timing
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
for L in "" $R/llamarec_amd/lib/abl/libllamarec_oldattn.so; do
  echo "== library: ${L:-product}"
  LLAMAREC_LIB=$L timeout -k 10 400 python -m pytest tests/test_gpu_llama.py -q -s -k "eight_layers_long_prompts_parity_vs_oracle and 0 or full_depth_full_width_parity" 2>&1 | grep "rms vs\|passed\|failed" || exit 1
  LLAMAREC_LIB=$L timeout -k 10 250 python tools/diag/latency_mode_gap.py 2>&1 | tail -1
  LLAMAREC_LIB=$L timeout -k 10 200 python tools/bench_attn.py 2 2>&1 | grep "variant 2"
done
