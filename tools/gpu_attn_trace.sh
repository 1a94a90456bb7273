#!/bin/bash
# box-local experiment build of the attention file, then the per-workgroup timeline (tools/attn_wg_trace.py) with two and
# with one workgroup per CU: bash tools/gpu_attn_trace.sh <tag>
set -e -o pipefail
OUT=gpurun_out/${1:-attn_trace}; mkdir -p $OUT
rm -f llamarec_amd/lib/obj/llama_attn.o && make -C llamarec_amd/csrc -j16 EXPERIMENTS=1 > $OUT/make.log 2>&1 || { tail -5 $OUT/make.log; exit 1; }
for s in "22 740" "2 8192"; do
  timeout -k 10 120 python tools/attn_wg_trace.py $s 2>&1 | grep -v amdgpu.ids | tee -a $OUT/trace.log
  LR_ATTN_ONE_PER_CU=1 timeout -k 10 120 python tools/attn_wg_trace.py $s 2>&1 | grep -v amdgpu.ids | sed 's/^/[one per CU] /' | tee -a $OUT/trace.log
done
