#!/bin/bash
# attention tests + microbench of the listed variants with the product build, then a box-local experiment build of the
# attention file and the per-workgroup timeline: bash tools/gpu_attn_trace.sh <tag> [variants, default 2]
set -e -o pipefail
OUT=gpurun_out/${1:-attn_trace}; mkdir -p $OUT
VARS=${2:-2}
timeout -k 10 600 python -m pytest tests/test_gpu_llama.py -m gpu -q -x -k "attention" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
timeout -k 10 300 python tools/bench_attn.py $VARS 2>&1 | grep "attention variant" | tee $OUT/bench.log
rm -f llamarec_amd/lib/obj/llama_attn.o && make -C llamarec_amd/csrc -j16 EXPERIMENTS=1 > $OUT/make.log 2>&1 || { tail -5 $OUT/make.log; exit 1; }
for v in ${VARS//,/ }; do for s in "22 740"; do timeout -k 10 120 python tools/attn_wg_trace.py $s $v 2>&1 | grep -v amdgpu.ids | tee -a $OUT/trace.log; done; done
