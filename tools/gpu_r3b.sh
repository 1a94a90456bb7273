#!/bin/bash
# round 3, GEMM diagnostics: per-epilogue cost, then in-kernel stamps from a box-local EXPERIMENTS build
OUT=gpurun_out/${1:-r3b}
mkdir -p $OUT
timeout -k 10 200 python tools/bench_gemm_epi.py 32768 5 > $OUT/epi.log 2>&1; cat $OUT/epi.log | grep -v amdgpu.ids
timeout -k 10 300 python tools/bench_eval_pipeline.py --users 3000 --gpu > $OUT/evalpipe.log 2>&1; grep -v amdgpu.ids $OUT/evalpipe.log | tail -4
rm -f llamarec_amd/lib/obj/llama_gemm.o && make -C llamarec_amd/csrc -j16 EXPERIMENTS=1 > $OUT/make.log 2>&1 || { tail -5 $OUT/make.log; exit 1; }
timeout -k 10 200 python tools/gemm_stamps.py 32768 all > $OUT/stamps_32768.log 2>&1; grep -v amdgpu.ids $OUT/stamps_32768.log
timeout -k 10 100 python tools/gemm_stamps.py 2048 qkv > $OUT/stamps_2048.log 2>&1; grep -v amdgpu.ids $OUT/stamps_2048.log
timeout -k 10 100 python tools/gemm_stamps.py 1024 o > $OUT/stamps_1024.log 2>&1; grep -v amdgpu.ids $OUT/stamps_1024.log
