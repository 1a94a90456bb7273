#!/bin/bash
# same-box A/B of the attention K/V staging: buffer descriptors (product) vs per-lane pointers (-DFA_STAGE_PTR, box-local build)
for r in 1 2; do
  echo "--- descriptors"; python tools/bench_attn.py 2>&1 | grep -v amdgpu
  rm -f llamarec_amd/lib/obj/llama_attn.o
  make -C llamarec_amd/csrc -j16 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -DFA_STAGE_PTR" > /dev/null 2>&1 || exit 1
  echo "--- per-lane pointers"; python tools/bench_attn.py 2>&1 | grep -v amdgpu
  rm -f llamarec_amd/lib/obj/llama_attn.o
  make -C llamarec_amd/csrc -j16 > /dev/null 2>&1 || exit 1
done
