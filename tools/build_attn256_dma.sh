#!/bin/bash
# timing experiment libraries of attention variant 3 with the block's 8 LDS-DMA pieces placed in other gaps (results stay correct: the
# pieces only have to be issued inside the block): llamarec_amd/lib/abl/libllamarec_dma_<name>.so
set -e
R=$(cd $(dirname $0)/.. && pwd); C=$R/llamarec_amd/csrc; L=$R/llamarec_amd/lib; mkdir -p $L/abl
(cd $C && make -s)
OBJS=$(ls $L/obj/*.o | grep -v llama_attn256.o)
build() {
  inc=$L/abl/body_dma_$1.inc
  A2_DMA_GAPS=$2 A2_OUT=$inc python $R/tools/gen_attn256.py
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-variable -Wno-unused-value -fno-slp-vectorize \
     -mllvm -amdgpu-spill-vgpr-to-agpr=0 "-DA2_BODY_INC=\"$inc\"" -c $C/llama_attn256.hip -o $L/abl/attn256_dma_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/abl/libllamarec_dma_$1.so $OBJS $L/abl/attn256_dma_$1.o
  echo "built dma_$1 ($2)"
}
build first8 1,2,3,4,5,6,7,8
build mid 9,11,13,15,17,19,20,21
build split 9,12,15,18,44,47,50,53
build every3 41,44,47,50,53,56,59,62
build late1 48,49,50,51,52,53,54,55
