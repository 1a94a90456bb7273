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
build spread8 2,10,18,26,34,42,50,58
build spread4 1,5,9,13,17,21,25,29
build late 41,43,45,47,49,51,53,55
build pairs 1,2,9,10,17,18,25,26
