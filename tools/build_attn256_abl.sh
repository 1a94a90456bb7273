#!/bin/bash
# Builds timing-only ablations of attention variant 3 into llamarec_amd/lib/abl/libllamarec_<name>.so (never the product
# library; select one with LLAMAREC_LIB). Runs here (hipcc cross-compiles); tools/gpu_attn256_abl.sh times them on the GPU.
set -e
R=$(cd $(dirname $0)/.. && pwd)
C=$R/llamarec_amd/csrc; L=$R/llamarec_amd/lib; mkdir -p $L/abl
(cd $C && make -s)
OBJS=$(ls $L/obj/*.o | grep -v llama_attn256.o)
for name in ${@:-nodma notr nokr noe nomax nobk nowait mfmaonly}; do
  inc=$L/abl/body_$name.inc
  if [ $name = stamps ]; then A2_STAMPS=${A2_STAMPS:-1} A2_OUT=$inc python $R/tools/gen_attn256.py; extra="-DA2_STAMPS";
  else A2_ABL=$name A2_OUT=$inc python $R/tools/gen_attn256.py; extra=""; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-variable -Wno-unused-value -fno-slp-vectorize \
     -mllvm -amdgpu-spill-vgpr-to-agpr=0 "-DA2_BODY_INC=\"$inc\"" $extra -c $C/llama_attn256.hip -o $L/abl/attn256_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/abl/libllamarec_$name.so $OBJS $L/abl/attn256_$name.o
  echo built $name
done
