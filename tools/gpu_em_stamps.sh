#!/bin/bash
# s_memtime anatomy of em_layer_kernel from an EXPERIMENTS build (box-local rebuild: the product library is not touched)
rm -f llamarec_amd/lib/obj/lru_encoder_mfma.o && make -C llamarec_amd/csrc -j16 EXPERIMENTS=1 > /dev/null 2>&1
python tools/em_stamps.py beauty 22332 2>&1 | grep -v amdgpu.ids | tail -8
