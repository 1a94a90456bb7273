#!/bin/bash
# ablations of the bound pass at Synth-1M (ad-hoc builds of lru_topk_bf16.hip with -DTK_ABL=n linked into SEPARATE
# libraries, selected with LLAMAREC_LIB: the product library is never touched). 1 = no 16-register maximum, 2 = fragments
# read from LDS once per stage, 3 = both.   usage: bash tools/gpu_r4_stage1_abl.sh <tag>
OUT=gpurun_out/${1:-r4abl}; mkdir -p $OUT
L=$(pwd)/llamarec_amd/lib; C=$(pwd)/llamarec_amd/csrc
mkdir -p $L/abl
for a in 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-honor-nans -DTK_ABL=$a -c $C/lru_topk_bf16.hip -o $L/abl/bf16_$a.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/abl/lib_abl$a.so $(ls $L/obj/*.o | grep -v lru_topk_bf16.o) $L/abl/bf16_$a.o || exit 1
done
cd /tmp && export TMPDIR=/tmp
for a in 0 1 2 3; do
  lib=$L/libllamarec_mi355x.so; [ $a -gt 0 ] && lib=$L/abl/lib_abl$a.so
  export LLAMAREC_LIB=$lib
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/kt$a -- python3 $GRAFT_REPO_ROOT/tools/bench_stage1.py synth-1m > $GRAFT_REPO_ROOT/$OUT/kt$a.log 2>&1 || exit 1
  echo "ablation $a: $(grep item_bound_kernel $(find $GRAFT_REPO_ROOT/$OUT/kt$a -name '*kernel_stats.csv' | head -1) | cut -d, -f1-4)"
done
