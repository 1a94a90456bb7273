#!/bin/bash
# timing-only ablations of item_topk_kernel (outputs are wrong in the ablated builds): box-local rebuilds
for a in 0 1 2 3; do
  rm -f llamarec_amd/lib/obj/lru_topk.o
  make -C llamarec_amd/csrc -j16 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -DTK_ABLATE=$a" > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "=== ablate=$a (0 none, 1 no history walker, 2 no filter at all, 3 no list store)"
  cd /tmp && export TMPDIR=/tmp && LR_TOPK_CHUNKS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abl$a -- python3 $GRAFT_REPO_ROOT/tools/bench_stage1.py beauty > $GRAFT_REPO_ROOT/gpurun_out/abl$a.log 2>&1
  cd $GRAFT_REPO_ROOT; python tools/kstats.py $(find gpurun_out/abl$a -name '*kernel_stats.csv' | head -1) 1 6 | tail -1
done
