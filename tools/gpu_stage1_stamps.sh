#!/bin/bash
# per-kernel breakdown (rocprofv3) of the Beauty-shape retrieve for several chunk counts, then the s_memtime anatomy of
# item_topk_kernel from an EXPERIMENTS build (box-local rebuild: the product library is not touched)
OUT=gpurun_out/${1:-s1stamps}
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_lru.py -m gpu -q -x > $OUT/tests.log 2>&1; rc=$?; tail -2 $OUT/tests.log; [ $rc -eq 0 ] || { echo "pytest rc=$rc: stopping"; exit 1; }
for c in 0 1 4; do
  cd /tmp && export TMPDIR=/tmp && LR_TOPK_CHUNKS=$c rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kt$c -- python3 $R/tools/bench_stage1.py beauty > $R/$OUT/kt$c.log 2>&1
  cd $R; echo "=== chunks=$c (0 = geometry model)"; grep beauty $OUT/kt$c.log; python tools/kstats.py $(find $OUT/kt$c -name '*kernel_stats.csv' | head -1) 6 6 2>/dev/null
done
rm -f llamarec_amd/lib/obj/lru_topk.o && make -C llamarec_amd/csrc -j16 EXPERIMENTS=1 > /dev/null 2>&1
for c in 1 4; do echo "=== stamps chunks=$c"; LR_TOPK_CHUNKS=$c python tools/topk_stamps.py beauty 22332 2>&1 | grep -v amdgpu.ids | tail -6; done
