#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}   # set before any cd: a missing variable must not turn into /gpurun_out and /tools paths
# Counters over the stand-alone GEMM harness (tools/diag/gemm_bm.hip, quick mode = the product's tile on the two batch shapes):
# what the vector-memory path of a CU looks like under the K loop -- texture-addresser busy, L1 stalls, and the L2 read latency a CU sees.
# Separate --pmc passes (no trace domains besides --kernel-trace), outputs under gpurun_out/gemm_harness_pmc/.
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/gemm_harness_pmc
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/gemm_bm $R/tools/diag/gemm_bm.hip
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
grep -o -E "\b(TA_[A-Z0-9_]+|TCP_[A-Z0-9_]+|TD_[A-Z0-9_]+)\b" $OUT/counters.txt | sort -u > $OUT/ta_tcp_names.txt || true
wc -l $OUT/ta_tcp_names.txt
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  echo "== pass $i: $set"
  timeout -k 10 180 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- /tmp/gemm_bm quick > $OUT/p$i.log 2>&1 || echo "pass $i failed (see p$i.log)"
  tail -2 $OUT/p$i.log | cut -c1-200
done
ls $OUT
