#!/bin/bash
# names of the hipBLASLt kernels torch.matmul picks for the prefill GEMM shapes (a yardstick only): rocprofv3 --kernel-trace --stats
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/yardstick; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/tools/yardstick_names.py > $OUT/run.log 2>&1 || exit 1
cd $R; python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/yardstick/kt/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'][:400], '| calls', r['Calls'], '| avg us', round(float(r['AverageNs']) / 1e3, 1))
PY
grep "TF/s" $OUT/run.log
