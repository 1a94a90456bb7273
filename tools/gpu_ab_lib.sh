#!/bin/bash
# same-box A/B of two builds of the library: llamarec_amd/lib/libllamarec_old.so (built from the previous commit) against
# the current one. GPU tests first (current build), then alternating GEMM micro-benchmarks and bench.py runs.
OUT=gpurun_out/${1:-ablib}
mkdir -p $OUT
L=llamarec_amd/lib
timeout -k 10 900 python -m pytest tests/test_gpu_llama.py tests/test_gpu_edge_cases.py -m gpu -q -x 2>&1 | tail -3 || exit 1
cp $L/libllamarec_mi355x.so $L/libllamarec_new.so
for i in 1 2; do
  for which in old new; do
    cp $L/libllamarec_$which.so $L/libllamarec_mi355x.so
    echo "== $which $i"
    timeout -k 10 200 python tools/bench_gemm.py 4 16384 2>&1 | grep "TF/s"
    timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-shapes > $OUT/${which}_$i.json 2>/dev/null || exit 1
    python - $OUT/${which}_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("%.2f users/s  %.2f ms/step  gemm %.0f TF/s share %.3f" % (d["value"], d["ms_per_step"], r["achieved"], r["share_of_step_time"]))
PY
  done
done
cp $L/libllamarec_new.so $L/libllamarec_mi355x.so
