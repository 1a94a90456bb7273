#!/bin/bash
# same-box A/B of two builds of the library: llamarec_amd/lib/libllamarec_old.so (built from the previous commit) against
# the current one. GPU tests first (current build), then alternating GEMM micro-benchmarks and bench.py runs.
# The build under test is selected with LLAMAREC_LIB (llamarec_amd/_lib.py): the product library is never overwritten, so a
# failed or timed-out run cannot leave the OLD build installed.
set -o pipefail
OUT=gpurun_out/${1:-ablib}
mkdir -p $OUT
L=$(pwd)/llamarec_amd/lib
timeout -k 10 900 python -m pytest tests/test_gpu_llama.py tests/test_gpu_edge_cases.py -m gpu -q -x > $OUT/tests.log 2>&1
rc=$?
tail -3 $OUT/tests.log
[ $rc -eq 0 ] || { echo "pytest rc=$rc: stopping"; exit 1; }
for i in 1 2; do
  for which in old new; do
    lib=$L/libllamarec_mi355x.so
    [ $which = old ] && lib=$L/libllamarec_old.so
    echo "== $which $i"
    LLAMAREC_LIB=$lib timeout -k 10 200 python tools/bench_gemm.py 4 ${AB_M:-32768} 2>&1 | grep "TF/s"
    LLAMAREC_LIB=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-shapes > $OUT/${which}_$i.json 2>$OUT/${which}_$i.err || { echo "bench failed"; tail -3 $OUT/${which}_$i.err; exit 1; }
    python - $OUT/${which}_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("%.2f users/s  %.2f ms/step  gemm %.0f TF/s share %.3f" % (d["value"], d["ms_per_step"], r["achieved"], r["share_of_step_time"]),
      {k: round(v["tflops"]) for k, v in (r.get("per_shape") or {}).items() if v["launches"] > 40})
PY
  done
done
