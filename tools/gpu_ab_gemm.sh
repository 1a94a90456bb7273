#!/bin/bash
# same-box A/B of two builds of the library on the GEMM micro-benchmark with real epilogues and on the bench line:
# llamarec_amd/lib/libllamarec_old.so (previous commit, built by hand) vs the current build; LLAMAREC_LIB selects the build.
set -o pipefail
OUT=gpurun_out/${1:-abgemm}
mkdir -p $OUT
L=$(pwd)/llamarec_amd/lib
for i in 1 2 3; do
  for which in old new; do
    lib=$L/libllamarec_mi355x.so
    [ $which = old ] && lib=$L/libllamarec_old.so
    echo "== $which $i"
    LLAMAREC_LIB=$lib timeout -k 10 200 python tools/bench_gemm_epi.py 32768 4 2>&1 | grep "TF/s" | sed 's/M=32768 //'
    LLAMAREC_LIB=$lib timeout -k 10 200 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-other-shapes > $OUT/${which}_$i.json 2>$OUT/${which}_$i.err || { echo "bench failed"; tail -3 $OUT/${which}_$i.err; exit 1; }
    python - $OUT/${which}_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("%.2f users/s  %.2f ms/step  gemm %.0f TF/s frac %.4f" % (d["value"], d["ms_per_step"], r["achieved"], r["frac"]),
      {k: round(v["tflops"]) for k, v in (r.get("per_shape") or {}).items() if v["launches"] > 40})
PY
  done
done
