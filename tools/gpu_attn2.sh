#!/bin/bash
# attention loop: correctness (all stage-2 tests), then the stand-alone timing and the bench line
OUT=gpurun_out/${1:-attn2}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_llama.py tests/test_gpu_llama_train.py tests/test_gpu_edge_cases.py -m gpu -q -x > $OUT/tests.log 2>&1
rc=$?
tail -8 $OUT/tests.log
[ $rc -eq 0 ] || { echo "pytest rc=$rc: stopping"; exit 1; }
for i in 1 2; do
  for which in old new; do
    lib=$(pwd)/llamarec_amd/lib/libllamarec_mi355x.so
    [ $which = old ] && lib=$(pwd)/llamarec_amd/lib/libllamarec_old.so
    echo "== $which $i"
    LLAMAREC_LIB=$lib timeout -k 10 200 python tools/bench_attn.py 2>&1 | grep "TF/s"
    LLAMAREC_LIB=$lib timeout -k 10 200 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-other-shapes > $OUT/${which}_$i.json 2>$OUT/${which}_$i.err || { echo "bench failed"; tail -3 $OUT/${which}_$i.err; exit 1; }
    python - $OUT/${which}_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("%.2f users/s  %.2f ms/step  gemm %.0f TF/s frac %.4f  attn %.0f TF/s share %.4f" % (d["value"], d["ms_per_step"], r["achieved"], r["frac"], d["attention_tflops"], d["attention_share_of_step_time"]))
PY
  done
done
