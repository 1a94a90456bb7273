#!/bin/bash
# round 3, first GPU call: full GPU suite, then the default bench line
OUT=gpurun_out/${1:-r3a}
mkdir -p $OUT
timeout -k 10 700 python -m pytest tests -m gpu -q -x --durations=8 > $OUT/tests.log 2>&1
rc=$?
tail -25 $OUT/tests.log
echo "pytest rc=$rc"
[ $rc -le 1 ] || exit $rc
timeout -k 10 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { echo "bench default failed"; tail -15 $OUT/bench_default.err; exit 1; }
python - $OUT/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("%.2f users/s  %.2f ms/step  gemm %.0f TF/s frac %.4f share %.3f traffic %.3g" % (d["value"], d["ms_per_step"], r["achieved"], r["frac"], r["share_of_step_time"], r["traffic"] or 0))
print({k: (round(v["tflops"]), v["launches"], round(v["avg_ms"],3)) for k, v in r["per_shape"].items()})
print("metrics", {k:v for k,v in d["metrics"].items() if k!="labels"})
print("lora", d.get("lora_train_shape"))
print("ml100k", d.get("ml100k_shape"))
print("cpu", {k:v for k,v in d["cpu_baseline"].items() if k in ("value","cores","stage1_users_per_s","stage2_users_per_s_extrapolated","config1_ml100k_retriever_only")})
print("parity", {k:v for k,v in d["parity"].items() if not k.startswith("stage1_metrics_")})
PY
