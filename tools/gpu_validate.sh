#!/bin/bash
# round validation (bash tools/gpu_validate.sh <tag>): full GPU suite, DP rehearsal (gloo, 2 ranks on one card), default bench line
OUT=gpurun_out/${1:-validate}
mkdir -p $OUT
timeout -k 10 800 python -m pytest tests -m gpu -q -x --durations=5 > $OUT/tests.log 2>&1
rc=$?
tail -12 $OUT/tests.log
[ $rc -eq 0 ] || { echo "pytest rc=$rc: stopping"; exit 1; }
timeout -k 10 600 bash tools/rehearse_dp.sh > $OUT/rehearse.log 2>&1; echo "rehearse rc=$?"; tail -8 $OUT/rehearse.log | cut -c1-400
for tb in; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --token-budget $tb --no-cpu-baseline --no-other-shapes > $OUT/tb_$tb.json 2> $OUT/tb_$tb.err || { echo "bench tb=$tb failed"; tail -5 $OUT/tb_$tb.err; continue; }
  python - $OUT/tb_$tb.json $tb <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("budget %s: %.2f users/s  %.2f ms/step  gemm %.0f TF/s frac %.4f attn %.0f" % (sys.argv[2], d["value"], d["ms_per_step"], r["achieved"], r["frac"], d["attention_tflops"]))
PY
done
timeout -k 10 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { echo "bench default failed"; tail -15 $OUT/bench_default.err; exit 1; }
python - $OUT/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("default: %.2f users/s  %.2f ms/step  gemm %.0f TF/s frac %.4f share %.3f traffic %.3g" % (d["value"], d["ms_per_step"], r["achieved"], r["frac"], r["share_of_step_time"], r["traffic"] or 0))
print({k: (round(v["tflops"]), v["launches"]) for k, v in r["per_shape"].items()})
print("lora", {k: d["lora_train_shape"][k] for k in ("ms_per_step","tokens_per_s","tokens_per_step","workspace_allocations_in_timed_loop")})
print("parity ok", d["parity"]["ok"], "metrics match", d["metrics"]["retrieve_matches_expected"], "cpu", round(d["cpu_baseline"]["value"],4))
PY
