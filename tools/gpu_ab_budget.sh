#!/bin/bash
# same-box A/B of the token budget per step (alternating runs)
OUT=gpurun_out/${1:-abbudget}
mkdir -p $OUT
for i in 1 2 3; do
  for b in 16384 32768; do
    st=$((b == 16384 ? 20 : 10))
    timeout -k 10 200 python bench.py --steps $st --warmup 2 --token-budget $b --no-cpu-baseline --no-other-shapes > $OUT/b${b}_$i.json 2>/dev/null || exit 1
  done
done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$OUT/b*_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(os.path.basename(f), "%.2f users/s  %.2f ms/step  users/step %.1f  gemm %.0f TF/s share %.3f  attn %.0f TF/s share %.3f" % (d["value"], d["ms_per_step"], d["config"]["users_per_step"], r["achieved"], r["share_of_step_time"], d["attention_tflops"], d["attention_share_of_step_time"]))
PY
