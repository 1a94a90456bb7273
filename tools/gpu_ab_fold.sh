#!/bin/bash
# same-box A/B: RMSNorm as its own pass vs folded into the GEMMs (alternating, 20 steps each)
OUT=gpurun_out/${1:-abfold}
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1
rc=$?
tail -4 $OUT/tests.log
[ $rc -le 1 ] || exit $rc
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-shapes > $OUT/plain_$i.json 2>/dev/null || exit 1
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-shapes --fold-norms > $OUT/fold_$i.json 2>/dev/null || exit 1
done
python - <<'PY'
import json,glob,os,sys
out=sys.argv[1] if len(sys.argv)>1 else None
for f in sorted(glob.glob(os.environ.get("OUTDIR","gpurun_out")+"/*/plain_*.json")+glob.glob(os.environ.get("OUTDIR","gpurun_out")+"/*/fold_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(os.path.basename(f), "%.2f users/s  %.2f ms/step  gemm %.0f TF/s share %.3f  attn %.0f TF/s share %.3f" % (d["value"], d["ms_per_step"], r["achieved"], r["share_of_step_time"], d["attention_tflops"], d["attention_share_of_step_time"]))
PY
