for w in ml-100k games synth-1m; do
  timeout -k 10 300 python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-other-shapes > gpurun_out/wl_$w.json 2> gpurun_out/wl_$w.err || { echo "$w FAILED"; tail -5 gpurun_out/wl_$w.err; }
  python - gpurun_out/wl_$w.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; c=d["config"]
print("%s: %.1f users/s  %.1f ms/step  users/step %.1f rows/step %.0f gemm %.0f TF/s attn %.0f TF/s stage1 %.3f ms/step" % (c["workload"][:12], d["value"], d["ms_per_step"], c["users_per_step"], c["mean_rows_per_step"], r["achieved"], d["attention_tflops"], d["stage1_ms_per_step"]))
PY
done
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --dist-backend gloo --steps 4 --warmup 1 --no-cpu-baseline --no-other-shapes 2>/dev/null | tail -1 | cut -c1-400
