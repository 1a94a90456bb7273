#!/bin/bash
# Same-box A/B of the folded RMSNorm (run on the GPU box): default / --fold-norms / --fold-norms with the statistic sweeps of
# layers 1.. skipped (timing-only experiment library: what the step would take if the row sums came for free from the o / down
# residual epilogues). Two interleaved rounds.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/foldab
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-shapes --no-item-roofline"
for round in 1 2; do
  timeout -k 10 400 $B > gpurun_out/foldab/a$round.log 2>&1 || exit 1
  echo "round $round default done"
  timeout -k 10 400 $B --fold-norms > gpurun_out/foldab/b$round.log 2>&1 || exit 1
  echo "round $round folded done"
  LLAMAREC_LIB=$R/llamarec_amd/lib/abl/libllamarec_foldexp.so LR_EXP_RSTD_ONCE=1 timeout -k 10 400 $B --fold-norms > gpurun_out/foldab/c$round.log 2>&1 || exit 1
  echo "round $round folded, sweeps skipped done"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/foldab/*.log")):
    line = [l for l in open(f) if l.startswith('{"metric"')][-1]
    j = json.loads(line)
    ps = j["roofline"]["per_shape"]
    print(f[-6:-4], "users/s %.2f  ms/step %.2f  gemm %.0f TF/s  " % (j["value"], j["ms_per_step"], j["roofline"]["achieved"]) +
          "  ".join("%s %.0f" % (k, v["tflops"]) for k, v in ps.items() if v["launches"] > 20))
PY
