#!/bin/bash
# what the card reports while the product GEMM runs back to back on model-like operands: power cap, socket power, shader clock
# (rocm-smi as an ordinary user: read-only). Samples every 2 s beside tools/bench_gemm_data.py.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
rocm-smi --showmaxpower 2>&1 | grep -i "max graphics"
timeout -k 10 120 python tools/bench_gemm_data.py > gpurun_out/power_probe_gemm.log 2>&1 &
for i in $(seq 1 45); do
  p=$(rocm-smi --showpower --showclocks 2>&1 | grep -i "Socket Graphics Package Power\|sclk" | sed 's/.*: //' | tr '\n' ' ')
  n=$(grep -c operands gpurun_out/power_probe_gemm.log 2>/dev/null)
  echo "t=$((2*i))s bench lines so far $n: $p"
  sleep 2
  kill -0 %1 2>/dev/null || break
done
wait
cat gpurun_out/power_probe_gemm.log | grep operands
