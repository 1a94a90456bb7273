#!/bin/bash
# round 3: GEMM epilogue + attention changes: tests, micro-benchmarks, bench line, stamps
OUT=gpurun_out/${1:-r3d}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_llama.py tests/test_gpu_edge_cases.py tests/test_gpu_llama_train.py tests/test_gpu_entrypoints.py -m gpu -q -x > $OUT/tests.log 2>&1
rc=$?
tail -12 $OUT/tests.log
[ $rc -eq 0 ] || { echo "pytest rc=$rc: stopping"; exit 1; }
timeout -k 10 200 python tools/bench_gemm_epi.py 32768 5 > $OUT/epi.log 2>&1; grep -v amdgpu.ids $OUT/epi.log
timeout -k 10 200 python tools/bench_attn.py > $OUT/attn.log 2>&1; grep -v amdgpu.ids $OUT/attn.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-shapes > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("%.2f users/s  %.2f ms/step  gemm %.0f TF/s frac %.4f share %.3f attn %.0f TF/s" % (d["value"], d["ms_per_step"], r["achieved"], r["frac"], r["share_of_step_time"], d["attention_tflops"]))
print({k: (round(v["tflops"]), v["launches"], round(v["avg_ms"],3)) for k, v in r["per_shape"].items()})
PY
rm -f llamarec_amd/lib/obj/llama_gemm.o && make -C llamarec_amd/csrc -j16 EXPERIMENTS=1 > $OUT/make.log 2>&1 || { tail -5 $OUT/make.log; exit 1; }
timeout -k 10 200 python tools/gemm_stamps.py 32768 all > $OUT/stamps_32768.log 2>&1; grep -v amdgpu.ids $OUT/stamps_32768.log
