mkdir -p gpurun_out/attn6
for s in "64 256" "32 512" "16 1024" "8 2048" "4 4096" "2 8192"; do timeout -k 10 120 python tools/bench_attn.py $s 0 2>&1 | grep "attention variant" >> gpurun_out/attn6/lens.log; done
cat gpurun_out/attn6/lens.log
bash tools/gpu_online_prof.sh online1 > gpurun_out/online1.log 2>&1; cat gpurun_out/online1.log | grep -v amdgpu.ids | tail -45
