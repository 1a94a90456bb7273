#!/bin/bash
# first GPU run of attention variant 3: correctness, then timing; each step under its own timeout
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 240 python tools/try_attn256.py > gpurun_out/try256.log 2>&1; rc=$?
tail -30 gpurun_out/try256.log
[ $rc -eq 0 ] || { echo "try_attn256 rc=$rc"; exit $rc; }
grep -q "^OK" gpurun_out/try256.log || { echo "parity failed: no timing"; exit 1; }
timeout -k 10 300 python tools/bench_attn.py 2,3 > gpurun_out/bench_attn256.log 2>&1; rc=$?
tail -20 gpurun_out/bench_attn256.log
exit $rc
