#!/bin/bash
# deterministic training mode (run on the GPU box): the training tests, then the step's cost with and without it
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_lru_train.py -x -q 2>&1 | tail -15 || exit 1
for d in 0 1 0 1; do timeout -k 10 300 python tools/bench_train.py --deterministic $d 2>&1 | grep "^graph=" | sed "s/^/det=$d /"; done
