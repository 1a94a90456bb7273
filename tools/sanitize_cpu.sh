#!/bin/bash
# CPU sanitizer job: the C oracle and the HOST side of the library -- lr_lru_pack,
# lr_metrics_from_histogram, lr_common_prefix_len, the weight pack helpers, error plumbing -- built with
# -fsanitize=address,undefined (ROCm's clang for both, one sanitizer runtime per process) and driven by the
# `-m "not gpu"` tests that reach them. Device code is NOT sanitized (GPU ASan is unavailable on this pool) and this job
# never runs on the GPU box. Usage: bash tools/sanitize_cpu.sh   (exit code = pytest's; log: gpurun_out/sanitize_cpu.log)
set -o pipefail
cd "$(dirname "$0")/.."
make -C llamarec_amd/csrc -j8 SAN=1 > /dev/null || exit 1
make -C oracle SAN=1 > /dev/null || exit 1
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
mkdir -p gpurun_out
# detect_leaks=0: the interpreter itself "leaks" by design; halt_on_error + abort_on_error make a finding fail the test run
LD_PRELOAD=$RT \
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1 \
UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
LLAMAREC_LIB=$(pwd)/llamarec_amd/lib/san/libllamarec_mi355x.so \
LR_ORACLE_LIB=$(pwd)/oracle/_san/liblr_oracle.so \
python -m pytest -q -x -m "not gpu" -p no:cacheprovider \
  tests/test_abi_symbols.py tests/test_oracle_lru_golden.py tests/test_host_logic.py tests/test_packing.py tests/test_sanitize_targets.py \
  2>&1 | tee gpurun_out/sanitize_cpu.log
