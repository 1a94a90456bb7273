#!/bin/bash
# stamps of the diagnostic attention build on the Beauty token-budget step (43 prompts, 32 768 tokens); $1: library suffix
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
V=${1:-stamps}
L="722 693 821 742 701 604 717 885 763 916 633 707 981 738 691 820 730 848 672 625 1031 678 730 722 952 724 676 676 949 847 695 725 943 922 664 656 713 663 676 745 735 830 807"
LLAMAREC_LIB=$R/llamarec_amd/lib/abl/libllamarec_$V.so timeout -k 10 120 python tools/attn256_stamps.py $L > gpurun_out/attn256_${V}_beauty.txt 2>&1; rc=$?
grep -v amdgpu.ids gpurun_out/attn256_${V}_beauty.txt
grep -q "Memory access fault" gpurun_out/attn256_${V}_beauty.txt && exit 9
exit $rc
