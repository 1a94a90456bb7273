#!/bin/bash
# where do the attention kernel's wave cycles go? one PMC pass (SQ counters only, --kernel-trace; no other trace domain)
OUT=$(pwd)/gpurun_out/${1:-attnpmc}
mkdir -p $OUT
R=$(pwd)
timeout -k 10 300 python -m pytest tests/test_gpu_llama.py -m gpu -q -x -k "attention" > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p1 -- python3 $R/tools/bench_attn.py 16 740 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/p2 -- python3 $R/tools/bench_attn.py 16 740 > $OUT/p2.log 2>&1
cd $R
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ("p1", "p2"):
    f = glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True)
    if not f:
        print(p, "no counter csv"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "attn_mfma128" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(p, {k: round(sum(v) / len(v)) for k, v in acc.items()}, "launches", len(next(iter(acc.values()), [])))
PY
