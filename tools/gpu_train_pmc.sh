#!/bin/bash
# retriever training step (Beauty): SQ / GRBM counters of its kernels in two rocprofv3 --pmc passes (no other trace domain)
OUT=$(pwd)/gpurun_out/${1:-trpmc}
mkdir -p $OUT
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p1 -- python3 $R/tools/bench_train.py --only beauty --graph 0 --iters 5 > $OUT/p1.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/p2 -- python3 $R/tools/bench_train.py --only beauty --graph 0 --iters 5 > $OUT/p2.log 2>&1 || exit 1
cd $R
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ("p1", "p2"):
    f = glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True)
    if not f:
        print(p, "no counter csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("ts_") or k.startswith("tb_") or "scan" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(p, k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
