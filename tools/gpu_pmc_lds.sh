#!/bin/bash
# LDS bank-conflict counters of the GEMM micro-benchmark (one --pmc pass with --kernel-trace only)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmclds}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc -- python3 $GRAFT_REPO_ROOT/tools/bench_gemm.py 4 16384 > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 - $OUT <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/pmc/*/*counter_collection.csv')[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'][:40]
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in acc.items():
    if 'gemm256' in k: print(k, {c:"%.3g"%x for c,x in v.items()})
PY
