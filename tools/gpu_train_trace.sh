#!/bin/bash
# per-dispatch timeline of ONE retriever training step (Beauty shape): rocprofv3 --kernel-trace, then the launches of the last
# pass in order with their durations and grids.   usage: bash tools/gpu_train_trace.sh <tag>
OUT=$(pwd)/gpurun_out/${1:-trtrace}; mkdir -p $OUT; R=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 $R/tools/bench_train.py --only beauty --graph 0 --iters 5 > $OUT/log.txt 2>&1 || exit 1
cd $R
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# one pass = from a tr_pass_init_kernel launch to the next
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("tr_pass_init")]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
tot = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    grid = f'{int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"])}x{int(r["Grid_Size_Y"])//max(1,int(r["Workgroup_Size_Y"]))}x{int(r["Grid_Size_Z"])//max(1,int(r["Workgroup_Size_Z"]))}' if "Grid_Size_X" in r else r.get("Grid_Size", "")
    print(f"{(s - t0) / 1e3:8.1f} us  +gap {(s - prev_end) / 1e3:5.1f}  dur {(e - s) / 1e3:6.1f} us  {name:40s} wgs {grid}")
    prev_end = e
    tot += e - s
print(f"pass: {(prev_end - t0) / 1e3:.1f} us wall, {tot / 1e3:.1f} us in kernels, {b - a} launches")
PY
