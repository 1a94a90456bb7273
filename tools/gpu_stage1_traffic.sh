#!/bin/bash
# Fabric-side traffic and L2 hit rate of the stage-1 kernels at Synth-1M (4 096 users x 1 M items): one --pmc pass per
# counter family (FETCH_SIZE and WRITE_SIZE cannot share a pass), --kernel-trace only. Summary -> gpurun_out/stage1_traffic/summary.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/stage1_traffic; mkdir -p $OUT
cd $R
timeout -k 10 200 python tools/bench_stage1.py synth-1m > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
grep -v amdgpu.ids $OUT/bench.log | tail -6
cd /tmp && export TMPDIR=/tmp
for pass in "a FETCH_SIZE" "w WRITE_SIZE" "b TCC_HIT_sum TCC_MISS_sum" "c GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
  set -- $pass; tag=$1; shift
  echo "[pmc] $*"
  timeout -k 10 200 rocprofv3 --pmc $* --kernel-trace --output-format csv -d $OUT/$tag -- python3 $R/tools/bench_stage1.py synth-1m > $OUT/$tag.log 2>&1 || { echo "pass failed"; grep -v "^    @" $OUT/$tag.log | tail -5; exit 1; }
done
cd $R
python - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for p in "awbc":
    f = glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True)
    if not f: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not any(t in k for t in ("item_", "cand_", "bound_", "em_")): continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, v in acc.items():
        for c, x in v.items(): res[k][c] = sum(x) / len(x)
        res[k]["dur_us_" + p] = sum(dur[k].values()) / len(dur[k]) / 1e3
for k, e in sorted(res.items()):
    if "FETCH_SIZE" in e: e["fetch_MB_x2"] = e["FETCH_SIZE"] * 1024 * 2 / 1e6
    if "WRITE_SIZE" in e: e["write_MB"] = e["WRITE_SIZE"] * 1024 / 1e6
    if "TCC_HIT_sum" in e: e["l2_hit"] = e["TCC_HIT_sum"] / max(1.0, e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
    if "GRBM_GUI_ACTIVE" in e: e["mfma_busy"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * e["GRBM_GUI_ACTIVE"] / 8)
    print(k, {x: round(y, 3) for x, y in e.items() if x in ("fetch_MB_x2", "write_MB", "l2_hit", "mfma_busy", "dur_us_a", "dur_us_c")})
PY
