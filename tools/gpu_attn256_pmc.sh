#!/bin/bash
# L2 / fabric traffic of the attention kernels on the Beauty token-budget step (variants 2 and 3), one PMC pass per counter group
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/attn256_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 2 3; do
  for pass in "a FETCH_SIZE" "w WRITE_SIZE" "b TCC_HIT_sum TCC_MISS_sum" "c GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
    set -- $pass; tag=$1; shift
    echo "[pmc] variant $v: $*"
    timeout -k 10 150 rocprofv3 --pmc $* --kernel-trace --output-format csv -d $OUT/v${v}_$tag -- python3 $R/tools/bench_attn.py beauty $v > $OUT/v${v}_$tag.log 2>&1 || { echo "pass failed"; grep -v "^    @" $OUT/v${v}_$tag.log | tail -5; exit 1; }
  done
done
cd $R
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for v in (2, 3):
    res = {}
    for p in "awbc":
        f = glob.glob(f"{out}/v{v}_{p}/**/*counter_collection.csv", recursive=True)
        if not f: continue
        acc = collections.defaultdict(list); dur = {}
        for r in csv.DictReader(open(f[0])):
            if "attn_mfma" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for k, x in acc.items(): res[k] = sum(x) / len(x)
        if dur: res["dur_us_" + p] = sum(dur.values()) / len(dur) / 1e3
    if "FETCH_SIZE" in res: res["fetch_MB_x2"] = res["FETCH_SIZE"] * 1024 * 2 / 1e6
    if "WRITE_SIZE" in res: res["write_MB"] = res["WRITE_SIZE"] * 1024 / 1e6
    if "TCC_HIT_sum" in res: res["l2_hit"] = res["TCC_HIT_sum"] / (res["TCC_HIT_sum"] + res["TCC_MISS_sum"])
    if "GRBM_GUI_ACTIVE" in res:
        g = res["GRBM_GUI_ACTIVE"] / 8
        res["mfma_busy"] = res["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * g); res["clock_ghz"] = g / (res["dur_us_c"] * 1e-6) / 1e9
    print("variant", v, {k: round(x, 3) for k, x in res.items()})
PY
