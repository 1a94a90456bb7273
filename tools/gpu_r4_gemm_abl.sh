#!/bin/bash
# MFMA busy x effective clock of the 256-tile GEMM over staging ablations, per shipped shape.
# Arms: 0 product; 1 no DMA after the prologue; 2 no fragment reads after the first K tile; 3 neither; 4 every workgroup
# streams tile (0,0)'s panels (same DMA / LDS traffic, nothing crosses the fabric). Ad-hoc builds of llama_gemm.hip with
# -DG2_ABL=n are linked into SEPARATE libraries and selected with LLAMAREC_LIB (the product library is never touched).
# Per arm: one un-profiled timing run (TF/s) and one rocprofv3 --pmc pass (GRBM_GUI_ACTIVE, SQ_VALU_MFMA_BUSY_CYCLES).
# usage: bash tools/gpu_r4_gemm_abl.sh <tag> [M]
OUT=gpurun_out/${1:-r4gabl}; M=${2:-32768}; mkdir -p $OUT
R=$(pwd); L=$R/llamarec_amd/lib; C=$R/llamarec_amd/csrc
mkdir -p $L/abl
for a in 1 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DG2_ABL=$a -c $C/llama_gemm.hip -o $L/abl/gemm_$a.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/abl/lib_gabl$a.so $(ls $L/obj/*.o | grep -v llama_gemm.o) $L/abl/gemm_$a.o || exit 1
done
cd /tmp && export TMPDIR=/tmp
for a in 0 1 2 3 4; do
  lib=$L/libllamarec_mi355x.so; [ $a -gt 0 ] && lib=$L/abl/lib_gabl$a.so
  export LLAMAREC_LIB=$lib
  echo "== arm $a (un-profiled)"; timeout -k 10 200 python3 $R/tools/bench_gemm.py 4 $M 2>&1 | grep "TF/s" | tee $R/$OUT/time_$a.log
  timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/$OUT/pmc$a -- python3 $R/tools/bench_gemm.py 4 $M > $R/$OUT/pmc_$a.log 2>&1 || exit 1
done
cd $R
python3 - $OUT <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
names = {48: "qkv", 16: "o/down", 86: "gate_up"}
res = {}
for a in range(5):
    f = glob.glob(f"{out}/pmc{a}/**/*counter_collection.csv", recursive=True)
    if not f:
        continue
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "gemm256rb" not in r["Kernel_Name"]:
            continue
        key = (int(r["Grid_Size"]) // 512, r["Dispatch_Id"])
        rows[key][r["Counter_Name"]] = float(r["Counter_Value"])
        rows[key]["dur"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for (wgs, _), c in rows.items():
        for k, v in c.items():
            agg[wgs][k].append(v)
    for wgs, c in sorted(agg.items()):
        n = len(c["dur"])
        d = sorted(c["dur"])[n // 2]                      # median duration, ns
        g = sorted(c["GRBM_GUI_ACTIVE"])[n // 2] / 8.0      # cycles per XCD
        b = sorted(c["SQ_VALU_MFMA_BUSY_CYCLES"])[n // 2]
        clock = g / d                                       # GHz
        busy = b / (1024.0 * g)
        res[f"arm{a}_wgs{wgs}"] = {"launches": n, "median_us": d / 1e3, "clock_ghz": clock, "mfma_busy": busy, "busy_x_clock": busy * clock}
        print(f"arm {a} wgs {wgs:6d}: {n:3d} launches  {d/1e3:8.1f} us  clock {clock:.3f} GHz  MFMA busy {busy:.3f}  busy x clock {busy*clock:.3f}")
json.dump(res, open(f"{out}/busy_clock.json", "w"), indent=1, sort_keys=True)
PY
